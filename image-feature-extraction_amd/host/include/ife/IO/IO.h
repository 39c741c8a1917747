// IO.h -- the text formats at the tool boundary, with the names of the reference's
// include/ife/IO/IO.h: StringPair (:12), writeSequenceAsText (:24-41) and readPairList
// (declared :112, body in its src/IO/IO.cxx:20-41).  Plain host code, header-only.
#ifndef __IO_h
#define __IO_h

#include <fstream>
#include <iostream>
#include <limits>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "ife/Util/String.h"

typedef std::pair<std::string, std::string> StringPair;

// "v0,v1,...": elements through operator<< with the stream's current formatting, sep
// between them, nothing after the last.
template <typename InputIt>
std::ostream &writeSequenceAsText(std::ostream &out, InputIt begin, InputIt end, char sep = ',') {
  for (InputIt it = begin; it != end; ++it) {
    if (it != begin) out << sep;
    out << *it;
  }
  return out;
}

// Numbers separated by sep (the rows of a histogram specification, tools/MakeBag.cxx:330-332;
// interface of IO.h:60-70): read while extraction succeeds, skipping through the next sep.
template <typename ElemT, typename CharT, typename OutputIt>
void readTextSequence(std::istream &is, OutputIt out, CharT sep = ',') {
  ElemT elem;
  while (is >> elem) {
    *out++ = elem;
    is.ignore(std::numeric_limits<std::streamsize>::max(), sep);
  }
}

// One "image<sep>mask" pair per line; blank lines skipped; a line without the separator
// throws std::invalid_argument; the first field is trimmed of blanks, the second of
// blanks, tabs and line ends.
inline std::vector<StringPair> readPairList(std::string inPath, char sep = ',') {
  std::ifstream is(inPath.c_str());
  std::vector<StringPair> pairs;
  std::string line;
  while (std::getline(is, line)) {
    if (line.empty()) continue;
    const std::string::size_type pos = line.find(sep);
    if (pos == std::string::npos) throw std::invalid_argument("Line does not contain a separator");
    pairs.push_back(StringPair(trim(line.substr(0, pos)), trim(line.substr(pos + 1), " \r\n\t")));
  }
  return pairs;
}

#endif
