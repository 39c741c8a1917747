// String.h -- trim / split with the signatures of the reference's include/ife/Util/String.h
// (:19, :34; bodies in its src/Util/String.cxx), header-only here.
#ifndef __String_h
#define __String_h

#include <string>
#include <vector>

// Remove every character of `chars` from both ends of s.
inline std::string trim(std::string s, std::string chars = " ") {
  const std::string::size_type a = s.find_first_not_of(chars);
  if (a == std::string::npos) return std::string();
  const std::string::size_type b = s.find_last_not_of(chars);
  return s.substr(a, b - a + 1);
}

// Tokens of s between occurrences of delim; an empty trailing token is dropped.
inline std::vector<std::string> split(std::string s, char delim) {
  std::vector<std::string> tokens;
  std::string::size_type start = 0;
  for (std::string::size_type pos = s.find(delim); pos != std::string::npos; pos = s.find(delim, start)) {
    tokens.push_back(s.substr(start, pos - start));
    start = pos + 1;
  }
  if (start < s.size()) tokens.push_back(s.substr(start));
  return tokens;
}

#endif
