// ROIReader.h -- reads region-of-interest files, "[x, y, z][sx, sy, sz]" per line, with the
// interface of the reference's ROIReader<TRegion> (include/ife/IO/ROIReader.h:8-22, parsing
// rules of ROIReader.hxx:28-47: optional header line, everything up to '[' skipped, fields
// separated by ',', a record counts only if the stream is still good after its last field).
#ifndef __ROIReader_h
#define __ROIReader_h

#include <fstream>
#include <iterator>
#include <limits>
#include <string>
#include <vector>

template <typename TRegion>
class ROIReader {
 public:
  typedef TRegion RegionType;
  typedef typename RegionType::SizeType SizeType;
  typedef typename RegionType::IndexType IndexType;

  static std::vector<RegionType> read(std::string path, bool header = true) {
    std::vector<RegionType> rois;
    read(path, std::back_inserter(rois), header);
    return rois;
  }
  template <typename OutputIter>
  static void read(std::string path, OutputIter it, bool header = true) {
    std::ifstream is(path.c_str());
    read(is, it, header);
  }
  template <typename OutputIter>
  static void read(std::istream &is, OutputIter it, bool header = true) {
    const std::streamsize all = std::numeric_limits<std::streamsize>::max();
    if (header) is.ignore(all, '\n');
    while (is.good()) {
      IndexType start;
      SizeType size;
      const char after[6] = {',', ',', '[', ',', ',', '\n'};
      is.ignore(all, '[');
      for (int k = 0; k < 6; ++k) {
        if (k < 3) is >> start[k];
        else is >> size[k - 3];
        is.ignore(all, after[k]);
      }
      if (is.good()) *it++ = RegionType(start, size);
    }
  }
};

#endif
