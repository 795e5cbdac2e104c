// f360/gaze_view_points.h -- reader/writer of the reference's gaze-trace text format
// (src/gaze_view_points.{h,cc}):  one sample per line,
//     frame,<n>,forward,<x>,<y>,eye,<x>,<y>
// "forward" is the view point, "eye" the gaze point, both as fractions of the frame; the
// pred_* fields hold the PREVIOUS sample (the reference's one-frame-late prediction,
// src/gaze_view_points.cc:24-31).  Lines that do not contain the pattern are skipped.
#pragma once

#include <cstdio>
#include <fstream>
#include <iostream>
#include <regex>
#include <string>
#include <vector>

class GazeViewPoints {
 public:
  struct GazeViewPoint {
    unsigned int frame = 0;
    float view_point[2] = {0, 0};
    float gaze_point[2] = {0, 0};
    float pred_view_point[2] = {0, 0};
    float pred_gaze_point[2] = {0, 0};
  };

  std::vector<GazeViewPoint> points;

  GazeViewPoints() = default;
  explicit GazeViewPoints(const std::string &file_path) {
    std::ifstream file(file_path);
    if (!file.good()) {
      std::cerr << "Cannot open file: " << file_path << std::endl;
      return;
    }
    static const char *kNum = R"(([-+]?\d*\.?\d+(?:[eE][-+]?\d+)?))";
    const std::regex sample(std::string("frame,(\\d+),forward,") + kNum + "," + kNum + ",eye," +
                            kNum + "," + kNum);
    std::string line;
    while (std::getline(file, line)) {
      std::smatch m;
      if (!std::regex_search(line, m, sample)) continue;
      GazeViewPoint p;
      p.frame = (unsigned int)std::stoul(m.str(1));
      p.view_point[0] = std::stof(m.str(2));
      p.view_point[1] = std::stof(m.str(3));
      p.gaze_point[0] = std::stof(m.str(4));
      p.gaze_point[1] = std::stof(m.str(5));
      const GazeViewPoint &prev = points.empty() ? p : points.back();
      for (int k = 0; k < 2; ++k) {
        p.pred_view_point[k] = prev.view_point[k];
        p.pred_gaze_point[k] = prev.gaze_point[k];
      }
      points.push_back(p);
    }
  }

  // Writes the same format (9 significant digits: floats round-trip).
  static bool Write(const std::string &file_path, const std::vector<GazeViewPoint> &pts) {
    FILE *f = std::fopen(file_path.c_str(), "w");
    if (!f) return false;
    for (const GazeViewPoint &p : pts)
      std::fprintf(f, "frame,%u,forward,%.9g,%.9g,eye,%.9g,%.9g\n", p.frame, p.view_point[0],
                   p.view_point[1], p.gaze_point[0], p.gaze_point[1]);
    std::fclose(f);
    return true;
  }
};
