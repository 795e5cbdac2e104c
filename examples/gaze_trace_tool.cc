// gaze_trace_tool.cc -- reads a gaze trace in the reference's text format with
// f360/gaze_view_points.h and prints what was parsed; with a second argument, rewrites it.
//   ./gaze_trace_tool <in.txt> [out.txt]
#include <cstdio>

#include "f360/gaze_view_points.h"

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  GazeViewPoints trace(argv[1]);
  for (const auto &p : trace.points)
    std::printf("%u %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", p.frame, p.view_point[0],
                p.view_point[1], p.gaze_point[0], p.gaze_point[1], p.pred_view_point[0],
                p.pred_view_point[1], p.pred_gaze_point[0], p.pred_gaze_point[1]);
  if (argc > 2 && !GazeViewPoints::Write(argv[2], trace.points)) return 1;
  return 0;
}
