// dump_frames_ppm.cc -- visual check of the hot path, counterpart of the PNG dumps of the
// reference's offline tool (src/run_satlogrectilinear.cc:169,240,326,415 via src/save_frame.h):
// a structured synthetic frame (gradients + checker + rings, so the warp is visible) goes
// through EncodeFrameGPU -> SampleFrameRectGPU -> InterpolateFrameRectGPU and every stage is
// written as a binary PPM with include/f360/save_frame.h.
//
//   ./dump_frames_ppm <outdir> [width height center_x center_y]
// writes <outdir>/source.ppm, reduced.ppm, unwarped.ppm and prints one JSON line.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "f360/parameters.h"
#include "f360/sat_decoder.h"
#include "f360/sat_encoder.h"
#include "f360/save_frame.h"

struct CodecContext {
  int width, height;
};

int main(int argc, char **argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s <outdir> [width height center_x center_y]\n", argv[0]);
    return EXIT_FAILURE;
  }
  const std::string outdir = argv[1];
  const int width = argc > 3 ? atoi(argv[2]) : 1920, height = argc > 3 ? atoi(argv[3]) : 1080;
  const float cx = argc > 5 ? (float)atof(argv[4]) : 0.65f, cy = argc > 5 ? (float)atof(argv[5]) : 0.75f;
  const int out_w = f360_reduced_size(width), out_h = f360_reduced_size(height);

  std::vector<uint8_t> frame((size_t)4 * width * height);
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      uint8_t *p = &frame[((size_t)y * width + x) * 4];
      const double r = std::hypot(x - cx * width, y - cy * height);
      p[0] = (uint8_t)(x * 255 / (width > 1 ? width - 1 : 1));
      p[1] = (uint8_t)(y * 255 / (height > 1 ? height - 1 : 1));
      p[2] = (uint8_t)((((x / 32) + (y / 32)) & 1) * 90 + (((int)(r / 24)) & 1) * 120 + 20);
      p[3] = 0;
    }

  OpenCLManager cl_manager;
  if (cl_manager.InitializeContext() != 0) return EXIT_FAILURE;
  SATEncoder sat_encoder(&cl_manager);
  SATDecoder sat_decoder(&cl_manager);
  CodecContext codec = {width, height};
  cl::Buffer cl_source(cl_manager.context, CL_MEM_READ_WRITE, frame.size());
  cl::Buffer cl_sat(cl_manager.context, CL_MEM_READ_WRITE, (size_t)12 * width * height);
  cl::Buffer cl_reduced(cl_manager.context, CL_MEM_READ_WRITE, (size_t)4 * out_w * out_h);
  cl::Buffer cl_full(cl_manager.context, CL_MEM_READ_WRITE, frame.size());
  cl::copy(cl_manager.command_queue, frame.begin(), frame.end(), cl_source);
  std::vector<uint8_t> zeros((size_t)4 * out_w * out_h, 0);
  cl::copy(cl_manager.command_queue, zeros.begin(), zeros.end(), cl_reduced);
  sat_decoder.InitializeGrid(out_w, out_h, width, height);
  sat_encoder.EncodeFrameGPU(cl_sat(), cl_source(), width, height, 4 * width);
  sat_decoder.SampleFrameRectGPU(cl_reduced(), out_w, out_h, 4 * out_w, cl_sat(), &codec, cx, cy);
  sat_decoder.InterpolateFrameRectGPU(cl_full(), width, height, 4 * width, cl_reduced(), out_w, out_h,
                                      4 * out_w, cx, cy);
  clFinish(cl_manager.command_queue);

  bool ok = SaveFramePPM(frame.data(), width, height, 4 * width, 4, outdir + "/source");
  ok = SaveDeviceFramePPM(&cl_manager, cl_reduced(), out_w, out_h, 4 * out_w, 4, outdir + "/reduced") && ok;
  ok = SaveDeviceFramePPM(&cl_manager, cl_full(), width, height, 4 * width, 4, outdir + "/unwarped") && ok;
  std::printf("{\"ok\": %s, \"width\": %d, \"height\": %d, \"reduced\": [%d, %d], \"gaze\": [%.9g, %.9g]}\n",
              ok ? "true" : "false", width, height, out_w, out_h, cx, cy);
  return ok ? EXIT_SUCCESS : EXIT_FAILURE;
}
