// save_frame.h -- counterpart of the reference's SaveFramePNG (src/save_frame.h:15,72) for the
// visual checks of its offline tool (src/run_satlogrectilinear.cc:169,240,326,415), without
// libavcodec: binary PPM (P6), which every image viewer and `convert x.ppm x.png` read.
//
//   SaveFramePPM(frame, path)                    any frame-shaped struct with data[0] /
//                                                linesize[0] / width / height (RGB0 or RGB24)
//   SaveFramePPM(bytes, w, h, linesize, bpp, path)
//   SaveDeviceFramePPM(&cl_manager, cl_mem, w, h, linesize, bpp, path)   device buffer: blocking
//                                                copy to the host through the C ABI first
// Returns true on success; a failure is reported on std::cerr like the reference's helpers.
#pragma once

#include <cstdint>
#include <cstdio>
#include <iostream>
#include <string>
#include <vector>

#include "f360.h"
#include "f360/opencl_manager.h"

inline bool SaveFramePPM(const uint8_t *pixels, int width, int height, int linesize,
                         int bytes_per_pixel, std::string output_filepath) {
  if (!pixels || width < 1 || height < 1 || bytes_per_pixel < 3 ||
      linesize < width * bytes_per_pixel) {
    std::cerr << "SaveFramePPM: bad frame" << std::endl;
    return false;
  }
  if (output_filepath.size() < 4 ||
      output_filepath.compare(output_filepath.size() - 4, 4, ".ppm") != 0)
    output_filepath += ".ppm";
  FILE *f = std::fopen(output_filepath.c_str(), "wb");
  if (!f) {
    std::cerr << "SaveFramePPM: cannot open " << output_filepath << std::endl;
    return false;
  }
  std::fprintf(f, "P6\n%d %d\n255\n", width, height);
  std::vector<uint8_t> row((size_t)3 * width);
  bool ok = true;
  for (int y = 0; y < height && ok; ++y) {
    const uint8_t *src = pixels + (size_t)y * linesize;
    for (int x = 0; x < width; ++x) {
      row[3 * (size_t)x + 0] = src[(size_t)x * bytes_per_pixel + 0];
      row[3 * (size_t)x + 1] = src[(size_t)x * bytes_per_pixel + 1];
      row[3 * (size_t)x + 2] = src[(size_t)x * bytes_per_pixel + 2];
    }
    ok = std::fwrite(row.data(), 1, row.size(), f) == row.size();
  }
  ok = (std::fclose(f) == 0) && ok;
  if (!ok) std::cerr << "SaveFramePPM: short write to " << output_filepath << std::endl;
  return ok;
}

// AVFrame-shaped: packed RGB0 / RGB24 in data[0]; bytes per pixel = linesize[0] / width, the
// rule the reference's kernels use (src/sat_encoder_encode_kernels.cl:9)
template <class Frame>
inline bool SaveFramePPM(const Frame *frame, const std::string &output_filepath) {
  if (!frame || frame->width < 1) return false;
  return SaveFramePPM(frame->data[0], frame->width, frame->height, frame->linesize[0],
                      frame->linesize[0] / frame->width, output_filepath);
}

inline bool SaveDeviceFramePPM(OpenCLManager *cl_manager, cl_mem device_frame, int width,
                               int height, int linesize, int bytes_per_pixel,
                               const std::string &output_filepath) {
  if (!cl_manager || !cl_manager->command_queue.ctx() || !device_frame || height < 1 || linesize < 1) {
    std::cerr << "SaveDeviceFramePPM: bad argument" << std::endl;
    return false;
  }
  std::vector<uint8_t> host((size_t)linesize * height);
  if (f360_memcpy_d2h(cl_manager->command_queue.ctx(), host.data(), device_frame, host.size()) != F360_OK) {
    std::cerr << "SaveDeviceFramePPM: " << f360_last_error_string() << std::endl;
    return false;
  }
  return SaveFramePPM(host.data(), width, height, linesize, bytes_per_pixel, output_filepath);
}
