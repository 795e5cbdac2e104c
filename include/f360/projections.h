// f360/projections.h -- drop-in for the reference's src/projections.h:21-36.
// The header of the reference names the 2nd/3rd parameters (target_height,
// target_width) while its .cc defines (target_width, target_height)
// (src/projections.cc:51-55); callers pass width first, which is what this
// signature says.
#pragma once

#include <cstdint>
#include <iostream>

#include "opencl_manager.h"

class Projections {
 private:
  bool use_OpenCL = false;
  OpenCLManager *cl_manager = nullptr;

 public:
  explicit Projections(OpenCLManager *manager)
      : use_OpenCL(manager != nullptr), cl_manager(manager) {}

  void GnomonicProjection(cl_mem cl_target_buffer, int target_width, int target_height,
                          int target_linesize, cl_mem cl_source_buffer, int source_width,
                          int source_height, int source_linesize, float center_x,
                          float center_y) {
    if (!use_OpenCL) {
      std::cerr << "GnomonicProjection Not initialized with OpenCL" << std::endl;
      return;
    }
    const int ret = f360_gnomonic(cl_manager->context.get(),
                                  static_cast<uint8_t *>(cl_target_buffer), target_width,
                                  target_height, target_linesize,
                                  static_cast<const uint8_t *>(cl_source_buffer), source_width,
                                  source_height, source_linesize, center_x, center_y);
    if (ret != F360_OK)
      std::cerr << "[GnomonicProjection] Gnomonic kernel launch failed:" << ret << " "
                << OpenCLManager::GetCLErrorString(ret) << std::endl;
  }
};
