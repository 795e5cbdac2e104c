// f360/image_sampler.h -- drop-in for the reference's src/image_sampler.h:30-102
// (device methods).  Not provided: the CPU twins, and the image-pyramid pair
// GenerateImagePyramid / SampleFrameLogPolarGPUFromImagePyramid whose kernel
// source is absent from the reference (src/image_sampler.cc:125-126).
#pragma once

#include <cstdint>
#include <cstdlib>
#include <iostream>

#include "cpu_twins.h"
#include "opencl_manager.h"

#ifndef ROUND_UP_TO
// Rounds x up to the nearest y. E.g. ROUND_UP_TO(5, 8) == 8 (src/image_sampler.h:21)
#define ROUND_UP_TO(x, y) (y * ((x + y - 1) / y))
#endif

class ImageSampler {
 private:
  OpenCLManager *cl_manager = nullptr;
  f360_image_sampler *impl = nullptr;
  bool use_opencl = false;

  ImageSampler(const ImageSampler &) = delete;
  ImageSampler &operator=(const ImageSampler &) = delete;

  void report(const char *where, int ret) const {
    if (ret != F360_OK)
      std::cerr << "[ImageSampler::" << where << "] kernel launch failed:" << ret << " "
                << f360_last_error_string() << std::endl;
  }

 public:
  ImageSampler() = default;
  explicit ImageSampler(OpenCLManager *manager) : cl_manager(manager) {
    if (manager && f360_is_create(manager->context.get(), &impl) == F360_OK) {
      use_opencl = true;
    } else if (manager) {
      std::cerr << "Failed to create sampler state: " << f360_last_error_string() << std::endl;
      exit(EXIT_FAILURE);
    }
  }
  ~ImageSampler() { f360_is_destroy(impl); }

  // The reference has these two views on the CPU only (ExpandSampledFrameRectCPU,
  // src/image_sampler.cc:358-419; ExpandSampledFrameLogPolarCPU, :623-666; AVFrame arguments);
  // same results on device buffers.
  void ExpandSampledFrameRectGPU(cl_mem cl_target_buffer, int target_width, int target_height,
                                 int target_linesize, cl_mem cl_source_buffer, int source_width,
                                 int source_height, int source_linesize, float center_x,
                                 float center_y) {
    Expand(false, cl_target_buffer, target_width, target_height, target_linesize,
           cl_source_buffer, source_width, source_height, source_linesize, center_x, center_y);
  }
  void ExpandSampledFrameLogPolarGPU(cl_mem cl_target_buffer, int target_width,
                                     int target_height, int target_linesize,
                                     cl_mem cl_source_buffer, int source_width,
                                     int source_height, int source_linesize, float center_x,
                                     float center_y) {
    Expand(true, cl_target_buffer, target_width, target_height, target_linesize,
           cl_source_buffer, source_width, source_height, source_linesize, center_x, center_y);
  }

  void InitializeGrid(int target_width, int target_height, int source_width, int source_height) {
    if (use_opencl)
      report("InitializeGrid", f360_is_initialize_grid(impl, target_width, target_height,
                                                       source_width, source_height));
  }
  void InitializeLogpolarGrid(int target_width, int target_height, int source_width,
                              int source_height) {
    if (use_opencl)
      report("InitializeLogpolarGrid",
             f360_is_initialize_logpolar_grid(impl, target_width, target_height, source_width,
                                              source_height));
  }
  void SampleFrameRectGPU(cl_mem cl_target_buffer, int target_width, int target_height,
                          int target_linesize, cl_mem cl_source_buffer, int source_width,
                          int source_height, int source_linesize, float center_x,
                          float center_y) {
    if (!use_opencl) {
      std::cerr << "[ImageSampler::SampleFrameRectGPU] Not initialized with OpenCL" << std::endl;
      return;
    }
    report("SampleFrameRectGPU",
           f360_is_sample_rect(impl, static_cast<uint8_t *>(cl_target_buffer), target_width,
                               target_height, target_linesize,
                               static_cast<const uint8_t *>(cl_source_buffer), source_width,
                               source_height, source_linesize, center_x, center_y));
  }
  void SampleFrameLogPolarGPU(cl_mem cl_target_buffer, int target_width, int target_height,
                              int target_linesize, cl_mem cl_source_buffer, int source_width,
                              int source_height, int source_linesize, float center_x,
                              float center_y) {
    if (!use_opencl) {
      std::cerr << "[ImageSampler::SampleFrameLogPolarGPU] Not initialized with OpenCL"
                << std::endl;
      return;
    }
    report("SampleFrameLogPolarGPU",
           f360_is_sample_logpolar(impl, static_cast<uint8_t *>(cl_target_buffer), target_width,
                                   target_height, target_linesize,
                                   static_cast<const uint8_t *>(cl_source_buffer), source_width,
                                   source_height, source_linesize, center_x, center_y));
  }
  void InterpolateFrameLogPolarGPU(cl_mem cl_target_buffer, int target_width, int target_height,
                                   int target_linesize, cl_mem cl_source_buffer,
                                   int source_width, int source_height, int source_linesize,
                                   float center_x, float center_y) {
    if (!use_opencl) {
      std::cerr << "[ImageSampler::InterpolateFrameLogPolarGPU] Not initialized with OpenCL"
                << std::endl;
      return;
    }
    report("InterpolateFrameLogPolarGPU",
           f360_is_interpolate_logpolar(impl, static_cast<uint8_t *>(cl_target_buffer),
                                        target_width, target_height, target_linesize,
                                        static_cast<const uint8_t *>(cl_source_buffer),
                                        source_width, source_height, source_linesize, center_x,
                                        center_y));
  }
  void ApplyLogPolarGaussianBlur(cl_mem cl_target_buffer, int target_width, int target_height,
                                 int target_linesize, cl_mem cl_source_buffer) {
    if (!use_opencl) {
      std::cerr << "[ImageSampler::ApplyLogPolarGaussianBlur] Not initialized with OpenCL"
                << std::endl;
      return;
    }
    report("ApplyLogPolarGaussianBlur",
           f360_is_logpolar_gaussian_blur(impl, static_cast<uint8_t *>(cl_target_buffer),
                                          target_width, target_height, target_linesize,
                                          static_cast<const uint8_t *>(cl_source_buffer)));
  }

 private:
  void Expand(bool logpolar, cl_mem cl_target_buffer, int target_width, int target_height,
              int target_linesize, cl_mem cl_source_buffer, int source_width, int source_height,
              int source_linesize, float center_x, float center_y) {
    if (!cl_manager) {
      std::cerr << "[ImageSampler::ExpandSampledFrame] Not initialized with OpenCL" << std::endl;
      return;
    }
    auto fn = logpolar ? f360_expand_logpolar : f360_expand_rect;
    const int ret = fn(cl_manager->context.get(), static_cast<uint8_t *>(cl_target_buffer),
                       target_width, target_height, target_linesize,
                       static_cast<const uint8_t *>(cl_source_buffer), source_width,
                       source_height, source_linesize, center_x, center_y);
    if (ret != F360_OK)
      std::cerr << "[ImageSampler::ExpandSampledFrame] kernel launch failed:" << ret << " "
                << OpenCLManager::GetCLErrorString(ret) << std::endl;
  }

 public:
  // ---- host twins (f360/cpu_twins.h), templated on the frame / codec-context types
  // src/image_sampler.cc:302-356 ("Untested" there): point sample of a uint32 buffer with
  // four elements per pixel
  template <class Frame, class CodecContext>
  void SampleFrameRectCPU(Frame *target_frame, uint32_t *buffer, CodecContext *codec_ctx,
                          float center_x, float center_y) {
    f360cpu::sample_rect_point(target_frame->data[0], target_frame->width, target_frame->height,
                               target_frame->linesize[0], buffer, codec_ctx->width,
                               codec_ctx->height, center_x, center_y);
  }
  // src/image_sampler.cc:358-419
  template <class Frame>
  void ExpandSampledFrameRectCPU(Frame *target_frame, Frame *source_frame, float center_x,
                                 float center_y) {
    f360cpu::expand_rect(target_frame->data[0], target_frame->width, target_frame->height,
                         target_frame->linesize[0], source_frame->data[0], source_frame->width,
                         source_frame->height, source_frame->linesize[0], center_x, center_y);
  }
  // src/image_sampler.cc:421-575
  template <class Frame>
  void InterpolateFrameRectCPU(Frame *target_frame, Frame *source_frame, float center_x,
                               float center_y) {
    f360cpu::interpolate_rect(target_frame->data[0], target_frame->width, target_frame->height,
                              target_frame->linesize[0], source_frame->data[0],
                              source_frame->width, source_frame->height,
                              source_frame->linesize[0], center_x, center_y);
  }
  // src/image_sampler.cc:623-666
  template <class Frame>
  void ExpandSampledFrameLogPolarCPU(Frame *target_frame, Frame *source_frame, float center_x,
                                     float center_y) {
    f360cpu::expand_logpolar(target_frame->data[0], target_frame->width, target_frame->height,
                             target_frame->linesize[0], source_frame->data[0],
                             source_frame->width, source_frame->height,
                             source_frame->linesize[0], center_x, center_y);
  }
  // src/image_sampler.cc:668-778
  template <class Frame>
  void InterpolateFrameLogPolarCPU(Frame *target_frame, Frame *source_frame, float center_x,
                                   float center_y) {
    f360cpu::interpolate_logpolar(target_frame->data[0], target_frame->width,
                                  target_frame->height, target_frame->linesize[0],
                                  source_frame->data[0], source_frame->width,
                                  source_frame->height, source_frame->linesize[0], center_x,
                                  center_y);
  }
};
