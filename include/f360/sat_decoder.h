// f360/sat_decoder.h -- drop-in for the reference's src/sat_decoder.h:20-83
// (device methods).  The AVCodecContext* parameter of SampleFrameRectGPU is a
// template: the reference reads only ->width / ->height (src/sat_decoder.cc
// :328-329), so a real AVCodecContext or any struct with those fields works.
// The CPU twins DecodeFrameCPU / ExpandSampledFrameRectCPU / InterpolateFrameRectCPU
// are the host implementations of f360/cpu_twins.h.  Not provided: SampleFrameRectCPU
// (src/sat_decoder.cc:400-532 reads its 3-element table with a 4-element row
// stride -- wrong rows, out of bounds below 3/4 of the height: SURVEY.md 8a-10) and
// the experimental SVD pair CreateReducedSAT / SampleFrameFromReducedSAT and
// SampleFrameRectGPU360 (no caller in the reference; SURVEY.md 2.1 #2f, 2.2).
#pragma once

#include <cstdint>
#include <cstdlib>
#include <iostream>

#include "cpu_twins.h"
#include "opencl_manager.h"

class SATDecoder {
 private:
  OpenCLManager *cl_manager = nullptr;
  f360_sat_decoder *impl = nullptr;
  bool use_opencl = false;

  SATDecoder(const SATDecoder &) = delete;
  SATDecoder &operator=(const SATDecoder &) = delete;

 public:
  SATDecoder() = default;  // src/sat_decoder.cc:4
  explicit SATDecoder(OpenCLManager *manager) : cl_manager(manager) {  // :6-127
    if (manager && f360_satdec_create(manager->context.get(), &impl) == F360_OK) {
      use_opencl = true;
    } else if (manager) {
      std::cerr << "Failed to create decoder state: " << f360_last_error_string() << std::endl;
      exit(EXIT_FAILURE);  // the reference exits when its programs fail to build (:41-62)
    }
  }
  ~SATDecoder() { f360_satdec_destroy(impl); }

  // src/sat_decoder.cc:139-174
  void InitializeGrid(int target_width, int target_height, int source_width, int source_height) {
    if (!use_opencl) return;
    const int ret = f360_satdec_initialize_grid(impl, target_width, target_height, source_width,
                                                source_height);
    if (ret != F360_OK)
      std::cerr << "[SATDecoder::InitializeGrid] Launch Kernel Failed; " << ret << std::endl;
  }

  // src/sat_decoder.cc:176-210 (the reference launch always fails with work_dim 0; this runs)
  void DecodeFrameGPU(cl_mem cl_target_buffer, int target_linesize, cl_mem cl_source_buffer,
                      int width, int height) {
    if (!use_opencl) {
      std::cerr << "[SATDecoder::DecodeFrameGPU] Not initialized with OpenCL" << std::endl;
      return;
    }
    const int ret = f360_satdec_decode(impl, static_cast<uint8_t *>(cl_target_buffer),
                                       target_linesize,
                                       static_cast<const uint32_t *>(cl_source_buffer), width,
                                       height);
    if (ret != F360_OK)
      std::cerr << "[SATDecoder::DecodeFrameGPU] decode kernel launch failed:" << ret << std::endl;
  }

  // src/sat_decoder.cc:301-348
  template <class CodecContext>
  void SampleFrameRectGPU(cl_mem cl_target_buffer, int target_width, int target_height,
                          int target_linesize, cl_mem cl_source_buffer, CodecContext *codec_ctx,
                          float center_x, float center_y) {
    if (!use_opencl) {
      std::cerr << "[SATDecoder::SampleFrameRectGPU] Not initialized with OpenCL" << std::endl;
      return;
    }
    const int ret = f360_satdec_sample_rect(
        impl, static_cast<uint8_t *>(cl_target_buffer), target_width, target_height,
        target_linesize, static_cast<const uint32_t *>(cl_source_buffer), codec_ctx->width,
        codec_ctx->height, center_x, center_y);
    if (ret != F360_OK)
      std::cerr << "[SATDecoder::SampleFrameRectGPU] Sample rect kernel launch failed:" << ret
                << " " << OpenCLManager::GetCLErrorString(ret) << std::endl;
  }

  // Not in the reference: `count` frames in shared launches -- target k is table k sampled at
  // (centers_xy[2k], centers_xy[2k+1]); the counterpart of SATEncoder::EncodeFramesGPU.
  template <class CodecContext>
  void SampleFramesRectGPU(int count, cl_mem const *cl_target_buffers, int target_width,
                           int target_height, int target_linesize,
                           cl_mem const *cl_source_buffers, CodecContext *codec_ctx,
                           const float *centers_xy) {
    if (!use_opencl) {
      std::cerr << "[SATDecoder::SampleFramesRectGPU] Not initialized with OpenCL" << std::endl;
      return;
    }
    const int ret = f360_satdec_sample_rect_frames(
        impl, reinterpret_cast<uint8_t *const *>(cl_target_buffers), count, target_width,
        target_height, target_linesize,
        reinterpret_cast<const uint32_t *const *>(cl_source_buffers), codec_ctx->width,
        codec_ctx->height, centers_xy);
    if (ret != F360_OK)
      std::cerr << "[SATDecoder::SampleFramesRectGPU] Sample rect kernel launch failed:" << ret
                << " " << OpenCLManager::GetCLErrorString(ret) << std::endl;
  }

  // Not in the reference: SATEncoder::EncodeFramesGPU + SampleFramesRectGPU for `count` frames
  // whose gaze is known before the encode -- the offline modes, which read it from a trace
  // (src/run_satlogrectilinear.cc:932-938); NOT the server loop, which reads the latest gaze
  // after the encode and the tick (src/video_server.cc:324-328) and keeps the two calls.  The
  // same tables and reduced frames; the reduced pixels are produced during the encoder's pass
  // (strip walker from 23 8K frames per call on, band writer below) and no table is read back.
  template <class CodecContext>
  void EncodeSampleFramesGPU(int count, cl_mem const *cl_target_buffers, int target_width,
                             int target_height, int target_linesize, cl_mem const *cl_tables,
                             cl_mem const *cl_source_frames, CodecContext *codec_ctx,
                             int source_linesize, const float *centers_xy) {
    if (!use_opencl) {
      std::cerr << "[SATDecoder::EncodeSampleFramesGPU] Not initialized with OpenCL" << std::endl;
      return;
    }
    const int ret = f360_satdec_encode_sample_frames(
        impl, reinterpret_cast<uint8_t *const *>(cl_target_buffers),
        reinterpret_cast<uint32_t *const *>(cl_tables),
        reinterpret_cast<const uint8_t *const *>(cl_source_frames), count, target_width,
        target_height, target_linesize, codec_ctx->width, codec_ctx->height, source_linesize,
        centers_xy);
    if (ret != F360_OK)
      std::cerr << "[SATDecoder::EncodeSampleFramesGPU] kernel launch failed:" << ret << " "
                << OpenCLManager::GetCLErrorString(ret) << std::endl;
  }

  // The same from the decoder's planar YUV 4:2:0 frames (SATEncoder::EncodeFramesYUV420PGPU's
  // planes: one linesize per plane kind for all frames).
  template <class CodecContext>
  void EncodeSampleFramesYUV420PGPU(int count, cl_mem const *cl_target_buffers, int target_width,
                                    int target_height, int target_linesize,
                                    cl_mem const *cl_tables, cl_mem const *cl_y,
                                    cl_mem const *cl_u, cl_mem const *cl_v, int y_linesize,
                                    int u_linesize, int v_linesize, CodecContext *codec_ctx,
                                    const float *centers_xy) {
    if (!use_opencl) {
      std::cerr << "[SATDecoder::EncodeSampleFramesYUV420PGPU] Not initialized with OpenCL"
                << std::endl;
      return;
    }
    const int ret = f360_satdec_encode_sample_frames_yuv420p(
        impl, reinterpret_cast<uint8_t *const *>(cl_target_buffers),
        reinterpret_cast<uint32_t *const *>(cl_tables),
        reinterpret_cast<const uint8_t *const *>(cl_y),
        reinterpret_cast<const uint8_t *const *>(cl_u),
        reinterpret_cast<const uint8_t *const *>(cl_v), y_linesize, u_linesize, v_linesize, count,
        target_width, target_height, target_linesize, codec_ctx->width, codec_ctx->height,
        centers_xy);
    if (ret != F360_OK)
      std::cerr << "[SATDecoder::EncodeSampleFramesYUV420PGPU] kernel launch failed:" << ret
                << " " << OpenCLManager::GetCLErrorString(ret) << std::endl;
  }

  // The reduced frames alone (no tables): FoveateFrameRectGPU below for `count` frames.
  void FoveateFramesRectGPU(int count, cl_mem const *cl_target_buffers, int target_width,
                            int target_height, int target_linesize,
                            cl_mem const *cl_source_frames, int source_width, int source_height,
                            int source_linesize, const float *centers_xy) {
    if (!use_opencl) {
      std::cerr << "[SATDecoder::FoveateFramesRectGPU] Not initialized with OpenCL" << std::endl;
      return;
    }
    const int ret = f360_satdec_foveate_rect_frames(
        impl, reinterpret_cast<uint8_t *const *>(cl_target_buffers),
        reinterpret_cast<const uint8_t *const *>(cl_source_frames), count, target_width,
        target_height, target_linesize, source_width, source_height, source_linesize, centers_xy);
    if (ret != F360_OK)
      std::cerr << "[SATDecoder::FoveateFramesRectGPU] kernel launch failed:" << ret << " "
                << OpenCLManager::GetCLErrorString(ret) << std::endl;
  }

  // ... and from planar YUV 4:2:0 frames.
  void FoveateFramesRectYUV420PGPU(int count, cl_mem const *cl_target_buffers, int target_width,
                                   int target_height, int target_linesize, cl_mem const *cl_y,
                                   cl_mem const *cl_u, cl_mem const *cl_v, int y_linesize,
                                   int u_linesize, int v_linesize, int source_width,
                                   int source_height, const float *centers_xy) {
    if (!use_opencl) {
      std::cerr << "[SATDecoder::FoveateFramesRectYUV420PGPU] Not initialized with OpenCL"
                << std::endl;
      return;
    }
    const int ret = f360_satdec_foveate_rect_frames_yuv420p(
        impl, reinterpret_cast<uint8_t *const *>(cl_target_buffers),
        reinterpret_cast<const uint8_t *const *>(cl_y),
        reinterpret_cast<const uint8_t *const *>(cl_u),
        reinterpret_cast<const uint8_t *const *>(cl_v), y_linesize, u_linesize, v_linesize, count,
        target_width, target_height, target_linesize, source_width, source_height, centers_xy);
    if (ret != F360_OK)
      std::cerr << "[SATDecoder::FoveateFramesRectYUV420PGPU] kernel launch failed:" << ret << " "
                << OpenCLManager::GetCLErrorString(ret) << std::endl;
  }

  // Not in the reference: EncodeFrameGPU + SampleFrameRectGPU fused for a gaze known before the
  // encode (its offline modes, src/run_satlogrectilinear.cc:926-938); same bytes, no table.
  void FoveateFrameRectGPU(cl_mem cl_target_buffer, int target_width, int target_height,
                           int target_linesize, cl_mem cl_source_frame, int source_width,
                           int source_height, int source_linesize, float center_x,
                           float center_y) {
    if (!use_opencl) {
      std::cerr << "[SATDecoder::FoveateFrameRectGPU] Not initialized with OpenCL" << std::endl;
      return;
    }
    const int ret = f360_satdec_foveate_rect(
        impl, static_cast<uint8_t *>(cl_target_buffer), target_width, target_height,
        target_linesize, static_cast<const uint8_t *>(cl_source_frame), source_width,
        source_height, source_linesize, center_x, center_y);
    if (ret != F360_OK)
      std::cerr << "[SATDecoder::FoveateFrameRectGPU] kernel launch failed:" << ret << " "
                << OpenCLManager::GetCLErrorString(ret) << std::endl;
  }

  // The same from a decoder's yuv420p planes (f360_sat_encode_yuv420p's requirements).
  void FoveateFrameRectYUV420PGPU(cl_mem cl_target_buffer, int target_width, int target_height,
                                  int target_linesize, cl_mem cl_y, cl_mem cl_u, cl_mem cl_v,
                                  int y_linesize, int u_linesize, int v_linesize,
                                  int source_width, int source_height, float center_x,
                                  float center_y) {
    if (!use_opencl) {
      std::cerr << "[SATDecoder::FoveateFrameRectYUV420PGPU] Not initialized with OpenCL"
                << std::endl;
      return;
    }
    const int ret = f360_satdec_foveate_rect_yuv420p(
        impl, static_cast<uint8_t *>(cl_target_buffer), target_width, target_height,
        target_linesize, static_cast<const uint8_t *>(cl_y), static_cast<const uint8_t *>(cl_u),
        static_cast<const uint8_t *>(cl_v), y_linesize, u_linesize, v_linesize, source_width,
        source_height, center_x, center_y);
    if (ret != F360_OK)
      std::cerr << "[SATDecoder::FoveateFrameRectYUV420PGPU] kernel launch failed:" << ret
                << " " << OpenCLManager::GetCLErrorString(ret) << std::endl;
  }

  // The reference has this view on the CPU only (ExpandSampledFrameRectCPU,
  // src/sat_decoder.cc:555-616, AVFrame arguments); same result on device buffers.
  void ExpandSampledFrameRectGPU(cl_mem cl_target_buffer, int target_width, int target_height,
                                 int target_linesize, cl_mem cl_source_buffer, int source_width,
                                 int source_height, int source_linesize, float center_x,
                                 float center_y) {
    if (!use_opencl) {
      std::cerr << "[SATDecoder::ExpandSampledFrameRectGPU] Not initialized with OpenCL"
                << std::endl;
      return;
    }
    const int ret = f360_expand_rect(
        cl_manager->context.get(), static_cast<uint8_t *>(cl_target_buffer), target_width,
        target_height, target_linesize, static_cast<const uint8_t *>(cl_source_buffer),
        source_width, source_height, source_linesize, center_x, center_y);
    if (ret != F360_OK)
      std::cerr << "[SATDecoder::ExpandSampledFrameRectGPU] kernel launch failed:" << ret << " "
                << OpenCLManager::GetCLErrorString(ret) << std::endl;
  }

  // src/sat_decoder.cc:887-928
  void InterpolateFrameRectGPU(cl_mem cl_target_buffer, int target_width, int target_height,
                               int target_linesize, cl_mem cl_source_buffer, int source_width,
                               int source_height, int source_linesize, float center_x,
                               float center_y) {
    if (!use_opencl) {
      std::cerr << "[SATDecoder::InterpolateFrameRectGPU] Not initialized with OpenCL"
                << std::endl;
      return;
    }
    const int ret = f360_satdec_interpolate_rect(
        impl, static_cast<uint8_t *>(cl_target_buffer), target_width, target_height,
        target_linesize, static_cast<const uint8_t *>(cl_source_buffer), source_width,
        source_height, source_linesize, center_x, center_y);
    if (ret != F360_OK) {
      std::cerr << "[SATDecoder::InterpolateFrameRectGPU] interpolate kernel launch failed:"
                << ret << " " << OpenCLManager::GetCLErrorString(ret) << std::endl;
      exit(EXIT_FAILURE);  // :924
    }
  }

  // ---- host twins (f360/cpu_twins.h), templated on the frame / codec-context types: anything
  // with ->width, ->height, ->data[0], ->linesize[0] (an AVFrame, an AVCodecContext)
  // src/sat_decoder.cc:212-299 (prints the size like the reference does)
  template <class Frame, class CodecContext>
  void DecodeFrameCPU(Frame *target_frame, uint32_t *buffer, CodecContext *codec_ctx) {
    std::cout << "width: " << codec_ctx->width << ", height: " << codec_ctx->height << std::endl;
    f360cpu::decode_frame(target_frame->data[0], target_frame->linesize[0], buffer,
                          codec_ctx->width, codec_ctx->height);
  }
  // src/sat_decoder.cc:555-616
  template <class Frame>
  void ExpandSampledFrameRectCPU(Frame *target_frame, Frame *source_frame, float center_x,
                                 float center_y) {
    f360cpu::expand_rect(target_frame->data[0], target_frame->width, target_frame->height,
                         target_frame->linesize[0], source_frame->data[0], source_frame->width,
                         source_frame->height, source_frame->linesize[0], center_x, center_y);
  }
  // src/sat_decoder.cc:618-772
  template <class Frame>
  void InterpolateFrameRectCPU(Frame *target_frame, Frame *source_frame, float center_x,
                               float center_y) {
    f360cpu::interpolate_rect(target_frame->data[0], target_frame->width, target_frame->height,
                              target_frame->linesize[0], source_frame->data[0],
                              source_frame->width, source_frame->height,
                              source_frame->linesize[0], center_x, center_y);
  }
};
