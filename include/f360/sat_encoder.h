// f360/sat_encoder.h -- drop-in for the reference's src/sat_encoder.h:22-43.
// Same class name, constructors and GPU method signature; the body calls the
// HIP engine through the C ABI.  EncodeFrameCPU (src/sat_encoder.cc:137-185) is
// the host implementation of f360/cpu_twins.h, templated on the codec-context /
// frame types (anything with ->width, ->height / ->data[0], ->linesize[0]).
#pragma once

#include <cstdint>
#include <iostream>

#include "cpu_twins.h"
#include "opencl_manager.h"

class SATEncoder {
 private:
  bool use_OpenCL = false;
  OpenCLManager *cl_manager = nullptr;

 public:
  SATEncoder() = default;                               // src/sat_encoder.cc:3
  explicit SATEncoder(OpenCLManager *manager)           // src/sat_encoder.cc:5-52
      : use_OpenCL(manager != nullptr), cl_manager(manager) {}

  // src/sat_encoder.cc:67-135.  Asynchronous: enqueues on the manager's queue.
  void EncodeFrameGPU(cl_mem cl_target_buffer, cl_mem cl_source_buffer, int source_width,
                      int source_height, int source_linesize) {
    if (!use_OpenCL) {
      std::cerr << "[SATEncoder::EncodeFrameGPU] Not initialized with OpenCL" << std::endl;
      return;
    }
    const int ret = f360_sat_encode(cl_manager->context.get(),
                                    static_cast<uint32_t *>(cl_target_buffer),
                                    static_cast<const uint8_t *>(cl_source_buffer), source_width,
                                    source_height, source_linesize);
    if (ret != F360_OK)
      std::cerr << "[SATEncoder::EncodeFrameGPU] kernel launch failed:" << ret << " "
                << f360_last_error_string() << std::endl;
  }

  // Not in the reference: `count` frames of one geometry in shared launches (table k of source
  // k), what small frames and many connections per GPU want (f360_sat_encode_batch).
  void EncodeFramesGPU(int count, cl_mem const *cl_target_buffers, cl_mem const *cl_source_buffers,
                       int source_width, int source_height, int source_linesize) {
    if (!use_OpenCL) {
      std::cerr << "[SATEncoder::EncodeFramesGPU] Not initialized with OpenCL" << std::endl;
      return;
    }
    const int ret = f360_sat_encode_batch(
        cl_manager->context.get(), count, reinterpret_cast<uint32_t *const *>(cl_target_buffers),
        reinterpret_cast<const uint8_t *const *>(cl_source_buffers), source_width, source_height,
        source_linesize);
    if (ret != F360_OK)
      std::cerr << "[SATEncoder::EncodeFramesGPU] kernel launch failed:" << ret << " "
                << f360_last_error_string() << std::endl;
  }

  // Not in the reference: the tables for EncodeFramesGPU calls of `count` frames, placed for the
  // read-once encoder (f360_sat_tables_alloc: groups of tables are drawn, one launch is timed
  // into each, the fastest are kept).  Returns the pool to hand to FreeTablesGPU, or nullptr.
  f360_table_pool *AllocateTablesGPU(int count, cl_mem *cl_target_buffers_out, int source_width,
                                     int source_height) {
    if (!use_OpenCL) {
      std::cerr << "[SATEncoder::AllocateTablesGPU] Not initialized with OpenCL" << std::endl;
      return nullptr;
    }
    f360_table_pool *pool = nullptr;
    const int ret = f360_sat_tables_alloc(cl_manager->context.get(), source_width, source_height,
                                          count, reinterpret_cast<uint32_t **>(cl_target_buffers_out),
                                          &pool);
    if (ret != F360_OK) {
      std::cerr << "[SATEncoder::AllocateTablesGPU] failed:" << ret << " "
                << f360_last_error_string() << std::endl;
      return nullptr;
    }
    return pool;
  }
  void FreeTablesGPU(f360_table_pool *pool) {
    if (use_OpenCL) (void)f360_sat_tables_free(cl_manager->context.get(), pool);
  }

  // The same from a decoder's planes (frames of one decoder share their linesizes).
  void EncodeFramesYUV420PGPU(int count, cl_mem const *cl_target_buffers, cl_mem const *cl_y,
                              cl_mem const *cl_u, cl_mem const *cl_v, int y_linesize,
                              int u_linesize, int v_linesize, int source_width,
                              int source_height) {
    if (!use_OpenCL) {
      std::cerr << "[SATEncoder::EncodeFramesYUV420PGPU] Not initialized with OpenCL" << std::endl;
      return;
    }
    const int ret = f360_sat_encode_yuv420p_batch(
        cl_manager->context.get(), count, reinterpret_cast<uint32_t *const *>(cl_target_buffers),
        reinterpret_cast<const uint8_t *const *>(cl_y),
        reinterpret_cast<const uint8_t *const *>(cl_u),
        reinterpret_cast<const uint8_t *const *>(cl_v), y_linesize, u_linesize, v_linesize,
        source_width, source_height);
    if (ret != F360_OK)
      std::cerr << "[SATEncoder::EncodeFramesYUV420PGPU] kernel launch failed:" << ret << " "
                << f360_last_error_string() << std::endl;
  }

  // src/sat_encoder.cc:137-185: the same table on the host (uint32 [height][width][3]).
  template <class CodecContext, class Frame>
  void EncodeFrameCPU(uint32_t *target_frame, CodecContext *codec_ctx, Frame *frame) {
    f360cpu::encode_frame(target_frame, codec_ctx->width, codec_ctx->height, frame->data[0],
                          frame->linesize[0]);
  }

  // Not in the reference: the table of the RGB0 frame sws_scale would make of a decoder's
  // yuv420p planes (src/video_decoder.cc:222-224), computed from the planes directly.
  void EncodeFrameYUV420PGPU(cl_mem cl_target_buffer, cl_mem cl_y, cl_mem cl_u, cl_mem cl_v,
                             int y_linesize, int u_linesize, int v_linesize, int source_width,
                             int source_height) {
    if (!use_OpenCL) {
      std::cerr << "[SATEncoder::EncodeFrameYUV420PGPU] Not initialized with OpenCL" << std::endl;
      return;
    }
    const int ret = f360_sat_encode_yuv420p(
        cl_manager->context.get(), static_cast<uint32_t *>(cl_target_buffer),
        static_cast<const uint8_t *>(cl_y), static_cast<const uint8_t *>(cl_u),
        static_cast<const uint8_t *>(cl_v), y_linesize, u_linesize, v_linesize, source_width,
        source_height);
    if (ret != F360_OK)
      std::cerr << "[SATEncoder::EncodeFrameYUV420PGPU] kernel launch failed:" << ret << " "
                << f360_last_error_string() << std::endl;
  }
};
