// run_satlogrectilinear_synth.cc -- the reference's offline tool call sequences on the HIP
// engine, through the drop-in class headers (include/f360/*.h).
//
// Mirrors the device part of /root/reference/src/run_satlogrectilinear.cc:
//   "interpolate_sampled"   (:360-415)  decode -> H2D -> EncodeFrameGPU -> SampleFrameRectGPU -> D2H
//   "foveate_no_encoding"   (:857-960)  per frame: H2D -> EncodeFrameGPU -> SampleFrameRectGPU
//                                        -> InterpolateFrameRectGPU back into the SOURCE buffer -> D2H
// with a synthetic LCG frame source in place of VideoDecoder (no FFmpeg here) and an FNV-1a digest
// in place of SaveFramePNG / NVENC.  The code between the markers is written exactly as a caller
// of the reference classes would write it (cl::Buffer, cl::copy, cl_manager.command_queue, ...).
//
//   ./run_satlogrectilinear_synth <mode> [width height frames]
// prints one JSON line with the digests the parity test compares with the oracle's.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "f360/image_sampler.h"
#include "f360/parameters.h"
#include "f360/projections.h"
#include "f360/sat_decoder.h"
#include "f360/sat_encoder.h"

struct CodecContext {  // stands in for AVCodecContext: only width / height are read
  int width, height;
};

static void lcg_fill(std::vector<uint8_t> &buf, uint32_t seed) {
  uint32_t s = seed;
  for (auto &b : buf) {
    s = s * 1664525u + 1013904223u;
    b = (uint8_t)(s >> 24);
  }
}

static uint64_t fnv1a64(const void *p, size_t n) {
  const uint8_t *b = (const uint8_t *)p;
  uint64_t h = 0xcbf29ce484222325ull;
  for (size_t k = 0; k < n; ++k) {
    h ^= b[k];
    h *= 0x100000001b3ull;
  }
  return h;
}

int main(int argc, char **argv) {
  const std::string mode = argc > 1 ? argv[1] : "interpolate_sampled";
  const int width = argc > 3 ? atoi(argv[2]) : 1920;
  const int height = argc > 3 ? atoi(argv[3]) : 1080;
  const int frames = argc > 4 ? atoi(argv[4]) : 1;

  // ---- reference-style caller code --------------------------------------------------------
  OpenCLManager cl_manager;
  if (cl_manager.InitializeContext() != 0) return EXIT_FAILURE;
  SATEncoder sat_encoder(&cl_manager);
  SATDecoder sat_decoder(&cl_manager);

  CodecContext codec = {width, height};
  CodecContext *source_codec_ctx = &codec;
  const int reduced_width = f360_reduced_size(width);    // 16 * ceil(width / 1.8 / 16)
  const int reduced_height = f360_reduced_size(height);
  sat_decoder.InitializeGrid(reduced_width, reduced_height, source_codec_ctx->width,
                             source_codec_ctx->height);

  const int linesize = 4 * width;  // AV_PIX_FMT_RGB0
  const int rect_linesize = 4 * reduced_width;
  std::vector<uint8_t> rgb_frame((size_t)linesize * height);
  std::vector<uint8_t> rect_frame((size_t)rect_linesize * reduced_height, 0xA5);

  const int source_frame_size = linesize * height;
  cl::Buffer cl_source_frame(cl_manager.context, CL_MEM_READ_WRITE, source_frame_size);
  const size_t sat_buffer_size = (size_t)width * height * 3 * sizeof(uint32_t);
  cl::Buffer cl_sat_buffer(cl_manager.context, CL_MEM_READ_WRITE, sat_buffer_size);
  const int cl_output_buffer_size = reduced_height * rect_linesize;
  cl::Buffer cl_output_buffer(cl_manager.context, CL_MEM_READ_WRITE, cl_output_buffer_size);

  uint64_t digest_rect = 0, digest_full = 0, digest_sat = 0;
  if (mode == "encode_sample_frames") {
    // the server loop's three steps (gaze known, EncodeFrameGPU, SampleFrameRectGPU,
    // src/video_server.cc:287-345) for all the frames in ONE call through the class header; the
    // digests are those of the per-frame loop below.  "sat.walk" 1: the read-once encoder -- and
    // with it the one-pass form of this call -- whatever the frame count.
    f360_ctx_set_option(cl_manager.context(), "sat.walk", 1);
    std::vector<cl::Buffer> sources, tables, outputs;
    std::vector<cl_mem> src_ptr, sat_ptr, out_ptr;
    std::vector<float> centers;
    for (int frame = 0; frame < frames; ++frame) {
      lcg_fill(rgb_frame, 12345u + (uint32_t)frame);
      sources.emplace_back(cl_manager.context, CL_MEM_READ_WRITE, source_frame_size);
      tables.emplace_back(cl_manager.context, CL_MEM_READ_WRITE, sat_buffer_size);
      outputs.emplace_back(cl_manager.context, CL_MEM_READ_WRITE, cl_output_buffer_size);
      cl_int ret = cl::copy(cl_manager.command_queue, rgb_frame.data(),
                            rgb_frame.data() + source_frame_size, sources.back());
      ret |= cl::copy(cl_manager.command_queue, rect_frame.data(),
                      rect_frame.data() + cl_output_buffer_size, outputs.back());
      if (ret != CL_SUCCESS) return EXIT_FAILURE;
      centers.push_back(frames == 1 ? 0.5f : 0.25f + 0.5f * frame / frames);
      centers.push_back(0.5f);
    }
    for (int frame = 0; frame < frames; ++frame) {
      src_ptr.push_back(sources[frame]());
      sat_ptr.push_back(tables[frame]());
      out_ptr.push_back(outputs[frame]());
    }
    sat_decoder.EncodeSampleFramesGPU(frames, out_ptr.data(), reduced_width, reduced_height,
                                      rect_linesize, sat_ptr.data(), src_ptr.data(),
                                      source_codec_ctx, linesize, centers.data());
    clFinish(cl_manager.command_queue());
    for (int frame = 0; frame < frames; ++frame) {
      std::vector<uint8_t> out_rect(rect_frame.size());
      if (cl::copy(cl_manager.command_queue, outputs[frame], out_rect.data(),
                   out_rect.data() + cl_output_buffer_size) != CL_SUCCESS)
        return EXIT_FAILURE;
      digest_rect ^= fnv1a64(out_rect.data(), out_rect.size()) + frame;
    }
    std::vector<uint32_t> sat((size_t)width * height * 3);
    cl::copy(cl_manager.command_queue, tables[0], sat.data(), sat.data() + sat.size());
    digest_sat = fnv1a64(sat.data(), sat.size() * sizeof(uint32_t));
  }
  for (int frame = 0; frame < frames && mode != "encode_sample_frames"; ++frame) {
    lcg_fill(rgb_frame, 12345u + (uint32_t)frame);  // VideoDecoder::GetFrame stand-in
    const float center_x = frames == 1 ? 0.5f : 0.25f + 0.5f * frame / frames;
    const float center_y = 0.5f;

    cl_int ret = cl::copy(cl_manager.command_queue, rgb_frame.data(),
                          rgb_frame.data() + source_frame_size, cl_source_frame);
    ret |= cl::copy(cl_manager.command_queue, rect_frame.data(),
                    rect_frame.data() + cl_output_buffer_size, cl_output_buffer);
    if (ret != CL_SUCCESS) {
      std::cerr << "Failed to copy frame to GPU" << std::endl;
      return EXIT_FAILURE;
    }
    if (mode == "planar_expand") {
      // the decoder's yuv420p planes instead of the sws_scale'd RGB0 frame: Y, U, V are the
      // first bytes of the synthetic buffer (tight rows), the RGB0 buffer is only the expand target
      const size_t y_bytes = (size_t)width * height, c_bytes = y_bytes / 4;
      cl::Buffer cl_y(cl_manager.context, CL_MEM_READ_ONLY, y_bytes);
      cl::Buffer cl_u(cl_manager.context, CL_MEM_READ_ONLY, c_bytes);
      cl::Buffer cl_v(cl_manager.context, CL_MEM_READ_ONLY, c_bytes);
      const uint8_t *planes = rgb_frame.data();
      cl::copy(cl_manager.command_queue, planes, planes + y_bytes, cl_y);
      cl::copy(cl_manager.command_queue, planes + y_bytes, planes + y_bytes + c_bytes, cl_u);
      cl::copy(cl_manager.command_queue, planes + y_bytes + c_bytes, planes + y_bytes + 2 * c_bytes,
               cl_v);
      sat_encoder.EncodeFrameYUV420PGPU(cl_sat_buffer(), cl_y(), cl_u(), cl_v(), width, width / 2,
                                        width / 2, width, height);
      clFinish(cl_manager.command_queue());  // the plane buffers go out of scope below
    } else {
      sat_encoder.EncodeFrameGPU(cl_sat_buffer(), cl_source_frame(), width, height, linesize);
    }
    sat_decoder.SampleFrameRectGPU(cl_output_buffer(), reduced_width, reduced_height,
                                   rect_linesize, cl_sat_buffer(), source_codec_ctx, center_x,
                                   center_y);
    if (mode == "foveate_no_encoding")
      sat_decoder.InterpolateFrameRectGPU(cl_source_frame(), width, height, linesize,
                                          cl_output_buffer(), reduced_width, reduced_height,
                                          rect_linesize, center_x, center_y);
    if (mode == "planar_expand")  // debug view: reduced pixels scattered back over the frame
      sat_decoder.ExpandSampledFrameRectGPU(cl_source_frame(), width, height, linesize,
                                            cl_output_buffer(), reduced_width, reduced_height,
                                            rect_linesize, center_x, center_y);
    clFlush(cl_manager.command_queue());
    clFinish(cl_manager.command_queue());
    std::vector<uint8_t> out_rect(rect_frame.size());
    ret = cl::copy(cl_manager.command_queue, cl_output_buffer, out_rect.data(),
                   out_rect.data() + cl_output_buffer_size);
    if (ret != CL_SUCCESS) {
      std::cout << "Failed to copy frame off of GPU" << std::endl;
      return EXIT_FAILURE;
    }
    digest_rect ^= fnv1a64(out_rect.data(), out_rect.size()) + frame;
    if (mode == "foveate_no_encoding" || mode == "planar_expand") {
      ret = cl::copy(cl_manager.command_queue, cl_source_frame, rgb_frame.data(),
                     rgb_frame.data() + source_frame_size);
      digest_full ^= fnv1a64(rgb_frame.data(), rgb_frame.size()) + frame;
    }
    if (frame == 0) {
      std::vector<uint32_t> sat((size_t)width * height * 3);
      cl::copy(cl_manager.command_queue, cl_sat_buffer, sat.data(), sat.data() + sat.size());
      digest_sat = fnv1a64(sat.data(), sat.size() * sizeof(uint32_t));
    }
  }
  // ---- end of reference-style caller code ---------------------------------------------------
  printf("{\"mode\": \"%s\", \"width\": %d, \"height\": %d, \"frames\": %d, \"sat\": \"%016llx\", "
         "\"rect\": \"%016llx\", \"full\": \"%016llx\"}\n",
         mode.c_str(), width, height, frames, (unsigned long long)digest_sat,
         (unsigned long long)digest_rect, (unsigned long long)digest_full);
  return EXIT_SUCCESS;
}
