// send_frame_loop_synth.cc -- the reference's per-client streaming loop on the HIP engine.
//
// Mirrors VideoServer::InitializeConnectionData + SendFrameLoop
// (/root/reference/src/video_server.cc:45-86,197-427): one thread and one
// OpenCLManager / SATEncoder / SATDecoder per client; per tick
//     get frame -> cl::copy H2D -> EncodeFrameGPU -> clFinish
//     -> sleep until the tick -> read the LATEST gaze -> SampleFrameRectGPU -> cl::copy D2H
//     -> encoder sink
// with a synthetic source (pre-staged pinned frames instead of VideoDecoder), the gaze
// coming from a GazeViewPoints trace (instead of websocket messages) and a null sink
// (instead of NVENC + fMP4 + websocket).  Reports per-frame busy latency (everything but
// the sleep) and aggregate input Mpixels/s: BASELINE.json config 5.
//
//   ./send_frame_loop_synth <clients> <fps> <frames> <width> <height> [trace.txt] [gpus] [rgb0|yuv420p in] [rgb0|yuv420p out] [plan]
//                           [rgb0|yuv420p (source layout)] [rgb0|yuv420p (delivered layout)]
// client c runs on GPU c % gpus.  Prints one JSON line.  With "yuv420p" as the delivered layout
// the reduced frame is converted on the device (f360_rgb0_to_yuv420p, the sws_scale of
// VideoEncoder::EncodeFrame, src/video_encoder.cc:380-395) and 1.5 instead of 4 bytes per pixel
// come back over PCIe; the digest then covers the three planes.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "f360/gaze_view_points.h"
#include "f360/parameters.h"
#include "f360/sat_decoder.h"
#include "f360/sat_encoder.h"

struct CodecContext {
  int width, height;
};

struct ClientResult {
  std::vector<double> latency_ms;
  double loop_s = 0;  // first tick .. last frame delivered (set-up excluded)
  uint64_t last_digest = 0;
  float last_gaze[2] = {0, 0};
  int frames = 0;
};

static uint64_t fnv1a64(const void *p, size_t n) {
  const uint8_t *b = (const uint8_t *)p;
  uint64_t h = 0xcbf29ce484222325ull;
  for (size_t k = 0; k < n; ++k) {
    h ^= b[k];
    h *= 0x100000001b3ull;
  }
  return h;
}

static void client_loop(int client, int device, double fps, int frames, int width, int height,
                        const GazeViewPoints *trace, bool planar, bool planar_out,
                        ClientResult *out) {
  using clock = std::chrono::high_resolution_clock;
  // ---- InitializeConnectionData (video_server.cc:62-66) --------------------------------
  OpenCLManager cl_manager;
  cl_manager.device_index = device;
  if (cl_manager.InitializeContext() != 0) return;
  SATEncoder sat_encoder(&cl_manager);
  SATDecoder sat_decoder(&cl_manager);
  CodecContext codec = {width, height};
  const int out_w = f360_reduced_size(width), out_h = f360_reduced_size(height);
  sat_decoder.InitializeGrid(out_w, out_h, width, height);

  // ---- SendFrameLoop buffers (video_server.cc:224-232) -----------------------------------
  const int linesize = 4 * width, out_linesize = 4 * out_w;
  // planar: the decoder's yuv420p frame (Y, U, V back to back, tight rows) is uploaded as it is --
  // 1.5 instead of 4 bytes per pixel over PCIe -- and converted inside the encode kernels
  const size_t y_bytes = (size_t)width * height, c_bytes = y_bytes / 4;
  const size_t cl_source_frame_size = planar ? y_bytes + 2 * c_bytes : (size_t)linesize * height;
  cl::Buffer cl_source_frame(cl_manager.context, CL_MEM_READ_WRITE, cl_source_frame_size);
  cl::Buffer cl_sat_buffer(cl_manager.context, CL_MEM_READ_WRITE,
                           (size_t)3 * width * height * sizeof(uint32_t));
  const size_t cl_output_buffer_size = (size_t)out_linesize * out_h;
  cl::Buffer cl_output_buffer(cl_manager.context, CL_MEM_READ_WRITE, cl_output_buffer_size);
  // delivered as planes: Y, U, V back to back, tight rows
  const size_t oy_bytes = (size_t)out_w * out_h, oc_bytes = oy_bytes / 4;
  const size_t delivered_size = planar_out ? oy_bytes + 2 * oc_bytes : cl_output_buffer_size;
  cl::Buffer cl_yuv_buffer(cl_manager.context, CL_MEM_READ_WRITE, planar_out ? delivered_size : 16);

  // synthetic decoder: a few pre-staged pinned frames, cycled
  const int pool = 3;
  uint8_t *staged[pool];
  uint8_t *output_frame = nullptr;
  for (int k = 0; k < pool; ++k) {
    if (f360_host_alloc_pinned(cl_source_frame_size, (void **)&staged[k]) != F360_OK) return;
    uint32_t s = 12345u + 1000u * client + k;
    for (size_t b = 0; b < cl_source_frame_size; ++b) {
      s = s * 1664525u + 1013904223u;
      staged[k][b] = (uint8_t)(s >> 24);
    }
  }
  if (f360_host_alloc_pinned(cl_output_buffer_size, (void **)&output_frame) != F360_OK) return;
  std::memset(output_frame, 0, cl_output_buffer_size);
  cl::copy(cl_manager.command_queue, output_frame, output_frame + cl_output_buffer_size,
           cl_output_buffer);
  std::memset(output_frame, 0, delivered_size);

  auto checkpoint_time = clock::now();
  const auto loop_start = checkpoint_time;
  const double tick_ms = 1000.0 / fps;
  for (int frame_number = 0; frame_number < frames; ++frame_number) {
    const auto t0 = clock::now();
    uint8_t *rgb = staged[frame_number % pool];  // video_decoder->GetFrame(rgb_frame, RGB0)
    cl_int ret = cl::copy(cl_manager.command_queue, rgb, rgb + cl_source_frame_size, cl_source_frame);
    if (planar) {
      uint8_t *base = static_cast<uint8_t *>(cl_source_frame());
      sat_encoder.EncodeFrameYUV420PGPU(cl_sat_buffer(), base, base + y_bytes,
                                        base + y_bytes + c_bytes, width, width / 2, width / 2,
                                        width, height);
    } else {
      sat_encoder.EncodeFrameGPU(cl_sat_buffer(), cl_source_frame(), width, height, linesize);
    }
    clFlush(cl_manager.command_queue());
    clFinish(cl_manager.command_queue());
    const auto t1 = clock::now();

    // Sleep until we're ready (video_server.cc:310-318)
    const double since =
        std::chrono::duration<double, std::milli>(clock::now() - checkpoint_time).count();
    if (tick_ms - since > 0)
      std::this_thread::sleep_for(std::chrono::duration<double, std::milli>(tick_ms - since));

    // Done sleeping. Grab the latest gaze position (video_server.cc:325-328)
    float center_x = 0.5f, center_y = 0.5f;
    if (trace && !trace->points.empty()) {
      const auto &p = trace->points[(size_t)(frame_number + 17 * client) % trace->points.size()];
      center_x = p.gaze_point[0];
      center_y = p.gaze_point[1];
    }
    checkpoint_time = clock::now();
    const auto t2 = clock::now();
    sat_decoder.SampleFrameRectGPU(cl_output_buffer(), out_w, out_h, out_linesize, cl_sat_buffer(),
                                   &codec, center_x, center_y);
    if (planar_out) {
      uint8_t *yb = static_cast<uint8_t *>(cl_yuv_buffer());
      if (f360_rgb0_to_yuv420p(cl_manager.command_queue.ctx(), yb, yb + oy_bytes,
                               yb + oy_bytes + oc_bytes, out_w, out_w / 2, out_w / 2,
                               static_cast<const uint8_t *>(cl_output_buffer()), out_linesize,
                               out_w, out_h) != F360_OK) {
        std::cerr << "f360_rgb0_to_yuv420p: " << f360_last_error_string() << std::endl;
        exit(EXIT_FAILURE);
      }
      ret |= cl::copy(cl_manager.command_queue, cl_yuv_buffer, output_frame,
                      output_frame + delivered_size);
    } else {
      ret |= cl::copy(cl_manager.command_queue, cl_output_buffer, output_frame,
                      output_frame + cl_output_buffer_size);
    }
    const auto t3 = clock::now();
    if (ret != CL_SUCCESS) {
      std::cerr << "Failed to copy output frame out. " << ret << std::endl;
      exit(EXIT_FAILURE);
    }
    // video_encoder->EncodeFrame(...) / websocket send: null sink
    out->latency_ms.push_back(std::chrono::duration<double, std::milli>((t1 - t0) + (t3 - t2)).count());
    out->last_gaze[0] = center_x;
    out->last_gaze[1] = center_y;
    ++out->frames;
  }
  out->loop_s = std::chrono::duration<double>(clock::now() - loop_start).count();
  out->last_digest = fnv1a64(output_frame, delivered_size);
  for (int k = 0; k < pool; ++k) f360_host_free_pinned(staged[k]);
  f360_host_free_pinned(output_frame);
}

int main(int argc, char **argv) {
  const int clients = argc > 1 ? atoi(argv[1]) : 8;
  const double fps = argc > 2 ? atof(argv[2]) : 60.0;
  const int frames = argc > 3 ? atoi(argv[3]) : 120;
  const int width = argc > 4 ? atoi(argv[4]) : 7680;
  const int height = argc > 5 ? atoi(argv[5]) : 3840;
  const std::string trace_path = argc > 6 ? argv[6] : "";
  int gpus = argc > 7 ? atoi(argv[7]) : 0;
  const bool planar = argc > 8 && std::string(argv[8]) == "yuv420p";
  const bool planar_out = argc > 9 && std::string(argv[9]) == "yuv420p";
  // "plan" as the 10th argument: print which device every client's thread would open and
  // what it would allocate there, without touching a GPU (the 8-GPU form of config 5 --
  // client c <-> GPU c, src/video_server.cc:62-66 -- can be checked on a box without eight)
  const bool plan_only = argc > 10 && std::string(argv[10]) == "plan";
  if (plan_only) {
    if (gpus <= 0) {
      std::cerr << "plan: the GPU count must be given explicitly" << std::endl;
      return EXIT_FAILURE;
    }
    const int rw = 16 * (int)std::ceil(width / 1.8 / 16), rh = 16 * (int)std::ceil(height / 1.8 / 16);
    const size_t in_bytes = planar ? (size_t)width * height * 3 / 2 : (size_t)width * height * 4;
    const size_t out_bytes = planar_out ? (size_t)rw * rh * 3 / 2 : (size_t)rw * rh * 4;
    printf("{\"plan\": true, \"clients\": %d, \"gpus\": %d, \"client_device\": [", clients, gpus);
    for (int c = 0; c < clients; ++c) printf("%s%d", c ? ", " : "", c % gpus);
    printf("], \"clients_per_device\": [");
    for (int g = 0; g < gpus; ++g)
      printf("%s%d", g ? ", " : "", clients / gpus + (g < clients % gpus ? 1 : 0));
    printf("], \"reduced\": [%d, %d], \"device_bytes_per_client\": %zu, "
           "\"upload_bytes_per_frame\": %zu, \"download_bytes_per_frame\": %zu}\n",
           rw, rh, in_bytes + (size_t)width * height * 12 + (size_t)rw * rh * 4 + out_bytes, in_bytes,
           out_bytes);
    return EXIT_SUCCESS;
  }
  if (gpus <= 0 && f360_device_count(&gpus) != F360_OK) {
    std::cerr << f360_last_error_string() << std::endl;
    return EXIT_FAILURE;
  }
  GazeViewPoints trace;
  if (!trace_path.empty()) trace = GazeViewPoints(trace_path);

  std::vector<ClientResult> results((size_t)clients);
  std::vector<std::thread> threads;
  const auto t0 = std::chrono::high_resolution_clock::now();
  for (int c = 0; c < clients; ++c)
    threads.emplace_back(client_loop, c, c % gpus, fps, frames, width, height,
                         trace.points.empty() ? nullptr : &trace, planar, planar_out,
                         &results[(size_t)c]);
  for (auto &t : threads) t.join();
  const double wall_s =
      std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();

  std::vector<double> all;
  long total_frames = 0;
  double fps_sum = 0, loop_max = 0;
  for (const auto &r : results) {
    if (r.loop_s > 0) fps_sum += r.frames / r.loop_s;
    loop_max = std::max(loop_max, r.loop_s);
    // skip each client's first frame (context warm-up)
    for (size_t k = 1; k < r.latency_ms.size(); ++k) all.push_back(r.latency_ms[k]);
    total_frames += r.frames;
  }
  std::sort(all.begin(), all.end());
  auto pct = [&](double q) { return all.empty() ? 0.0 : all[(size_t)std::min<double>(all.size() - 1, q * all.size())]; };
  printf("{\"source\": \"%s\", \"delivered\": \"%s\", \"clients\": %d, \"gpus\": %d, \"fps_target\": %.1f, \"frames_per_client\": %d, \"width\": %d, "
         "\"height\": %d, \"wall_s_with_setup\": %.3f, \"fps_achieved_per_client\": %.2f, \"mpix_per_s\": %.1f, "
         "\"latency_ms_p50\": %.3f, \"latency_ms_p99\": %.3f, \"latency_ms_max\": %.3f, "
         "\"client0_last_gaze\": [%.9g, %.9g], \"client0_last_digest\": \"%016llx\", \"last_digests\": [",
         planar ? "yuv420p" : "rgb0", planar_out ? "yuv420p" : "rgb0", clients, gpus, fps, frames, width,
         height, wall_s,
         fps_sum / clients,
         loop_max > 0 ? (double)total_frames * width * height / 1e6 / loop_max : 0.0, pct(0.50), pct(0.99),
         all.empty() ? 0.0 : all.back(), results[0].last_gaze[0], results[0].last_gaze[1],
         (unsigned long long)results[0].last_digest);
  // every client's delivered bytes (digest of its last output buffer), for the parity test
  for (size_t c = 0; c < results.size(); ++c)
    printf("%s\"%016llx\"", c ? ", " : "", (unsigned long long)results[c].last_digest);
  printf("]}\n");
  return EXIT_SUCCESS;
}
