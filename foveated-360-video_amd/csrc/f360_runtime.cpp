// f360_runtime.cpp -- context / buffers / events: the part of the C ABI that
// replaces the reference's OpenCLManager (src/opencl_manager.{h,cc}) and the
// cl::Buffer / cl::copy / clFinish calls its callers make directly
// (src/video_server.cc:224-232,298-303,342-345).
#include <cstring>
#include <string>

#include "f360_internal.h"

namespace f360 {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int DevBuf::reserve(size_t n) {
  if (n <= bytes && p) return F360_OK;
  release();
  F360_HIP_TRY(hipMalloc(&p, n));
  bytes = n;
  return F360_OK;
}

void DevBuf::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  bytes = 0;
}

}  // namespace f360

using f360::set_error;

int f360::side_stream(f360_ctx *ctx) {
  if (ctx->side) return F360_OK;
  {  // (creating a stream and events is not something to do inside a capture)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    F360_HIP_TRY(hipStreamIsCapturing(ctx->stream, &cap));
    F360_REQUIRE(cap == hipStreamCaptureStatusNone,
                 "the context's side stream must be created but the stream is being captured; "
                 "run the same call once before the capture");
  }
  F360_HIP_TRY(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
  F360_HIP_TRY(hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming));
  F360_HIP_TRY(hipEventCreateWithFlags(&ctx->side_join, hipEventDisableTiming));
  return F360_OK;
}

extern "C" {

int f360_version(void) { return F360_VERSION_MAJOR * 100 + F360_VERSION_MINOR; }

const char *f360_last_error_string(void) { return f360::g_err; }

const char *f360_status_string(int status) {
  switch (status) {
    case F360_OK: return "F360_OK";
    case F360_ERR_INVALID_ARG: return "F360_ERR_INVALID_ARG";
    case F360_ERR_NO_DEVICE: return "F360_ERR_NO_DEVICE";
    case F360_ERR_HIP: return "F360_ERR_HIP";
    case F360_ERR_OOM: return "F360_ERR_OOM";
    case F360_ERR_NOT_INITIALIZED: return "F360_ERR_NOT_INITIALIZED";
    default: return "F360_ERR_UNKNOWN";
  }
}

int f360_device_count(int *count) {
  F360_REQUIRE(count, "f360_device_count: null output");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    *count = 0;
    set_error("no HIP device visible (%s)", hipGetErrorString(e));
    return F360_ERR_NO_DEVICE;
  }
  *count = n;
  return F360_OK;
}

static int ctx_create_common(int device, void *stream, bool borrow,
                             f360_ctx **out) {
  F360_REQUIRE(out, "f360_ctx_create: null output");
  *out = nullptr;
  int n = 0;
  int st = f360_device_count(&n);
  if (st != F360_OK) return st;
  F360_REQUIRE(device >= 0 && device < n, "f360_ctx_create: device %d of %d",
               device, n);
  f360::DeviceGuard guard(device);  // the caller's current device is left as it was
  F360_HIP_TRY(guard.status());
  f360_ctx *ctx = new f360_ctx();
  ctx->device = device;
  if (borrow) {
    ctx->stream = static_cast<hipStream_t>(stream);
    ctx->owns_stream = false;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
      delete ctx;
      return F360_ERR_HIP;
    }
    ctx->owns_stream = true;
  }
  *out = ctx;
  return F360_OK;
}

int f360_ctx_create(int device, f360_ctx **out) {
  return ctx_create_common(device, nullptr, false, out);
}

int f360_ctx_create_on_stream(int device, void *hip_stream, f360_ctx **out) {
  return ctx_create_common(device, hip_stream, true, out);
}

int f360_ctx_destroy(f360_ctx *ctx) {
  if (!ctx) return F360_OK;
  f360::DeviceGuard guard(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ctx->enc.ws.release();
  ctx->enc.walk_chain.release();
  ctx->enc.walk_state.release();
  ctx->enc.walk_plan.release();
  if (ctx->enc.walk_err_host) (void)hipHostFree(ctx->enc.walk_err_host);
  ctx->ex_tables.release();
  ctx->gn_table.release();
  ctx->gn_gtab.release();
  ctx->gn_counters.release();
  ctx->ex_keys.release();
  for (const f360::ProfSpan &s : ctx->prof_pending) {
    (void)hipEventDestroy(s.a);
    (void)hipEventDestroy(s.b);
  }
  for (hipEvent_t e : ctx->prof_free) (void)hipEventDestroy(e);
  if (ctx->side) {
    (void)hipStreamSynchronize(ctx->side);
    (void)hipStreamDestroy(ctx->side);
  }
  if (ctx->side_fork) (void)hipEventDestroy(ctx->side_fork);
  if (ctx->side_join) (void)hipEventDestroy(ctx->side_join);
  if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return F360_OK;
}

int f360_ctx_device(const f360_ctx *ctx, int *device) {
  F360_REQUIRE(ctx && device, "f360_ctx_device: null argument");
  *device = ctx->device;
  return F360_OK;
}

int f360_ctx_stream(const f360_ctx *ctx, void **hip_stream) {
  F360_REQUIRE(ctx && hip_stream, "f360_ctx_stream: null argument");
  *hip_stream = static_cast<void *>(ctx->stream);
  return F360_OK;
}

int f360_sync(f360_ctx *ctx) {
  F360_REQUIRE(ctx, "f360_sync: null context");
  F360_BIND_DEVICE(ctx);
  F360_HIP_TRY(hipStreamSynchronize(ctx->stream));
  // (a strip of the read-once encoder whose hand-off wait timed out finishes alone and exactly:
  // nothing to report here; f360_debug_walk_recoveries counts them)
  return F360_OK;
}

int f360_malloc(f360_ctx *ctx, size_t bytes, void **dptr) {
  F360_REQUIRE(ctx && dptr, "f360_malloc: null argument");
  *dptr = nullptr;
  F360_BIND_DEVICE(ctx);
  F360_HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
  return F360_OK;
}

int f360_free(f360_ctx *ctx, void *dptr) {
  F360_REQUIRE(ctx, "f360_free: null context");
  if (!dptr) return F360_OK;
  F360_BIND_DEVICE(ctx);
  F360_HIP_TRY(hipStreamSynchronize(ctx->stream));
  F360_HIP_TRY(hipFree(dptr));
  return F360_OK;
}

int f360_memset(f360_ctx *ctx, void *dptr, int value, size_t bytes) {
  F360_REQUIRE(ctx && dptr, "f360_memset: null argument");
  F360_BIND_DEVICE(ctx);
  F360_HIP_TRY(hipMemsetAsync(dptr, value, bytes, ctx->stream));
  return F360_OK;
}

int f360_memcpy_h2d_async(f360_ctx *ctx, void *dst_dev, const void *src_host,
                          size_t bytes) {
  F360_REQUIRE(ctx && dst_dev && src_host, "f360_memcpy_h2d: null argument");
  F360_BIND_DEVICE(ctx);
  F360_HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice,
                              ctx->stream));
  return F360_OK;
}

int f360_memcpy_d2h_async(f360_ctx *ctx, void *dst_host, const void *src_dev,
                          size_t bytes) {
  F360_REQUIRE(ctx && dst_host && src_dev, "f360_memcpy_d2h: null argument");
  F360_BIND_DEVICE(ctx);
  F360_HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost,
                              ctx->stream));
  return F360_OK;
}

int f360_memcpy_h2d(f360_ctx *ctx, void *dst_dev, const void *src_host,
                    size_t bytes) {
  int st = f360_memcpy_h2d_async(ctx, dst_dev, src_host, bytes);
  if (st != F360_OK) return st;
  return f360_sync(ctx);
}

int f360_memcpy_d2h(f360_ctx *ctx, void *dst_host, const void *src_dev,
                    size_t bytes) {
  int st = f360_memcpy_d2h_async(ctx, dst_host, src_dev, bytes);
  if (st != F360_OK) return st;
  return f360_sync(ctx);
}

int f360_host_alloc_pinned(size_t bytes, void **hptr) {
  F360_REQUIRE(hptr, "f360_host_alloc_pinned: null output");
  F360_HIP_TRY(hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault));
  return F360_OK;
}

int f360_host_free_pinned(void *hptr) {
  if (!hptr) return F360_OK;
  F360_HIP_TRY(hipHostFree(hptr));
  return F360_OK;
}

int f360_event_create(f360_ctx *ctx, f360_event **ev) {
  F360_REQUIRE(ctx && ev, "f360_event_create: null argument");
  F360_BIND_DEVICE(ctx);
  f360_event *e = new f360_event();
  hipError_t r = hipEventCreate(&e->ev);
  if (r != hipSuccess) {
    set_error("hipEventCreate failed: %s", hipGetErrorString(r));
    delete e;
    return F360_ERR_HIP;
  }
  *ev = e;
  return F360_OK;
}

int f360_event_destroy(f360_event *ev) {
  if (!ev) return F360_OK;
  (void)hipEventDestroy(ev->ev);
  delete ev;
  return F360_OK;
}

int f360_event_record(f360_ctx *ctx, f360_event *ev) {
  F360_REQUIRE(ctx && ev, "f360_event_record: null argument");
  F360_BIND_DEVICE(ctx);
  F360_HIP_TRY(hipEventRecord(ev->ev, ctx->stream));
  return F360_OK;
}

int f360_event_elapsed_ms(f360_event *start, f360_event *stop, float *ms) {
  F360_REQUIRE(start && stop && ms, "f360_event_elapsed_ms: null argument");
  F360_HIP_TRY(hipEventSynchronize(stop->ev));
  F360_HIP_TRY(hipEventElapsedTime(ms, start->ev, stop->ev));
  return F360_OK;
}

struct OptionSlot {
  const char *key;
  int f360_ctx::*field;
};
static const OptionSlot kOptions[] = {
    {"sat.band_rows", &f360_ctx::opt_band_rows},
    {"sat.sb_bands", &f360_ctx::opt_sb_bands},
    {"sample.variant", &f360_ctx::opt_sample_variant},
    {"sample.rows", &f360_ctx::opt_walk_rows},
    {"sample.srows", &f360_ctx::opt_stream_rows},
    {"sat.batch_mb", &f360_ctx::opt_batch_mb},
    {"sample.fpl", &f360_ctx::opt_sample_fpl},
    {"sat.walk", &f360_ctx::opt_walk},
    {"sat.walk_units", &f360_ctx::opt_walk_units},
    {"sat.walk_frames", &f360_ctx::opt_walk_frames},
    {"debug.ablate", &f360_ctx::opt_ablate},
    {"debug.walk_spin", &f360_ctx::opt_walk_spin},
    {"debug.walk_mute", &f360_ctx::opt_walk_mute},
    {"interp.rows", &f360_ctx::opt_interp_rows},
    {"interp.staged", &f360_ctx::opt_interp_staged},
    {"yuv.model", &f360_ctx::opt_yuv_model},
    {"yuv.r2y_rows", &f360_ctx::opt_r2y_rows},
    {"fov.piggyback", &f360_ctx::opt_fov_piggyback},
    {"fuse.walk", &f360_ctx::opt_fuse_walk},
    {"fuse.band", &f360_ctx::opt_fuse_band},
    {"sat.pipeline", &f360_ctx::opt_pipeline},
    {"sat.pool_mb", &f360_ctx::opt_pool_mb},
    {"debug.fuse_force", &f360_ctx::opt_fuse_force},
    {"gnomonic.table", &f360_ctx::opt_gnomonic_table},
    {"gnomonic.guard", &f360_ctx::opt_gnomonic_guard},
    {"is.lp_table", &f360_ctx::opt_lp_table},
    {"is.lp_lds", &f360_ctx::opt_lp_lds},
    {"is.xcd_bands", &f360_ctx::opt_xcd_bands},
};

int f360_ctx_set_option(f360_ctx *ctx, const char *key, int value) {
  F360_REQUIRE(ctx && key, "f360_ctx_set_option: null argument");
  for (const OptionSlot &s : kOptions)
    if (std::strcmp(s.key, key) == 0) {
      if (s.field == &f360_ctx::opt_band_rows)
        F360_REQUIRE(value == 0 || value == 8 || value == 16 || value == 32 || value == 64,
                     "sat.band_rows must be 0 (automatic), 8, 16, 32 or 64 (got %d)", value);
      if (s.field == &f360_ctx::opt_sb_bands)
        F360_REQUIRE(value >= -1 && value <= 64, "sat.sb_bands out of range: %d",
                     value);
      if (s.field == &f360_ctx::opt_yuv_model)
        F360_REQUIRE(value == 0 || value == 1,
                     "yuv.model must be 0 (libswscale C tables) or 1 (libswscale x86): %d", value);
      if (s.field == &f360_ctx::opt_r2y_rows)
        F360_REQUIRE(value >= -1 && value <= 4096, "yuv.r2y_rows out of range -1..4096: %d", value);
      if (s.field == &f360_ctx::opt_walk_rows)
        F360_REQUIRE(value >= 1 && value <= 4096, "sample.rows out of range: %d", value);
      if (s.field == &f360_ctx::opt_stream_rows)
        F360_REQUIRE(value >= 0 && value <= 64, "sample.srows out of range 0..64: %d", value);
      if (s.field == &f360_ctx::opt_walk)
        F360_REQUIRE(value >= -1 && value <= 1, "sat.walk must be -1 (automatic), 0 or 1: %d", value);
      if (s.field == &f360_ctx::opt_walk_frames)
        F360_REQUIRE(value >= 0 && value <= 64, "sat.walk_frames out of range 0..64: %d", value);
      if (s.field == &f360_ctx::opt_walk_units)
        F360_REQUIRE(value >= 1, "sat.walk_units must be >= 1: %d", value);
      if (s.field == &f360_ctx::opt_sample_fpl)
        F360_REQUIRE(value >= 1 && value <= 64, "sample.fpl out of range 1..64: %d", value);
      if (s.field == &f360_ctx::opt_xcd_bands)
        F360_REQUIRE(value >= 0 && value <= 2, "is.xcd_bands must be 0, 1 (automatic) or 2: %d", value);
      if (s.field == &f360_ctx::opt_lp_lds)
        F360_REQUIRE(value == 0 || value == 1 || value == 256 || value == 512 || value == 1024,
                     "is.lp_lds must be 0 (off), 1 (on) or a workgroup size 256 / 512 / 1024: %d",
                     value);
      if (s.field == &f360_ctx::opt_sample_variant)
        F360_REQUIRE(value >= 0 && value <= 2, "sample.variant must be 0, 1 or 2: %d", value);
      ctx->*(s.field) = value;
      return F360_OK;
    }
  set_error("unknown option '%s'", key);
  return F360_ERR_INVALID_ARG;
}

int f360_ctx_get_option(const f360_ctx *ctx, const char *key, int *value) {
  F360_REQUIRE(ctx && key && value, "f360_ctx_get_option: null argument");
  for (const OptionSlot &s : kOptions)
    if (std::strcmp(s.key, key) == 0) {
      *value = ctx->*(s.field);
      return F360_OK;
    }
  set_error("unknown option '%s'", key);
  return F360_ERR_INVALID_ARG;
}

static const char *const kKernelNames[f360::kKernelCount] = {
    "sat_reduce_kernel",        "sat_carry_kernel",       "sat_write_kernel",
    "sat_walk_kernel",
    "sample_rect_kernel",       "interpolate_rect_kernel", "decode_kernel",
    "is_sample_rect_kernel",    "is_sample_logpolar_kernel",
    "is_interpolate_logpolar_kernel", "is_blur_kernel",   "gnomonic_kernel",
    "foveate_maps_kernel",      "sample_compact_kernel",  "yuv420p_to_rgb0_kernel",
    "rgb0_to_yuv420p_kernel",   "expand_kernel",          "walk_fuse_plan_kernel",
    "walk_fuse_fix_kernel",
    "sat_write_fuse_kernel"};

int f360_kernel_count(void) { return f360::kKernelCount; }

const char *f360_kernel_name(int kernel_id) {
  if (kernel_id < 0 || kernel_id >= f360::kKernelCount) return "";
  return kKernelNames[kernel_id];
}

int f360_ctx_profile_arm(f360_ctx *ctx, int calls) {
  F360_REQUIRE(ctx && calls >= 0, "f360_ctx_profile_arm: bad argument");
  F360_BIND_DEVICE(ctx);
  ctx->prof_armed = calls;
  return F360_OK;
}

static int profile_collect(f360_ctx *ctx) {
  for (const f360::ProfSpan &s : ctx->prof_pending) {
    F360_HIP_TRY(hipEventSynchronize(s.b));
    float ms = 0.0f;
    F360_HIP_TRY(hipEventElapsedTime(&ms, s.a, s.b));
    ctx->prof_ms[s.kid] += (double)ms;
    ctx->prof_launches[s.kid] += 1;
    ctx->prof_frames[s.kid] += s.frames;
    ctx->prof_free.push_back(s.a);
    ctx->prof_free.push_back(s.b);
  }
  ctx->prof_pending.clear();
  return F360_OK;
}

int f360_ctx_profile_read(f360_ctx *ctx, int kernel_id, double *total_ms,
                          int *launches) {
  F360_REQUIRE(ctx && total_ms && launches && kernel_id >= 0 &&
                   kernel_id < f360::kKernelCount,
               "f360_ctx_profile_read: bad argument");
  F360_BIND_DEVICE(ctx);
  int st = profile_collect(ctx);
  if (st != F360_OK) return st;
  *total_ms = ctx->prof_ms[kernel_id];
  *launches = ctx->prof_launches[kernel_id];
  return F360_OK;
}

int f360_ctx_profile_frames(f360_ctx *ctx, int kernel_id, int *frames) {
  F360_REQUIRE(ctx && frames && kernel_id >= 0 && kernel_id < f360::kKernelCount,
               "f360_ctx_profile_frames: bad argument");
  F360_BIND_DEVICE(ctx);
  int st = profile_collect(ctx);
  if (st != F360_OK) return st;
  *frames = ctx->prof_frames[kernel_id];
  return F360_OK;
}

int f360_ctx_profile_reset(f360_ctx *ctx) {
  F360_REQUIRE(ctx, "f360_ctx_profile_reset: null context");
  F360_BIND_DEVICE(ctx);
  int st = profile_collect(ctx);
  if (st != F360_OK) return st;
  for (int k = 0; k < f360::kKernelCount; ++k) {
    ctx->prof_ms[k] = 0.0;
    ctx->prof_launches[k] = 0;
    ctx->prof_frames[k] = 0;
  }
  return F360_OK;
}

}  // extern "C"
