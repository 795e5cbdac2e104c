// sat_walk.hip -- host side of the read-once batched encoder (sat_walk.h) and its tables-only
// instantiations: f360_sat_encode_batch / _yuv420p_batch with enough frames to fill the device.
#include "sat_walk.h"

using namespace f360::sat;

namespace f360 {
namespace sat {

// Whether a batched call of `count` frames takes the read-once encoder ("sat.walk").
bool walk_wanted(const f360_ctx *ctx, int count, int width) {
  if (ctx->opt_walk == 0 || width > f360::kMaxDim) return false;
  if (ctx->opt_walk == 1) return true;
  const long strips = (width + kStripPx - 1) / kStripPx;
  return (long)count * strips >= ctx->opt_walk_units;
}

// Frames per launch ("sat.walk_frames", 0 = automatic): one workgroup per CU -- one strip owner
// per SIMD -- is the sweet spot (32 frames at 8K: 78.4 us per frame against 82.8 for 64 in one
// launch on the same box: with two owners per SIMD the per-batch jitter that the hand-off
// chain accumulates is five times larger), so a larger call runs as
// several launches of about 1024 units, never fewer units than the call's own frames allow, and
// of equal size (65 frames: 33 + 32, not 64 + 1 -- a launch of one frame would be 30 strip
// owners on an empty device).
// A launch must not exceed the device by a little: 35 frames (1050 units) take 3.8 ms -- the
// 26 units of the second round walk their 480 batches alone -- where 32 take 2.6; and a launch
// costs its serial chain (height / 8 batches of ~3.5 us: 1.7 ms at 8K) however few frames it
// holds, so 40 frames as one oversubscribed launch (4.3 ms) or as two of 20 (4.1 ms) are both
// no better than the three kernels (profiles/round4_few_frames.txt).
int walk_frames_per_launch(const f360_ctx *ctx, int count, int width) {
  const int nstrips = (width + kStripPx - 1) / kStripPx;
  int max_frames = ctx->opt_walk_frames > 0 ? ctx->opt_walk_frames : std::max(1024 / nstrips, 1);
  max_frames = std::min(max_frames, kWalkFrames);
  const int nlaunch = (count + max_frames - 1) / max_frames;
  return (count + nlaunch - 1) / nlaunch;
}

// What every launch of a call shares: the hand-off buffers, the state words, (encode + sample)
// the plan / side buffers, and the kernel arguments but for the launch's own frames.
int walk_prepare(f360_ctx *ctx, int count, const f360::YuvPlanes *yuvs, int width, int height,
                 int linesize, const f360::SatFuse *fuse, WalkSetup &ws) {
  f360::SatEncodePlan &p = ctx->enc;
  const int nstrips = (width + kStripPx - 1) / kStripPx;
  const int nb = (height + kRowUnroll - 1) / kRowUnroll;
  const int per_launch = walk_frames_per_launch(ctx, count, width);
  // Everything below that allocates, clears or synchronises is illegal while the stream is
  // being captured into a hipGraph, and a captured launch keeps the hand-off buffer's address:
  // warm up eagerly with the largest geometry and frame count first.
  const size_t gran_bytes = (size_t)per_launch * nstrips * nb * kWalkLanes * 8;
  const size_t chain_bytes = gran_bytes + (size_t)per_launch * nstrips * 64;
  // encode + sample: a row plan per frame of a launch (one word per table row, whole batches)
  const int plan_stride = nb * kRowUnroll;
  const int pmax = (ctx->opt_fuse_force & 1) ? 1 : std::max(1, std::min(3 * (nstrips - 1), kFixCols));
  const size_t side_stride = fuse ? (size_t)fuse->out_h * pmax * 6 : 0;  // dwords per frame
  const size_t plan_words = (size_t)per_launch * plan_stride;
  const size_t plan_bytes =
      fuse ? (plan_words + (size_t)per_launch * kSpixWords + (size_t)per_launch * side_stride) * 4
           : 0;
  if (!p.walk_state.p || !p.walk_err_host || chain_bytes > p.walk_chain.bytes ||
      plan_bytes > p.walk_plan.bytes) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    F360_HIP_TRY(hipStreamIsCapturing(ctx->stream, &cap));
    F360_REQUIRE(cap == hipStreamCaptureStatusNone,
                 "f360_sat_encode_batch: the read-once encoder must allocate its hand-off buffers "
                 "(%zu bytes) but the stream is being captured; run the same call once before "
                 "the capture", chain_bytes);
  }
  // state words: zero ticket / done, serial 1; the launches advance them
  if (!p.walk_state.p) {
    int st = p.walk_state.reserve(64);
    if (st != F360_OK) return st;
    const WalkState init{0u, 0u, 1ull};
    F360_HIP_TRY(hipMemsetAsync(p.walk_state.p, 0, 64, ctx->stream));
    F360_HIP_TRY(hipMemcpyAsync(p.walk_state.p, &init, sizeof(init), hipMemcpyHostToDevice,
                                ctx->stream));
    F360_HIP_TRY(hipStreamSynchronize(ctx->stream));  // `init` is a stack object
  }
  if (!p.walk_err_host) {
    void *h = nullptr, *d = nullptr;
    F360_HIP_TRY(hipHostMalloc(&h, 64, hipHostMallocMapped));
    *static_cast<uint32_t *>(h) = 0;
    F360_HIP_TRY(hipHostGetDevicePointer(&d, h, 0));
    p.walk_err_host = static_cast<uint32_t *>(h);
    p.walk_err_dev = static_cast<uint32_t *>(d);
  }
  // granules: zeroed when (re)allocated -- a tag is never 0 -- and never again
  // (+ 64 bytes per unit behind the granules: the debug statistics of debug.ablate bit 8)
  if (chain_bytes > p.walk_chain.bytes) {
    if (p.walk_chain.p) F360_HIP_TRY(hipStreamSynchronize(ctx->stream));
    int st = p.walk_chain.reserve(chain_bytes);
    if (st != F360_OK) return st;
    F360_HIP_TRY(hipMemsetAsync(p.walk_chain.p, 0, p.walk_chain.bytes, ctx->stream));
  }
  if (plan_bytes > p.walk_plan.bytes) {
    if (p.walk_plan.p) F360_HIP_TRY(hipStreamSynchronize(ctx->stream));
    int st = p.walk_plan.reserve(plan_bytes);
    if (st != F360_OK) return st;
  }

  EncodeArgs &a = ws.a;
  a = EncodeArgs{};
  a.width = width;
  a.height = height;
  a.linesize = linesize;
  a.bpp = 4;
  a.nstrips = nstrips;
  a.ablate = ctx->opt_ablate;
  a.yuv = yuvs ? yuvs[0] : f360::YuvPlanes{nullptr, nullptr, nullptr, 0, 0, 0};
  if (yuvs)
    f360::build_yuv2rgb_consts(a.k);
  else
    a.k = f360::YuvConsts{};
  a.walk_nbatches = nb;
  a.walk = p.walk_state.as<WalkState>();
  a.walk_chain = p.walk_chain.as<unsigned long long>();
  a.walk_err = p.walk_err_dev;
  a.walk_spin = ctx->opt_walk_spin > 0 ? (uint32_t)ctx->opt_walk_spin : kWalkSpinDefault;
  a.walk_mute = ctx->opt_walk_mute - 1;
  a.walk_stats = reinterpret_cast<unsigned long long *>(p.walk_chain.as<uint8_t>() + gran_bytes);
  p.walk_stats_units = per_launch * nstrips;
  p.walk_stats_offset = gran_bytes;
  ws.nstrips = nstrips;
  ws.nb = nb;
  ws.per_launch = per_launch;
  ws.yuv_src = !yuvs ? 0 : ctx->opt_yuv_model == 1 ? kSrcYuvSwsX86 : kSrcYuvSwsC;
  ws.plan_stride = plan_stride;
  ws.pmax = pmax;
  ws.plan_words = plan_words;
  ws.side_stride = side_stride;
  return F360_OK;
}

// frames [k0, k0 + n) of a call as one launch's by-value argument (unused slots repeat frame k0)
void walk_fill_batch(WalkBatch &wb, int k0, int n, uint32_t *const *sats,
                     const uint8_t *const *srcs, const f360::YuvPlanes *yuvs) {
  for (int k = 0; k < kWalkFrames; ++k) {
    const int q = k0 + (k < n ? k : 0);
    wb.src[k] = yuvs ? yuvs[q].y : srcs[q];
    wb.sat[k] = sats ? sats[q] : nullptr;  // (null: one pass without tables)
    wb.u[k] = yuvs ? yuvs[q].u : nullptr;
    wb.v[k] = yuvs ? yuvs[q].v : nullptr;
  }
}

// f360_sat_encode_batch / _yuv420p_batch on the read-once encoder: launches of up to kWalkFrames
// frames.  The caller has checked the arguments and that every buffer allows 16-byte accesses.
int sat_encode_walk(f360_ctx *ctx, int count, uint32_t *const *sats, const uint8_t *const *srcs,
                    const f360::YuvPlanes *yuvs, int width, int height, int linesize, bool prof) {
  WalkSetup ws;
  int st = walk_prepare(ctx, count, yuvs, width, height, linesize, nullptr, ws);
  if (st != F360_OK) return st;
  EncodeArgs &a = ws.a;
  for (int k0 = 0; k0 < count; k0 += ws.per_launch) {
    const int n = std::min(count - k0, ws.per_launch);
    WalkBatch wb;
    walk_fill_batch(wb, k0, n, sats, srcs, yuvs);
    a.walk_units = n * ws.nstrips;
    const dim3 grid((a.walk_units + kWalkWaves - 1) / kWalkWaves);
    const dim3 block(64 * kWalkWaves);
    f360::KernelSpan span(ctx, f360::kSatWalk, prof, n);
#define F360_WALK_LAUNCH(SRC)                                                                   \
  hipLaunchKernelGGL((sat_walk_kernel<SRC, 2>), grid, block, 0, ctx->stream, a, wb, WalkNoFuse{})
    if (ws.yuv_src == kSrcYuvSwsX86)
      F360_WALK_LAUNCH(kSrcYuvSwsX86);
    else if (ws.yuv_src == kSrcYuvSwsC)
      F360_WALK_LAUNCH(kSrcYuvSwsC);
    else
      F360_WALK_LAUNCH(kSrcRgb0);
#undef F360_WALK_LAUNCH
  }
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}

}  // namespace sat
}  // namespace f360

// Debug: the per-unit statistics of the last read-once launch that ran with debug.ablate bit 8
// ({start, end} in 100 MHz ticks, slow-path waits | shader cycles << 16, polls spent waiting); returns the unit count.
extern "C" int f360_debug_walk_stats(f360_ctx *ctx, unsigned long long *out, int max_units) {
  F360_REQUIRE(ctx && out && max_units >= 0, "f360_debug_walk_stats: bad argument");
  F360_BIND_DEVICE(ctx);
  const f360::SatEncodePlan &p = ctx->enc;
  const int n = std::min(max_units, p.walk_stats_units);
  if (n <= 0 || !p.walk_chain.p) return 0;
  F360_HIP_TRY(hipStreamSynchronize(ctx->stream));
  F360_HIP_TRY(hipMemcpy(out, p.walk_chain.as<uint8_t>() + p.walk_stats_offset, (size_t)n * 64,
                         hipMemcpyDeviceToHost));
  return n;
}

// Strips of read-once launches that gave up waiting for their hand-off and finished alone (the
// tables are right either way) since the last call of this function; blocks until the stream
// has drained.
extern "C" int f360_debug_walk_recoveries(f360_ctx *ctx, unsigned *count_out);
extern "C" int f360_ctx_handoff_recoveries(f360_ctx *ctx, unsigned *count_out) {
  return f360_debug_walk_recoveries(ctx, count_out);
}
extern "C" int f360_debug_walk_recoveries(f360_ctx *ctx, unsigned *count_out) {
  F360_REQUIRE(ctx && count_out, "f360_debug_walk_recoveries: bad argument");
  F360_BIND_DEVICE(ctx);
  *count_out = 0;
  f360::SatEncodePlan &p = ctx->enc;
  if (!p.walk_err_host) return F360_OK;
  F360_HIP_TRY(hipStreamSynchronize(ctx->stream));
  *count_out = *p.walk_err_host;
  *p.walk_err_host = 0;
  return F360_OK;
}
