// f360_internal.h -- shared declarations of the HIP engine behind include/f360.h.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <initializer_list>
#include <vector>

#include "f360.h"
#include "yuv_device.h"

namespace f360 {

void set_error(const char *fmt, ...);

#define F360_HIP_TRY(expr)                                                     \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess) {                                                    \
      ::f360::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                        __FILE__, __LINE__);                                   \
      return _e == hipErrorOutOfMemory ? F360_ERR_OOM : F360_ERR_HIP;          \
    }                                                                          \
  } while (0)

#define F360_REQUIRE(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      ::f360::set_error(__VA_ARGS__);    \
      return F360_ERR_INVALID_ARG;       \
    }                                    \
  } while (0)

// Upper bound on any frame dimension taken by the entry points that build per-axis tables on the
// host: the grids are 16-bit offsets (the reference's short2), so nothing larger can be meant,
// and a wild value must come back as an error code, not as std::bad_alloc through a C boundary.
constexpr int kMaxDim = 1 << 16;
inline bool dims_ok(std::initializer_list<int> dims) {
  for (int d : dims)
    if (d > kMaxDim) return false;
  return true;
}

// Makes `device` the calling thread's current HIP device for the lifetime of the guard and puts
// the previous one back afterwards: every C-ABI entry that allocates, copies or launches binds
// its context's device this way, so one thread may hold contexts on several GPUs.
class DeviceGuard {
 public:
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
    if (prev_ != device) {
      status_ = hipSetDevice(device);
      changed_ = status_ == hipSuccess;
    }
  }
  ~DeviceGuard() {
    if (changed_ && prev_ >= 0) (void)hipSetDevice(prev_);
  }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
  hipError_t status() const { return status_; }

 private:
  int prev_ = -1;
  bool changed_ = false;
  hipError_t status_ = hipSuccess;
};
#define F360_BIND_DEVICE(ctx_expr)                         \
  ::f360::DeviceGuard _f360_guard((ctx_expr)->device);     \
  F360_HIP_TRY(_f360_guard.status())

// A device allocation owned by an engine object.
struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  int reserve(size_t n);  // grows (never shrinks); contents undefined after growth
  void release();
  template <class T> T *as() const { return static_cast<T *>(p); }
};

// Scratch of the SAT encoder for one frame geometry (see sat_three.hip, sat_walk.hip).
struct SatEncodePlan {
  int width = 0, height = 0;
  int band_rows = 0;   // TH: rows per band (tile height of the writer kernel)
  int sb_bands = 0;    // bands per super-band (tile height of the reducer)
  int nstrips = 0, nbands = 0, nsb = 0;
  int wp3 = 0;         // 3 * padded width (padded to whole 256-pixel strips)
  DevBuf ws;           // one allocation, carved below (`frames` slices of ws_stride elements)
  int frames = 0;      // frames the scratch can hold at once (f360_sat_encode_batch)
  size_t ws_stride = 0;
  uint32_t *lp = nullptr;        // [nbands][wp3]  column sums above the band, inside its super-band
  uint32_t *sbtotal = nullptr;   // [nsb][wp3]     column sums of each super-band
  uint32_t *sbprefix = nullptr;  // [nsb][wp3]     column sums of all super-bands above
  uint32_t *rowsum = nullptr;    // [nstrips][height][3]
  uint32_t *rowcarry = nullptr;  // [nstrips][height][3] sum of the strips to the left
  uint32_t *tiletotal = nullptr; // [nstrips][nbands][3]
  uint32_t *tprefix = nullptr;   // [nstrips][nbands][3] sum of the tiles to the left
  // read-once batched encoder (sat_walk_kernel): hand-off granules, the launch state words
  // (device memory, advanced by the launches themselves) and the host-mapped error word
  DevBuf walk_chain, walk_state;
  DevBuf walk_plan;  // encode + sample: per frame of a launch, one plan word per table row
  uint32_t *walk_err_host = nullptr, *walk_err_dev = nullptr;
  int walk_stats_units = 0;       // debug statistics behind the granules (debug.ablate bit 8)
  size_t walk_stats_offset = 0;
};

}  // namespace f360

namespace f360 {
#ifdef __HIPCC__
// Workgroups are dealt to the 8 XCDs (each with its own L2) in launch order.  A kernel whose
// neighbouring output rows gather from neighbouring source rows therefore pulls every source
// sector through up to eight L2s.  This remap gives XCD k the k-th contiguous run of blocks in
// row-major order -- a band of output rows -- instead of every eighth block ("is.xcd_bands").
__device__ __forceinline__ void xcd_band_block(bool on, int &bx, int &by) {
  bx = blockIdx.x;
  by = blockIdx.y;
  const uint32_t gx = gridDim.x, total = gx * gridDim.y;
  if (!on || total < 64) return;
  const uint32_t lid = blockIdx.y * gx + blockIdx.x, k = lid & 7;
  const uint32_t idx = k * (total >> 3) + min(k, total & 7) + (lid >> 3);
  by = (int)(idx / gx);
  bx = (int)(idx - (uint32_t)by * gx);
}
#endif
}  // namespace f360

namespace f360 {
// Kernel ids of the per-kernel timing facility (f360_ctx_profile_*).
enum KernelId {
  kSatReduce = 0,
  kSatCarry,
  kSatWrite,
  kSatWalk,
  kSampleRect,
  kInterpolateRect,
  kSatDecode,
  kIsSampleRect,
  kIsSampleLogpolar,
  kIsInterpolateLogpolar,
  kIsBlur,
  kGnomonic,
  kFovMaps,
  kFovSample,
  kYuvToRgb,
  kRgbToYuv,
  kExpand,
  kWalkFusePlan,
  kWalkFuseFix,
  kSatWriteFuse,
  kKernelCount
};
struct ProfSpan {
  int kid;
  hipEvent_t a, b;
  int frames;  // frames the launch covered (batched calls)
};
}  // namespace f360

struct f360_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  f360::SatEncodePlan enc;
  // options (f360_ctx_set_option)
  int opt_band_rows = 0;       // "sat.band_rows": 0 (by frame size) | 16 | 32 | 64
  int opt_sb_bands = -1;       // "sat.sb_bands": bands per reducer wave (-1: 1 for planar sources and 64-row bands, else 2; 0: as few super-bands as 32)
  int opt_sample_variant = 2;  // "sample.variant": 0 per-pixel, 1 column walker, 2 tile streamer (falls back to the walker where it does not apply)
  int opt_walk_rows = 8;       // "sample.rows": reduced rows per wave of the column walker
  int opt_stream_rows = 0;     // "sample.srows": reduced rows per wave of the tile streamer, <= 64; 0 = 4 for a single frame's launch, 8 for the batched launches
  int opt_sample_fpl = 16;     // "sample.fpl": frames per launch of f360_satdec_sample_rect_frames (1..64)
  int opt_batch_mb = 180;      // "sat.batch_mb": source bytes (MB) a batched encoder launch may cover
  int opt_walk = -1;           // "sat.walk": batched encodes read the frame once (sat_walk_kernel): -1 = when the batch fills the device ("sat.walk_units"), 0 never, 1 whenever the layout allows
  int opt_walk_units = 690;    // "sat.walk_units": (frame, strip) units a batch needs before sat.walk = -1 takes the read-once encoder (23 frames at 8K: a launch takes at least its 480 serial batches of ~3.5 us whatever its frame count, so below ~22 frames the three kernels' 101 us per frame win; profiles/round4_few_frames.txt)
  int opt_pool_mb = 0;         // "sat.pool_mb": most device memory (MB) f360_sat_tables_alloc may hold at any time while it draws, scratch frames included; 0 = a third of what is free when the call starts (never less than the tables asked for plus one group)
  int opt_walk_frames = 0;     // "sat.walk_frames": most frames one read-once launch takes (1..64); 0 = about 1024 strip owners, one per SIMD
  int opt_interp_staged = 1;   // "interp.staged": the un-warp computes the vertical lerps once per reduced column (wave-private LDS) instead of per output pixel
  int opt_interp_rows = 0;     // "interp.rows": output rows per wave of the un-warp, 0 = by size
  int opt_r2y_rows = 0;        // "yuv.r2y_rows": chroma rows a wave of the RGB0 -> yuv420p converter walks down; 0 = by frame size (16 / 8 / 4, small frames: the kernel with one chroma row per thread), -1 = always that kernel
  int opt_fuse_force = 0;      // "debug.fuse_force": tests only -- force the one-pass forms' rare branches (results stay exact): bit 0 = side rows of one pixel (more straddling boxes than they hold: the fix-up takes every row), bit 1 = no listed leftover rows (the fix-up finds them itself), bit 2 = the strip walker's helpers keep one round of pixels in registers and take their tail loop for the rest
  int opt_walk_spin = 0;       // "debug.walk_spin": polls a strip's hand-off wait may take before it finishes alone; 0 = 65536
  int opt_walk_mute = 0;       // "debug.walk_mute": test only -- unit (value - 1) of every read-once launch publishes no hand-off, so its right neighbour times out; 0 = none
  int opt_ablate = 0;          // "debug.ablate": timing experiments, breaks results
  int opt_xcd_bands = 1;       // "is.xcd_bands": the point samplers give each XCD a band of output rows instead of every eighth workgroup: 0 never, 1 where it pays (log-rectilinear sampler, sources of 64 MB and more), 2 always
  int opt_lp_lds = 1;          // "is.lp_lds": log-polar un-warp keeps its axis tables in LDS (needs is.lp_table); 0 off, 1 on, 256 / 512 / 1024 = on with that workgroup size
  int opt_lp_table = 1;        // "is.lp_table": log-polar un-warp reads its inverse map from a per-geometry table
  int opt_gnomonic_table = 1;  // "gnomonic.table": view-independent terms of the remap read from a per-geometry table: 0 none, 1 five planes (x, y, rho, sin, cos)
  int opt_gnomonic_guard = 1;  // "gnomonic.guard": texel indices from a cheap float evaluation wherever its error bound decides them, the exact chain for the others (64 at a time)
  int opt_fuse_walk = 1;       // "fuse.walk": f360_satdec_encode_sample_frames samples inside the read-once encoder's pass wherever it applies; 0 = always the two calls
  int opt_fuse_band = 1;       // "fuse.band": f360_satdec_encode_sample_frames calls too small for the read-once encoder sample inside the three-kernel encoder's table writer (sat_write_fuse_kernel): 1 = from four frames per call on (where the launch groups, pipelined over the side stream, beat the two calls: +2 % at 4 frames of 8K, +8 % at 8-22; one and two frames are faster as the two calls, profiles/round5_band_one_pass.txt), 2 = always, 0 = never (the two calls)
  int opt_fov_piggyback = 1;   // "fov.piggyback": lattice maps of the fused path as extra workgroups of the reducer
  int opt_yuv_model = 1;       // "yuv.model": libswscale converter to reproduce, 0 C tables, 1 x86 MMX
  // "expand" debug views (expand.hip): per-geometry axis tables and the ordering keys of the
  // log-polar scatter
  int ex_w = 0, ex_h = 0, ex_tw = 0, ex_th = 0, ex_kind = -1;
  f360::DevBuf ex_tables, ex_keys;
  // gnomonic remap: view-independent per-pixel terms of one target geometry (projections.hip)
  int gn_w = 0, gn_h = 0;
  f360::DevBuf gn_table;
  // index-guarded remap ("gnomonic.guard"): rho / sin / cos planes + the two screen axes of one
  // target geometry; a debug word (rejected pixels of the last counted launch)
  int gn_gw = 0, gn_gh = 0, gn_cus = 0, gn_wg_per_cu = 0;
  f360::DevBuf gn_gtab, gn_counters;
  // Side stream of batched calls that pipeline their frames (f360::side_stream): frame k + 1's
  // reducer and carry pass run beside frame k's table writer.  Created on first use; forked
  // from and joined back into `stream` with the two events inside the call, so from outside the
  // call is still "enqueue on one in-order stream".
  hipStream_t side = nullptr;
  hipEvent_t side_fork = nullptr, side_join = nullptr;
  int opt_pipeline = 1;  // "sat.pipeline": batched calls on the three-kernel encoder alternate their launch groups between the context's stream and the side stream: 1 = the band writer's one pass only (+14 %; the plain three kernels gain nothing), 2 = every batched call, 0 = one stream.  (Reducers + carry passes on one stream and writers on the other, tied by an event pair per group, was slower than one stream: profiles/round5_band_one_pass.txt)
  // per-kernel HIP-event timing of sampled calls (f360_ctx_profile_arm/read)
  int prof_armed = 0;
  std::vector<f360::ProfSpan> prof_pending;
  std::vector<hipEvent_t> prof_free;
  double prof_ms[f360::kKernelCount] = {};
  int prof_launches[f360::kKernelCount] = {};
  int prof_frames[f360::kKernelCount] = {};
};

namespace f360 {
// Brackets one kernel launch with HIP events on the context's stream when the
// current call is being sampled.
class KernelSpan {
 public:
  KernelSpan(f360_ctx *ctx, int kid, bool on, int frames = 1, hipStream_t stream = nullptr)
      : ctx_(ctx), kid_(kid), on_(on), frames_(frames), stream_(stream ? stream : ctx->stream) {
    if (!on_) return;
    a_ = take();
    b_ = take();
    if (!a_ || !b_) { on_ = false; return; }
    (void)hipEventRecord(a_, stream_);
  }
  ~KernelSpan() {
    if (!on_) return;
    (void)hipEventRecord(b_, stream_);
    ctx_->prof_pending.push_back(ProfSpan{kid_, a_, b_, frames_});
  }
 private:
  hipEvent_t take() {
    if (!ctx_->prof_free.empty()) {
      hipEvent_t e = ctx_->prof_free.back();
      ctx_->prof_free.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
  }
  f360_ctx *ctx_;
  int kid_;
  bool on_;
  int frames_;
  hipStream_t stream_;
  hipEvent_t a_ = nullptr, b_ = nullptr;
};
// true when this call is sampled; consumes one armed call
inline bool take_profile_slot(f360_ctx *ctx) {
  if (ctx->prof_armed <= 0) return false;
  --ctx->prof_armed;
  return true;
}
}  // namespace f360

namespace f360 {
// Fused foveation: what the table writer emits instead of the full table.
struct FovMaps;
struct SatEmit {
  const int *xmap, *ymap;
  uint32_t *corners;
  int corner_stride;
  const FovMaps *maps;  // non-null: the reducer's launch also computes the lattice maps
};
// The context's side stream and its fork / join events, created on first use (null + error set
// on failure).
int side_stream(f360_ctx *ctx);
// Makes the three-kernel encoder's scratch hold `frames` slices of this geometry (sat_three.hip).
int sat_encode_reserve(f360_ctx *ctx, int width, int height, int frames, bool planar = false);
struct SatBandFuse;  // sat_fuse_dev.h: the table writer also emits the reduced pixels of its tile
// Where one launch group of a pipelined batched call goes: the stream (null: the context's) and
// which of the `slots` slices of the encoder's scratch, each `slot_frames` frames large, it uses.
struct SatLaunch {
  hipStream_t stream = nullptr;
  int slot = 0, slots = 1, slot_frames = 0;  // (slot_frames 0: the launch's own frame count)
};
// `yuv` non-null: the pixels come from three planes (src_dev / linesize unused)
// `count` > 0: a batch of frames of one geometry (tables sats[k] of sources srcs[k]; sat_dev /
// src_dev unused, no emit; `yuvs` non-null: frame k's planes, all with yuvs[0]'s linesizes).  `profile`: -1 = take a profile slot if one is armed,
// 0 / 1 = the caller already decided (one slot per batched call, however many launches)
int sat_encode_impl(f360_ctx *ctx, uint32_t *sat_dev, const uint8_t *src_dev, int width,
                    int height, int linesize, const SatEmit *emit, const YuvPlanes *yuv,
                    int count = 0, uint32_t *const *sats = nullptr,
                    const uint8_t *const *srcs = nullptr, int profile = -1,
                    const YuvPlanes *yuvs = nullptr, const SatBandFuse *band_fuse = nullptr,
                    const SatLaunch *where = nullptr);
// The launch groups of a batched call on the three-kernel encoder, `ngroups` of at most
// `group_frames` frames: group g is a chain reducer -> carry pass -> table writer, the frames of
// a call are independent, and beside one group's writer (write-bound) the next group's reducer
// (read-bound, a third of a frame's time) and carry pass (latency-bound) run almost for free.
// So the groups alternate between the context's stream and its side stream ("sat.pipeline"),
// each with its own slice of the encoder's scratch; forked here, joined before returning: from
// outside the call is still work enqueued on one in-order stream (and capturable as such).
// `launch(g, where)` enqueues group g.
template <class Launch>
int sat_pipelined_groups(f360_ctx *ctx, int width, int height, bool planar, int ngroups,
                         int group_frames, bool one_pass, Launch &&launch) {
  // (the plain three kernels gain nothing from it -- their writer is bandwidth-bound, and a
  // reducer beside it takes what it gives -- so they stay on one stream unless "sat.pipeline" is 2)
  const bool pipelined = (ctx->opt_pipeline == 2 || (ctx->opt_pipeline == 1 && one_pass)) && ngroups >= 2;
  int st = F360_OK;
  if (pipelined) {
    st = side_stream(ctx);
    if (st != F360_OK) return st;
    // (both slices exist before anything is forked: growing the scratch synchronises)
    st = sat_encode_reserve(ctx, width, height, 2 * group_frames, planar);
    if (st != F360_OK) return st;
    F360_HIP_TRY(hipEventRecord(ctx->side_fork, ctx->stream));
    F360_HIP_TRY(hipStreamWaitEvent(ctx->side, ctx->side_fork, 0));
  }
  SatLaunch where;
  where.slots = pipelined ? 2 : 1;
  where.slot_frames = group_frames;
  for (int g = 0; g < ngroups && st == F360_OK; ++g) {
    where.slot = pipelined ? g & 1 : 0;
    where.stream = pipelined && (g & 1) ? ctx->side : nullptr;
    st = launch(g, where);
  }
  if (pipelined) {  // (joined whatever happened above: nothing may be left running on the side)
    F360_HIP_TRY(hipEventRecord(ctx->side_join, ctx->side));
    F360_HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->side_join, 0));
  }
  return st;
}
}  // namespace f360

namespace f360 {
// Encode + sample in one pass over the frames (sat_fuse.hip: sat_walk_kernel<.., true>): where
// the reduced frames go, the gaze of every frame, the decoder's 1-D grid factors (device).
struct SatFuse {
  uint8_t *const *dsts;
  const float *centers_xy;
  const int16_t *gx, *gy;
  int out_w, out_h, dst_linesize;
};
// `yuv` / `yuvs` non-null: planar sources (srcs / linesize unused; every frame's planes with
// yuvs[0]'s linesizes)
bool sat_encode_sample_applies(const f360_ctx *ctx, int count, int width, int height,
                               int linesize, int out_w, int out_h, int dst_linesize,
                               const YuvPlanes *yuv = nullptr);
int sat_encode_sample_walk(f360_ctx *ctx, int count, uint32_t *const *sats,
                           const uint8_t *const *srcs, const YuvPlanes *yuvs, int width,
                           int height, int linesize, const SatFuse &fuse, bool prof);
// The context's side stream and its fork / join events, created on first use (null + error set
// on failure).
int side_stream(f360_ctx *ctx);
// Makes the three-kernel encoder's scratch hold `frames` slices of this geometry (sat_three.hip).

// The same for calls the read-once encoder does not take (1 .. 22 8K frames): the three-kernel
// encoder with its table writer in one-pass form (sat_band_fuse.hip); RGB0 frames or planes.
bool sat_encode_sample_band_applies(const f360_ctx *ctx, int count, int width, int height,
                                    int linesize, int out_w, int out_h, int dst_linesize,
                                    const YuvPlanes *yuv = nullptr);
int sat_encode_sample_band(f360_ctx *ctx, int count, uint32_t *const *sats,
                           const uint8_t *const *srcs, const YuvPlanes *yuvs, int width, int height,
                           int linesize, const SatFuse &fuse, bool prof);
}  // namespace f360

struct f360_event {
  hipEvent_t ev = nullptr;
};

struct f360_sat_decoder {
  f360_ctx *ctx = nullptr;
  // log-rectilinear midpoint grid, 1-D factors (create_grid_kernel is separable)
  int gw = 0, gh = 0, sw = 0, sh = 0;  // target / source geometry of the grid
  std::vector<int16_t> gx_host, gy_host;  // gw+1 / gh+1 entries
  f360::DevBuf gx_dev, gy_dev;
  // tile streamer: inverse of the x grid, largest corner step (+1, rounded to 4)
  f360::DevBuf lbx_dev;
  std::vector<int> lbx_host;
  int lb_dmin = 0, lb_n = 0, halo = 0;
  bool stream_ok = false;
  // fused foveation (f360_satdec_foveate_rect): per-gaze lattice maps and the compact corners
  f360::DevBuf fov_maps, fov_corners;
  // inverse-map tables of the interpolate kernel, indexed by pixel offset from
  // the gaze centre (geometry-only; see sat_decoder.hip)
  int it_w = 0, it_h = 0, it_rw = 0, it_rh = 0;  // geometry they were built for
  int it_dx = 0, it_dy = 0;                      // covered offset range [-d, d]
  f360::DevBuf itx_dev, ity_dev;
};

struct f360_image_sampler {
  f360_ctx *ctx = nullptr;
  int gw = 0, gh = 0, sw = 0, sh = 0;
  std::vector<int16_t> gx_host, gy_host;  // gw / gh entries
  f360::DevBuf gx_dev, gy_dev;
  int lw = 0, lh = 0, lsw = 0, lsh = 0;   // log-polar grid geometry
  std::vector<float> lrad_host, lcos_host, lsin_host;
  f360::DevBuf lrad_dev, lcos_dev, lsin_dev;
  // interpolate_logpolar tables (per source geometry)
  int iw = 0, ih = 0;
  f360::DevBuf irad_dev, icos_dev, isin_dev;  // float[iw], double[ih], double[ih]
  // interpolate_logpolar: offset -> reduced-buffer coordinate table of one geometry
  int lpt_w = 0, lpt_h = 0, lpt_sw = 0, lpt_sh = 0;
  f360::DevBuf lpt_dev;
  bool lp_lds_ready = false;  // LDS-table un-warp: kernel attribute set, CU count known
  int lp_cus = 0;
};
