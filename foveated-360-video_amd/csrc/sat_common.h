// sat_common.h -- device helpers and argument blocks shared by the SAT encoders of this
// directory: the three-kernel encoder (sat_three.hip), the read-once strip walker (sat_walk.h,
// sat_walk.hip), its one-pass encode + sample form (sat_fuse_dev.h, sat_fuse.hip) and the band
// writer's one-pass form (sat_band_fuse.hip).
//
// Replaces SATEncoder::EncodeFrameGPU (src/sat_encoder.cc:67-135) and its three OpenCL kernels
// copy_image / scan_rows / scan_columns (src/sat_encoder_encode_kernels.cl:1-20,44-58,60-74).
// Output is bit-identical: uint32 addition is associative mod 2^32, so any summation order gives
// the reference's integers.
//
// Everything here is inline device code or plain data, in a named namespace, so that kernels
// and host functions of different translation units agree on the types they exchange.
#pragma once

#include <algorithm>
#include <string>
#include <vector>

#include "f360_internal.h"
#include "fov_maps.h"
#include "host_tables.h"

namespace f360 {
namespace sat {

// Where K1 and K3 take their pixels from (template parameter SRC).  The planar sources
// convert in registers with libswscale's arithmetic (yuv_device.h), so the table equals the
// one of the RGB0 frame sws_scale would have produced, without that frame ever existing.
enum { kSrcBytes = 0, kSrcRgb0 = 1, kSrcYuvSwsC = 2, kSrcYuvSwsX86 = 3 };

constexpr int kLanePx = 4;                 // pixels per lane
constexpr int kStripPx = 64 * kLanePx;     // pixels per wave-row
constexpr int kWavesPerBlock = 4;
constexpr int kRowUnroll = 8;              // rows whose loads are issued together
#ifndef F360_REDUCE_DEPTH
#define F360_REDUCE_DEPTH 2
#endif
constexpr int kReduceDepth = F360_REDUCE_DEPTH;  // batches a reducer wave keeps in flight

// ---- wave64 DPP helpers ---------------------------------------------------
// dpp_ctrl: 0x110+n row_shr:n, 0x142 row_bcast:15, 0x143 row_bcast:31.
#define F360_DPP_ADD(v, ctrl, row_mask)                                        \
  (v) += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (row_mask), \
                                               0xf, false)

// Inclusive prefix sum over the 64 lanes of a wave (mod 2^32).
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
  F360_DPP_ADD(v, 0x111, 0xf);  // within each row of 16 lanes: Kogge-Stone
  F360_DPP_ADD(v, 0x112, 0xf);
  F360_DPP_ADD(v, 0x114, 0xf);
  F360_DPP_ADD(v, 0x118, 0xf);
  F360_DPP_ADD(v, 0x142, 0xa);  // rows 1,3 += last lane of rows 0,2
  F360_DPP_ADD(v, 0x143, 0xc);  // rows 2,3 += lane 31
  return v;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned 16 B
struct __attribute__((packed, aligned(4))) u32x3 {
  uint32_t x, y, z;
};

// LDS byte address in, 16 bytes per lane.  The reads carry their own wait
// (hipcc does not count memory operations issued from inline asm).
__device__ __forceinline__ void lds_write_b128(uint32_t addr, u32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_read3_b128(uint32_t addr, u32x4 &a, u32x4 &b,
                                               u32x4 &c) {
  asm volatile(
      "ds_read_b128 %0, %3\n\t"
      "ds_read_b128 %1, %3 offset:1024\n\t"
      "ds_read_b128 %2, %3 offset:2048\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(a), "=&v"(b), "=&v"(c)
      : "v"(addr)
      : "memory");
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lds_write_b64(uint32_t addr, uint32_t a, uint32_t b) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(u32x2{a, b}) : "memory");
}
__device__ __forceinline__ uint32_t lds_read_b32(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

// A 16-byte global store the compiler does not see as a memory operation.  Loads and stores share
// one counter (vmcnt) on this hardware and complete out of order relative to each other, so
// with a store pending the compiler turns every wait for a load into vmcnt(0), i.e. into a wait
// for the acknowledgement of all earlier stores as well.  Hidden from its bookkeeping, a wait
// for a load is vmcnt(number of younger loads): still sufficient (loads complete in order; the
// extra pending stores can only make the wait longer, never shorter) and it no longer drains
// the stores.  The s_nop covers the store-data hazard (data > 8 bytes overwritten next).
__device__ __forceinline__ void global_store_b128_uncounted(uint32_t *p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

typedef uint32_t u32x3v __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void global_store_b96_uncounted(uint32_t *p, uint32_t x, uint32_t y,
                                                           uint32_t z) {
  asm volatile("global_store_dwordx3 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(u32x3v{x, y, z})
               : "memory");
}

// The same, non-temporal: the table (354 MB at 8K) is written once and read back much later by
// another kernel, so its lines should not linger in L2 / Infinity Cache as dirty data -- their
// deferred write-back is otherwise paid by whichever kernel runs next (the next frame's reducer:
// 38-41 us in the pipeline against 26 us alone).  With `nt` the reducer takes 29 us, the writer
// itself 80 instead of 83 us and the whole path gains 8-12 %.  (Not for the fused path's corner
// array, which the compact sampler reads back at once: measured neutral to slightly worse.)
__device__ __forceinline__ void global_store_b128_uncounted_nt(uint32_t *p, u32x4 v) {
#ifdef F360_NO_NT_STORES
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#else
#ifndef F360_NT_BITS
#define F360_NT_BITS "sc0 sc1 nt"  // A/B: "nt" 174.3, "sc0 sc1 nt" 176.5, "sc0 sc1" alone 160.7 Gpix/s
#endif
  asm volatile("global_store_dwordx4 %0, %1, off " F360_NT_BITS "\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#endif
}

constexpr int kEncBatch = 16;  // frames per batched launch (f360_sat_encode_batch)

struct EncodeArgs {
  uint32_t *sat;
  const uint8_t *src;
  int width, height, linesize, bpp;
  int band_rows, sb_bands, nstrips, nbands, nsb, wp3;
  uint32_t *lp, *sbtotal, *sbprefix, *rowsum, *rowcarry, *tiletotal, *tprefix;
  int ablate;  // timing experiments only (results are wrong when non-zero)
  // STORE == 2 (fused foveation): instead of the table, emit only the entries at the
  // lattice rows / columns a given gaze will sample
  const int *xmap, *ymap;  // source column / row -> compact index, -1 when unused
  uint32_t *corners;       // [compact row][corner_stride][3]
  int corner_stride;
  // SRC >= kSrcYuvSwsC: the three planes and the conversion constants
  f360::YuvPlanes yuv;
  f360::YuvConsts k;
  // fused foveation: workgroups past `reduce_blocks` of the reducer's grid compute the lattice
  // maps (one per axis) while the others reduce
  int reduce_blocks, has_maps;
  f360::FovMaps maps;
  // batched frames: blockIdx.y selects the frame -- its source, its table and its slice of the
  // scratch arrays (frame f's start f * ws_stride elements after frame 0's)
  int nbatch;
  size_t ws_stride;
  // read-once batched encoder (sat_walk_kernel): (frame, strip) units of the launch, 8-row
  // batches per strip, and the hand-off state (see WalkState)
  int walk_units, walk_nbatches;
  struct WalkState *walk;
  unsigned long long *walk_chain;  // [unit][batch][24] {tag:40 | row prefix:24}
  uint32_t *walk_err;              // host-visible: strips that gave up waiting and finished alone
  uint32_t walk_spin;              // polls a hand-off wait may take before the strip goes it alone
  int walk_mute;                   // test only ("debug.walk_mute"): this unit publishes nothing; -1
  unsigned long long *walk_stats;  // debug.ablate bit 8: per unit {start, end, slow polls, spins}
};
// (a kernel argument of its own: inside EncodeArgs the arrays keep the compiler from taking
// that struct apart, it lands in scratch memory and the row loops wait on vmcnt(0))
struct EncodeBatch {
  const uint8_t *src[kEncBatch];  // packed source, or the luma plane
  uint32_t *sat[kEncBatch];
  const uint8_t *u[kEncBatch], *v[kEncBatch];  // planar sources
};

// What a workgroup works on: the call's one frame, or frame blockIdx.y of a batch.
struct EncodeFrame {
  const uint8_t *src;
  uint32_t *sat;
  size_t ws;  // offset of the frame's scratch slice, in elements
  const uint8_t *y, *u, *v;  // planar sources (linesizes are the call's, in EncodeArgs::yuv)
};
// Constant indices and scalar selects: a dynamic index into a by-value argument makes the
// compiler copy it to scratch memory.
#define F360_ENCODE_FRAME(fr, a, b)                                          \
  EncodeFrame fr{a.src, a.sat, 0, a.yuv.y, a.yuv.u, a.yuv.v};                \
  if (a.nbatch != 0) {                                                       \
    const int f_ = (int)blockIdx.y;                                          \
    fr.src = b.src[0];                                                       \
    fr.sat = b.sat[0];                                                       \
    fr.u = b.u[0];                                                           \
    fr.v = b.v[0];                                                           \
    _Pragma("unroll") for (int k_ = 1; k_ < kEncBatch; ++k_) if (f_ == k_) { \
      fr.src = b.src[k_];                                                    \
      fr.sat = b.sat[k_];                                                    \
      fr.u = b.u[k_];                                                        \
      fr.v = b.v[k_];                                                        \
    }                                                                        \
    fr.y = fr.src;                                                           \
    fr.ws = (size_t)f_ * a.ws_stride;                                        \
  }

// kRowUnroll rows of a lane's four pixels as loaded; planar sources convert at use, so that
// the loads of a whole batch stay in flight.
template <bool YUV>
struct RowBatchT {
  uint4 raw[kRowUnroll];  // packed R | G<<8 | B<<16 dwords
};
template <>
struct RowBatchT<true> {
  uint32_t y4[kRowUnroll];      // four luma bytes per row
  uint32_t uv[kRowUnroll / 2];  // per row pair: U0 | U1<<8 | V0<<16 | V1<<24
};
template <int SRC>
using RowBatch = RowBatchT<(SRC >= kSrcYuvSwsC)>;

template <int SRC>
__device__ __forceinline__ void batch_pixels(const EncodeArgs &a, const RowBatch<SRC> &b,
                                             int r, uint32_t (&v)[4]) {
  if constexpr (SRC >= kSrcYuvSwsC) {
    f360::yuv_pixels4<SRC - kSrcYuvSwsC>(a.k, b.y4[r], b.uv[r >> 1], v);
  } else {
    v[0] = b.raw[r].x;
    v[1] = b.raw[r].y;
    v[2] = b.raw[r].z;
    v[3] = b.raw[r].w;
  }
}

// rows [y, y + kRowUnroll) of a planar source, y a multiple of kRowUnroll; rows past y_last
// (odd: the last row of the frame or of the caller's run of rows) re-read that row -- what lies
// past it is masked or never stored by the callers
template <int SRC>
__device__ __forceinline__ void load_yuv_batch(const EncodeArgs &a, const EncodeFrame &fr,
                                               RowBatch<SRC> &b, int y, int x0, int y_last) {
  if constexpr (SRC >= kSrcYuvSwsC) {
    const int xc = min(x0, a.width - kLanePx);
#pragma unroll
    for (int r = 0; r < kRowUnroll; ++r)
      b.y4[r] = *reinterpret_cast<const uint32_t *>(
          fr.y + (size_t)min(y + r, y_last) * a.yuv.y_linesize + xc);
#pragma unroll
    for (int r = 0; r < kRowUnroll / 2; ++r) {
      const size_t crow = (size_t)min((y >> 1) + r, y_last >> 1);
      const uint32_t u =
          *reinterpret_cast<const uint16_t *>(fr.u + crow * a.yuv.u_linesize + (xc >> 1));
      const uint32_t v =
          *reinterpret_cast<const uint16_t *>(fr.v + crow * a.yuv.v_linesize + (xc >> 1));
      b.uv[r] = u | (v << 16);
    }
  }
}

// Four pixels of one row as packed R | G<<8 | B<<16 dwords (0 beyond the row).
template <int SRC>
__device__ __forceinline__ uint4 load_px4(const uint8_t *src, int width, int y,
                                          int x0, int linesize, int bpp) {
  if (SRC == kSrcRgb0) {
    if (x0 < width)
      return *reinterpret_cast<const uint4 *>(src + (size_t)y * linesize +
                                              (size_t)x0 * 4);
    return make_uint4(0, 0, 0, 0);
  }
  uint32_t v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = 0;
    if (x0 + k < width) {
      const uint8_t *p = src + (size_t)y * linesize + (size_t)(x0 + k) * bpp;
      v[k] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
    }
  }
  return make_uint4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ void unpack_px4(const uint4 &raw, uint32_t (&c)[12]) {
  const uint32_t v[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    c[3 * k + 0] = v[k] & 0xffu;
    c[3 * k + 1] = (v[k] >> 8) & 0xffu;
    c[3 * k + 2] = (v[k] >> 16) & 0xffu;
  }
}

__device__ __forceinline__ void store12(uint32_t *dst, const uint32_t (&a)[12]) {
  uint4 *d = reinterpret_cast<uint4 *>(dst);
  d[0] = make_uint4(a[0], a[1], a[2], a[3]);
  d[1] = make_uint4(a[4], a[5], a[6], a[7]);
  d[2] = make_uint4(a[8], a[9], a[10], a[11]);
}

__device__ __forceinline__ void load12(const uint32_t *src, uint32_t (&a)[12]) {
  const uint4 *s = reinterpret_cast<const uint4 *>(src);
  const uint4 q0 = s[0], q1 = s[1], q2 = s[2];
  a[0] = q0.x; a[1] = q0.y; a[2] = q0.z; a[3] = q0.w;
  a[4] = q1.x; a[5] = q1.y; a[6] = q1.z; a[7] = q1.w;
  a[8] = q2.x; a[9] = q2.y; a[10] = q2.z; a[11] = q2.w;
}

template <int SRC>
__device__ __forceinline__ void reduce_load_batch(const EncodeArgs &a, const EncodeFrame &fr,
                                                  RowBatch<SRC> &b, int y, int x0, int y_last) {
  if constexpr (SRC >= kSrcYuvSwsC) {
    load_yuv_batch<SRC>(a, fr, b, y, x0, y_last);
  } else if constexpr (SRC == kSrcRgb0) {
    // branch-free: rows past the wave's last row re-read that row (a cache hit), validity is
    // applied by the caller's masks
    const int xc = min(x0, a.width - kLanePx);
    const uint8_t *p = fr.src + (size_t)xc * 4;
#pragma unroll
    for (int r = 0; r < kRowUnroll; ++r)
      b.raw[r] = *reinterpret_cast<const uint4 *>(p + (size_t)min(y + r, y_last) * a.linesize);
  } else {
#pragma unroll
    for (int r = 0; r < kRowUnroll; ++r)
      b.raw[r] = (y + r < a.height)
                     ? load_px4<kSrcBytes>(fr.src, a.width, y + r, x0, a.linesize, a.bpp)
                     : make_uint4(0, 0, 0, 0);
  }
}

}  // namespace sat
}  // namespace f360
