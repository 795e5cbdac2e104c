// fov_maps.h -- shared by sat_decoder.hip and the SAT encoders (sat_three.hip, sat_fuse.hip): one axis of the SAT sampler's box
// rule, and the lattice maps of the fused foveation path.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace f360 {

// One axis of sample_rect_kernel (src/sat_decoder_sample_rect_kernel.cl:168-204): box corner
// `hi`, its lower partner `lo`, and whether the pixel is processed at all as far as this axis
// is concerned.
struct AxisBox {
  int hi, lo;
  bool ok;
};

__device__ __forceinline__ AxisBox sample_axis(int centre, int d_hi, int d_lo, int size,
                                               bool wraps) {
  int hi = centre + d_hi, lo = centre + d_lo;
  if (wraps) {  // only x wraps (:181-187); the y wrap is commented out
    if (hi >= size && lo >= size) {
      hi -= size;
      lo -= size;
    } else if (hi < 0 && lo < 0) {
      hi += size;
      lo += size;
    }
  }
  AxisBox b;
  b.ok = (hi >= 0 && hi < size) || (lo >= 0 && lo < size);
  b.hi = min(max(hi, 1), size - 1);
  b.lo = min(max(lo, 0), b.hi - 1);
  return b;
}

// The three channel quotients of one box, exact.  Common case (all operands
// < 2^22): ONE hardware reciprocal (1 ulp) shared by the channels, a float
// multiply per channel -- off by at most one, since n/d < 2^22 and the relative
// error is < 2^-22 -- then the exact remainder decides the +-1 correction.
static __device__ __noinline__ uint3 udiv3_slow(uint3 n, uint32_t d) {
  return make_uint3(n.x / d, n.y / d, n.z / d);
}
__device__ __forceinline__ uint32_t udiv_by_rcp(uint32_t n, float inv, uint32_t d) {
  uint32_t q = (uint32_t)((float)n * inv);
  const uint32_t r = n - __umul24(q, d);
  if ((int32_t)r < 0) q -= 1;
  else if (r >= d) q += 1;
  return q;
}
__device__ __forceinline__ uint3 udiv3_exact(uint3 n, uint32_t d) {
  if (((n.x | n.y | n.z | d) >> 22) != 0) return udiv3_slow(n, d);
  const float inv = __builtin_amdgcn_rcpf((float)d);
  return make_uint3(udiv_by_rcp(n.x, inv, d), udiv_by_rcp(n.y, inv, d),
                    udiv_by_rcp(n.z, inv, d));
}

// Fused foveation (SURVEY.md 8f-1 i): which table rows / columns a gaze samples, numbered.
struct FovMaps {
  const int16_t *gx, *gy;
  int cxp, cyp, src_w, src_h, out_w, out_h;
  int *xmap, *ymap;          // source column / row -> compact index or -1
  int *ihx, *ilx, *dxw;      // per reduced column: compact index of hi / lo corner, box width
  int *ihy, *ily, *dyw;      // per reduced row
};

// The maps of ONE axis (axis 0: x, 1: y) by one 256-thread workgroup.  Runs as two extra
// workgroups of the reducer's launch (sat_three.hip): the reducer does not need the maps, the
// table writer does, so the ~13 us this used to take as a kernel of its own disappear behind
// the reducer.  LDS: `flags` (one byte per source column / row) and `ranks` (int16), both
// kFovLdsEntries long; longer axes use the global map in place.
constexpr int kFovLdsEntries = 8192;
constexpr int kFovThreads = 256;
constexpr int kFovPerThread = 20;  // boxes a thread keeps in registers: axes up to 5120 outputs

__device__ __forceinline__ void fov_maps_axis(const FovMaps &m, int axis, uint8_t *flags_lds,
                                              int16_t *ranks_lds, int *part_lds) {
  const bool is_x = axis == 0;
  const int size = is_x ? m.src_w : m.src_h, n_out = is_x ? m.out_w : m.out_h;
  const int16_t *g = is_x ? m.gx : m.gy;
  const int centre = is_x ? m.cxp : m.cyp;
  int *map = is_x ? m.xmap : m.ymap;
  int *ih = is_x ? m.ihx : m.ihy, *il = is_x ? m.ilx : m.ily, *dw = is_x ? m.dxw : m.dyw;
  const bool in_lds = size <= kFovLdsEntries;
  const int t = threadIdx.x;
  // this thread's boxes (reduced pixels t, t + 256, ...): every grid load is issued before
  // anything depends on one, and the boxes stay in registers for the last phase -- otherwise the
  // workgroup is a chain of memory round trips that outlasts the reducer it hides behind
  const bool in_regs = n_out <= kFovPerThread * kFovThreads;
  int16_t g0[kFovPerThread], g1[kFovPerThread];
#pragma unroll
  for (int k = 0; k < kFovPerThread; ++k) {
    const int i = min(t + k * kFovThreads, n_out - 1);
    g0[k] = g[i];
    g1[k] = g[i + 1];
  }
  if (in_lds) {
    for (int x = t; x < size; x += kFovThreads) flags_lds[x] = 0;
  } else {
    for (int x = t; x < size; x += kFovThreads) map[x] = 0;
  }
  __syncthreads();
  AxisBox box[kFovPerThread];
#pragma unroll
  for (int k = 0; k < kFovPerThread; ++k) {
    box[k] = sample_axis(centre, g1[k], g0[k], size, is_x);
    if (t + k * kFovThreads < n_out && box[k].ok) {
      if (in_lds) {
        flags_lds[box[k].hi] = 1;
        flags_lds[box[k].lo] = 1;
      } else {
        map[box[k].hi] = 1;
        map[box[k].lo] = 1;
      }
    }
  }
  for (int i = t + kFovPerThread * kFovThreads; i < n_out; i += kFovThreads) {  // longer axes
    const AxisBox b = sample_axis(centre, g[i + 1], g[i], size, is_x);
    if (b.ok) {
      if (in_lds) {
        flags_lds[b.hi] = 1;
        flags_lds[b.lo] = 1;
      } else {
        map[b.hi] = 1;
        map[b.lo] = 1;
      }
    }
  }
  __syncthreads();
  // rank the used entries: per-thread chunk counts, a wave scan, a scan over the four waves
  const int chunk = (size + kFovThreads - 1) / kFovThreads;
  const int a = min(t * chunk, size), e = min(a + chunk, size);
  int cnt = 0;
  for (int x = a; x < e; ++x) cnt += in_lds ? (int)flags_lds[x] : map[x];
  int incl = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if ((t & 63) >= off) incl += v;
  }
  if ((t & 63) == 63) part_lds[t >> 6] = incl;
  __syncthreads();
  int run = incl - cnt;
  for (int w = 0; w < (t >> 6); ++w) run += part_lds[w];
  for (int x = a; x < e; ++x) {
    const int f = in_lds ? (int)flags_lds[x] : map[x];
    const int rank = f ? run : -1;
    if (in_lds) ranks_lds[x] = (int16_t)rank;
    map[x] = rank;
    run += f;
  }
  __syncthreads();
  if (in_regs) {
#pragma unroll
    for (int k = 0; k < kFovPerThread; ++k) {
      const int i = t + k * kFovThreads;
      if (i < n_out) {
        const AxisBox &b = box[k];
        ih[i] = b.ok ? (in_lds ? (int)ranks_lds[b.hi] : map[b.hi]) : -1;
        il[i] = b.ok ? (in_lds ? (int)ranks_lds[b.lo] : map[b.lo]) : -1;
        dw[i] = b.hi - b.lo;
      }
    }
    return;
  }
  for (int i = t; i < n_out; i += kFovThreads) {
    const AxisBox b = sample_axis(centre, g[i + 1], g[i], size, is_x);
    ih[i] = b.ok ? (in_lds ? (int)ranks_lds[b.hi] : map[b.hi]) : -1;
    il[i] = b.ok ? (in_lds ? (int)ranks_lds[b.lo] : map[b.lo]) : -1;
    dw[i] = b.hi - b.lo;
  }
}

}  // namespace f360
