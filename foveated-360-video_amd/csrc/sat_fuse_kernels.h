// sat_fuse_kernels.h -- the two small kernels around a one-pass encode + sample launch: the
// strip walker's row plan before it (the band writer has its own, sat_band_fuse.hip) and the
// fix-up after it, which both one-pass forms share.
#pragma once

#include "sat_walk.h"

namespace f360 {
namespace sat {

// The row plan of encode + sample, one workgroup per frame.  Reduced row j is the box of table
// rows (lo, hi] (fov_maps.h: sample_axis); with the grid's offsets strictly increasing -- checked
// on the host -- lo and hi are non-decreasing in j and lo(j + 1) = hi(j) everywhere except where
// the clamps of sat_decoder_sample_rect_kernel.cl:201-204 bite, next to the frame's top and
// bottom edge.  A strip owner keeps ONE snapshot, so row j can be emitted by the walk iff
//   * no processed row snapshots strictly inside (lo, hi): with lo non-decreasing only the rows
//     just below can, and
//   * no earlier reduced row already emits at table row hi (two reduced rows clamped onto the
//     frame's last table row).
// Such rows get their marks -- EMIT | j | height at hi, SNAP at lo -- and every other processed
// row is left to walk_fuse_fix_kernel, which recognises it by the missing mark.
// (a template only so that the header can be included by several translation units)
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void walk_fuse_plan_kernel(const int16_t *__restrict__ gy,
                                                             int out_h, int src_w, int src_h,
                                                             uint32_t *__restrict__ rowplan,
                                                             int plan_stride,
                                                             const WalkFuse wf) {
  uint32_t *plan = rowplan + (size_t)blockIdx.x * plan_stride;
  uint32_t *sp = wf.spix + (size_t)blockIdx.x * kSpixWords;
  const int cyp = wf.cyp[blockIdx.x];
  __shared__ int count, nleft;
  if (threadIdx.x == 0) count = nleft = 0;
  for (int y = threadIdx.x; y < plan_stride; y += 256) plan[y] = 0;
  __syncthreads();
  for (int j = threadIdx.x; j < out_h; j += 256) {
    const f360::AxisBox b = f360::sample_axis(cyp, gy[j + 1], gy[j], src_h, false);
    if (!b.ok) continue;
    bool fused = true;
    for (int d = 1; d <= 3; ++d) {
      if (j + d < out_h) {
        const f360::AxisBox n = f360::sample_axis(cyp, gy[j + d + 1], gy[j + d], src_h, false);
        if (n.ok && n.lo > b.lo && n.lo < b.hi) fused = false;
      }
      if (j - d >= 0) {
        const f360::AxisBox p = f360::sample_axis(cyp, gy[j - d + 1], gy[j - d], src_h, false);
        if (p.ok && p.hi == b.hi) fused = false;
      }
    }
    if (fused) {
      atomicOr(&plan[b.hi], kFuseEmit | (uint32_t)j | ((uint32_t)(b.hi - b.lo) << 16));
      atomicOr(&plan[b.lo], kFuseSnap);
    } else {
      const int k = atomicAdd(&nleft, 1);
      if (k < wf.lrows_max) sp[kSpixLrows + 1 + k] = (uint32_t)j;
    }
  }
  // the reduced columns whose box straddles two strips (in any order: the position in this
  // list is the pixel's slot in the side rows, for the helpers and for the fix-up alike)
  const int cxp = wf.cxp[blockIdx.x];
  for (int i = threadIdx.x; i < wf.out_w; i += 256) {
    const f360::AxisBox bx = f360::sample_axis(cxp, wf.gx[i + 1], wf.gx[i], src_w, true);
    if (bx.ok && (bx.hi >> 8) != (bx.lo >> 8)) {
      const int k = atomicAdd(&count, 1);
      if (k < kFixCols) {
        sp[1 + 3 * k] = (uint32_t)i;
        sp[2 + 3 * k] = (uint32_t)bx.hi;
        sp[3 + 3 * k] = (uint32_t)bx.lo;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    sp[0] = (uint32_t)count;
    sp[kSpixLrows] = (uint32_t)nleft;
  }
}

// What the strip owners' helpers left out: in the reduced rows they emitted, the pixels whose box
// straddles two strips, from the D values the two strips' helpers put into the side rows; every
// processed pixel of the other reduced rows (the plan kernel's comment), from the finished
// table with sample_rect_kernel's arithmetic (sat_decoder.hip) -- or, when no table was asked
// for, as plain sums over the box's source pixels.
// (`yuv_model` -1: RGB0 sources; 0 / 1: planes, libswscale's C / x86 arithmetic)
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void walk_fuse_fix_kernel(const WalkBatch wb, const WalkFuse wf,
                                                            int src_w, int src_h,
                                                            int src_linesize, int yuv_model,
                                                            const f360::YuvPlanes yl,
                                                            const f360::YuvConsts yk) {
  const int f = blockIdx.y;
  const int cxp = wf.cxp[f], cyp = wf.cyp[f];
  const uint32_t *sat = wb.sat[f];
  uint8_t *dst = wf.dst[f];
  const uint32_t *plan = wf.rowplan + (size_t)f * wf.plan_stride;
  const uint32_t *sp = wf.spix + (size_t)f * kSpixWords;
  const uint32_t *side = wf.side + (size_t)f * wf.side_stride;
  const int npix = (int)sp[0];
  const int nleft = (int)sp[kSpixLrows];
  auto store = [&](int i, int j, uint3 q) {
    uint8_t *o = dst + (size_t)j * wf.dst_linesize + (size_t)i * 4;
    // (plain stores: these pixels' neighbours were written long ago, so each one is a partial
    // write of a cold line -- left in L2 they cost 3.2 us per 8K frame, written through 7.1)
    *reinterpret_cast<uint16_t *>(o) = (uint16_t)((q.x & 0xffu) | ((q.y & 0xffu) << 8));
    o[2] = (uint8_t)q.z;
  };
  auto from_table = [&](int i, int j, const f360::AxisBox &by) {
    const f360::AxisBox bx = f360::sample_axis(cxp, wf.gx[i + 1], wf.gx[i], src_w, true);
    if (!bx.ok) return;
    if (sat == nullptr) {  // no table was written: the box from the (converted) pixels
      const uint8_t *src = wb.src[f];
      uint3 n = make_uint3(0, 0, 0);
      for (int y = by.lo + 1; y <= by.hi; ++y)
        for (int x = bx.lo + 1; x <= bx.hi; ++x) {
          uint32_t v;
          if (yuv_model < 0) {
            v = *reinterpret_cast<const uint32_t *>(src + (size_t)y * src_linesize + 4 * x);
          } else {  // (wb.src is the luma plane; the chroma sample of the 2x2 block, yuv_device.h)
            const int Y = src[(size_t)y * yl.y_linesize + x];
            const int U = wb.u[f][(size_t)(y >> 1) * yl.u_linesize + (x >> 1)];
            const int V = wb.v[f][(size_t)(y >> 1) * yl.v_linesize + (x >> 1)];
            v = yuv_model == 0 ? f360::yuv_pixel<0>(yk, Y, f360::chroma_terms<0>(yk, U, V))
                               : f360::yuv_pixel<1>(yk, Y, f360::chroma_terms<1>(yk, U, V));
          }
          n.x += v & 0xffu;
          n.y += (v >> 8) & 0xffu;
          n.z += (v >> 16) & 0xffu;
        }
      store(i, j, f360::udiv3_exact(n, (uint32_t)((bx.hi - bx.lo) * (by.hi - by.lo))));
      return;
    }
    auto at = [&](int y, int x) {
      const uint32_t *p = sat + ((size_t)y * src_w + x) * 3;
      return make_uint3(p[0], p[1], p[2]);
    };
    const uint3 br = at(by.hi, bx.hi), tr = at(by.lo, bx.hi), tl = at(by.lo, bx.lo),
                bl = at(by.hi, bx.lo);
    store(i, j,
          f360::udiv3_exact(make_uint3(br.x - tr.x + tl.x - bl.x, br.y - tr.y + tl.y - bl.y,
                                       br.z - tr.z + tl.z - bl.z),
                            (uint32_t)((bx.hi - bx.lo) * (by.hi - by.lo))));
  };
  // workgroups past the straddling pixels': 256 columns of one listed leftover row each
  const int nsb = (wf.out_h * wf.pmax + 255) / 256;
  if ((int)blockIdx.x >= nsb) {
    const int ncc = (wf.out_w + 255) / 256;
    const int idx = blockIdx.x - nsb, lr = idx / ncc, i = (idx - lr * ncc) * 256 + threadIdx.x;
    if (lr >= min(nleft, wf.lrows_max) || nleft > wf.lrows_max || npix > wf.pmax || i >= wf.out_w) return;
    const int j = (int)sp[kSpixLrows + 1 + lr];
    from_table(i, j, f360::sample_axis(cyp, wf.gy[j + 1], wf.gy[j], src_h, false));
    return;
  }
  // one thread per (reduced row, straddling pixel): no thread waits for more than its own loads
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int j = t / wf.pmax, k = t - j * wf.pmax;
  if (j >= wf.out_h) return;
  const f360::AxisBox by = f360::sample_axis(cyp, wf.gy[j + 1], wf.gy[j], src_h, false);
  if (!by.ok) return;
  const uint32_t pr = plan[by.hi];
  const bool emitted = (pr & kFuseEmit) && (int)(pr & 0xffffu) == j && npix <= wf.pmax;
  if (emitted) {
    if (k >= npix) return;
    const uint32_t *v = side + ((size_t)j * npix + k) * 6;
    const uint32_t area = (sp[2 + 3 * k] - sp[3 + 3 * k]) * (uint32_t)(by.hi - by.lo);
    store((int)sp[1 + 3 * k], j,
          f360::udiv3_exact(make_uint3(v[0] - v[3], v[1] - v[4], v[2] - v[5]), area));
  } else if (nleft > wf.lrows_max || npix > wf.pmax) {
    // (more leftover rows than the list holds: this row's pmax threads take the whole row)
    for (int i = k; i < wf.out_w; i += wf.pmax) from_table(i, j, by);
  }
}


}  // namespace sat
}  // namespace f360
