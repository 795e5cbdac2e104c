// sat_band_fuse.hip -- encode + sample in one pass for calls too small for the strip walker:
// one frame (the per-frame loop of the reference's offline tool, run_satlogrectilinear.cc:926-938,
// where the gaze comes from a trace before the encode) up to the 22 8K frames below
// "sat.walk_units" (8 per rank when BASELINE config 4's 64 frames are sharded over 8 GPUs).
//
// The three-kernel encoder's table writer (sat_three.hip: one wave per band x strip tile) holds,
// at every row of its tile, the table row of its strip in registers -- exactly what a strip owner
// of the read-once encoder holds.  So the same arithmetic applies (sat_fuse_dev.h): a tile keeps a
// snapshot of its table row at a box's upper edge `lo`, forms D = row - snapshot at the lower edge
// `hi`, and a reduced pixel whose two columns lie in the strip is (D[hi_x] - D[lo_x]) / area
// (src/sat_decoder_sample_rect_kernel.cl:206-217, all in uint32).  The table is written exactly
// as by sat_write_kernel; what disappears is the sampler's pass over the table (233 MB of reads
// per 8K frame for 146.6 MB algorithmic).
//
// What differs from the strip walker's one pass:
//   * Tiles are independent -- no hand-off chain runs through them -- and a launch holds seven
//     waves per CU at 8K, so a tile does its own gathering (no helper wave, no mailbox): the rows
//     of the other waves on its SIMD hide it.  D rows go through the 3 KiB of LDS the table
//     row's store re-staging has just released.
//   * A tile starts with the table row above its band as its snapshot (the prologue computes
//     that row anyway), so every box whose rows lie inside one band is emitted; a box that
//     crosses a band boundary -- at most one per boundary, none in the fovea -- is left to the
//     fix-up kernel, which samples it from the finished table (walk_fuse_plan_kernel<true>).
//   * Which reduced pixels a strip owns is listed once per frame by the plan kernel, not worked
//     out per tile (a strip has 60 tiles at 8K).
// Boxes that straddle two strips and the rows the plan could not mark go through the same side
// rows and the same fix-up kernel as the strip walker's (sat_fuse_kernels.h).
#include "sat_fuse_kernels.h"
#include "sat_walk.h"

using namespace f360::sat;

namespace {

// One row of a tile turned into reduced pixels.  `dlds`: the wave's D row (3 KiB, 12 bytes per
// column), `plds`: the row's source pixels (1 KiB) -- both written by the caller just before --
// `elds`: the strip's pixel list as the prologue left it, 8 bytes per pixel:
//   {hi * 12 | lo * 12 << 12 | width << 24,  reduced column * 4 | hi << 22}
// in rounds of 64, the boxes one column wide first (`n_unit` of them).  Rounds run one after the
// other, one LDS round trip each: a tile shares its SIMD with other tiles' waves, which is what
// hides the round trips -- keeping all rounds in registers as the strip walker's helper does
// costs 300 VGPRs here, i.e. one wave per SIMD.
__device__ __forceinline__ void band_emit_row(uint32_t pr, const WalkFuse &wf, uint8_t *dst,
                                              uint32_t dlds, uint32_t plds, uint32_t elds,
                                              int lane, int n_ent, int n_unit, uint32_t max_dxw,
                                              bool exports, const int (&xcol)[3],
                                              const int (&xslot)[3], int npix, uint32_t *side) {
  const uint32_t dy = (pr >> 16) & 0x3ffu;
  const bool one_row = dy == 1u;
  uint8_t *orow = dst + (size_t)(pr & 0xffffu) * wf.dst_linesize;
  const bool quick = dy * max_dxw <= 2048u;  // the float quotient is exact (tests/test_fuse_div.py)
  const float inv_dy = __builtin_amdgcn_rcpf((float)dy);
  // Every asm statement below ends with its own wait: nothing is in flight between two of them,
  // so the compiler may move or copy their results as it likes.  A round's statement issues the
  // NEXT round's entry read together with this round's gathers -- one LDS round trip per round.
  u32x2 es;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)"
               : "=v"(es)
               : "v"(elds + (uint32_t)lane * 8u)
               : "memory");
  for (int e0 = 0; e0 < n_ent; e0 += 64) {
    const uint32_t eoff = es.x, estore = es.y;
    const bool valid = e0 + lane < n_ent;
    // (the list area is one round longer than 768 entries: the read past the last round is legal)
    const uint32_t next = elds + (uint32_t)(e0 + 64 + lane) * 8u;
    if (one_row && e0 + 64 <= n_unit) {  // the fovea: a reduced pixel IS a source pixel
      uint32_t p;
      asm volatile("ds_read_b64 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(es), "=&v"(p)
                   : "v"(next), "v"(plds + (estore >> 22) * 4u)
                   : "memory");
      asm volatile(  // R, G from the low half, B from byte 2
          "global_store_short %0, %1, %2 nt\n\t"
          "global_store_byte_d16_hi %0, %1, %2 offset:2 nt" ::"v"(estore & 0x3fffffu),
          "v"(p), "s"(orow)
          : "memory");
      continue;
    }
    u32x2 h01, l01;
    uint32_t h2, l2;
    asm volatile(
        "ds_read_b64 %0, %5\n\t"
        "ds_read2_b32 %1, %6 offset1:1\n\t"
        "ds_read_b32 %2, %6 offset:8\n\t"
        "ds_read2_b32 %3, %7 offset1:1\n\t"
        "ds_read_b32 %4, %7 offset:8\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(es), "=&v"(h01), "=&v"(h2), "=&v"(l01), "=&v"(l2)
        : "v"(next), "v"(dlds + (eoff & 0xfffu)), "v"(dlds + ((eoff >> 12) & 0xfffu))
        : "memory");
    const uint3 n = make_uint3(h01.x - l01.x, h01.y - l01.y, h2 - l2);
    const uint32_t dxw = eoff >> 24;
    if (quick) {
      const float inv = __builtin_amdgcn_rcpf((float)dxw) * inv_dy;
      const uint32_t qx = (uint32_t)__builtin_fmaf((float)n.x, inv, 0x1p-12f);
      const uint32_t qy = (uint32_t)__builtin_fmaf((float)n.y, inv, 0x1p-12f);
      const uint32_t qz = (uint32_t)__builtin_fmaf((float)n.z, inv, 0x1p-12f);
      if (valid) fuse_store_rgb(orow, estore & 0x3fffffu, qx | (qy << 8), qz);
    } else {
      const uint3 q = fuse_div3(n, dxw * dy);
      if (valid) fuse_store_rgb(orow, estore & 0x3fffffu, (q.x & 0xffu) | ((q.y & 0xffu) << 8), q.z);
    }
  }
  if (exports) {  // this strip's columns of the boxes that straddle two strips
    u32x2 x01[3];
    uint32_t x2[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
      asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read_b32 %1, %2 offset:8\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(x01[k]), "=&v"(x2[k])
                   : "v"(dlds + (uint32_t)xcol[k] * 12u)
                   : "memory");
    uint32_t *srow = side + (size_t)(pr & 0xffffu) * npix * 6;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (xslot[k] >= 0)
        asm volatile("global_store_dwordx3 %0, %1, %2" ::"v"((uint32_t)xslot[k] * 12u),
                     "v"(u32x3v{x01[k].x, x01[k].y, x2[k]}), "s"(srow)
                     : "memory");
  }
}

// The table writer of sat_three.hip (RGB0 frames, LDS-staged non-temporal stores) with the
// reduced pixels of its tile emitted on the way.  `frame0`: index of the launch's first frame in
// the call's per-frame arrays of `wf`.
template <int SRC>
__global__ __launch_bounds__(64 * kWavesPerBlock) void sat_write_fuse_kernel(
    const EncodeArgs a, const EncodeBatch eb, const WalkFuse wf, int frame0) {
  static_assert(SRC == kSrcRgb0, "the band writer's one pass takes RGB0 frames");
  // per wave: 3 KiB store staging / D row, 1 KiB source-pixel row, 6 KiB pixel list (+ one
  // round of slack: the emit loop requests the entry of the round after the last)
  constexpr int kWaveDwords = 4 * kStripPx + 2 * (kFuseEntries + 64);
  __shared__ __attribute__((aligned(16))) uint32_t stage[kWavesPerBlock * kWaveDwords];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tile = __builtin_amdgcn_readfirstlane((int)blockIdx.x * kWavesPerBlock + wave);
  if (tile >= a.nstrips * a.nbands) return;  // 1-D grid over the tiles, see sat_reduce_kernel
  F360_ENCODE_FRAME(fr, a, eb)
  const int band = tile / a.nstrips;
  const int strip = tile - band * a.nstrips;
  const int x0 = strip * kStripPx + lane * kLanePx;
  const int sb = band / a.sb_bands;
  const int f = frame0 + (a.nbatch != 0 ? (int)blockIdx.y : 0);
  const uint32_t mine = (uint32_t)reinterpret_cast<uintptr_t>(stage) + wave * kWaveDwords * 4;
  const uint32_t plds = mine + 3 * kStripPx * 4;
  const uint32_t elds = plds + kStripPx * 4;

  const int y_end = min((band + 1) * a.band_rows, a.height);
  const uint32_t *rc = a.rowcarry + fr.ws + (size_t)strip * a.height * 3;
  const uint32_t *plan = wf.rowplan + (size_t)f * wf.plan_stride;
  // A batch = kRowUnroll rows of pixels, their row carries (lanes 0..23) and their plan words
  // (lanes 32..39) in one more load, broadcast with v_readlane; no branch around any load
  auto load_batch = [&](RowBatch<SRC> &raw, uint32_t &carry, int y) {
    reduce_load_batch<SRC>(a, fr, raw, y, x0, y_end - 1);
    const uint32_t *cp = rc + min(y * 3 + min(lane, 3 * kRowUnroll - 1), a.height * 3 - 1);
    if (lane >= 32) cp = plan + min(y + min(lane - 32, kRowUnroll - 1), wf.plan_stride - 1);
    carry = *cp;
  };
  RowBatch<SRC> buf_a, buf_b;
  uint32_t carry_a, carry_b;
  const int y_begin = band * a.band_rows;
  load_batch(buf_a, carry_a, y_begin);
  load_batch(buf_b, carry_b, y_begin + kRowUnroll);

  // --- the strip's reduced pixels (plan kernel's list) and its straddling columns -----------
  const uint32_t *ent = wf.ent + ((size_t)f * a.nstrips + strip) * kBandEntStride;
  const int n_ent = (int)ent[0];
  const uint32_t max_dxw = ent[1];
  const int n_unit = (int)ent[2];
  for (int e0 = 0; e0 < n_ent; e0 += 64) {
    const int e = e0 + lane;
    const uint32_t en = e < n_ent ? ent[kBandEntHead + e] : 0x00000100u;  // (hi 0, lo 1: harmless)
    const uint32_t hi = en & 255u, lo = (en >> 8) & 255u;
    const uint32_t dxw = e < n_ent ? hi - lo : 1u;
    lds_write_b64(elds + (uint32_t)e * 8u, (hi * 12u) | ((lo * 12u) << 12) | (dxw << 24),
                  ((en >> 16) * 4u) | (hi << 22));
  }
  const uint32_t *sp = wf.spix + (size_t)f * kSpixWords;
  int npix = (int)sp[0];
  if (npix > wf.pmax) npix = 0;  // (more than the side rows hold: the fix-up takes every row)
  int xcol[3], xslot[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int q = lane + 64 * k;
    xcol[k] = 0;
    xslot[k] = -1;
    const int hi = (int)sp[2 + 3 * min(q, kFixCols - 1)], lo = (int)sp[3 + 3 * min(q, kFixCols - 1)];
    if (q < npix) {
      if ((hi >> 8) == strip) {
        xcol[k] = hi & 255;
        xslot[k] = 2 * q;
      } else if ((lo >> 8) == strip) {
        xcol[k] = lo & 255;
        xslot[k] = 2 * q + 1;
      }
    }
  }
  const bool exports = __any(xslot[0] >= 0) || __any(xslot[1] >= 0) || __any(xslot[2] >= 0);
  uint8_t *dst = wf.dst[f];
  uint32_t *side = wf.side + (size_t)f * wf.side_stride;

  // --- table row just above the band, for this lane's 4 pixels -------------
  uint32_t acc[12];
  {
    uint32_t t0[12], t1[12];
    load12(a.sbprefix + fr.ws + (size_t)sb * a.wp3 + (size_t)x0 * 3, t1);
    if (a.sb_bands == 1) {
#pragma unroll
      for (int e = 0; e < 12; ++e) acc[e] = t1[e];
    } else {
      load12(a.lp + fr.ws + (size_t)band * a.wp3 + (size_t)x0 * 3, t0);
#pragma unroll
      for (int e = 0; e < 12; ++e) acc[e] = t0[e] + t1[e];
    }
  }
  uint32_t corner[3] = {0, 0, 0};
  for (int b = lane; b < band; b += 64) {
    const uint32_t *tp = a.tprefix + fr.ws + ((size_t)strip * a.nbands + b) * 3;
    corner[0] += tp[0];
    corner[1] += tp[1];
    corner[2] += tp[2];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    corner[c] = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(corner[c]), 63);
    acc[3 + c] += acc[c];
    acc[6 + c] += acc[3 + c];
    acc[9 + c] += acc[6 + c];
    const uint32_t excl = wave_scan_incl(acc[9 + c]) - acc[9 + c] + corner[c];
    acc[c] += excl;
    acc[3 + c] += excl;
    acc[6 + c] += excl;
    acc[9 + c] += excl;
  }
  // the snapshot a box with its upper edge on the row above the band needs
  uint32_t snap[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) snap[e] = acc[e];

  const int row_dwords = a.width * 3;
  const int base = strip * kStripPx * 3;
  auto write_batch = [&](const RowBatch<SRC> &raw, uint32_t carry, int y) {
#pragma unroll
    for (int r = 0; r < kRowUnroll; ++r) {
      if (y + r >= y_end) break;
      uint32_t c[12], px[4];
      batch_pixels<SRC>(a, raw, r, px);
      unpack_px4(make_uint4(px[0], px[1], px[2], px[3]), c);
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        c[3 * k + 0] += c[3 * k - 3];
        c[3 * k + 1] += c[3 * k - 2];
        c[3 * k + 2] += c[3 * k - 1];
      }
      const uint32_t inc_rg = wave_scan_incl(c[9] | (c[10] << 16));
      const uint32_t inc_b = wave_scan_incl(c[11]);
      const uint32_t base_r =
          (inc_rg & 0xffffu) - c[9] + (uint32_t)__builtin_amdgcn_readlane((int)carry, 3 * r);
      const uint32_t base_g =
          (inc_rg >> 16) - c[10] + (uint32_t)__builtin_amdgcn_readlane((int)carry, 3 * r + 1);
      const uint32_t base_b =
          inc_b - c[11] + (uint32_t)__builtin_amdgcn_readlane((int)carry, 3 * r + 2);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[3 * k + 0] += c[3 * k + 0] + base_r;
        acc[3 * k + 1] += c[3 * k + 1] + base_g;
        acc[3 * k + 2] += c[3 * k + 2] + base_b;
      }
      uint32_t *row = fr.sat + (size_t)(y + r) * a.width * 3;
      lds_write_b128(mine + lane * 48, u32x4{acc[0], acc[1], acc[2], acc[3]});
      lds_write_b128(mine + lane * 48 + 16, u32x4{acc[4], acc[5], acc[6], acc[7]});
      lds_write_b128(mine + lane * 48 + 32, u32x4{acc[8], acc[9], acc[10], acc[11]});
      u32x4 v[3];
      lds_read3_b128(mine + lane * 16, v[0], v[1], v[2]);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int off = q * 256 + lane * 4;
        if (base + off < row_dwords)  // width % 4 == 0 -> whole 16 B in range
          global_store_b128_uncounted_nt(row + base + off, v[q]);
      }
      const uint32_t pr = (uint32_t)__builtin_amdgcn_readlane((int)carry, 32 + r);  // plan[y + r]
      if (pr & kFuseEmit) {
        // D = this row - snapshot into the staging slice (its reads have returned), the row's
        // pixels beside it where the fovea's short cut can apply
        lds_write_b128(mine + lane * 48, u32x4{acc[0] - snap[0], acc[1] - snap[1],
                                                acc[2] - snap[2], acc[3] - snap[3]});
        lds_write_b128(mine + lane * 48 + 16, u32x4{acc[4] - snap[4], acc[5] - snap[5],
                                                     acc[6] - snap[6], acc[7] - snap[7]});
        lds_write_b128(mine + lane * 48 + 32, u32x4{acc[8] - snap[8], acc[9] - snap[9],
                                                     acc[10] - snap[10], acc[11] - snap[11]});
        if (((pr >> 16) & 0x3ffu) == 1u)
          lds_write_b128(plds + lane * 16, u32x4{px[0], px[1], px[2], px[3]});
        band_emit_row(pr, wf, dst, mine, plds, elds, lane, n_ent, n_unit, max_dxw, exports, xcol,
                      xslot, npix, side);
      }
      if (pr & kFuseSnap) {
#pragma unroll
        for (int e = 0; e < 12; ++e) snap[e] = acc[e];
      }
    }
  };
  for (int y = y_begin; y < y_end; y += 2 * kRowUnroll) {
    write_batch(buf_a, carry_a, y);
    load_batch(buf_a, carry_a, y + 2 * kRowUnroll);
    if (y + kRowUnroll < y_end) write_batch(buf_b, carry_b, y + kRowUnroll);
    load_batch(buf_b, carry_b, y + 3 * kRowUnroll);
  }
}

}  // namespace

// The table writer's launch of a three-kernel encode (sat_three.hip: sat_encode_impl), one-pass
// form.  `grid` / frames as for sat_write_kernel.
void f360::sat::launch_write_fuse(f360_ctx *ctx, const EncodeArgs &a, const EncodeBatch &eb,
                                  dim3 grid, const f360::SatBandFuse &bf) {
  hipLaunchKernelGGL((sat_write_fuse_kernel<kSrcRgb0>), grid, dim3(64 * kWavesPerBlock), 0,
                     ctx->stream, a, eb, bf.wf, bf.frame0);
}

// Whether f360_satdec_encode_sample_frames can take the band writer's one pass for this call
// ("fuse.band"; the caller has already found the strip walker's form not applicable).
bool f360::sat_encode_sample_band_applies(const f360_ctx *ctx, int width, int height,
                                          int linesize, int out_w, int out_h, int dst_linesize) {
  return ctx->opt_fuse_band != 0 && linesize / width == 4 && linesize % 16 == 0 &&
         width % 4 == 0 && width <= f360::kMaxDim &&
         (size_t)width * height * 3 < ((size_t)1 << 31) && out_w < 65536 && out_h < 65536 &&
         (width + kStripPx - 1) / kStripPx <= kFixCols / 4 && dst_linesize % 4 == 0 &&
         dst_linesize >= 4 * out_w;
}

// f360_satdec_encode_sample_frames on the three-kernel encoder: one plan launch and one fix-up
// launch per (up to kWalkFrames) frames, between them the encoder's launches of "sat.batch_mb"
// each -- reducer, carry pass, and the table writer in its one-pass form.
int f360::sat_encode_sample_band(f360_ctx *ctx, int count, uint32_t *const *sats,
                                 const uint8_t *const *srcs, int width, int height, int linesize,
                                 const f360::SatFuse &fuse, bool prof) {
  f360::SatEncodePlan &p = ctx->enc;
  const int nstrips = (width + kStripPx - 1) / kStripPx;
  const int plan_stride = ((height + kRowUnroll - 1) / kRowUnroll) * kRowUnroll;
  const int pmax = (ctx->opt_fuse_force & 1) ? 1 : std::max(1, std::min(3 * (nstrips - 1), kFixCols));
  const size_t side_stride = (size_t)fuse.out_h * pmax * 6;  // dwords per frame
  const size_t frame_bytes = (size_t)linesize * height;
  const int per_launch = (int)std::min<size_t>(
      std::max<size_t>(((size_t)std::max(ctx->opt_batch_mb, 1) << 20) / frame_bytes, 1),
      (size_t)kEncBatch);
  for (int c0 = 0; c0 < count; c0 += kWalkFrames) {
    const int n = std::min(count - c0, kWalkFrames);
    const size_t words = (size_t)n * plan_stride + (size_t)n * kSpixWords + (size_t)n * side_stride +
                         (size_t)n * nstrips * kBandEntStride;
    if (words * 4 > p.walk_plan.bytes) {
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      F360_HIP_TRY(hipStreamIsCapturing(ctx->stream, &cap));
      F360_REQUIRE(cap == hipStreamCaptureStatusNone,
                   "f360_satdec_encode_sample_frames: the plan buffers (%zu bytes) must be "
                   "allocated but the stream is being captured; run the same call once before "
                   "the capture", words * 4);
      if (p.walk_plan.p) F360_HIP_TRY(hipStreamSynchronize(ctx->stream));
      int st = p.walk_plan.reserve(words * 4);
      if (st != F360_OK) return st;
    }
    f360::SatBandFuse bf;
    WalkFuse &wf = bf.wf;
    WalkBatch wb;
    walk_fill_batch(wb, c0, n, sats, srcs, nullptr);
    for (int k = 0; k < kWalkFrames; ++k) {
      const int q = c0 + (k < n ? k : 0);
      wf.dst[k] = fuse.dsts[q];
      wf.cxp[k] = (int)(fuse.centers_xy[2 * q] * (float)width);  // sat_decoder_sample_rect_kernel.cl:176-179
      wf.cyp[k] = (int)(fuse.centers_xy[2 * q + 1] * (float)height);
    }
    wf.gx = fuse.gx;
    wf.gy = fuse.gy;
    wf.rowplan = p.walk_plan.as<uint32_t>();
    wf.plan_stride = plan_stride;
    wf.out_w = fuse.out_w;
    wf.out_h = fuse.out_h;
    wf.dst_linesize = fuse.dst_linesize;
    wf.spix = wf.rowplan + (size_t)n * plan_stride;
    wf.side = wf.spix + (size_t)n * kSpixWords;
    wf.pmax = pmax;
    wf.side_stride = side_stride;
    wf.lrows_max = (ctx->opt_fuse_force & 2) ? 0 : kFixLrowsBand;
    wf.ent = wf.side + (size_t)n * side_stride;
    // (the band height is the encoder plan's: make sure it exists before the plan kernel runs)
    int st = f360_sat_encode_prepare(ctx, width, height);
    if (st != F360_OK) return st;
    wf.band_rows = p.band_rows;
    {
      f360::KernelSpan span(ctx, f360::kWalkFusePlan, prof, n);
      hipLaunchKernelGGL(walk_fuse_plan_kernel<true>, dim3(n), dim3(256), 0, ctx->stream, wf.gy,
                         wf.out_h, width, height, wf.rowplan, plan_stride, wf);
    }
    for (int k0 = 0; k0 < n; k0 += per_launch) {
      const int m = std::min(n - k0, per_launch);
      bf.frame0 = k0;
      st = f360::sat_encode_impl(ctx, nullptr, nullptr, width, height, linesize, nullptr, nullptr,
                                 m, sats + c0 + k0, srcs + c0 + k0, prof ? 1 : 0, nullptr, &bf);
      if (st != F360_OK) return st;
      if (ctx->enc.band_rows != wf.band_rows) {  // (cannot happen: same geometry, same options)
        f360::set_error("f360_satdec_encode_sample_frames: the encoder plan changed under the call");
        return F360_ERR_INVALID_ARG;
      }
    }
    {
      f360::KernelSpan span(ctx, f360::kWalkFuseFix, prof, n);
      hipLaunchKernelGGL(walk_fuse_fix_kernel<0>,
                         dim3((wf.out_h * pmax + 255) / 256 +
                                  kFixLrowsBand * ((wf.out_w + 255) / 256), n),
                         dim3(256), 0, ctx->stream, wb, wf, width, height, linesize, -1,
                         f360::YuvPlanes{nullptr, nullptr, nullptr, 0, 0, 0}, f360::YuvConsts{});
    }
  }
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}
