// sat_fuse_dev.h -- device side of "encode + sample in one pass": the row plan's bits, the
// argument block, and the helper wave that turns an owner's D rows into reduced pixels.  Used by
// the strip walker (sat_walk.h: sat_walk_kernel<.., true>) and by the band writer's one-pass form
// (sat_band_fuse.hip).
#pragma once

#include "sat_common.h"

namespace f360 {
namespace sat {

// ---- encode + sample in one pass (f360_satdec_encode_sample_frames) ---------------------------
// For callers that know the gaze of a frame before its table is built -- the reference's offline
// modes take it from a trace, src/run_satlogrectilinear.cc:932-938; its server does NOT: it
// encodes, sleeps to the tick and then reads the latest gaze (src/video_server.cc:296-303,
// 324-328,336) and therefore keeps the two calls -- a strip owner has, at every table row, the
// whole row of its strip in registers.  A reduced pixel is a box of the table,
//   (S[hi_y][hi_x] - S[lo_y][hi_x]) - (S[hi_y][lo_x] - S[lo_y][lo_x])) / area
// (src/sat_decoder_sample_rect_kernel.cl:206-217), so a strip that keeps a copy of its table row
// at row lo_y ("snapshot") can, at row hi_y, form D = row - snapshot and emit every reduced pixel
// whose two columns lie inside the strip: (D[hi_x] - D[lo_x]) / area.  The table is written
// exactly as before; what disappears is the sampler's pass over it (197 of its 233 MB per 8K
// frame are table rows read back).  Which rows snapshot and which emit is a per-frame row plan
// (walk_fuse_plan_kernel).  The gathering is not done by the strip owners -- they are the serial
// chain of the launch -- but by a helper wave per owner that takes the D rows from LDS
// (walk_fuse_helper); which pixels a strip owns the helper works out when it starts.  What this
// leaves out is finished by walk_fuse_fix_kernel: boxes that straddle two strips (three per strip
// boundary at most), from the halves the two strips' helpers export, and the reduced rows whose
// boxes overlap their neighbours' at the frame's top and bottom edge (none or one per frame),
// from the finished table.  Sources: RGB0 (source-pixel rows for the fovea, snapshot in
// registers) and planar YUV 4:2:0 (snapshot in LDS).  Null table pointers: the same launch
// without the table (f360_satdec_foveate_rect_frames).
constexpr uint32_t kFuseEmit = 1u << 31;  // row plan: this table row is the lower edge of a
                                          // reduced row (bits 0-15: which, bits 16-25: box height)
constexpr uint32_t kFuseSnap = 1u << 30;  // row plan: snapshot this table row (after emitting)
constexpr int kFuseOwners = kWalkWaves;  // strip owners per workgroup of the one-pass kernel, + as many helpers
constexpr int kFuseEntries = 3 * kStripPx;  // reduced pixels a strip can own: <= 256 per wrap class
constexpr int kFuseRawRows = kFuseEntries + kRowUnroll * 3 * kStripPx;  // (offset of the pixel rows)
constexpr int kFuseWaveDwords = kFuseRawRows + kRowUnroll * kStripPx;  // + a D row and a pixel row per batch row

struct WalkFuse {
  uint8_t *dst[kWalkFrames];
  int cxp[kWalkFrames], cyp[kWalkFrames];
  const int16_t *gx, *gy;
  uint32_t *rowplan;  // [frame][plan_stride]
  int plan_stride, out_w, out_h, dst_linesize;
  // boxes that straddle two strips: the list per frame ({count, then i, hi, lo per pixel}) and,
  // per emitted reduced row, the D values of their two columns ([row][pixel][hi | lo][3])
  uint32_t *spix, *side;
  int pmax;
  size_t side_stride;  // dwords per frame
  int lrows_max;       // reduced rows the fix-up can take from a list (more: it takes every row)
  int force_tail;      // tests ("debug.fuse_force" bit 2): a helper keeps ONE round of its pixels in
                       // registers, the rest take the tail loop (else only strips of > 320 pixels do)
  // the band writer's one pass (sat_band_fuse.hip): rows per band, the per-strip pixel lists
  int band_rows;
  uint32_t *ent;  // [frame][strip][kBandEntStride]
};
constexpr int kFixCols = 256;  // straddling pixels of a frame: <= 3 per strip boundary
constexpr int kSpixWords = 1024;  // 1 + 3 * kFixCols, rounded up; the tail: reduced rows the walk
constexpr int kSpixLrows = 800;   // cannot emit ({count, rows}), at most WalkFuse::lrows_max listed
constexpr int kFixLrowsWalk = 16;   // strip walker: rows clamped onto the frame's edges, 0 or 1 at 8K
constexpr int kFixLrowsBand = 200;  // band writer: also one box per band boundary at most
// band writer: per (frame, strip) {head: wide boxes listed, widest box, ...} | for every column
// the reduced pixel of the one-column box that ends there (byte offset in the reduced row, or
// ~0) | the wide boxes {hi column : 8 | lo column : 8 | reduced column : 16}
constexpr int kBandEntHead = 64;
constexpr int kBandEntUnit = kBandEntHead;           // word offset of the per-column map
constexpr int kBandEntWide = kBandEntHead + 256;     // word offset of the list of wide boxes
constexpr int kBandEntStride = kBandEntWide + 3 * 256;
struct WalkNoFuse {};
}  // namespace sat
// what sat_encode_impl hands to the table writer's launch in one-pass form (sat_band_fuse.hip)
struct SatBandFuse {
  sat::WalkFuse wf;
  int frame0;  // index of the launch's first frame in wf's per-frame arrays
};
namespace sat {
void launch_write_fuse(f360_ctx *ctx, hipStream_t stream, const EncodeArgs &a,
                       const EncodeBatch &eb, dim3 grid, const f360::SatBandFuse &bf, int src_kind);
template <bool FUSE> struct WalkFuseArg { typedef WalkNoFuse type; };
template <> struct WalkFuseArg<true> { typedef WalkFuse type; };

typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));

// Three bytes of a reduced pixel (the fourth is not ours, sat_decoder_sample_rect_kernel.cl:212),
// hidden from the compiler's vmcnt bookkeeping like the table stores.
__device__ __forceinline__ void fuse_store_rgb(uint8_t *row, uint32_t off, uint32_t rg,
                                               uint32_t b) {
  asm volatile(
      "global_store_short %0, %1, %3 nt\n\t"
      "global_store_byte %0, %2, %3 offset:2 nt" ::"v"(off), "v"(rg), "v"(b), "s"(row)
      : "memory");
}

// Three consecutive dwords of LDS at a dword-aligned byte address, no wait: the caller issues a
// row's worth of these and waits once.
__device__ __forceinline__ void fuse_lds_read3(uint32_t addr, u32x2 &a01, uint32_t &a2) {
  asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read_b32 %1, %2 offset:8"
               : "=&v"(a01), "=&v"(a2)
               : "v"(addr)
               : "memory");
}

// The three quotients of one box, exact; operands of 2^22 and more (boxes of > 16k pixels) take
// the integer division inline -- a call would cost the walker its register allocation.
__device__ __forceinline__ uint3 fuse_div3(uint3 n, uint32_t d) {
  if (((n.x | n.y | n.z | d) >> 22) != 0) return make_uint3(n.x / d, n.y / d, n.z / d);
  const float inv = __builtin_amdgcn_rcpf((float)d);
  return make_uint3(f360::udiv_by_rcp(n.x, inv, d), f360::udiv_by_rcp(n.y, inv, d),
                    f360::udiv_by_rcp(n.z, inv, d));
}

// The helper wave of strip owner `unit`: the reduced pixels whose box lies inside the strip --
// {hi column : 8 | lo column : 8 | reduced column : 16}, worked out once -- and then, batch by
// batch, for every row the plan marks EMIT: wait for the owner's D row in slot r (mailbox word r
// = the plan word), one pixel per lane and round, hand the slot back.  The helper waits for
// nothing but its owner, and the owner only ever waits for a slot of the batch before.
//
// A row must take the helper less than the owner takes over its own (~0.45 us), so the first
// kFuseRounds * 64 pixels of the strip (all of them, outside pathological geometries) live in
// registers as LDS offsets, and a row's gathers are all issued before the first is used: one
// LDS round trip per row, not two per round.  Quotients: n <= 255 * area and q = n / area <= 255,
// so for area <= 2048 the float product n * (1/width) * (1/height), biased by 2^-12, truncates
// to q exactly -- every rounding together moves it by < 2^-13, a true fraction is at least
// 1/2048 below the next integer (tests/test_fuse_div.py walks every case); larger boxes divide.
constexpr int kFuseRounds = 5;

template <int NR, bool PIX>
__device__ __forceinline__ void walk_fuse_rows(const EncodeArgs &a, const WalkFuse &wf,
                                               const uint32_t *plan, uint8_t *dst, int lane,
                                               const uint32_t *ent, int n_ent,
                                               const uint32_t *drows, uint32_t mbox,
                                               uint32_t max_dxw, const int (&xcol)[3],
                                               const int (&xslot)[3], int npix, uint32_t *side,
                                               int unit) {
  const bool exports_k[3] = {(bool)__any(xslot[0] >= 0), (bool)__any(xslot[1] >= 0),
                             (bool)__any(xslot[2] >= 0)};
  const bool exports = exports_k[0] || exports_k[1] || exports_k[2];
  // this lane's pixel of round k: D-row byte offsets of its two columns, box width, target
  uint32_t eoff[NR], estore[NR];
  float einv[NR];
  bool valid[NR], unit_wide[NR];  // (unit_wide: every pixel of the round is one column wide)
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int e = lane + 64 * k;
    valid[k] = e < n_ent;
    const uint32_t en = valid[k] ? ent[e] : 0x00000100u;  // (hi 0, lo 1: width taken as 1 below)
    const uint32_t hi = en & 255u, lo = (en >> 8) & 255u;
    const uint32_t dxw = valid[k] ? hi - lo : 1u;
    eoff[k] = (hi * 12u) | ((lo * 12u) << 12) | (dxw << 24);
    estore[k] = ((en >> 16) * 4u) | (hi << 22);  // (reduced column * 4 < 2^18; the hi column again)
    einv[k] = __builtin_amdgcn_rcpf((float)dxw);
    unit_wide[k] = __all(dxw == 1u);
  }
  const bool timed = a.ablate & 256;
  unsigned long long wait_cycles = 0, work_cycles = 0, rows_done = 0;
  for (int t = 0; t < a.walk_nbatches; ++t) {
    // The plan words of the batch by a scalar load with its own wait: nothing in this loop may
    // make the compiler wait on the vector memory counter -- the pixel stores are hidden from
    // it, and a vmcnt(0) for a plan word waits for the stores of the row before as well (that
    // was 800 of a row's 1600 cycles).
    u32x8 pw;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(pw)
                 : "s"(plan + (size_t)t * kRowUnroll)
                 : "memory");
    uint32_t pv = 0;
#pragma unroll
    for (int r = 0; r < kRowUnroll; ++r) pv = lane == r ? pw[r] : pv;
    for (int r = 0; r < kRowUnroll; ++r) {
      const uint32_t pr = (uint32_t)__builtin_amdgcn_readlane((int)pv, r);
      if (!(pr & kFuseEmit)) continue;
      const unsigned long long c0 = timed ? __builtin_amdgcn_s_memtime() : 0;
      // (two quick looks, then long naps: a helper of a sparse strip waits most of the time,
      // and every look costs its SIMD -- its owner's SIMD -- issue slots and an LDS access)
      for (int looks = 0; lds_read_b32(mbox + r * 4) != pr; ++looks) {
        if (looks < 2)
          __builtin_amdgcn_s_sleep(1);
        else
          __builtin_amdgcn_s_sleep(8);
      }
      const unsigned long long c1 = timed ? __builtin_amdgcn_s_memtime() : 0;
      const uint32_t dy = (pr >> 16) & 0x3ffu;
      uint8_t *orow = dst + (size_t)(pr & 0xffffu) * wf.dst_linesize;
      const uint32_t *d = drows + r * (3 * kStripPx);
      // every gather of the row -- two columns per pixel, the columns to export -- issued
      // before anything waits: left to the compiler the rounds wait one after the other, and a
      // row costs five LDS round trips instead of one (1600 against 1100 cycles for 267 boxes)
      const uint32_t dlds = (uint32_t)reinterpret_cast<uintptr_t>(d);
      u32x2 h01[NR], l01[NR], x01[3] = {};
      uint32_t h2[NR], l2[NR], x2[3] = {};
      const bool one_row = dy == 1u;
      const uint32_t plds = (uint32_t)reinterpret_cast<uintptr_t>(
          drows + kRowUnroll * 3 * kStripPx + r * kStripPx);  // the source pixels of the row
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        if (PIX && one_row && unit_wide[k]) {  // the fovea: a reduced pixel IS a source pixel
          asm volatile("ds_read_b32 %0, %1" : "=v"(h2[k]) : "v"(plds + (estore[k] >> 22) * 4u) : "memory");
        } else {
          fuse_lds_read3(dlds + (eoff[k] & 0xfffu), h01[k], h2[k]);
          fuse_lds_read3(dlds + ((eoff[k] >> 12) & 0xfffu), l01[k], l2[k]);
        }
      }
      if (exports) {
#pragma unroll
        for (int k = 0; k < 3; ++k)  // (every lane reads -- column 0 if it has nothing to export)
          if (exports_k[k]) fuse_lds_read3(dlds + (uint32_t)xcol[k] * 12u, x01[k], x2[k]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < NR; ++k)
        asm volatile("" : "+v"(h01[k]), "+v"(h2[k]), "+v"(l01[k]), "+v"(l2[k]));
#pragma unroll
      for (int k = 0; k < 3; ++k) asm volatile("" : "+v"(x01[k]), "+v"(x2[k]));
      {
        uint3 n[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k)
          n[k] = make_uint3(h01[k].x - l01[k].x, h01[k].y - l01[k].y, h2[k] - l2[k]);
        if (dy * max_dxw <= 2048u) {
          const float inv_dy = __builtin_amdgcn_rcpf((float)dy);
#pragma unroll
          for (int k = 0; k < NR; ++k) {
            if (one_row && unit_wide[k]) {
              if (PIX) {  // the posted source pixel: R, G from the low half, B from byte 2
                if (valid[k])
                  asm volatile(
                      "global_store_short %0, %1, %2 nt\n\t"
                      "global_store_byte_d16_hi %0, %1, %2 offset:2 nt" ::"v"(estore[k] & 0x3fffffu),
                      "v"(h2[k]), "s"(orow)
                      : "memory");
              } else if (valid[k]) {  // (planar sources post no pixel rows) a box of one: n itself
                fuse_store_rgb(orow, estore[k] & 0x3fffffu, n[k].x | (n[k].y << 8), n[k].z);
              }
              continue;
            }
            const float inv = einv[k] * inv_dy;
            const uint32_t qx = (uint32_t)__builtin_fmaf((float)n[k].x, inv, 0x1p-12f);
            const uint32_t qy = (uint32_t)__builtin_fmaf((float)n[k].y, inv, 0x1p-12f);
            const uint32_t qz = (uint32_t)__builtin_fmaf((float)n[k].z, inv, 0x1p-12f);
            if (valid[k]) fuse_store_rgb(orow, estore[k] & 0x3fffffu, qx | (qy << 8), qz);
          }
        } else {
#pragma unroll
          for (int k = 0; k < NR; ++k) {
            const uint3 q = fuse_div3(n[k], (eoff[k] >> 24) * dy);
            if (valid[k]) fuse_store_rgb(orow, estore[k] & 0x3fffffu, (q.x & 0xffu) | ((q.y & 0xffu) << 8), q.z);
          }
        }
        for (int e = lane + 64 * NR; e < n_ent; e += 64) {  // (NR == kFuseRounds only)
          const uint32_t en = ent[e];
          const uint32_t hi = en & 255u, lo = (en >> 8) & 255u;
          const uint32_t *ph = d + hi * 3, *pl = d + lo * 3;
          const uint3 q = fuse_div3(make_uint3(ph[0] - pl[0], ph[1] - pl[1], ph[2] - pl[2]),
                                    (hi - lo) * dy);
          fuse_store_rgb(orow, (en >> 16) * 4, (q.x & 0xffu) | ((q.y & 0xffu) << 8), q.z);
        }
      }
      if (exports) {  // this strip's columns of the boxes that straddle two strips
        uint32_t *srow = side + (size_t)(pr & 0xffffu) * npix * 6;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (xslot[k] >= 0)
            asm volatile("global_store_dwordx3 %0, %1, %2" ::"v"((uint32_t)xslot[k] * 12u),
                         "v"(u32x3v{x01[k].x, x01[k].y, x2[k]}), "s"(srow)
                         : "memory");
      }
      // (the stores above took their data from the D row: every read of it has returned)
      asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" ::"v"(mbox + r * 4), "v"(0u)
                   : "memory");
      if (timed) {
        wait_cycles += c1 - c0;
        work_cycles += __builtin_amdgcn_s_memtime() - c1;
        ++rows_done;
      }
    }
  }
  if (timed && lane == 0) {  // debug.ablate bit 8: the helper's half of the unit's statistics
    ulonglong2 *st = reinterpret_cast<ulonglong2 *>(a.walk_stats + (size_t)unit * 8 + 4);
    st[0] = make_ulonglong2(wait_cycles, work_cycles);
    st[1] = make_ulonglong2(rows_done, (unsigned long long)n_ent);
  }
}

template <bool PIX>
__device__ __forceinline__ void walk_fuse_helper(const EncodeArgs &a, const WalkFuse &wf,
                                                 int unit, int lane, uint32_t *ent,
                                                 uint32_t *box) {
  const int f = unit / a.nstrips;
  const int strip = unit - f * a.nstrips;
  const uint32_t *drows = ent + kFuseEntries;
  const uint32_t mbox = (uint32_t)reinterpret_cast<uintptr_t>(box);
  const uint32_t *plan = wf.rowplan + (size_t)f * wf.plan_stride;
  uint8_t *dst = wf.dst[f];
  const int cxp = wf.cxp[f];
  // (the boxes one column wide first, then the others, each group by reduced column: whole
  // rounds of the former take the fovea's short cut, and a round's stores stay consecutive)
  int n_ent = 0;
  uint32_t max_dxw = 1;
  for (int pass = 0; pass < 2; ++pass)
    for (int i0 = 0; i0 < wf.out_w; i0 += 64) {
      const int i = i0 + lane, ic = min(i, wf.out_w - 1);
      const f360::AxisBox bx = f360::sample_axis(cxp, wf.gx[ic + 1], wf.gx[ic], a.width, true);
      const bool own = i < wf.out_w && bx.ok && (bx.hi >> 8) == strip && (bx.lo >> 8) == strip &&
                       (bx.hi - bx.lo == 1) == (pass == 0);
      const unsigned long long m = __ballot(own);
      if (own) {
        ent[n_ent + __popcll(m & ((1ull << lane) - 1))] =
            (uint32_t)(bx.hi & 255) | ((uint32_t)(bx.lo & 255) << 8) | ((uint32_t)i << 16);
        max_dxw = max(max_dxw, (uint32_t)(bx.hi - bx.lo));
      }
      n_ent += __popcll(m);
    }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) max_dxw = max(max_dxw, (uint32_t)__shfl_xor((int)max_dxw, off, 64));
  max_dxw = (uint32_t)__builtin_amdgcn_readfirstlane((int)max_dxw);
  // the straddling boxes with a column in this strip: lane, round -> {column, slot in the row}
  const uint32_t *sp = wf.spix + (size_t)f * kSpixWords;
  int npix = (int)sp[0];
  if (npix > wf.pmax) npix = 0;  // (more than the side rows hold: the fix-up takes every row)
  int xcol[3], xslot[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int q = lane + 64 * k;
    xcol[k] = 0;
    xslot[k] = -1;
    if (q < npix) {
      const int hi = (int)sp[2 + 3 * q], lo = (int)sp[3 + 3 * q];
      if ((hi >> 8) == strip) {
        xcol[k] = hi & 255;
        xslot[k] = 2 * q;
      } else if ((lo >> 8) == strip) {
        xcol[k] = lo & 255;
        xslot[k] = 2 * q + 1;
      }
    }
  }
  uint32_t *side = wf.side + (size_t)f * wf.side_stride;
#define F360_FUSE_ROWS(NR)                                                                  \
  walk_fuse_rows<NR, PIX>(a, wf, plan, dst, lane, ent, n_ent, drows, mbox, max_dxw, xcol, xslot, \
                     npix, side, unit)
  if (n_ent <= 64 || wf.force_tail) F360_FUSE_ROWS(1);
  else if (n_ent <= 128) F360_FUSE_ROWS(2);
  else if (n_ent <= 192) F360_FUSE_ROWS(3);
  else if (n_ent <= 256) F360_FUSE_ROWS(4);
  else F360_FUSE_ROWS(kFuseRounds);
#undef F360_FUSE_ROWS
}

}  // namespace sat
}  // namespace f360
