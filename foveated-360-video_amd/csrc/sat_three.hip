// sat_three.hip -- summed-area-table encode for gfx950 (MI355X).
//
// Replaces SATEncoder::EncodeFrameGPU (src/sat_encoder.cc:67-135) and its three
// OpenCL kernels copy_image / scan_rows / scan_columns
// (src/sat_encoder_encode_kernels.cl:1-20,44-58,60-74).  Output is bit-identical:
// uint32 addition is associative mod 2^32, so any summation order gives the
// reference's integers.
//
// Design (DESIGN.md "SAT encode"): reduce -> carry scan -> write.
//   A wave owns a strip of 256 pixels (64 lanes x 4 RGB0 pixels = one 16-byte
//   load per lane and row).  The frame is cut into bands of `band_rows` rows
//   and super-bands of `sb_bands` bands.
//   K1 sat_reduce : one wave per (strip, super-band) walks down the super-band
//                   and emits the column sums above each band (inside the
//                   super-band), the super-band column sums, every row's
//                   strip sum and every tile's sum.        reads 4 B/px
//   K2 sat_carry  : exclusive prefixes of those small arrays across
//                   super-bands / across strips.           ~1 % of the traffic
//   K3 sat_write  : one wave per (strip, band) re-reads its pixels, rebuilds
//                   the row prefix with a DPP wave scan, adds the carried-in
//                   column / row / corner sums and writes the final uint32x3
//                   table once.                     reads 4 B/px, writes 12 B/px
//   Total HBM traffic ~20.6 B/px against 16 B/px compulsory (the reference's
//   three passes move 64 B/px).
#include "sat_common.h"
#include "sat_walk.h"
#include "sat_reduce.h"

using namespace f360::sat;

namespace {

// ---- K2: exclusive prefixes of the carry arrays ------------------------------
// out[k][i] = sum_{k' < k} in[k'][i]   for i < n, k < K
struct ScanSeg {
  const uint32_t *in;
  uint32_t *out;
  int n, K, nblocks;
};
// (a batched launch: blockIdx.y = frame, whose arrays start `frame_stride` elements apart)

// One round of at most 32 loads per thread: a segment with more than 32 rows is split into
// `parts` row ranges handled by different threads of the workgroup (the workgroup then covers
// 256 / parts columns); a part's base is the sum of the parts before it, passed through LDS.
// (Two rounds in one thread: 8.8 us for the 60 super-bands of an 8K frame; 64 loads in flight: 11.4.)
__device__ __forceinline__ void carry_scan_segment(const ScanSeg &s, int blk, uint32_t *totals) {
  if (s.K > 128) {  // very tall / wide frames: rounds of 32 in one thread
    const int i = blk * 256 + (int)threadIdx.x;
    if (i >= s.n) return;
    uint32_t run = 0;
    for (int k = 0; k < s.K; k += 32) {
      uint32_t t[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) t[q] = (k + q < s.K) ? s.in[(size_t)(k + q) * s.n + i] : 0u;
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        if (k + q < s.K) s.out[(size_t)(k + q) * s.n + i] = run;
        run += t[q];
      }
    }
    return;
  }
  const int parts = (s.K + 31) / 32;           // 1, 2 (8K), 3 or 4
  const int cols = 256 / parts;                // columns per workgroup
  const int part = (int)threadIdx.x / cols, c = (int)threadIdx.x - part * cols;
  const int i = blk * cols + c;
  const int rows = (s.K + parts - 1) / parts;  // rows per part, <= 32
  const int k0 = part * rows, k1 = min(k0 + rows, s.K);
  const bool live = part < parts && i < s.n;
  uint32_t t[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) t[q] = (live && k0 + q < k1) ? s.in[(size_t)(k0 + q) * s.n + i] : 0u;
  uint32_t base = 0;
  if (parts > 1) {  // wave-uniform
    uint32_t sum = 0;
#pragma unroll
    for (int q = 0; q < 32; ++q) sum += t[q];
    totals[threadIdx.x] = sum;
    __syncthreads();
    for (int p = 0; p < part; ++p) base += totals[p * cols + c];
  }
  if (!live) return;
  uint32_t run = base;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    if (k0 + q < k1) s.out[(size_t)(k0 + q) * s.n + i] = run;
    run += t[q];
  }
}

__global__ __launch_bounds__(256) void sat_carry_kernel(ScanSeg a, ScanSeg b, ScanSeg c,
                                                        size_t frame_stride) {
  __shared__ uint32_t totals[256];
  const size_t shift = (size_t)blockIdx.y * frame_stride;
  a.in += shift; a.out += shift;
  b.in += shift; b.out += shift;
  c.in += shift; c.out += shift;
  int blk = blockIdx.x;
  if (blk < a.nblocks) {
    carry_scan_segment(a, blk, totals);
    return;
  }
  blk -= a.nblocks;
  if (blk < b.nblocks) {
    carry_scan_segment(b, blk, totals);
    return;
  }
  blk -= b.nblocks;
  carry_scan_segment(c, blk, totals);
}

// ---- K3: final table ----------------------------------------------------------
// STORE 0: three 16-byte stores per lane at a 48-byte lane stride.
// STORE 1: re-stage the row through wave-private LDS so that each store
//          instruction writes 1 KiB contiguous.
template <int SRC, int STORE>
__global__ __launch_bounds__(64 * kWavesPerBlock) void sat_write_kernel(
    const EncodeArgs a, const EncodeBatch eb) {
  constexpr bool VEC = SRC != kSrcBytes;  // 16-byte accesses allowed (width % 4 == 0, aligned)
  __shared__ __attribute__((aligned(16))) uint32_t
      stage[STORE >= 1 ? kWavesPerBlock * 3 * kStripPx : 4];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  int tile = __builtin_amdgcn_readfirstlane((int)blockIdx.x * kWavesPerBlock + wave);
  if (tile >= a.nstrips * a.nbands) return;  // 1-D grid over the tiles, see sat_reduce_kernel
  F360_ENCODE_FRAME(fr, a, eb)
  const int band = tile / a.nstrips;
  const int strip = tile - band * a.nstrips;
  const int x0 = strip * kStripPx + lane * kLanePx;
  const int sb = band / a.sb_bands;
  int xm[4] = {-1, -1, -1, -1};
  bool dense = false;  // every pixel of the wave is a lattice column, ranks consecutive
  int xm_first = 0;
  if (STORE == 2) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (x0 + k < a.width) xm[k] = a.xmap[x0 + k];
    xm_first = __builtin_amdgcn_readfirstlane(xm[0]);
    dense = __all(xm[0] == xm_first + 4 * lane && xm[1] == xm[0] + 1 && xm[2] == xm[0] + 2 &&
                  xm[3] == xm[0] + 3 && xm_first >= 0);
  }

  const int y_end = min((band + 1) * a.band_rows, a.height);
  const uint32_t *rc = a.rowcarry + fr.ws + (size_t)strip * a.height * 3;
  // A batch = kRowUnroll rows of pixels plus their row carries (3 dwords per row, fetched by
  // lanes 0..23 in one load and broadcast with v_readlane: a per-row load of a wave-uniform
  // address would be one more vector-memory operation to wait for in every row).  No branch
  // around any load (addresses are clamped to the band's last row): see sat_reduce_kernel.
  auto load_batch = [&](RowBatch<SRC> &raw, uint32_t &carry, int y) {
    if constexpr (SRC == kSrcBytes) {
#pragma unroll
      for (int r = 0; r < kRowUnroll; ++r)
        raw.raw[r] = (y + r < y_end)
                         ? load_px4<SRC>(fr.src, a.width, y + r, x0, a.linesize, a.bpp)
                         : make_uint4(0, 0, 0, 0);
    } else {
      reduce_load_batch<SRC>(a, fr, raw, y, x0, y_end - 1);
    }
    const uint32_t *cp = rc + min(y * 3 + min(lane, 3 * kRowUnroll - 1), a.height * 3 - 1);
    if (STORE == 2 && lane >= 32)  // emit mode: lanes 32..39 fetch the rows' compact indices
      cp = reinterpret_cast<const uint32_t *>(a.ymap) +
           min(y + min(lane - 32, kRowUnroll - 1), a.height - 1);
    carry = *cp;
  };
  // The band's first two batches of pixels are requested BEFORE the prologue's own loads (the
  // carried-in column sums, the corner look-back): those used to be three memory round trips in
  // a row before the first pixel load was even issued; now everything is in flight together.
  RowBatch<SRC> buf_a, buf_b;
  uint32_t carry_a, carry_b;
  const int y_begin = band * a.band_rows;
  load_batch(buf_a, carry_a, y_begin);
  load_batch(buf_b, carry_b, y_begin + kRowUnroll);

  // --- table row just above the band, for this lane's 4 pixels -------------
  uint32_t acc[12];
  {
    uint32_t t0[12], t1[12];
    load12(a.sbprefix + fr.ws + (size_t)sb * a.wp3 + (size_t)x0 * 3, t1);
    if (a.sb_bands == 1) {  // a band is its own super-band: nothing above it inside
#pragma unroll
      for (int e = 0; e < 12; ++e) acc[e] = t1[e];
    } else {
      load12(a.lp + fr.ws + (size_t)band * a.wp3 + (size_t)x0 * 3, t0);
#pragma unroll
      for (int e = 0; e < 12; ++e) acc[e] = t0[e] + t1[e];
    }
  }
  // corner: every tile above and to the left
  uint32_t corner[3] = {0, 0, 0};
  for (int b = lane; b < band; b += 64) {
    const uint32_t *tp = a.tprefix + fr.ws + ((size_t)strip * a.nbands + b) * 3;
    corner[0] += tp[0];
    corner[1] += tp[1];
    corner[2] += tp[2];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    corner[c] = (uint32_t)__builtin_amdgcn_readlane(
        (int)wave_scan_incl(corner[c]), 63);
    // prefix along x of the column sums: inside the lane, then across lanes
    acc[3 + c] += acc[c];
    acc[6 + c] += acc[3 + c];
    acc[9 + c] += acc[6 + c];
    const uint32_t excl = wave_scan_incl(acc[9 + c]) - acc[9 + c] + corner[c];
    acc[c] += excl;
    acc[3 + c] += excl;
    acc[6 + c] += excl;
    acc[9 + c] += excl;
  }

  auto write_batch = [&](const RowBatch<SRC> &raw, uint32_t carry, int y) {
#pragma unroll
    for (int r = 0; r < kRowUnroll; ++r) {
      if (y + r >= y_end) break;
      uint32_t c[12];
      {
        uint32_t v[4];
        batch_pixels<SRC>(a, raw, r, v);
        unpack_px4(make_uint4(v[0], v[1], v[2], v[3]), c);
      }
      // inclusive prefix over the lane's 4 pixels
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
      if (STORE == 2) {
        const int py = __builtin_amdgcn_readlane((int)carry, 32 + r);  // a.ymap[y + r]
        if (py >= 0) {
          uint32_t *crow = a.corners + (size_t)py * a.corner_stride * 3;
          if (dense) {
            // the wave's 768 dwords are contiguous in the compact row: re-stage through LDS as
            // the table writer does and store 1 KiB per instruction (4-byte aligned only)
            const uint32_t mine =
                (uint32_t)reinterpret_cast<uintptr_t>(stage) + wave * 3 * kStripPx * 4;
            lds_write_b128(mine + lane * 48, u32x4{acc[0], acc[1], acc[2], acc[3]});
            lds_write_b128(mine + lane * 48 + 16, u32x4{acc[4], acc[5], acc[6], acc[7]});
            lds_write_b128(mine + lane * 48 + 32, u32x4{acc[8], acc[9], acc[10], acc[11]});
            u32x4 v[3];
            lds_read3_b128(mine + lane * 16, v[0], v[1], v[2]);
            uint32_t *dst = crow + (size_t)xm_first * 3;
#pragma unroll
            for (int q = 0; q < 3; ++q)
              global_store_b128_uncounted(dst + q * 256 + lane * 4, v[q]);
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (xm[k] >= 0) {  // one 12-byte store per lattice texel
                global_store_b96_uncounted(crow + (size_t)xm[k] * 3, acc[3 * k + 0],
                                           acc[3 * k + 1], acc[3 * k + 2]);
              }
          }
        }
        continue;
      }
      uint32_t *row = fr.sat + (size_t)(y + r) * a.width * 3;
      if (VEC && STORE == 0) {
        if (x0 < a.width) store12(row + (size_t)x0 * 3, acc);
      } else if (VEC && STORE == 1) {
        // Lanes exchange data through wave-private LDS.  One wave's LDS
        // operations execute in order, so no s_barrier is needed; the accesses
        // are inline asm because the compiler's memory model is per lane (it
        // deletes plain LDS stores that only OTHER lanes read back).
        const uint32_t mine =
            (uint32_t)reinterpret_cast<uintptr_t>(stage) + wave * 3 * kStripPx * 4;
        lds_write_b128(mine + lane * 48, u32x4{acc[0], acc[1], acc[2], acc[3]});
        lds_write_b128(mine + lane * 48 + 16, u32x4{acc[4], acc[5], acc[6], acc[7]});
        lds_write_b128(mine + lane * 48 + 32, u32x4{acc[8], acc[9], acc[10], acc[11]});
        u32x4 v[3];
        lds_read3_b128(mine + lane * 16, v[0], v[1], v[2]);
        const int row_dwords = a.width * 3;
        const int base = strip * kStripPx * 3;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int off = q * 256 + lane * 4;
          if (base + off < row_dwords)  // width % 4 == 0 -> whole 16 B in range
            global_store_b128_uncounted_nt(row + base + off, v[q]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (x0 + k < a.width) {
            row[(size_t)(x0 + k) * 3 + 0] = acc[3 * k + 0];
            row[(size_t)(x0 + k) * 3 + 1] = acc[3 * k + 1];
            row[(size_t)(x0 + k) * 3 + 2] = acc[3 * k + 2];
          }
      }
    }
  };
  // two batches alternate (a band is 2, 4 or 8 batches): the next one is in flight while this
  // one is scanned and stored
  for (int y = y_begin; y < y_end; y += 2 * kRowUnroll) {
    write_batch(buf_a, carry_a, y);
    load_batch(buf_a, carry_a, y + 2 * kRowUnroll);
    if (y + kRowUnroll < y_end) write_batch(buf_b, carry_b, y + kRowUnroll);
    load_batch(buf_b, carry_b, y + 3 * kRowUnroll);
  }
}

int ensure_plan(f360_ctx *ctx, int width, int height, bool planar = false, int frames = 1) {
  f360::SatEncodePlan &p = ctx->enc;
  // band height: the largest of 64 / 32 / 16 rows that still yields enough tiles (one wave
  // each in the writer) to fill 256 CUs -- 64 at 7680x3840, 16 at 3840x1920 and below
  int band_rows = ctx->opt_band_rows;
  if (band_rows == 0) {
    const int strips = (width + kStripPx - 1) / kStripPx;
    band_rows = 16;
    for (int cand : {64, 32})
      if ((long)strips * ((height + cand - 1) / cand) >= 1500) {
        band_rows = cand;
        break;
      }
  }
  // bands per reducer wave: at least 2, and few enough super-bands (<= 32) that the carry
  // kernel needs a single round of loads
  // (planar sources convert in the reducer, which makes it instruction-bound: one band per
  // wave doubles the waves, 45 -> 34 us at 8K)
  int sb_bands = ctx->opt_sb_bands;
  // (with 64-row bands one band per reducer wave also makes the reducer visit the frame in the
  // writer's tile order, and the writer's re-read then finds more of it in the caches: 80 -> 76 us
  // at 8K for +1.7 us in the carry kernel; with 16-row bands two bands per wave stay better)
  // (a small frame has too few reducer waves to fill the device with two bands each: 1080p
  // 272 waves of 32 rows against 544 of 16 -- reducer 11.6 -> 8.4 us, 11.8 -> 9.0 at 2560x1440)
  if (sb_bands < 0)
    sb_bands = (planar || band_rows == 64 || (long)width * height <= 4200000L) ? 1 : 2;
  if (sb_bands == 0) {
    const int nb = (height + band_rows - 1) / band_rows;
    sb_bands = (nb + 31) / 32;
    if (sb_bands < 2) sb_bands = 2;
  }
  // The scratch is carved for the layout with the most super-bands (one band each), so callers
  // that alternate planar and RGB0 sources on one context (different sb_bands at <= 4K) only
  // change two numbers: no re-carve, no synchronisation.
  if (p.width == width && p.height == height && p.band_rows == band_rows && p.ws.p &&
      p.frames >= frames) {
    p.sb_bands = sb_bands;
    p.nsb = (p.nbands + sb_bands - 1) / sb_bands;
    return F360_OK;
  }
  frames = std::max(frames, p.width == width && p.height == height ? p.frames : 1);
  // A geometry change (or more frames per launch group) re-carves the scratch: it allocates and
  // waits for work that may use the old one -- both illegal while the stream is being captured
  // into a graph, so say so instead of breaking the capture (warm up eagerly first).
  {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    F360_HIP_TRY(hipStreamIsCapturing(ctx->stream, &cap));
    F360_REQUIRE(cap == hipStreamCaptureStatusNone,
                 "f360_sat_encode: the encoder's scratch for %dx%d x %d frames must be allocated "
                 "but the stream is being captured; run the same call once before the capture",
                 width, height, frames);
  }
  if (p.ws.p) F360_HIP_TRY(hipStreamSynchronize(ctx->stream));
  p.width = width;
  p.height = height;
  p.band_rows = band_rows;
  p.sb_bands = sb_bands;
  p.nstrips = (width + kStripPx - 1) / kStripPx;
  p.nbands = (height + p.band_rows - 1) / p.band_rows;
  p.nsb = (p.nbands + p.sb_bands - 1) / p.sb_bands;
  p.wp3 = p.nstrips * kStripPx * 3;
  auto align = [](size_t n) { return (n + 63) & ~(size_t)63; };
  const size_t n_lp = align((size_t)p.nbands * p.wp3);
  const size_t n_sb = align((size_t)p.nbands * p.wp3);  // room for sb_bands = 1
  const size_t n_row = align((size_t)p.nstrips * height * 3);
  const size_t n_tile = align((size_t)p.nstrips * p.nbands * 3);
  const size_t total = n_lp + 2 * n_sb + 2 * n_row + 2 * n_tile;
  int st = p.ws.reserve(total * sizeof(uint32_t) * (size_t)frames);
  if (st != F360_OK) {
    p.width = p.height = 0;
    p.frames = 0;
    return st;
  }
  p.frames = frames;
  p.ws_stride = total;
  uint32_t *w = p.ws.as<uint32_t>();
  p.lp = w;             w += n_lp;
  p.sbtotal = w;        w += n_sb;
  p.sbprefix = w;       w += n_sb;
  p.rowsum = w;         w += n_row;
  p.rowcarry = w;       w += n_row;
  p.tiletotal = w;      w += n_tile;
  p.tprefix = w;
  return F360_OK;
}

}  // namespace

extern "C" int f360_sat_encode_batch_max(void) { return kEncBatch; }

extern "C" int f360_sat_encode_prepare(f360_ctx *ctx, int width, int height) {
  F360_REQUIRE(ctx, "f360_sat_encode_prepare: null context");
  F360_REQUIRE(width >= 1 && height >= 1, "f360_sat_encode_prepare: bad size %dx%d",
               width, height);
  F360_BIND_DEVICE(ctx);
  return ensure_plan(ctx, width, height);
}

namespace f360 {

int sat_encode_reserve(f360_ctx *ctx, int width, int height, int frames, bool planar) {
  return ensure_plan(ctx, width, height, planar, frames);
}

int sat_encode_impl(f360_ctx *ctx, uint32_t *sat_dev, const uint8_t *src_dev, int width,
                    int height, int linesize, const SatEmit *emit, const YuvPlanes *yuv,
                    int count, uint32_t *const *sats, const uint8_t *const *srcs, int profile,
                    const YuvPlanes *yuvs, const SatBandFuse *band_fuse, const SatLaunch *where) {
  F360_REQUIRE(ctx, "f360_sat_encode: null context");
  F360_BIND_DEVICE(ctx);
  if (count > 0) {
    F360_REQUIRE(count <= kEncBatch && sats && (srcs || yuvs) && !emit && !yuv,
                 "f360_sat_encode_batch: count %d outside 1..%d, or null arrays", count, kEncBatch);
    for (int k = 0; k < count; ++k) {
      F360_REQUIRE(sats[k] && (yuvs ? yuvs[k].y && yuvs[k].u && yuvs[k].v : srcs[k] != nullptr),
                   "f360_sat_encode_batch: null buffer %d", k);
      if (yuvs)  // one set of linesizes and alignments for the whole launch
        F360_REQUIRE(yuvs[k].y_linesize == yuvs[0].y_linesize &&
                         yuvs[k].u_linesize == yuvs[0].u_linesize &&
                         yuvs[k].v_linesize == yuvs[0].v_linesize &&
                         ((uintptr_t)yuvs[k].y % 4) == 0 && ((uintptr_t)yuvs[k].u % 2) == 0 &&
                         ((uintptr_t)yuvs[k].v % 2) == 0 && ((uintptr_t)sats[k] % 16) == 0,
                     "f360_sat_encode_yuv420p_batch: frame %d: other linesizes than frame 0, "
                     "or a misaligned plane / table", k);
    }
    sat_dev = sats[0];
    if (yuvs) yuv = &yuvs[0];
    else src_dev = srcs[0];
  }
  F360_REQUIRE((sat_dev || emit) && (src_dev || yuv), "f360_sat_encode: null buffer");
  F360_REQUIRE(width >= 1 && height >= 1, "f360_sat_encode: bad size %dx%d", width,
               height);
  const int bpp = yuv ? 4 : linesize / width;  // src/sat_encoder_encode_kernels.cl:9
  F360_REQUIRE(bpp >= 3, "f360_sat_encode: linesize %d gives %d bytes per pixel (need >= 3)",
               linesize, bpp);
  if (yuv) {
    F360_REQUIRE(yuv->y && yuv->u && yuv->v, "f360_sat_encode_yuv420p: null plane");
    // the planar path loads 4 luma bytes and 2 + 2 chroma bytes per lane and row
    F360_REQUIRE(width % 4 == 0 && height % 2 == 0,
                 "f360_sat_encode_yuv420p: size %dx%d (need width %% 4 == 0, even height)",
                 width, height);
    F360_REQUIRE(yuv->y_linesize >= width && yuv->u_linesize >= width / 2 &&
                     yuv->v_linesize >= width / 2 && yuv->y_linesize % 4 == 0 &&
                     yuv->u_linesize % 2 == 0 && yuv->v_linesize % 2 == 0,
                 "f360_sat_encode_yuv420p: linesizes %d/%d/%d (need >= row, y %% 4, u,v %% 2)",
                 yuv->y_linesize, yuv->u_linesize, yuv->v_linesize);
    F360_REQUIRE(((uintptr_t)yuv->y % 4) == 0 && ((uintptr_t)yuv->u % 2) == 0 &&
                     ((uintptr_t)yuv->v % 2) == 0 && (emit || ((uintptr_t)sat_dev % 16) == 0),
                 "f360_sat_encode_yuv420p: misaligned plane or table");
  }
  F360_REQUIRE((size_t)width * height * 3 < ((size_t)1 << 31),
               "f360_sat_encode: frame too large for 32-bit element indices");
  const int slots = where ? where->slots : 1, slot = where ? where->slot : 0;
  hipStream_t stream = where && where->stream ? where->stream : ctx->stream;
  const int slot_frames = where && where->slot_frames > 0 ? where->slot_frames : (count > 0 ? count : 1);
  int st = ensure_plan(ctx, width, height, yuv != nullptr, slot_frames * slots);
  if (st != F360_OK) return st;
  const f360::SatEncodePlan &p = ctx->enc;
  // (a pipelined call's launch groups alternate between slices of the scratch: group g + 1's
  // reducer writes while group g's table writer still reads)
  const size_t ws_off = (size_t)slot * slot_frames * p.ws_stride;

  EncodeArgs a;
  a.sat = sat_dev;
  a.src = src_dev;
  a.width = width;
  a.height = height;
  a.linesize = linesize;
  a.bpp = bpp;
  a.band_rows = p.band_rows;
  a.sb_bands = p.sb_bands;
  a.nstrips = p.nstrips;
  a.nbands = p.nbands;
  a.nsb = p.nsb;
  a.wp3 = p.wp3;
  a.lp = p.lp + ws_off;
  a.sbtotal = p.sbtotal + ws_off;
  a.sbprefix = p.sbprefix + ws_off;
  a.rowsum = p.rowsum + ws_off;
  a.rowcarry = p.rowcarry + ws_off;
  a.tiletotal = p.tiletotal + ws_off;
  a.tprefix = p.tprefix + ws_off;
  a.ablate = ctx->opt_ablate;
  a.xmap = emit ? emit->xmap : nullptr;
  a.ymap = emit ? emit->ymap : nullptr;
  a.corners = emit ? emit->corners : nullptr;
  a.corner_stride = emit ? emit->corner_stride : 0;
  a.has_maps = emit && emit->maps ? 1 : 0;
  if (a.has_maps) a.maps = *emit->maps;
  else a.maps = FovMaps{};
  a.yuv = yuv ? *yuv : YuvPlanes{nullptr, nullptr, nullptr, 0, 0, 0};
  if (yuv)
    build_yuv2rgb_consts(a.k);
  else
    a.k = YuvConsts{};

  a.nbatch = count;
  a.ws_stride = p.ws_stride;
  EncodeBatch eb;
  for (int k = 0; k < kEncBatch; ++k) {
    const int q = k < count ? k : 0;
    eb.src[k] = count <= 0 ? nullptr : yuvs ? yuvs[q].y : srcs[q];
    eb.sat[k] = count > 0 ? sats[q] : nullptr;
    eb.u[k] = count > 0 && yuvs ? yuvs[q].u : nullptr;
    eb.v[k] = count > 0 && yuvs ? yuvs[q].v : nullptr;
  }
  const unsigned frames = count > 0 ? (unsigned)count : 1u;

  const bool prof = profile < 0 ? f360::take_profile_slot(ctx) : profile != 0;
  bool vec = !yuv && bpp == 4 && (width % 4) == 0 && (linesize % 16) == 0 &&
             ((uintptr_t)src_dev % 16) == 0 && (emit || ((uintptr_t)sat_dev % 16) == 0);
  for (int k = 1; k < count && !yuvs; ++k)  // one kernel flavour for the whole batch
    vec = vec && ((uintptr_t)srcs[k] % 16) == 0 && ((uintptr_t)sats[k] % 16) == 0;
  const int yuv_src = !yuv ? 0 : ctx->opt_yuv_model == 1 ? kSrcYuvSwsX86 : kSrcYuvSwsC;
  const dim3 block(64 * kWavesPerBlock);
  const int reduce_blocks = (p.nstrips * p.nsb + kWavesPerBlock - 1) / kWavesPerBlock;
  a.reduce_blocks = reduce_blocks;
  const int blocks1 = reduce_blocks + (a.has_maps ? 2 : 0);

  {
    f360::KernelSpan span(ctx, f360::kSatReduce, prof, (int)frames, stream);
    if (yuv_src == kSrcYuvSwsX86)
      hipLaunchKernelGGL(sat_reduce_kernel<kSrcYuvSwsX86>, dim3(blocks1, frames), block, 0, stream, a, eb);
    else if (yuv_src == kSrcYuvSwsC)
      hipLaunchKernelGGL(sat_reduce_kernel<kSrcYuvSwsC>, dim3(blocks1, frames), block, 0, stream, a, eb);
    else if (vec)
      hipLaunchKernelGGL(sat_reduce_kernel<kSrcRgb0>, dim3(blocks1, frames), block, 0, stream, a, eb);
    else
      hipLaunchKernelGGL(sat_reduce_kernel<kSrcBytes>, dim3(blocks1, frames), block, 0, stream, a, eb);
  }

  if (ctx->opt_ablate & 8) return F360_OK;  // timing experiments: reducer only
  auto seg = [](const uint32_t *in, uint32_t *out, int n, int K) {
    const int parts = K > 128 ? 1 : (K + 31) / 32, cols = 256 / (parts < 1 ? 1 : parts);
    return ScanSeg{in, out, n, K, (n + cols - 1) / cols};
  };
  ScanSeg sa = seg(a.sbtotal, a.sbprefix, p.wp3, p.nsb);
  ScanSeg sb = seg(a.rowsum, a.rowcarry, height * 3, p.nstrips);
  ScanSeg sc = seg(a.tiletotal, a.tprefix, p.nbands * 3, p.nstrips);
  {
    f360::KernelSpan span(ctx, f360::kSatCarry, prof, (int)frames, stream);
    hipLaunchKernelGGL(sat_carry_kernel,
                       dim3(sa.nblocks + sb.nblocks + sc.nblocks, frames), dim3(256), 0,
                       stream, sa, sb, sc, p.ws_stride);
  }
  if (band_fuse) {  // the writer also emits the reduced pixels of its tile (sat_band_fuse.hip)
    F360_REQUIRE((vec || yuv_src) && !emit && count > 0,
                 "sat_encode_impl: the one-pass writer takes batches of aligned RGB0 frames or planes");
    f360::KernelSpan span(ctx, f360::kSatWriteFuse, prof, (int)frames, stream);
    launch_write_fuse(ctx, stream, a, eb,
                      dim3((p.nstrips * p.nbands + kWavesPerBlock - 1) / kWavesPerBlock, frames),
                      *band_fuse, yuv_src ? yuv_src : kSrcRgb0);
  } else {
    f360::KernelSpan span(ctx, f360::kSatWrite, prof, (int)frames, stream);
    const dim3 grid3((p.nstrips * p.nbands + kWavesPerBlock - 1) / kWavesPerBlock, frames);
    if (yuv_src == kSrcYuvSwsX86 && emit)
      hipLaunchKernelGGL((sat_write_kernel<kSrcYuvSwsX86, 2>), grid3, block, 0, stream, a, eb);
    else if (yuv_src == kSrcYuvSwsX86)
      hipLaunchKernelGGL((sat_write_kernel<kSrcYuvSwsX86, 1>), grid3, block, 0, stream, a, eb);
    else if (yuv_src == kSrcYuvSwsC && emit)
      hipLaunchKernelGGL((sat_write_kernel<kSrcYuvSwsC, 2>), grid3, block, 0, stream, a, eb);
    else if (yuv_src == kSrcYuvSwsC)
      hipLaunchKernelGGL((sat_write_kernel<kSrcYuvSwsC, 1>), grid3, block, 0, stream, a, eb);
    else if (emit && vec)
      hipLaunchKernelGGL((sat_write_kernel<kSrcRgb0, 2>), grid3, block, 0, stream, a, eb);
    else if (emit)
      hipLaunchKernelGGL((sat_write_kernel<kSrcBytes, 2>), grid3, block, 0, stream, a, eb);
    else if (!vec)
      hipLaunchKernelGGL((sat_write_kernel<kSrcBytes, 0>), grid3, block, 0, stream, a, eb);
    else  // (direct 48-byte-stride stores for RGB0 frames, "sat.store" = 0, were an A/B switch
          // until round 4: 136 against 88 us at 8K)
      hipLaunchKernelGGL((sat_write_kernel<kSrcRgb0, 1>), grid3, block, 0, stream, a, eb);
  }
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}

}  // namespace f360

extern "C" int f360_sat_encode(f360_ctx *ctx, uint32_t *sat_dev, const uint8_t *src_dev,
                               int width, int height, int linesize) {
  F360_REQUIRE(sat_dev, "f360_sat_encode: null buffer");
  return f360::sat_encode_impl(ctx, sat_dev, src_dev, width, height, linesize, nullptr,
                               nullptr);
}

extern "C" int f360_sat_encode_batch(f360_ctx *ctx, int count, uint32_t *const *sat_dev,
                                     const uint8_t *const *src_dev, int width, int height,
                                     int linesize) {
  F360_REQUIRE(count >= 1 && sat_dev && src_dev && width >= 1 && height >= 1 && linesize >= 1,
               "f360_sat_encode_batch: bad arguments");
  // Frames per launch ("sat.batch_mb", 180 MB of source).  The frames of a launch are read
  // twice, by the reducer and then by the table writer, and the second read comes out of the
  // 256 MiB Infinity Cache only while they fit beside the rest of the traffic: two 8K RGB0
  // frames (236 MB) per launch and the writer takes 85-87 us per frame instead of 75, whereas
  // four 8K frames from planes (176 MB) keep it at 74 and six (264 MB) do not (80).  Sixteen
  // 1080p frames (133 MB) are 2.7 times faster than one at a time.
  const size_t frame_bytes = (size_t)linesize * height;
  const int per_launch = (int)std::min<size_t>(
      std::max<size_t>(((size_t)std::max(ctx ? ctx->opt_batch_mb : 180, 1) << 20) / frame_bytes, 1),
      (size_t)f360_sat_encode_batch_max());
  F360_REQUIRE(ctx, "f360_sat_encode_batch: null context");
  const int prof = f360::take_profile_slot(ctx) ? 1 : 0;  // one slot for the whole call
  // enough frames to fill the device with strip owners: the read-once encoder (sat_walk_kernel)
  if (walk_wanted(ctx, count, width) && linesize / width == 4 && width % 4 == 0 &&
      linesize % 16 == 0 && (size_t)width * height * 3 < ((size_t)1 << 31)) {
    bool ok = true;
    for (int k = 0; k < count && ok; ++k)
      ok = sat_dev[k] && src_dev[k] && ((uintptr_t)src_dev[k] % 16) == 0 &&
           ((uintptr_t)sat_dev[k] % 16) == 0;
    if (ok) {
      F360_BIND_DEVICE(ctx);
      return sat_encode_walk(ctx, count, sat_dev, src_dev, nullptr, width, height, linesize,
                             prof != 0);
    }
  }
  F360_BIND_DEVICE(ctx);
  return f360::sat_pipelined_groups(
      ctx, width, height, false, (count + per_launch - 1) / per_launch, std::min(per_launch, count), false,
      [&](int g, const f360::SatLaunch &where) {
        const int k = g * per_launch, n = std::min(count - k, per_launch);
        return f360::sat_encode_impl(ctx, nullptr, nullptr, width, height, linesize, nullptr, nullptr,
                                     n, sat_dev + k, src_dev + k, prof, nullptr, nullptr, &where);
      });
}

extern "C" int f360_sat_encode_yuv420p_batch(f360_ctx *ctx, int count, uint32_t *const *sat_dev,
                                             const uint8_t *const *y_dev,
                                             const uint8_t *const *u_dev,
                                             const uint8_t *const *v_dev, int y_linesize,
                                             int u_linesize, int v_linesize, int width,
                                             int height) {
  F360_REQUIRE(ctx && count >= 1 && sat_dev && y_dev && u_dev && v_dev && width >= 1 &&
                   height >= 1 && y_linesize >= 1,
               "f360_sat_encode_yuv420p_batch: bad arguments");
  // frames per launch: the same cache budget as f360_sat_encode_batch, on 1.5 bytes per pixel
  // (8K: four frames per launch -- reducer 31 -> 23 us and carry pass 7.4 -> 3 us per frame)
  const size_t frame_bytes = (size_t)y_linesize * height * 3 / 2;
  const int per_launch = (int)std::min<size_t>(
      std::max<size_t>(((size_t)std::max(ctx->opt_batch_mb, 1) << 20) / frame_bytes, 1),
      (size_t)kEncBatch);
  const int prof = f360::take_profile_slot(ctx) ? 1 : 0;  // one slot for the whole call
  std::vector<f360::YuvPlanes> planes((size_t)count);
  for (int k = 0; k < count; ++k)
    planes[(size_t)k] = f360::YuvPlanes{y_dev[k], u_dev[k], v_dev[k], y_linesize, u_linesize,
                                        v_linesize};
  // enough frames to fill the device: the read-once encoder, converting in registers as the
  // three-kernel one does (the same argument rules; anything else takes the old path, whose
  // checks then report what is wrong)
  if (walk_wanted(ctx, count, width) && width % 4 == 0 && height % 2 == 0 &&
      y_linesize >= width && u_linesize >= width / 2 && v_linesize >= width / 2 &&
      y_linesize % 4 == 0 && u_linesize % 2 == 0 && v_linesize % 2 == 0 &&
      (size_t)width * height * 3 < ((size_t)1 << 31)) {
    bool ok = true;
    for (int k = 0; k < count && ok; ++k)
      ok = sat_dev[k] && y_dev[k] && u_dev[k] && v_dev[k] && ((uintptr_t)y_dev[k] % 4) == 0 &&
           ((uintptr_t)u_dev[k] % 2) == 0 && ((uintptr_t)v_dev[k] % 2) == 0 &&
           ((uintptr_t)sat_dev[k] % 16) == 0;
    if (ok) {
      F360_BIND_DEVICE(ctx);
      return sat_encode_walk(ctx, count, sat_dev, nullptr, planes.data(), width, height, 0,
                             prof != 0);
    }
  }
  F360_BIND_DEVICE(ctx);
  return f360::sat_pipelined_groups(
      ctx, width, height, true, (count + per_launch - 1) / per_launch, std::min(per_launch, count), false,
      [&](int g, const f360::SatLaunch &where) {
        const int k = g * per_launch, n = std::min(count - k, per_launch);
        return f360::sat_encode_impl(ctx, nullptr, nullptr, width, height, 0, nullptr, nullptr, n,
                                     sat_dev + k, nullptr, prof, planes.data() + k, nullptr, &where);
      });
}

extern "C" int f360_sat_encode_yuv420p(f360_ctx *ctx, uint32_t *sat_dev, const uint8_t *y_dev,
                                       const uint8_t *u_dev, const uint8_t *v_dev,
                                       int y_linesize, int u_linesize, int v_linesize,
                                       int width, int height) {
  F360_REQUIRE(sat_dev, "f360_sat_encode_yuv420p: null buffer");
  const f360::YuvPlanes planes{y_dev, u_dev, v_dev, y_linesize, u_linesize, v_linesize};
  return f360::sat_encode_impl(ctx, sat_dev, nullptr, width, height, 0, nullptr, &planes);
}
