// sample_stream.h -- the SAT sampler as a row-streaming kernel (sample.variant 2, the default
// where it applies).  Included once by sat_decoder.hip, inside its anonymous namespace, after
// SampleArgs / sample_axis / udiv3_exact / the wave-private LDS helpers.
//
// Same arithmetic as sample_rect_kernel (src/sat_decoder_sample_rect_kernel.cl:138-241); only
// the way the table reaches the lanes differs.  At 8K a gaze needs ~1900 table rows and every
// 128-byte line of each, but a 12-byte gather uses 12 bytes of a 64-byte request in the
// periphery; here a wave owns a column tile of the table and a run of reduced rows, streams the
// tile's segment of every table row the run needs into wave-private LDS, and the lanes pick
// their corners there (the corners of the row before stay in registers; no workgroup barrier
// anywhere: a wave's LDS operations execute in order).  A first version with 256-texel tiles,
// register staging and 200 registers per wave showed what bounds such a kernel on this machine:
// not bytes and not arithmetic but how many waves a SIMD can hold -- a lone wave issues one
// instruction every 4+ cycles and serialises its own memory, LDS and ALU phases (s_memtime
// stamps: 45 us per wave for a 37 us memory floor).  This one is built
// for occupancy:
//   * the table row segment goes memory -> LDS directly (global_load_lds_dwordx4: no staging
//     registers, no ds_write), into a ring of NS 2 KiB slots per wave (NS - 1 rows in flight
//     while one is consumed; a CU needs ~100 KB in flight to keep HBM busy);
//   * a tile is 128 texels (1.5 KiB per row: one and a half load instructions, the idle half
//     of the second carries the halo the left corners need), so a wave has at most three
//     passes of 64 pixels per row and ~100 registers;
//   * a wave's candidates -- the tile's main range of reduced columns, then the few pixels of
//     the other ranges (the far periphery of the other side of the gaze wraps onto every
//     tile) -- are packed into consecutive lanes, so a periphery tile needs ONE pass;
//   * divisions use one reciprocal and, per channel, convert / add / multiply / convert:
//     (n + 0.5) * (1/d) truncated is floor(n/d) exactly whenever the quotient is below 256 and
//     d below 4096 (see ts_div); anything else in the wave takes the integer path.
// Pixels are stored 2 + 1 bytes each: the reference writes .xyz only (byte 3 and skipped pixels
// keep their values).  Whole 16-byte groups merged with the destination's old bytes through LDS
// ("sample.groups", rounds 2-3) measured equal -- a 3-of-4-byte store costs a read-modify-write at
// the DRAM whichever side does it -- and were removed in round 4 (EXPERIMENTS.md section 3).
#pragma once

#ifndef F360_TS_ST_BITS
#define F360_TS_ST_BITS " nt"  // A/B: plain stores ("") measured in profiles/round2_sampler_sweeps.txt
#endif
__device__ __forceinline__ void store_rgb_uncounted(uint8_t *base, uint32_t off, uint32_t rg,
                                                    uint32_t b) {
  asm volatile(
      "global_store_short %0, %1, %3" F360_TS_ST_BITS "\n\t"
      "global_store_byte %0, %2, %3 offset:2" F360_TS_ST_BITS ::"v"(off),
      "v"(rg), "v"(b), "s"(base)
      : "memory");
}
__device__ __forceinline__ void store_b128_uncounted(uint8_t *base, uint32_t off, u32x4_t v) {
  asm volatile("global_store_dwordx4 %0, %1, %2" F360_TS_ST_BITS "\n\ts_nop 1" ::"v"(off), "v"(v),
               "s"(base)
               : "memory");
}

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v = max(v, __shfl_xor(v, m, 64));
  return v;
}

// Two 12-byte texels (4-byte aligned) from LDS, no wait: ts_lds_wait_all() orders the use.
struct TsTexels {
  u32x2_t a01, b01;
  uint32_t a2, b2;
};
__device__ __forceinline__ void ts_lds_issue(uint32_t addr_a, uint32_t addr_b, TsTexels &t) {
  asm volatile(
      "ds_read2_b32 %0, %4 offset1:1\n\t"
      "ds_read_b32 %1, %4 offset:8\n\t"
      "ds_read2_b32 %2, %5 offset1:1\n\t"
      "ds_read_b32 %3, %5 offset:8"
      : "=&v"(t.a01), "=&v"(t.a2), "=&v"(t.b01), "=&v"(t.b2)
      : "v"(addr_a), "v"(addr_b)
      : "memory");
}
// One asm that makes every pending LDS read of the row visible: the texel registers of the N
// passes pass through it as read-write operands, so no use can be scheduled above the wait.
#define TS_T(i) "+v"(t[i].a01), "+v"(t[i].a2), "+v"(t[i].b01), "+v"(t[i].b2)
template <int N>
__device__ __forceinline__ void ts_lds_wait_all(TsTexels *t) {
  static_assert(N >= 1 && N <= 5, "one to five passes");
  if constexpr (N == 1) asm volatile("s_waitcnt lgkmcnt(0)" : TS_T(0) : : "memory");
  if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(0)" : TS_T(0), TS_T(1) : : "memory");
  if constexpr (N == 3)
    asm volatile("s_waitcnt lgkmcnt(0)" : TS_T(0), TS_T(1), TS_T(2) : : "memory");
  if constexpr (N == 4)
    asm volatile("s_waitcnt lgkmcnt(0)" : TS_T(0), TS_T(1), TS_T(2), TS_T(3) : : "memory");
  if constexpr (N == 5)
    asm volatile("s_waitcnt lgkmcnt(0)" : TS_T(0), TS_T(1), TS_T(2), TS_T(3), TS_T(4) : : "memory");
}
#undef TS_T

constexpr int kTsTile = 128;             // texels per tile
constexpr int kTsTileVecs = 96;          // 16-byte vectors of a full tile row segment
constexpr int kTsMaxHaloVecs = 32;       // 512 bytes = 42 texels
// ring slot: [0,1536) tile, [1536,2048) halo (right-aligned)
constexpr int kTsSlotBytes = 2048;
constexpr int kTsStageBytes = 128 * 4;  // the schedule is built here (up to 128 entries)
constexpr int kTsPasses = 3;             // 192 candidates: 128 main + 64 others
constexpr int kTsRanges = 5;
constexpr int kTsMaxRows = 64;

// floor(n / d) for the three channels with one reciprocal.  With inv = rcp(d) (1 ulp) the float
// product (n + 0.5) * inv has a relative error below 2^-22; when n / d < 256 its absolute error
// is below 2^-14, while (n + 0.5) / d keeps a distance of at least 0.5 / d > 2^-13 (d < 4096)
// from every integer -- so truncation gives the exact quotient.  The caller checks both bounds.
__device__ __forceinline__ uint32_t ts_div(uint32_t n, float inv) {
  return (uint32_t)(((float)n + 0.5f) * inv);
}

typedef const __attribute__((address_space(1))) void *ts_gptr;
typedef __attribute__((address_space(3))) void *ts_lptr;

// The rows of one wave: NP passes of 64 lanes, NS ring slots (NS - 1 rows in flight).  The row
// loop is not unrolled (the ring slot is a scalar), so the whole loop is a few hundred
// instructions of straight-line code.
template <int NS, int NP>
__device__ __forceinline__ void tile_stream_rows(
    const SampleArgs &a, int j0, int nsched, uint32_t sched_a, uint32_t sched_b,
    const char *row0, uint32_t off_a, uint32_t off_b, uint32_t lds0, uint8_t *lds_ptr,
    const int (&pi)[kTsPasses], const uint32_t (&off_hi)[kTsPasses],
    const uint32_t (&off_lo)[kTsPasses], const uint32_t (&dxw)[kTsPasses]) {
  constexpr int D = NS - 1;  // rows in flight
  constexpr int LPR = 2;     // vector-memory loads per row
  const uint32_t row_bytes = (uint32_t)a.src_w * 12u;
  const uint32_t out_stride = (uint32_t)a.out_stride_px * 4u;
  uint32_t pix_off[NP];
  bool edge[NP], any_edge[NP], unit_p[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    edge[p] = pi[p] >= 0;  // this lane writes a pixel in pass p
    any_edge[p] = __any(edge[p]);
    pix_off[p] = (uint32_t)max(pi[p], 0) * 4u;
    unit_p[p] = __all(dxw[p] == 1);  // every box of the pass one texel wide: the fovea
  }
  uint3 p_hi[NP], p_lo[NP];  // corners of the previous streamed row
#pragma unroll
  for (int p = 0; p < NP; ++p) p_hi[p] = p_lo[p] = make_uint3(0, 0, 0);

  auto entry_at = [&](int n) -> uint32_t {
    n = min(n, nsched - 1);
    return (uint32_t)__builtin_amdgcn_readlane((int)(n < 64 ? sched_a : sched_b), n & 63);
  };
  // one row: two LDS-direct loads of 1 KiB each
  auto issue = [&](int slot, uint32_t entry) {
    const char *row = row0 + (size_t)(entry & 0xffffu) * row_bytes;
    uint8_t *dst = lds_ptr + slot * kTsSlotBytes;
    __builtin_amdgcn_global_load_lds((ts_gptr)(row + off_a), (ts_lptr)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((ts_gptr)(row + off_b), (ts_lptr)(dst + 1024), 16, 0, 0);
  };
#pragma unroll
  for (int k = 0; k < D; ++k) issue(k, entry_at(k));
  int slot = 0, slot_pf = D;  // slot of row n, slot the row n + D goes to (row n - 1 has left it)
  for (int n = 0; n < nsched; ++n) {
    const uint32_t cur = entry_at(n);
    issue(slot_pf, entry_at(n + D));
    // all but the loads of the D younger rows have landed.  (Loads and stores retire from vmcnt
    // in issue order on gfx950 -- tools/vmcnt_order -- so the stores issued meanwhile could be
    // counted too; selecting the immediate at run time cost more than the stricter wait.)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D * LPR) : "memory");
    const uint32_t slot_base = lds0 + (uint32_t)slot * kTsSlotBytes;
    slot = slot + 1 == NS ? 0 : slot + 1;
    slot_pf = slot_pf + 1 == NS ? 0 : slot_pf + 1;
    TsTexels t[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) ts_lds_issue(slot_base + off_hi[p], slot_base + off_lo[p], t[p]);
    ts_lds_wait_all<NP>(t);
    const uint32_t dy = (cur >> 16) & 0xffu;
    const int rsel = (int)(cur >> 24);  // 0: top corners only
    const uint32_t row_off = (uint32_t)(j0 + rsel - 1) * out_stride;
    uint3 hi[NP], lo[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      hi[p] = make_uint3(t[p].a01.x, t[p].a01.y, t[p].a2);
      lo[p] = make_uint3(t[p].b01.x, t[p].b01.y, t[p].b2);
    }
    if (rsel != 0 && !(a.ablate & 32)) {
      uint3 s[NP];
      uint32_t d[NP];
      bool slow = false;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        s[p] = make_uint3(hi[p].x - p_hi[p].x + p_lo[p].x - lo[p].x,
                          hi[p].y - p_hi[p].y + p_lo[p].y - lo[p].y,
                          hi[p].z - p_hi[p].z + p_lo[p].z - lo[p].z);
        d[p] = dxw[p] * dy;
        // the float quotient is exact for quotients < 256 and d < 4096
        const uint32_t lim = d[p] << 8;
        slow = slow || s[p].x >= lim || s[p].y >= lim || s[p].z >= lim || d[p] >= 4096u;
      }
      uint3 q[NP];
      if (__builtin_expect(__any(slow), 0)) {
        // a wrapped table read with a degenerate box somewhere in the wave: integer division
#pragma unroll
        for (int p = 0; p < NP; ++p) q[p] = udiv3_exact(s[p], d[p]);
      } else {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          if (unit_p[p] && dy == 1) {  // 1x1 boxes: most pixels of the fovea
            q[p] = s[p];
          } else {
            const float inv = __builtin_amdgcn_rcpf((float)d[p]);
            q[p] = make_uint3(ts_div(s[p].x, inv), ts_div(s[p].y, inv), ts_div(s[p].z, inv));
          }
        }
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const uint32_t rg = (q[p].x & 0xffu) | ((q[p].y & 0xffu) << 8);
        if (any_edge[p] && !(a.ablate & 16)) {
          if (edge[p]) store_rgb_uncounted(a.dst, row_off + pix_off[p], rg, q[p].z);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      p_hi[p] = hi[p];
      p_lo[p] = lo[p];
    }
  }
  // nothing may still be landing in this wave's LDS when the workgroup's allocation is reused
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int NS>
__device__ __forceinline__ void tile_stream_body(const SampleArgs &a, int tile, int j0, int rows,
                                             uint32_t lds0, uint8_t *lds_ptr) {
  const int lane = threadIdx.x & 63;
  const int src_w = a.src_w;
  const int ntiles = (src_w + kTsTile - 1) / kTsTile;
  const uint32_t lds_stage = lds0 + NS * kTsSlotBytes;  // the schedule is built here
  const int x_tile = tile * kTsTile;

  // ---- candidate pixels: contiguous ranges of reduced columns whose unwrapped right corner
  // cxp + gx[i+1] lies in the tile's window -- wrap states k = 0, +1, -1 (the log-rectilinear
  // lattice spans +-W around the gaze, so the far periphery of the other side lands on every
  // tile), and for the last tile the pixels that straddle the seam (their corner is clamped
  // back to src_w - 1; wrap states 0 and +1, a box that was moved up by W cannot straddle).
  // A corner clamped up to 1 was 0: inside the first tile anyway.
  int cnt[kTsRanges], first[kTsRanges];
#pragma unroll
  for (int q = 0; q < kTsRanges; ++q) {
    const int k = (q == 1 || q == 4) ? 1 : (q == 2 ? -1 : 0);
    int lo_b = x_tile + k * src_w - a.cxp, hi_b = lo_b + kTsTile;
    if (q >= 3) {
      lo_b = hi_b;
      hi_b = lo_b + (tile == ntiles - 1 ? a.halo : 0);
    }
    const int g_lo = a.lbx[min(max(lo_b - a.lb_dmin, 0), a.lb_n - 1)];
    const int g_hi = a.lbx[min(max(hi_b - a.lb_dmin, 0), a.lb_n - 1)];
    const int i_lo = max(g_lo - 1, 0);  // grid index g is the right corner of pixel g - 1
    const int i_hi = max(min(g_hi - 1, a.out_w), i_lo);
    first[q] = i_lo;
    cnt[q] = i_hi - i_lo;
  }
  // main range = the largest (the host guarantees <= 128, and the others <= 64 in all); the
  // candidate list is the main range followed by the others
  int m = 0;
#pragma unroll
  for (int q = 1; q < kTsRanges; ++q)
    if (cnt[q] > cnt[m]) m = q;
  int cnt_main = 0, first_main = 0;
#pragma unroll
  for (int q = 0; q < kTsRanges; ++q)
    if (q == m) {
      cnt_main = cnt[q];
      first_main = first[q];
      cnt[q] = 0;  // what remains in cnt[] are the other ranges
    }
  const int total = cnt_main + cnt[0] + cnt[1] + cnt[2] + cnt[3] + cnt[4];
  if (total == 0) return;
  const int np = (total + 63) >> 6;

  int pi[kTsPasses];       // reduced column this wave writes, -1: none
  int ci[kTsPasses];       // candidate column (written or not), -1: none
  bool is_main[kTsPasses];
  int16_t g_hi[kTsPasses], g_lo[kTsPasses];
  // every pass's grid loads are issued before anything depends on one (clamped indices, no
  // load inside a branch: the set-up is one memory round trip)
#pragma unroll
  for (int p = 0; p < kTsPasses; ++p) {
    int q = p * 64 + lane;
    int i = -1;
    is_main[p] = q < cnt_main;
    if (is_main[p]) {
      i = first_main + q;
    } else {
      q -= cnt_main;
#pragma unroll
      for (int s = 0; s < kTsRanges; ++s) {
        if (i < 0 && q >= 0 && q < cnt[s]) i = first[s] + q;
        q -= cnt[s];
      }
    }
    ci[p] = i;
    const int ic = min(max(i, 0), a.out_w - 1);
    g_hi[p] = a.gx[ic + 1];
    g_lo[p] = a.gx[ic];
  }
  AxisBox bx[kTsPasses];
  int need_halo = 0;
  bool any_mine = false;
#pragma unroll
  for (int p = 0; p < kTsPasses; ++p) {
    pi[p] = -1;
    bx[p] = sample_axis(a.cxp, g_hi[p], g_lo[p], src_w, true);
    if (ci[p] >= 0) {
      if (bx[p].ok && bx[p].hi >= x_tile && bx[p].hi < x_tile + kTsTile) {
        pi[p] = ci[p];
        need_halo = max(need_halo, x_tile - bx[p].lo);
        any_mine = true;
      }
    }
  }
  if (!__any(any_mine)) return;
  // halo of this tile in 16-byte vectors (4 texels = 3 vectors), never reaching below x = 0
  const int halo_vecs = __builtin_amdgcn_readfirstlane(
      min((wave_max_i32(need_halo) + 3) / 4 * 3, x_tile * 3 / 4));
  const int hb = halo_vecs * 16;
  const int tile_vecs = (min(x_tile + kTsTile, src_w) - x_tile) * 3 / 4;  // width % 4 == 0
  uint32_t off_hi[kTsPasses], off_lo[kTsPasses], dxw[kTsPasses];
#pragma unroll
  for (int p = 0; p < kTsPasses; ++p) {
    off_hi[p] = off_lo[p] = 0;
    dxw[p] = 1;
    if (pi[p] >= 0) {
      off_hi[p] = (uint32_t)(bx[p].hi - x_tile) * 12u;
      const int dl = bx[p].lo - x_tile;  // < 0: in the halo, stored right-aligned behind the tile
      off_lo[p] = dl >= 0 ? (uint32_t)dl * 12u : (uint32_t)(kTsTileVecs * 16 + hb + dl * 12);
      dxw[p] = (uint32_t)(bx[p].hi - bx[p].lo);
    }
  }

  // ---- schedule: lane r owns reduced row j0 + r.  entry = table row | box height << 16 |
  // (r + 1) << 24, the last field 0 for a row that only provides the top corners of the box
  // below it.  A row whose top is not the bottom of the row before needs such an entry.
  int nsched;
  uint32_t sched_a, sched_b;
  {
    const int j = min(j0 + lane, a.out_h - 1);
    AxisBox by = sample_axis(a.cyp, a.gy[j + 1], a.gy[j], a.src_h, false);
    const bool ok = by.ok && lane < rows && j0 + lane < a.out_h;
    const int hi_above = __shfl_up(by.hi, 1, 64);
    const bool ok_above = __shfl_up((int)ok, 1, 64) != 0;
    const bool top = ok && (lane == 0 || !ok_above || hi_above != by.lo);
    const unsigned long long m_ok = __ballot(ok), m_top = __ballot(top);
    const unsigned long long below = (1ull << lane) - 1ull;
    const int pos = __popcll(m_ok & below) + __popcll(m_top & below);
    if (top) lds_store4(lds_stage + 4 * pos, (uint32_t)by.lo);
    if (ok)
      lds_store4(lds_stage + 4 * (pos + (top ? 1 : 0)),
                 (uint32_t)by.hi | ((uint32_t)(by.hi - by.lo) << 16) | ((uint32_t)(lane + 1) << 24));
    nsched = __popcll(m_ok) + __popcll(m_top);
    sched_a = lds_load4(lds_stage + 4 * lane);         // entries 0..63  (stale beyond nsched:
    sched_b = lds_load4(lds_stage + 4 * (64 + lane));  // entries 64..127 never selected)
  }
  if (nsched == 0) return;

  // ---- per-lane source offsets of the two loads of a row, relative to row0 = table +
  // x_tile * 12 - 512 (so that the halo's offsets are not negative; tile 0 has no halo and
  // never goes below its own start)
  const char *row0 = reinterpret_cast<const char *>(a.sat) + (size_t)x_tile * 12 - 512;
  const uint32_t off_a = 512u + (uint32_t)min(lane, tile_vecs - 1) * 16u;
  uint32_t off_b;
  if (lane < 32)
    off_b = 512u + (uint32_t)min(64 + lane, tile_vecs - 1) * 16u;
  else
    off_b = halo_vecs > 0 ? (uint32_t)(512 - hb + min(lane - 32, halo_vecs - 1) * 16) : 512u;

#define TS_ROWS(N) \
  tile_stream_rows<NS, N>(a, j0, nsched, sched_a, sched_b, row0, off_a, off_b, lds0, lds_ptr, pi, off_hi, off_lo, dxw)
  if (np <= 1) TS_ROWS(1);
  else if (np == 2) TS_ROWS(2);
  else TS_ROWS(3);
#undef TS_ROWS
}

// Work items = (row block, tile), four consecutive tiles per workgroup.
template <int NS>
__global__ __launch_bounds__(256) void sample_rect_stream_kernel(const SampleArgs a, int rows,
                                                                  int nblocks) {
  __shared__ __attribute__((aligned(16))) uint8_t stage[4][NS * kTsSlotBytes + kTsStageBytes];
  const int wave = threadIdx.x >> 6;
  const int ntiles = (a.src_w + kTsTile - 1) / kTsTile;
  const int g = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wave);
  if (g >= ntiles * nblocks) return;
  const int blk = g / ntiles;
  tile_stream_body<NS>(a, g - blk * ntiles, blk * rows, rows,
                       (uint32_t)reinterpret_cast<uintptr_t>(&stage[wave][0]), &stage[wave][0]);
}

