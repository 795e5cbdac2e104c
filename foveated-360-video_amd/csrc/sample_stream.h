// sample_stream.h -- the SAT sampler as a row-streaming kernel (variant 4, the default at
// sizes where it applies).  Included once by sat_decoder.hip, inside its anonymous namespace,
// after SampleArgs / sample_axis / the wave-private LDS helpers.
//
// Same arithmetic as sample_rect_kernel (src/sat_decoder_sample_rect_kernel.cl:138-241); only
// the way the table reaches the lanes differs.  At 8K a gaze needs ~1900 table rows and every
// 128-byte line of each, but a 12-byte gather uses 12 bytes of a 64-byte request in the
// periphery.  Here a wave owns a 256-texel column tile of the table (3 KiB per row, a multiple
// of both the texel and the 1 KiB a wave loads per instruction) and a RUN of reduced rows:
//   * set-up, once per wave: the reduced pixels whose right corner lies in the tile (from the
//     host-built inverse of the x grid, three wrap states), their LDS offsets, and -- computed
//     by the 64 lanes in parallel -- the list of table rows the run needs (the "schedule",
//     kept in two registers and read with v_readlane);
//   * per scheduled row: the tile's segment (+ the halo its left corners need) travels
//     memory -> registers (DEPTH rows in flight, branch-free clamped loads) -> wave-private LDS,
//     the lanes pick their two corners of that row from LDS, the two corners of the row before
//     are still in registers, and a row that closes a box produces its pixels;
//   * pixels leave as whole 16-byte groups: a tile's pixels are one contiguous range of reduced
//     columns, so they are packed through LDS into 4-pixel groups, merged with the destination's
//     old bytes (the reference writes .xyz only: byte 3 and skipped pixels keep their values;
//     the old groups are prefetched with the table rows) and stored with one 1-KiB-contiguous
//     instruction per row.  Range ends that do not fill a group, and the rare tiles whose
//     pixels form several ranges (frame seam), store 2 + 1 bytes per pixel instead.
// No workgroup barrier anywhere: a wave's LDS operations execute in order.
#pragma once

constexpr int kS4Tile = 256;       // texels per tile
constexpr int kS4MaxHalo = 64;     // texels (host checks the grid's largest step against it)
constexpr int kS4MaxRows = 64;     // reduced rows per wave (one schedule lane per row)
constexpr int kS4SegBytes = (kS4Tile + kS4MaxHalo) * 12;
constexpr int kS4OutBytes = 6 * 64 * 4;  // packed pixels of one row (up to 6 passes)
constexpr int kS4LdsBytes = kS4SegBytes + kS4OutBytes;

// Output stores the compiler does not count (see global_store_b128_uncounted in
// sat_encode.hip): scalar base + 32-bit byte offset, non-temporal.
__device__ __forceinline__ void store_rgb_uncounted(uint8_t *base, uint32_t off, uint32_t rg,
                                                    uint32_t b) {
  asm volatile(
      "global_store_short %0, %1, %3 nt\n\t"
      "global_store_byte %0, %2, %3 offset:2 nt" ::"v"(off),
      "v"(rg), "v"(b), "s"(base)
      : "memory");
}
__device__ __forceinline__ void store_b128_uncounted(uint8_t *base, uint32_t off, u32x4_t v) {
  asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base)
               : "memory");
}

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v = max(v, __shfl_xor(v, m, 64));
  return v;
}

// Two 12-byte texels (4-byte aligned) from LDS, no wait: s4_lds_wait<N>() orders the use.
struct S4Texels {
  u32x2_t a01, b01;
  uint32_t a2, b2;
};
__device__ __forceinline__ void s4_lds_issue(uint32_t addr_a, uint32_t addr_b, S4Texels &t) {
  asm volatile(
      "ds_read2_b32 %0, %4 offset1:1\n\t"
      "ds_read_b32 %1, %4 offset:8\n\t"
      "ds_read2_b32 %2, %5 offset1:1\n\t"
      "ds_read_b32 %3, %5 offset:8"
      : "=&v"(t.a01), "=&v"(t.a2), "=&v"(t.b01), "=&v"(t.b2)
      : "v"(addr_a), "v"(addr_b)
      : "memory");
}
template <int N>
__device__ __forceinline__ void s4_lds_wait(S4Texels &t) {
  asm volatile("s_waitcnt lgkmcnt(%4)"
               : "+v"(t.a01), "+v"(t.a2), "+v"(t.b01), "+v"(t.b2)
               : "n"(N)
               : "memory");
}

// wait until at most `younger` LDS operations issued after t's are outstanding (a constant
// once the pass loop is unrolled; the counter saturates at 15)
__device__ __forceinline__ void s4_lds_wait_younger(int younger, S4Texels &t) {
  switch (younger) {
    case 0: s4_lds_wait<0>(t); break;
    case 4: s4_lds_wait<4>(t); break;
    case 8: s4_lds_wait<8>(t); break;
    case 12: s4_lds_wait<12>(t); break;
    default: s4_lds_wait<15>(t); break;
  }
}

// q = n / d, exact, for n, d < 2^22 with a shared reciprocal (1 ulp): the float product is
// off by at most one, the exact remainder decides the correction.  24-bit multiplies.
__device__ __forceinline__ uint32_t s4_div(uint32_t n, float inv, uint32_t d) {
  uint32_t q = (uint32_t)((float)n * inv);
  const uint32_t r = n - (q & 0xffffffu) * (d & 0x3fffffu);
  if ((int32_t)r < 0) q -= 1;
  else if (r >= d) q += 1;
  return q;
}

template <int PASSES, int DEPTH, bool FAST>
__device__ __forceinline__ void stream4_rows(
    const SampleArgs &a, int j0, int nsched, uint32_t sched_a, uint32_t sched_b,
    const char *sat_tile, int tile_vecs, int halo_vecs, uint32_t lds_tile, uint32_t lds_out,
    const int (&pi)[PASSES], const int (&ci)[PASSES], const uint32_t (&off_hi)[PASSES],
    const uint32_t (&off_lo)[PASSES], const uint32_t (&dxw)[PASSES], int fg0, int nfull) {
  const int lane = threadIdx.x & 63;
  const uint32_t row_bytes = (uint32_t)a.src_w * 12u;
  const uint32_t out_stride = (uint32_t)a.out_stride_px * 4u;
  // per-lane byte offsets inside a table row (clamped: every load is issued by every lane)
  uint32_t voff[3];
#pragma unroll
  for (int v = 0; v < 3; ++v) voff[v] = (uint32_t)min(v * 64 + lane, tile_vecs - 1) * 16u;
  const int hv = min(lane, max(halo_vecs - 1, 0));
  const int halo_off = halo_vecs > 0 ? (hv - halo_vecs) * 16 : 0;  // bytes left of the tile
  const bool halo_lane = lane < halo_vecs;
  // FAST: lane L owns the 4-pixel group of reduced columns fg0 + 4L .. + 3
  const uint32_t grp_off = (uint32_t)(fg0 + 4 * min(lane, max(nfull - 1, 0))) * 4u;
  const bool grp_lane = lane < nfull;
  // candidates inside the whole groups are staged (a candidate that is not written stages 0,
  // so every slot of a group is defined); written pixels outside them are stored 2 + 1 bytes
  bool staged[PASSES], edge[PASSES];
  uint32_t stage_at[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int rel = ci[p] - fg0;
    staged[p] = FAST && ci[p] >= 0 && rel >= 0 && rel < 4 * nfull;
    edge[p] = pi[p] >= 0 && !staged[p];
    stage_at[p] = lds_out + (staged[p] ? (uint32_t)rel * 4u : 0u);
  }
  uint3 p_hi[PASSES], p_lo[PASSES];  // corners of the previous streamed row
  bool unit_w[PASSES];               // every box of the pass is one texel wide (the fovea)
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    p_hi[p] = p_lo[p] = make_uint3(0, 0, 0);
    unit_w[p] = __all(dxw[p] == 1);
  }

  auto entry_at = [&](int n) -> uint32_t {
    n = min(n, nsched - 1);
    return (uint32_t)__builtin_amdgcn_readlane((int)(n < 64 ? sched_a : sched_b), n & 63);
  };
  u32x4_t regs[DEPTH][5];  // [4]: the destination's old group, FAST only
  uint32_t ent[DEPTH];
  auto issue = [&](int k, uint32_t entry) {
    const char *row = sat_tile + (size_t)(entry & 0xffffu) * row_bytes;
#pragma unroll
    for (int v = 0; v < 3; ++v)
      regs[k][v] = *reinterpret_cast<const u32x4_t *>(row + voff[v]);
    regs[k][3] = *reinterpret_cast<const u32x4_t *>(row + halo_off);
    if (FAST) {
      const int rsel = (int)(entry >> 24);
      const uint32_t row_off = (uint32_t)(j0 + max(rsel - 1, 0)) * out_stride;
      regs[k][4] = *reinterpret_cast<const u32x4_t *>(a.dst + row_off + grp_off);
    }
  };
#pragma unroll
  for (int k = 0; k < DEPTH; ++k) {
    ent[k] = entry_at(k);
    issue(k, ent[k]);
    // keeps the rows' loads in issue order: the wait for row n counts the loads younger than it,
    // and a reordered prologue would turn the loop's first wait into vmcnt(0)
    asm volatile("" ::: "memory");
  }
  for (int n0 = 0; n0 < nsched; n0 += DEPTH) {
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
      const int n = n0 + k;
      const uint32_t cur = ent[k];
#pragma unroll
      for (int v = 0; v < 3; ++v)
        if (v * 64 + lane < tile_vecs)
          lds_store16(lds_tile + (uint32_t)(v * 64 + lane) * 16u, regs[k][v]);
      if (halo_lane) lds_store16(lds_tile + (uint32_t)halo_off, regs[k][3]);
      u32x4_t old = {0, 0, 0, 0};
      if (FAST) old = regs[k][4];
      ent[k] = entry_at(n + DEPTH);
      issue(k, ent[k]);
      if (n >= nsched) continue;
      S4Texels t[PASSES];
#pragma unroll
      for (int p = 0; p < PASSES; ++p) s4_lds_issue(off_hi[p], off_lo[p], t[p]);
      const uint32_t dy = (cur >> 16) & 0xffu;
      const int rsel = (int)(cur >> 24);  // 0: top corners only
      const uint32_t row_off = (uint32_t)(j0 + rsel - 1) * out_stride;
      const bool emit = rsel != 0 && !(a.ablate & 32);
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        s4_lds_wait_younger(4 * (PASSES - 1 - p), t[p]);
        const uint3 hi = make_uint3(t[p].a01.x, t[p].a01.y, t[p].a2);
        const uint3 lo = make_uint3(t[p].b01.x, t[p].b01.y, t[p].b2);
        if (emit) {
          uint3 s = make_uint3(hi.x - p_hi[p].x + p_lo[p].x - lo.x,
                               hi.y - p_hi[p].y + p_lo[p].y - lo.y,
                               hi.z - p_hi[p].z + p_lo[p].z - lo.z);
          const uint32_t d = dxw[p] * dy;
          uint3 q;
          if (unit_w[p] && dy == 1) {
            q = s;  // 1x1 boxes: more than half of the pixels at the benchmark geometries
          } else if (__builtin_expect(__any(((s.x | s.y | s.z | d) >> 22) != 0), 0)) {
            q = udiv3_exact(s, d);  // a wrapped table read with a degenerate box
          } else {
            const float inv = __builtin_amdgcn_rcpf((float)d);
            q = make_uint3(s4_div(s.x, inv, d), s4_div(s.y, inv, d), s4_div(s.z, inv, d));
          }
          const uint32_t rg = (q.x & 0xffu) | ((q.y & 0xffu) << 8);
          if (FAST && staged[p])  // 0xff in byte 3 marks a pixel that is written
            lds_store4(stage_at[p], pi[p] >= 0 ? (rg | ((q.z & 0xffu) << 16) | 0xff000000u) : 0u);
          if (edge[p] && !(a.ablate & 16))
            store_rgb_uncounted(a.dst, row_off + (uint32_t)pi[p] * 4u, rg, q.z);
        }
        p_hi[p] = hi;
        p_lo[p] = lo;
      }
      if (FAST && emit) {
        u32x4_t px;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(px)
                     : "v"(lds_out + (uint32_t)lane * 16u)
                     : "memory");
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const uint32_t m = (uint32_t)((int32_t)px[c] >> 31) & 0x00ffffffu;
          old[c] = (px[c] & m) | (old[c] & ~m);
        }
        if (grp_lane && !(a.ablate & 16)) store_b128_uncounted(a.dst, row_off + grp_off, old);
      }
    }
  }
}

template <int PASSES, int DEPTH>
__device__ __forceinline__ void stream4_body(const SampleArgs &a, int tile, int j0, int rows,
                                             uint32_t lds0) {
  const int lane = threadIdx.x & 63;
  const int src_w = a.src_w;
  const int ntiles = (src_w + kS4Tile - 1) / kS4Tile;
  const uint32_t lds_tile = lds0 + kS4MaxHalo * 12;  // LDS address of texel x_tile
  const uint32_t lds_out = lds0 + kS4SegBytes;
  const int x_tile = tile * kS4Tile;

  // ---- candidate pixels: wrap states k = 0, +1, -1, each a contiguous range of columns
  int cnt[3], first[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int k = q == 0 ? 0 : (q == 1 ? 1 : -1);
    // unwrapped right corner cxp + gx[i+1] in [lo_b, hi_b); the last tile also takes the
    // pixels that straddle the seam (their corner is clamped back to src_w - 1); a corner
    // clamped up to 1 was 0, inside the first tile anyway
    const int lo_b = x_tile + k * src_w - a.cxp;
    const int hi_b = lo_b + kS4Tile + (tile == ntiles - 1 ? a.halo : 0);
    const int g_lo = a.lbx[min(max(lo_b - a.lb_dmin, 0), a.lb_n - 1)];
    const int g_hi = a.lbx[min(max(hi_b - a.lb_dmin, 0), a.lb_n - 1)];
    const int i_lo = max(g_lo - 1, 0);  // grid index g is the right corner of pixel g - 1
    const int i_hi = max(min(g_hi - 1, a.out_w), i_lo);
    first[q] = i_lo;
    cnt[q] = i_hi - i_lo;
  }
  const int total = cnt[0] + cnt[1] + cnt[2];
  if (total == 0) return;

  int pi[PASSES];  // reduced column this wave writes, -1: none
  int ci[PASSES];  // candidate column (written or not), -1: none
  uint32_t off_hi[PASSES], off_lo[PASSES], dxw[PASSES];
  int need_halo = 0;
  bool any_mine = false, foreign = false;
  // every pass's grid loads are issued before anything depends on one (clamped indices, no
  // load inside a branch: the set-up is one memory round trip, not one per pass)
  int16_t g_hi[PASSES], g_lo[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    int q = p * 64 + lane;
    int i = -1;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if (i < 0 && q >= 0 && q < cnt[s]) i = first[s] + q;
      q -= cnt[s];
    }
    ci[p] = i;
    const int ic = min(max(i, 0), a.out_w - 1);
    g_hi[p] = a.gx[ic + 1];
    g_lo[p] = a.gx[ic];
  }
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    pi[p] = -1;
    off_hi[p] = off_lo[p] = lds_tile;
    dxw[p] = 1;
    if (ci[p] >= 0) {
      const AxisBox bx = sample_axis(a.cxp, g_hi[p], g_lo[p], src_w, true);
      if (bx.ok && bx.hi >= x_tile && bx.hi < x_tile + kS4Tile) {
        pi[p] = ci[p];
        off_hi[p] = lds_tile + (uint32_t)(bx.hi - x_tile) * 12u;
        off_lo[p] = lds_tile + (uint32_t)((bx.lo - x_tile) * 12);  // may lie in the halo
        dxw[p] = (uint32_t)(bx.hi - bx.lo);
        need_halo = max(need_halo, x_tile - bx.lo);
        any_mine = true;
      } else if (bx.ok) {
        foreign = true;  // another tile's pixel inside this tile's candidate range
      }
    }
  }
  if (!__any(any_mine)) return;
  // halo of this tile in 16-byte vectors (4 texels = 3 vectors), never reaching below x = 0
  const int halo_vecs = __builtin_amdgcn_readfirstlane(
      min((wave_max_i32(need_halo) + 3) / 4 * 3, x_tile * 3 / 4));
  const int tile_vecs = (min(x_tile + kS4Tile, src_w) - x_tile) * 3 / 4;  // width % 4 == 0

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
    if (top) lds_store4(lds_out + 4 * pos, (uint32_t)by.lo);
    if (ok)
      lds_store4(lds_out + 4 * (pos + (top ? 1 : 0)),
                 (uint32_t)by.hi | ((uint32_t)(by.hi - by.lo) << 16) | ((uint32_t)(lane + 1) << 24));
    nsched = __popcll(m_ok) + __popcll(m_top);
    sched_a = lds_load4(lds_out + 4 * lane);         // entries 0..63  (stale beyond nsched:
    sched_b = lds_load4(lds_out + 4 * (64 + lane));  // entries 64..127 never selected)
  }
  if (nsched == 0) return;

  const char *sat_tile = reinterpret_cast<const char *>(a.sat) + (size_t)x_tile * 12;
  // whole-group stores need ONE range of candidates and no foreign pixel in it
  const int nranges = (cnt[0] > 0) + (cnt[1] > 0) + (cnt[2] > 0);
  const int ia = cnt[0] > 0 ? first[0] : (cnt[1] > 0 ? first[1] : first[2]);
  const int fg0 = (ia + 3) & ~3;
  const int nfull = min((ia + total - fg0) >> 2, 64);  // one group per lane
  if (nranges == 1 && !__any(foreign) && nfull >= 1 && !(a.ablate & 64))
    stream4_rows<PASSES, DEPTH, true>(a, j0, nsched, sched_a, sched_b, sat_tile, tile_vecs,
                                      halo_vecs, lds_tile, lds_out, pi, ci, off_hi, off_lo, dxw,
                                      fg0, nfull);
  else
    stream4_rows<PASSES, DEPTH, false>(a, j0, nsched, sched_a, sched_b, sat_tile, tile_vecs,
                                       halo_vecs, lds_tile, lds_out, pi, ci, off_hi, off_lo, dxw,
                                       0, 0);
}

// Work items.  The frame's reduced rows are cut into blocks of `rows`; inside a block a light
// (periphery) tile is one item, a heavy tile -- one that overlaps the fovea's unit-step columns:
// four passes per row instead of one -- is `hsplit` items of rows / hsplit rows each, so that
// items carry about the same instruction count and no wave is the kernel's critical path.
// Heavy tiles are `th` consecutive tiles starting at `rot` (mod the tile count); any 0 <= th <=
// ntiles covers every (tile, row) exactly once.  Consecutive waves take items `istride` apart
// (coprime to the items of a block, about a quarter of them): the four waves of a workgroup --
// and with them every CU and SIMD, since all waves of the launch are resident at once and
// nothing rebalances them -- get the same mix of heavy and light items.
template <int PASSES, int DEPTH>
__global__ __launch_bounds__(256) void sample_rect_stream4_kernel(const SampleArgs a, int rows,
                                                                  int nblocks, int hsplit,
                                                                  int th, int rot, int istride) {
  __shared__ __attribute__((aligned(16))) uint8_t stage[4][kS4LdsBytes];
  const int wave = threadIdx.x >> 6;
  const int ntiles = (a.src_w + kS4Tile - 1) / kS4Tile;
  const int per_block = hsplit * th + (ntiles - th);
  const int g = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wave);
  if (g >= per_block * nblocks) return;
  const int blk = g / per_block;
  const int r = (int)(((unsigned)(g - blk * per_block) * (unsigned)istride) % (unsigned)per_block);
  int t, j0, nrows;
  if (r < hsplit * th) {
    t = r / hsplit;
    nrows = rows / hsplit;
    j0 = blk * rows + (r - t * hsplit) * nrows;
  } else {
    t = th + (r - hsplit * th);
    nrows = rows;
    j0 = blk * rows;
  }
  t += rot;
  if (t >= ntiles) t -= ntiles;
  if (j0 >= a.out_h) return;
  stream4_body<PASSES, DEPTH>(a, t, j0, nrows,
                              (uint32_t)reinterpret_cast<uintptr_t>(&stage[wave][0]));
}
