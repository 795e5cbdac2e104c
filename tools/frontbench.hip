// frontbench.hip -- the read-once encoder's WRITE pattern on its own (not part of the product):
// NF tables of 7680 x 3840 x 12 bytes written at the same time, one wave per (table, 256-pixel
// strip) walking down the rows -- 3 x 1 KiB stores per row, a full drain every 8 rows, like
// sat_walk_kernel -- against where the tables lie: one allocation each, or carved from one
// allocation at a given pitch.  profiles/round4_table_placement.txt has the encoder's own figures.
//   hipcc --offload-arch=gfx950 -O3 tools/frontbench.hip -o tools/frontbench && tools/frontbench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int W = 7680, H = 3840, NF = 32, STRIPS = W / 256;
constexpr size_t ROW = (size_t)W * 12, TAB = ROW * H;
struct Tabs {
  uint8_t *p[NF];
};
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: like the walker (drain every 8 rows); 1: never drain; 2: rows of a batch in reverse order;
// 3: units interleaved so that neighbouring waves belong to DIFFERENT tables; 5: every wave stays
// inside the first 16 rows of its table (1.4 MiB: one or two translation entries per table) and
// writes them over and over -- the same bytes per wave, no new pages
template <int MODE>
__global__ __launch_bounds__(256) void write_fronts(const Tabs t, int nf, int rows, int skew = 0) {
  const int lane = threadIdx.x & 63;
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= nf * STRIPS) return;
  int f, s;
  if (MODE == 3) {
    f = unit % nf;
    s = unit / nf;
  } else {
    f = unit / STRIPS;
    s = unit - f * STRIPS;
  }
  uint8_t *base = t.p[f] + (size_t)s * 3072 + (size_t)lane * 16;
  const u32x4 v{(uint32_t)unit, 1u, 2u, 3u};
  const int y_start = (f * skew) % rows & ~7;  // table f starts `skew` rows further down (wraps)
  for (int yy = 0; yy < rows; yy += 8) {
    int y = yy + y_start;
    if (y >= rows) y -= rows;
    if (MODE == 5) y &= 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int rr = MODE == 2 ? 7 - r : r;
      uint8_t *row = base + (size_t)(y + rr) * ROW;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(row + k * 1024), "v"(v) : "memory");
    }
    if (MODE != 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

template <int MODE>
static float run(const Tabs &t, int nf, const char *what, int skew = 0) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  float best = 1e30f, sum = 0;
  const int reps = 6;
  for (int r = 0; r < reps + 1; ++r) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(write_fronts<MODE>, dim3((nf * STRIPS + 3) / 4), dim3(256), 0, 0, t, nf, H, skew);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (r == 0) continue;
    best = ms < best ? ms : best;
    sum += ms;
  }
  if (skew) printf("  [skew %4d rows]", skew);
  printf("  %-46s mode %d: best %.1f us per table, mean %.1f  (%.2f TB/s)\n", what, MODE, best * 1e3f / nf,
         sum / reps * 1e3f / nf, (double)TAB * nf / (best * 1e-3) / 1e12);
  return best;
}

// One table written by 960 waves: wave (band, strip) walks 1/32 of the rows of its strip
__global__ __launch_bounds__(256) void write_one(uint8_t *tab) {
  const int lane = threadIdx.x & 63;
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= NF * STRIPS) return;
  const int band = unit / STRIPS, s = unit - band * STRIPS;
  uint8_t *base = tab + (size_t)s * 3072 + (size_t)lane * 16;
  const u32x4 v{(uint32_t)unit, 1u, 2u, 3u};
  const int y0 = band * (H / NF);
  for (int y = y0; y < y0 + H / NF; y += 8) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      uint8_t *row = base + (size_t)(y + r) * ROW;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(row + k * 1024), "v"(v) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

int per_bo_scan() {
  // is the write rate a property of each buffer object?  96 tables, each timed alone (32 row bands
  // x 30 strips in flight inside it), then 32 fronts over the 32 fastest and over the 32 slowest
  const int N = 96;
  std::vector<uint8_t *> bo(N);
  std::vector<float> us(N);
  for (int i = 0; i < N; ++i) CK(hipMalloc(&bo[i], TAB));
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int i = 0; i < N; ++i) {
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      (void)hipEventRecord(a);
      hipLaunchKernelGGL(write_one, dim3(NF * STRIPS / 4), dim3(256), 0, 0, bo[i]);
      (void)hipEventRecord(b);
      (void)hipEventSynchronize(b);
      float ms;
      (void)hipEventElapsedTime(&ms, a, b);
      if (r) best = ms < best ? ms : best;
    }
    us[i] = best * 1e3f;
  }
  std::vector<int> order(N);
  for (int i = 0; i < N; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int x, int y) { return us[x] < us[y]; });
  printf("per-table write alone, us (sorted):");
  for (int i = 0; i < N; ++i) printf(" %.1f", us[order[i]]);
  printf("\n");
  Tabs fast, slow, mid;
  for (int f = 0; f < NF; ++f) {
    fast.p[f] = bo[order[f]];
    slow.p[f] = bo[order[N - 1 - f]];
    mid.p[f] = bo[f];
  }
  run<0>(fast, NF, "32 fronts over the 32 fastest tables");
  run<0>(slow, NF, "32 fronts over the 32 slowest tables");
  run<0>(mid, NF, "32 fronts over the first 32 allocated");
  run<0>(fast, NF, "32 fronts over the 32 fastest tables (again)");
  return 0;
}

// Tables backed through the virtual-memory API: `chunk` bytes per physical handle (0 = one handle
// per table), all mapped into one reserved address range per table
static int vmm_table(uint8_t **out, size_t chunk) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
  if (chunk == 0) chunk = TAB;
  chunk = (chunk + gran - 1) / gran * gran;
  const size_t total = (TAB + chunk - 1) / chunk * chunk;
  void *va = nullptr;
  CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
  for (size_t off = 0; off < total; off += chunk) {
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, chunk, &prop, 0));
    CK(hipMemMap((char *)va + off, chunk, 0, h, 0));
  }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(va, total, &acc, 1));
  *out = (uint8_t *)va;
  return 0;
}

int vmm_scan() {
  Tabs sep;
  for (int f = 0; f < NF; ++f) CK(hipMalloc(&sep.p[f], TAB));
  run<0>(sep, NF, "hipMalloc, one allocation per table");
  for (size_t chunk : {(size_t)0, (size_t)2 << 20, (size_t)8 << 20, (size_t)16 << 20, (size_t)32 << 20, (size_t)32 << 20,
                       (size_t)48 << 20, (size_t)64 << 20, (size_t)128 << 20}) {
    Tabs t;
    for (int f = 0; f < NF; ++f)
      if (vmm_table(&t.p[f], chunk)) return 1;
    char what[96];
    snprintf(what, sizeof what, "hipMemCreate, %zu MiB per handle%s", chunk >> 20, chunk ? "" : " (one per table)");
    run<0>(t, NF, what);
  }
  run<0>(sep, NF, "hipMalloc, one allocation per table (again)");
  return 0;
}

// Round 5: is it the PHYSICAL interleave of what backs the tables?  32 tables of `chunk`-byte
// handles each; the handles of a table are (a) created and mapped in order, (b) created in order and
// mapped at SHUFFLED offsets of the table's address range, (c) created round-robin over the tables
// (table 0's first chunk, table 1's first chunk, ...: physically interleaved across tables) and
// mapped in order.  Same virtual layout, same handle size, same driver path in all three -- only the
// virtual-to-physical order differs.
static int vmm_tables(Tabs &t, size_t chunk, int how, unsigned seed) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
  chunk = (chunk + gran - 1) / gran * gran;
  const size_t n = (TAB + chunk - 1) / chunk, total = n * chunk;
  std::vector<std::vector<hipMemGenericAllocationHandle_t>> h(NF, std::vector<hipMemGenericAllocationHandle_t>(n));
  if (how == 2) {
    for (size_t k = 0; k < n; ++k)
      for (int f = 0; f < NF; ++f) CK(hipMemCreate(&h[f][k], chunk, &prop, 0));
  } else {
    for (int f = 0; f < NF; ++f)
      for (size_t k = 0; k < n; ++k) CK(hipMemCreate(&h[f][k], chunk, &prop, 0));
  }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  for (int f = 0; f < NF; ++f) {
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
    std::vector<size_t> slot(n);
    for (size_t k = 0; k < n; ++k) slot[k] = k;
    if (how == 1)
      for (size_t k = n - 1; k > 0; --k) {  // Fisher-Yates with an LCG
        seed = seed * 1664525u + 1013904223u;
        std::swap(slot[k], slot[(seed >> 8) % (k + 1)]);
      }
    for (size_t k = 0; k < n; ++k) CK(hipMemMap((char *)va + slot[k] * chunk, chunk, 0, h[f][k], 0));
    CK(hipMemSetAccess(va, total, &acc, 1));
    t.p[f] = (uint8_t *)va;
  }
  return 0;
}

int shuffle_scan() {
  Tabs sep;
  for (int f = 0; f < NF; ++f) CK(hipMalloc(&sep.p[f], TAB));
  run<0>(sep, NF, "hipMalloc, one allocation per table");
  const char *names[3] = {"in order", "shuffled inside each table", "created round-robin over the tables"};
  for (int rep = 0; rep < 2; ++rep)
    for (size_t chunk : {(size_t)2 << 20, (size_t)8 << 20, (size_t)32 << 20})
      for (int how = 0; how < 3; ++how) {
        Tabs t;
        if (vmm_tables(t, chunk, how, 12345u + rep)) return 1;
        char what[112];
        snprintf(what, sizeof what, "%zu MiB handles, %s", chunk >> 20, names[how]);
        run<0>(t, NF, what);
      }
  run<0>(sep, NF, "hipMalloc, one allocation per table (again)");
  return 0;
}

int pool_scan() {
  // ONE pool; tables carved at different base offsets and pitches inside it: the same physical
  // backing throughout, so whatever differs is decided by the addresses alone
  const size_t pool_bytes = (size_t)40 << 30;
  uint8_t *pool;
  CK(hipMalloc(&pool, pool_bytes));
  printf("pool at %p\n", (void *)pool);
  for (int rep = 0; rep < 2; ++rep)
    for (size_t base_mib : {0, 2, 4, 6, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 3, 5, 100}) {
      for (size_t extra_mib : {0, 5, 55}) {
        const size_t pitch = TAB + (extra_mib << 20);
        Tabs t;
        for (int f = 0; f < NF; ++f) t.p[f] = pool + (base_mib << 20) + (size_t)f * pitch;
        char what[96];
        snprintf(what, sizeof what, "pool + %zu MiB, pitch table + %zu MiB", base_mib, extra_mib);
        run<0>(t, NF, what);
      }
    }
  return 0;
}

int main(int argc, char **argv) {
  if (argc > 1 && argv[1][0] == 'p') return pool_scan();
  if (argc > 1 && argv[1][0] == 'b') return per_bo_scan();
  if (argc > 1 && argv[1][0] == 'v') return vmm_scan();
  if (argc > 1 && argv[1][0] == 's') return shuffle_scan();
  const bool scan = argc > 1;
  // A: one allocation per table
  Tabs sep;
  for (int f = 0; f < NF; ++f) CK(hipMalloc(&sep.p[f], TAB));
  printf("separate allocations, deltas:");
  for (int f = 1; f < 4; ++f) printf(" %lld", (long long)(sep.p[f] - sep.p[f - 1]));
  printf("\n");
  run<0>(sep, NF, "one allocation per table");
  run<1>(sep, NF, "one allocation per table");
  run<3>(sep, NF, "one allocation per table");
  for (int skew : {8, 40, 120}) run<0>(sep, NF, "one allocation per table", skew);
  // B: one slab, several pitches (argv[1] = "scan": many)
  std::vector<size_t> pitches = {TAB, (size_t)512 << 20, TAB + ((size_t)5 << 20), TAB + 92160, (size_t)1 << 30};
  if (scan) {
    pitches.clear();
    for (int mib : {0, 1, 2, 3, 4, 5, 6, 7, 9, 11, 13, 17, 21, 27, 37, 53, 67, 101}) pitches.push_back(TAB + ((size_t)mib << 20));
    for (int kib : {128, 256, 384, 640, 896, 1152}) pitches.push_back(TAB + ((size_t)kib << 10));
  }
  for (size_t pitch : pitches) {
    uint8_t *slab;
    CK(hipMalloc(&slab, pitch * NF));
    Tabs t;
    for (int f = 0; f < NF; ++f) t.p[f] = slab + (size_t)f * pitch;
    char what[96];
    snprintf(what, sizeof what, "one slab, pitch %zu (table + %lld)", pitch, (long long)(pitch - TAB));
    run<0>(t, NF, what);
    if (!scan) run<3>(t, NF, what);
    CK(hipFree(slab));
  }
  // C: one 512 MiB allocation per table
  Tabs big;
  for (int f = 0; f < NF; ++f) CK(hipMalloc(&big.p[f], (size_t)512 << 20));
  run<0>(big, NF, "one 512 MiB allocation per table");
  run<5>(sep, NF, "one allocation per table, 16 rows over and over");
  {
    uint8_t *slab;
    CK(hipMalloc(&slab, TAB * NF));
    Tabs t;
    for (int f = 0; f < NF; ++f) t.p[f] = slab + (size_t)f * TAB;
    run<0>(t, NF, "one slab");
    run<5>(t, NF, "one slab, 16 rows over and over");
    CK(hipFree(slab));
  }
  // D: fewer tables at a time with the same number of waves?  16 tables, 30 strips: half the waves
  run<0>(sep, 16, "one allocation per table, 16 tables");
  return 0;
}
