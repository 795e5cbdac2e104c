// table_pool.cpp -- tables for batched encodes, placed for the read-once encoder
// (f360_sat_tables_alloc / _free / _report).
#include "sat_walk.h"

using namespace f360::sat;

// ------------------------------------------------------------------------------------------
// Tables for batched encodes, placed for the read-once encoder (f360_sat_tables_alloc).
//
// A launch of sat_walk_kernel writes `group` tables (32 at 8K) at the same time, and the rate at
// which this device takes those writes depends on what backs the tables: 5.5 to 7 TB/s for the
// write pattern alone, in two clusters, by ALLOCATION -- not by address, pitch, row phase or
// allocation API (profiles/round4_table_placement.txt, tools/frontbench; round 5's experiment on
// the cause: profiles/round5_table_pool.txt).  User space can only measure it.  So the pool
// draws groups -- alternately one allocation per table and one slab -- times one encode launch
// of scratch frames into each (the second of two), keeps the fastest and gives the rest back.
// A new group is drawn while earlier ones are still held, so that it is backed by other memory
// than the ones already seen; how much may be held at once is BOUNDED ("sat.pool_mb", default a
// third of the device memory that is free when the call starts, never less than the tables the
// caller asked for plus one group): when a new draw would exceed the bound, the slowest group held
// beyond those worth keeping is given back first.  Drawing stops when enough groups lie within
// 4 % of the fastest seen and a slower cluster has been seen beside them (or eight more groups
// than needed have been drawn), after `nlaunch + 12` draws, or when memory runs out.  Calls too small for the read-once encoder get plain
// allocations.
struct f360_table_pool {
  std::vector<void *> allocs;   // what f360_sat_tables_free gives back
  std::string report;
};

extern "C" int f360_sat_tables_alloc(f360_ctx *ctx, int width, int height, int count,
                                     uint32_t **tables_out, f360_table_pool **pool_out) {
  F360_REQUIRE(ctx && tables_out && pool_out && count >= 1 && width >= 1 && height >= 1,
               "f360_sat_tables_alloc: bad argument");
  F360_REQUIRE(f360::dims_ok({width, height}), "f360_sat_tables_alloc: a dimension exceeds 65536");
  F360_BIND_DEVICE(ctx);
  *pool_out = nullptr;
  const size_t tb = (size_t)width * height * 12;
  auto *pool = new f360_table_pool;
  auto fail = [&](int st) {
    for (void *p : pool->allocs) (void)hipFree(p);
    delete pool;
    return st;
  };
  // (the conditions under which f360_sat_encode_batch takes the read-once encoder)
  const bool walks = walk_wanted(ctx, count, width) && width % 4 == 0 &&
                     (size_t)width * height * 3 < ((size_t)1 << 31);
  char line[200];
  if (!walks) {
    for (int k = 0; k < count; ++k) {
      void *p = nullptr;
      if (hipMalloc(&p, tb) != hipSuccess) {
        (void)hipGetLastError();
        f360::set_error("f360_sat_tables_alloc: out of device memory");
        return fail(F360_ERR_OOM);
      }
      pool->allocs.push_back(p);
      tables_out[k] = static_cast<uint32_t *>(p);
    }
    pool->report = "one allocation per table (the call is below the read-once encoder's threshold)";
    *pool_out = pool;
    return F360_OK;
  }
  const int group = walk_frames_per_launch(ctx, count, width);
  const int nlaunch = (count + group - 1) / group;
  const size_t group_bytes = tb * (size_t)group;
  // the bound on what the pool holds at any time, scratch frames included
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    free_b = total_b = 0;
  }
  const size_t fb = (size_t)width * height * 4;
  const size_t needed = group_bytes * (size_t)(nlaunch + 1) + fb * (size_t)group;
  size_t cap = ctx->opt_pool_mb > 0 ? (size_t)ctx->opt_pool_mb << 20 : free_b / 3;
  cap = std::max(cap, needed);
  const int hold_max = (int)std::max<size_t>((cap - fb * (size_t)group) / group_bytes, (size_t)(nlaunch + 1));
  struct Draw {
    float us = 0;
    bool slab = false;
    std::vector<void *> allocs;
    std::vector<uint32_t *> tabs;
  };
  std::vector<Draw> held;  // groups currently allocated
  auto drop = [&](Draw &d) {
    for (void *p : d.allocs) (void)hipFree(p);
    d.allocs.clear();
  };
  void *zero = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int st = F360_OK;
  // (sources: `group` frames' worth of scratch, contents irrelevant -- distinct frames, so that
  // the timed launch reads from memory as a real one does)
  if (hipMalloc(&zero, fb * group) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
      hipEventCreate(&e1) != hipSuccess) {
    (void)hipGetLastError();
    f360::set_error("f360_sat_tables_alloc: cannot allocate the scratch frames (%zu bytes) or "
                    "the timing events", fb * (size_t)group);
    st = F360_ERR_OOM;
  }
  std::vector<const uint8_t *> srcs((size_t)group);
  for (int k = 0; k < group; ++k) srcs[(size_t)k] = static_cast<const uint8_t *>(zero) + (size_t)k * fb;
  std::vector<float> seen;  // every draw's time, for the report and the stopping rule
  size_t held_peak = 0;
  int draws = 0, returned_early = 0;
  for (; st == F360_OK && draws < nlaunch + 12; ++draws) {
    // room for one more group?  give the slowest one back first (never one of the `nlaunch`
    // fastest: those are the ones the caller gets)
    if ((int)held.size() >= hold_max) {
      size_t worst = 0;
      for (size_t k = 1; k < held.size(); ++k)
        if (held[k].us > held[worst].us) worst = k;
      drop(held[worst]);
      held.erase(held.begin() + (long)worst);
      ++returned_early;
    }
    Draw d;
    d.slab = draws % 2 == 1;
    if (d.slab) {
      void *p = nullptr;
      if (hipMalloc(&p, group_bytes) != hipSuccess) {
        (void)hipGetLastError();
        break;  // out of memory: make do with what has been drawn
      }
      d.allocs.push_back(p);
      for (int k = 0; k < group; ++k)
        d.tabs.push_back(reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(p) + (size_t)k * tb));
    } else {
      bool ok = true;
      for (int k = 0; k < group && ok; ++k) {
        void *p = nullptr;
        ok = hipMalloc(&p, tb) == hipSuccess;
        if (ok) {
          d.allocs.push_back(p);
          d.tabs.push_back(static_cast<uint32_t *>(p));
        }
      }
      if (!ok) {
        (void)hipGetLastError();
        drop(d);
        break;
      }
    }
    for (int rep = 0; rep < 2 && st == F360_OK; ++rep) {  // the second launch is the measurement
      if (hipEventRecord(e0, ctx->stream) != hipSuccess) st = F360_ERR_HIP;
      if (st == F360_OK)
        st = sat_encode_walk(ctx, group, d.tabs.data(), srcs.data(), nullptr, width, height,
                             4 * width, false);
      if (st == F360_OK && (hipEventRecord(e1, ctx->stream) != hipSuccess ||
                            hipEventSynchronize(e1) != hipSuccess))
        st = F360_ERR_HIP;
      float ms = 0;
      if (st == F360_OK && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) st = F360_ERR_HIP;
      d.us = 1e3f * ms / (float)group;
    }
    if (st != F360_OK) {
      (void)hipGetLastError();
      f360::set_error("f360_sat_tables_alloc: timing a launch into a drawn group failed");
      drop(d);
      break;
    }
    snprintf(line, sizeof line, "%s%s %.1f", seen.empty() ? "" : ", ", d.slab ? "slab" : "separate", d.us);
    pool->report += line;
    seen.push_back(d.us);
    held.push_back(std::move(d));
    held_peak = std::max(held_peak, held.size() * group_bytes + fb * (size_t)group);
    // Enough good ones?  "Good" is relative to the fastest draw seen, so the rule holds on any
    // device and clock -- but a fastest draw is only known to be fast once a slower cluster has
    // been seen beside it (the two lie 10-15 % apart): on some boxes five draws in a row land in
    // the slow one (round 5: 90.4 .. 92.0 us, and the same command's next set-up drew 78).  So:
    // enough groups within 4 % of the fastest, at least three more draws than needed, and either
    // a spread of 6 % among the draws or eight more draws than needed.
    const float best = *std::min_element(seen.begin(), seen.end());
    const float worst_seen = *std::max_element(seen.begin(), seen.end());
    int good = 0;
    for (const Draw &x : held) good += x.us <= best * 1.04f;
    if (good >= nlaunch && draws + 1 >= nlaunch + 3 &&
        (worst_seen >= best * 1.06f || draws + 1 >= nlaunch + 8)) {
      ++draws;
      break;
    }
  }
  if (zero) (void)hipFree(zero);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st != F360_OK || (int)held.size() < nlaunch) {
    for (Draw &d : held) drop(d);
    if (st == F360_OK) {
      f360::set_error("f360_sat_tables_alloc: out of device memory (%d of %d groups of %zu bytes)",
                      (int)held.size(), nlaunch, group_bytes);
      st = F360_ERR_OOM;
    }
    return fail(st);
  }
  std::sort(held.begin(), held.end(), [](const Draw &a, const Draw &b) { return a.us < b.us; });
  snprintf(line, sizeof line,
           "us per frame of one launch into each of %d drawn groups of %d tables (at most %d held "
           "at once: %.1f of %.1f GB allowed, %d given back before the end): ",
           draws, group, hold_max, (double)held_peak / 1e9, (double)cap / 1e9, returned_early);
  pool->report = std::string(line) + pool->report + "; kept:";
  int k = 0;
  for (int g = 0; g < (int)held.size(); ++g) {
    if (g < nlaunch) {
      snprintf(line, sizeof line, " %s %.1f", held[g].slab ? "slab" : "separate", held[g].us);
      pool->report += line;
      for (void *p : held[g].allocs) pool->allocs.push_back(p);
      for (uint32_t *t : held[g].tabs)
        if (k < count) tables_out[k++] = t;
    } else {
      drop(held[g]);
    }
  }
  *pool_out = pool;
  return F360_OK;
}

extern "C" int f360_sat_tables_free(f360_ctx *ctx, f360_table_pool *pool) {
  F360_REQUIRE(ctx, "f360_sat_tables_free: null context");
  if (!pool) return F360_OK;
  F360_BIND_DEVICE(ctx);
  F360_HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (void *p : pool->allocs) (void)hipFree(p);
  delete pool;
  return F360_OK;
}

extern "C" const char *f360_sat_tables_report(const f360_table_pool *pool) {
  return pool ? pool->report.c_str() : "";
}
