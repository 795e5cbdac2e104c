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
// allocation API (profiles/round4_table_placement.txt, tools/frontbench).  User space can only
// measure it.  So the pool draws groups: every group allocated while all earlier ones are still
// held (so that it is backed by other memory), alternately as one allocation per table and as one
// slab, one encode launch of scratch frames timed into each (the second of two), the fastest
// ones kept, the rest given back.  Calls too small for the read-once encoder get plain allocations.
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
  const int strips = (width + kStripPx - 1) / kStripPx;
  // (the conditions under which f360_sat_encode_batch takes the read-once encoder)
  const bool walks = walk_wanted(ctx, count, width) && width % 4 == 0 &&
                     (size_t)width * height * 3 < ((size_t)1 << 31);
  char line[160];
  if (!walks) {
    for (int k = 0; k < count; ++k) {
      void *p = nullptr;
      if (hipMalloc(&p, tb) != hipSuccess) {
        (void)hipGetLastError();
        f360::set_error("f360_sat_tables_alloc: out of device memory");
        return fail(F360_ERR_HIP);
      }
      pool->allocs.push_back(p);
      tables_out[k] = static_cast<uint32_t *>(p);
    }
    pool->report = "one allocation per table (the call is below the read-once encoder's threshold)";
    *pool_out = pool;
    return F360_OK;
  }
  int max_frames = ctx->opt_walk_frames > 0 ? ctx->opt_walk_frames : std::max(1024 / strips, 1);
  max_frames = std::min(max_frames, kWalkFrames);
  const int nlaunch = (count + max_frames - 1) / max_frames;
  const int group = (count + nlaunch - 1) / nlaunch;
  struct Draw {
    float us = 0;
    bool slab = false;
    std::vector<void *> allocs;
    std::vector<uint32_t *> tabs;
  };
  std::vector<Draw> draws;
  auto drop = [&](Draw &d) {
    for (void *p : d.allocs) (void)hipFree(p);
    d.allocs.clear();
  };
  void *zero = nullptr;
  const size_t fb = (size_t)width * height * 4;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int st = F360_OK;
  // (sources: `group` frames' worth of scratch, contents irrelevant -- distinct frames, so that
  // the timed launch reads from memory as a real one does)
  if (hipMalloc(&zero, fb * group) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
      hipEventCreate(&e1) != hipSuccess)
    st = F360_ERR_HIP;
  const float good_us = 80.5f * (float)((double)width * height / (7680.0 * 3840.0));
  std::vector<const uint8_t *> srcs((size_t)group);
  for (int k = 0; k < group; ++k) srcs[(size_t)k] = static_cast<const uint8_t *>(zero) + (size_t)k * fb;
  // (up to nlaunch + 12 draws: on the boxes where most allocations draw badly -- 7 of 8 at
  // 86-94 us seen -- eight draws left a 1-in-4 chance of keeping a mediocre group; a draw is
  // 11 GB at 8K, held until the choice is made, and running out of memory just ends the drawing)
  for (int i = 0; st == F360_OK && i < nlaunch + 12; ++i) {
    Draw d;
    d.slab = i % 2 == 1;
    if (d.slab) {
      void *p = nullptr;
      if (hipMalloc(&p, tb * group) != hipSuccess) {
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
    snprintf(line, sizeof line, "%s%s %.1f", draws.empty() ? "" : ", ", d.slab ? "slab" : "separate",
             d.us);
    pool->report += line;
    draws.push_back(std::move(d));
    int good = 0;
    for (const Draw &x : draws) good += x.us <= good_us;
    if (good >= nlaunch) break;
  }
  if (zero) (void)hipFree(zero);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st != F360_OK || (int)draws.size() < nlaunch) {
    for (Draw &d : draws) drop(d);
    if (st == F360_OK) {
      f360::set_error("f360_sat_tables_alloc: out of device memory");
      st = F360_ERR_HIP;
    }
    return fail(st);
  }
  std::sort(draws.begin(), draws.end(), [](const Draw &a, const Draw &b) { return a.us < b.us; });
  pool->report = "us per frame of one launch into each drawn group of " + std::to_string(group) +
                 " tables: " + pool->report + "; kept:";
  int k = 0;
  for (int g = 0; g < (int)draws.size(); ++g) {
    if (g < nlaunch) {
      snprintf(line, sizeof line, " %s %.1f", draws[g].slab ? "slab" : "separate", draws[g].us);
      pool->report += line;
      for (void *p : draws[g].allocs) pool->allocs.push_back(p);
      for (uint32_t *t : draws[g].tabs)
        if (k < count) tables_out[k++] = t;
    } else {
      drop(draws[g]);
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
