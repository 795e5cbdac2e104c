// Builds the product's host-side table code (csrc/host_tables.cpp) under ASan + UBSan and walks
// it over small and benchmark-sized geometries.  CPU only (GPU sanitizers are not available).
#include <cstdio>
#include <vector>

#include "host_tables.h"

int main() {
  long checksum = 0;
  const int sizes[][4] = {{1, 1, 2, 2},       {2, 3, 4, 4},         {48, 32, 64, 32},
                          {1072, 608, 1920, 1080}, {4272, 2144, 7680, 3840}, {33, 77, 200, 120}};
  for (const auto &s : sizes) {
    std::vector<int16_t> g;
    f360::build_satdec_grid_axis(g, s[0], s[2]);
    checksum += g.front() + g.back() + (long)g.size();
    f360::build_satdec_grid_axis(g, s[1], s[3]);
    checksum += g.front() + g.back();
    f360::build_is_grid_axis(g, s[0], s[2]);
    checksum += g.front() + g.back();
    std::vector<float> r, c, sn;
    f360::build_logpolar_axes(r, c, sn, s[0], s[1]);
    checksum += (long)r.size() + (long)c.size();
    std::vector<double> cd, sd;
    f360::build_logpolar_inverse_axes(r, cd, sd, s[0], s[1]);
    std::vector<f360::InterpAxisEntry> t;
    f360::build_interp_axis(t, 2 * s[2], s[2], s[0]);
    checksum += t.front().u + t.back().dcalc + (long)t.size();
  }
  std::printf("ok %ld\n", checksum);
  return 0;
}
