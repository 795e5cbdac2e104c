// host_tables.cpp -- see host_tables.h.
#include "host_tables.h"

#include <cmath>

namespace f360 {
namespace {

// OpenCL float builtins as correctly rounded floats.
inline float exp_f(float x) { return (float)std::exp((double)x); }
inline float pow_f(float x, float y) { return (float)std::pow((double)x, (double)y); }
inline float log_f(float x) { return (float)std::log((double)x); }
inline float cos_f(float x) { return (float)std::cos((double)x); }
inline float sin_f(float x) { return (float)std::sin((double)x); }

inline int sgn(int v) { return (v > 0) - (v < 0); }
inline unsigned uabs(int v) { return v < 0 ? 0u - (unsigned)v : (unsigned)v; }

inline float lambda_of(int full) { return (float)full / (exp_f(1.0f) - 1.0f); }

// max((int)a, (int)(lambda * (exp(pow(2.0f*a/n, 4.0f)) - 1))), float math
inline int radial_f32(unsigned a, int n, float lambda) {
  const float t = 2.0f * (float)a / (float)n;
  const float e = exp_f(pow_f(t, 4.0f)) - 1.0f;
  const int v = (int)(lambda * e);
  return (int)a > v ? (int)a : v;
}
// same expression in double math (src/sat_decoder_interpolate_kernel.cl:56-65)
inline int radial_f64(unsigned a, int n, float lambda) {
  const double t = 2.0 * (double)a / (double)n;
  const double e = std::exp(std::pow(t, 4.0)) - 1.0;
  const int v = (int)((double)lambda * e);
  return (int)a > v ? (int)a : v;
}

}  // namespace

void build_yuv2rgb_consts(YuvConsts &k) {
  // ff_yuv2rgb_coeffs[SWS_CS_DEFAULT] (yuv2rgb.c:49-61): {crv, cbu, cgu, cgv} of ITU-R 601
  int64_t crv = 104597, cbu = 132201, cgu = -25675, cgv = -53279;
  // limited range: luma gain 255/219, black at 16 (yuv2rgb.c:813-815); contrast and
  // saturation are 1.0 and brightness 0, so :823-828 change nothing
  const int64_t cy = ((int64_t)(1 << 16) * 255) / 219;
  const int64_t oy = (int64_t)16 << 16;
  // 16-bit lanes of the MMX converter (roundToInt16, yuv2rgb.c:762-772,830-837)
  auto lane16 = [](int64_t f) {
    const int r = (int)((f + (1 << 15)) >> 16);
    return r < -0x7FFF ? -0x8000 : r > 0x7FFF ? 0x7FFF : r;
  };
  k.yc = lane16(cy * (1 << 13));
  k.vrc = lane16(crv * (1 << 13));
  k.ubc = lane16(cbu * (1 << 13));
  k.vgc = lane16(cgv * (1 << 13));
  k.ugc = lane16(cgu * (1 << 13));
  k.yoff = lane16(oy * (1 << 3));
  // C converter: chroma increments in table entries (yuv2rgb.c:846-850) ...
  crv = ((crv * (1 << 16)) + 0x8000) / cy;
  cbu = ((cbu * (1 << 16)) + 0x8000) / cy;
  cgu = ((cgu * (1 << 16)) + 0x8000) / cy;
  cgv = ((cgv * (1 << 16)) + 0x8000) / cy;
  k.crv = (int)crv;
  k.cbu = (int)cbu;
  k.cgu = (int)cgu;
  k.cgv = (int)cgv;
  // ... the pointer bias of fill_table / fill_gv_table (:743,:754; >> floors) ...
  k.r0 = -(int)(crv >> 9);
  k.b0 = -(int)(cbu >> 9);
  k.gu0 = -(int)(cgu >> 9);
  k.gv0 = -(int)(cgv >> 9);
  // ... and the table itself, entry i = clip8((yb + i * cy + 0x8000) >> 16) with
  // yb = -(384 << 16) - 512 * cy - oy (:978), read at i = yoffs + offsets + Y,
  // yoffs = 326 + 512 (:802)
  const int64_t yb = -((int64_t)384 << 16) - 512 * cy - oy;
  k.cy = (int)cy;
  k.c0 = (int)(yb + (326 + 512) * cy + 0x8000);
}

// These two follow host C++ of the reference (not OpenCL C), so they use the C library's float
// routines as that code does.
void build_expand_axis(std::vector<int32_t> &d, int n_reduced, int n_full) {
  d.resize((size_t)n_reduced);
  const float lambda = n_full / (std::exp(1.0f) - 1);
  for (int i = 0; i < n_reduced; ++i) {
    const int u = i - n_reduced / 2;
    const int a = u < 0 ? -u : u;
    const int b = (int)(lambda * (std::exp(std::pow(2.0 * a / n_reduced, 4.0)) - 1));
    d[(size_t)i] = (a > b ? a : b) * sgn(u);
  }
}

void build_expand_logpolar_axes(std::vector<float> &radius, std::vector<double> &cs,
                                std::vector<double> &sn, int src_w, int src_h) {
  radius.resize((size_t)src_w);
  cs.resize((size_t)src_h);
  sn.resize((size_t)src_h);
  const float alpha = 1.0f;
  for (int i = 0; i < src_w; ++i)
    radius[(size_t)i] = std::exp(10.0f * std::pow((float)i / src_w, alpha));
  for (int j = 0; j < src_h; ++j) {
    const double angle = (float)j / src_h * 2 * M_PI;
    cs[(size_t)j] = std::cos(angle);
    sn[(size_t)j] = std::sin(angle);
  }
}

void build_satdec_grid_axis(std::vector<int16_t> &g, int n_out, int n_src) {
  g.resize((size_t)n_out + 1);
  const float lambda = lambda_of(n_src);
  for (int t = 0; t <= n_out; ++t) {
    const int u = (t - 1) - n_out / 2;
    const int lo = radial_f32(uabs(u), n_out, lambda) * sgn(u);
    const int hi = radial_f32(uabs(u + 1), n_out, lambda) * sgn(u + 1);
    g[(size_t)t] = (int16_t)std::floor((float)(lo + hi) / 2.0f);
  }
}

void build_is_grid_axis(std::vector<int16_t> &g, int n_out, int n_src) {
  g.resize((size_t)n_out);
  const float lambda = lambda_of(n_src);
  for (int t = 0; t < n_out; ++t) {
    const int u = t - n_out / 2;
    g[(size_t)t] = (int16_t)(radial_f32(uabs(u), n_out, lambda) * sgn(u));
  }
}

void build_logpolar_axes(std::vector<float> &radius, std::vector<float> &cs,
                         std::vector<float> &sn, int out_w, int out_h) {
  radius.resize((size_t)out_w);
  cs.resize((size_t)out_h);
  sn.resize((size_t)out_h);
  for (int i = 0; i < out_w; ++i)
    radius[(size_t)i] = exp_f(10.0f * pow_f((float)i / (float)out_w, 1.0f));
  for (int j = 0; j < out_h; ++j) {
    // (float)((float)j / h * 2.0f * _PI), _PI = 3.14159265359 (double)
    const float a =
        (float)((double)((float)j / (float)out_h * 2.0f) * 3.14159265359);
    cs[(size_t)j] = cos_f(a);
    sn[(size_t)j] = sin_f(a);
  }
}

void build_logpolar_inverse_axes(std::vector<float> &radius,
                                 std::vector<double> &cs,
                                 std::vector<double> &sn, int src_w,
                                 int src_h) {
  radius.resize((size_t)src_w);
  cs.resize((size_t)src_h);
  sn.resize((size_t)src_h);
  for (int i = 0; i < src_w; ++i)
    radius[(size_t)i] = exp_f(10.0f * pow_f((float)i / (float)src_w, 1.0f));
  for (int j = 0; j < src_h; ++j) {
    const double a = (double)((float)j / (float)src_h * 2.0f) * M_PI;
    cs[(size_t)j] = std::cos(a);
    sn[(size_t)j] = std::sin(a);
  }
}

void build_interp_axis(std::vector<InterpAxisEntry> &t, int range, int n_full,
                       int n_reduced) {
  t.resize((size_t)2 * range + 1);
  const float lambda = lambda_of(n_full);
  for (int d = -range; d <= range; ++d) {
    // ceil(0.5 * n_reduced * pow(log(abs(d) / lambda + 1), 0.25f)) * sgn(d)
    const float lg = log_f((float)uabs(d) / lambda + 1.0f);
    const float pw = pow_f(lg, 0.25f);
    int u = (int)(std::ceil(0.5 * (double)n_reduced * (double)pw) *
                  (double)sgn(d));
    if (uabs(u) > uabs(d) || u == 0) u = d;
    InterpAxisEntry e;
    e.u = u;
    e.dcalc = radial_f64(uabs(u), n_reduced, lambda) * sgn(u);
    e.du = -sgn(d);
    e.dmin = radial_f32(uabs(u + e.du), n_reduced, lambda) * sgn(u);
    t[(size_t)(d + range)] = e;
  }
}

}  // namespace f360

// ---- C ABI: host-only exports (include/f360.h "host-only geometry tables") ----
#include <cstring>

#include "f360_internal.h"

extern "C" {

int f360_tables_satdec_grid_axis(int16_t *out, int n_out, int n_src) {
  F360_REQUIRE(out && n_out >= 1 && n_src >= 1, "f360_tables_satdec_grid_axis: bad argument");
  std::vector<int16_t> g;
  f360::build_satdec_grid_axis(g, n_out, n_src);
  std::memcpy(out, g.data(), g.size() * sizeof(int16_t));
  return F360_OK;
}

int f360_tables_is_grid_axis(int16_t *out, int n_out, int n_src) {
  F360_REQUIRE(out && n_out >= 1 && n_src >= 1, "f360_tables_is_grid_axis: bad argument");
  std::vector<int16_t> g;
  f360::build_is_grid_axis(g, n_out, n_src);
  std::memcpy(out, g.data(), g.size() * sizeof(int16_t));
  return F360_OK;
}

int f360_tables_logpolar_axes(float *radius, float *cs, float *sn, int out_w, int out_h) {
  F360_REQUIRE(radius && cs && sn && out_w >= 1 && out_h >= 1,
               "f360_tables_logpolar_axes: bad argument");
  std::vector<float> r, c, s;
  f360::build_logpolar_axes(r, c, s, out_w, out_h);
  std::memcpy(radius, r.data(), r.size() * sizeof(float));
  std::memcpy(cs, c.data(), c.size() * sizeof(float));
  std::memcpy(sn, s.data(), s.size() * sizeof(float));
  return F360_OK;
}

int f360_tables_yuv2rgb(int32_t *out16) {
  F360_REQUIRE(out16, "f360_tables_yuv2rgb: bad argument");
  f360::YuvConsts k;
  f360::build_yuv2rgb_consts(k);
  const int32_t v[16] = {k.cy,  k.c0, k.crv, k.cbu, k.cgu, k.cgv, k.r0,  k.gu0,
                         k.gv0, k.b0, k.yc,  k.vrc, k.ubc, k.vgc, k.ugc, k.yoff};
  std::memcpy(out16, v, sizeof(v));
  return F360_OK;
}

int f360_tables_interp_axis(int32_t *out, int range, int n_full, int n_reduced) {
  F360_REQUIRE(out && range >= 0 && n_full >= 1 && n_reduced >= 1,
               "f360_tables_interp_axis: bad argument");
  std::vector<f360::InterpAxisEntry> t;
  f360::build_interp_axis(t, range, n_full, n_reduced);
  std::memcpy(out, t.data(), t.size() * sizeof(t[0]));
  return F360_OK;
}

}  // extern "C"
