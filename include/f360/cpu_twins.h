// f360/cpu_twins.h -- host implementations of the reference's *CPU methods, so that every mode
// of src/run_satlogrectilinear.cc links against the drop-in classes: SATEncoder::EncodeFrameCPU
// (src/sat_encoder.cc:137-185), SATDecoder::DecodeFrameCPU / ExpandSampledFrameRectCPU /
// InterpolateFrameRectCPU (src/sat_decoder.cc:212-299,555-616,618-772) and the ImageSampler
// twins (src/image_sampler.cc:302-356,358-575,623-778).
//
// These are product code for the HOST side of the boundary (debug views and the tool's
// interpolate_sampled mode run them on AVFrames); they are plain C++, need no device and do not
// touch oracle/.  Each keeps the operand types of the function it replaces -- which
// sub-expressions are float and which double, truncating conversions, lerp in double
// (src/sat_decoder.h:42) -- because those decide individual bytes.  What differs is the shape:
// every exp / pow / log depends on ONE axis, so it is tabulated per column and per row before
// the pixel loop (the reference re-evaluates ~8 transcendentals per pixel), and the loops run
// row-major.  The results are identical; the scatter views keep the reference's "later write
// wins" order, which is separable too (largest source column, then largest source row).
// Where the reference indexes outside its source frame (no clamps: src/sat_decoder.cc:724-735,
// undefined behaviour near the frame edges), the index is clamped into the frame here.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <vector>

namespace f360cpu {

inline int sgn(int v) { return (v > 0) - (v < 0); }
// src/sat_decoder.h:41-42: float clamp, lerp evaluated in double and returned as float
inline float clampf(float a, float b, float c) { return std::min(std::max(a, b), c); }
inline float lerpf(float a, float b, float c) { return (float)(a * (1.0 - c) + b * c); }

// f(u) of the log-rectilinear map as the CPU methods write it: max(|u|, (int)(lambda *
// (exp(pow(2.0 * |u| / n_reduced, 4.0)) - 1))) * sgn(u), pow and exp in double, lambda float
inline int rect_forward(int u, float lambda, int n_reduced) {
  const int a = std::abs(u);
  const int far = (int)(lambda * (std::exp(std::pow(2.0 * a / n_reduced, 4.0)) - 1));
  return std::max(a, far) * sgn(u);
}
inline float rect_lambda(int n_full) { return n_full / (std::exp(1.0f) - 1); }

// ---------------------------------------------------------------- SATEncoder::EncodeFrameCPU
// uint32 table [height][width][3] of a packed frame; sums wrap mod 2^32 like the reference's.
inline void encode_frame(uint32_t *table, int width, int height, const uint8_t *frame,
                         int linesize) {
  const int bpp = linesize / width;
  const size_t row = (size_t)3 * width;
  for (int y = 0; y < height; ++y) {
    const uint8_t *src = frame + (size_t)y * linesize;
    uint32_t *dst = table + (size_t)y * row;
    const uint32_t *above = y > 0 ? dst - row : nullptr;
    uint32_t run[3] = {0, 0, 0};  // prefix along the row
    for (int x = 0; x < width; ++x)
      for (int c = 0; c < 3; ++c) {
        run[c] += src[(size_t)x * bpp + c];
        dst[(size_t)3 * x + c] = run[c] + (above ? above[(size_t)3 * x + c] : 0u);
      }
  }
}

// ---------------------------------------------------------------- SATDecoder::DecodeFrameCPU
// 1x1 boxes; the quotient is stored through a plain uint32 -> uint8 conversion.
inline void decode_frame(uint8_t *out, int out_linesize, const uint32_t *table, int width,
                         int height) {
  const int bpp = out_linesize / width;
  const size_t row = (size_t)3 * width;
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x)
      for (int c = 0; c < 3; ++c) {
        const size_t at = (size_t)y * row + (size_t)3 * x + c;
        uint32_t v = table[at];
        if (x > 0 && y > 0) v = v - table[at - row] + table[at - row - 3] - table[at - 3];
        else if (x > 0) v -= table[at - 3];
        else if (y > 0) v -= table[at - row];
        out[(size_t)y * out_linesize + (size_t)x * bpp + c] = (uint8_t)v;
      }
}

// ------------------------------------------- {SATDecoder,ImageSampler}::ExpandSampledFrameRectCPU
// Scatter of the reduced frame to its forward positions (a debug view).
inline void expand_rect(uint8_t *target, int tw, int th, int t_linesize, const uint8_t *source,
                        int sw, int sh, int s_linesize, float center_x, float center_y) {
  const int tbpp = t_linesize / tw, sbpp = s_linesize / sw;
  const float lx = rect_lambda(tw), ly = rect_lambda(th);
  std::vector<int> xs((size_t)sw), ys((size_t)sh);
  for (int i = 0; i < sw; ++i)
    xs[(size_t)i] = (int)(center_x * tw + rect_forward(i - sw / 2, lx, sw));
  for (int j = 0; j < sh; ++j)
    ys[(size_t)j] = (int)(center_y * th + rect_forward(j - sh / 2, ly, sh));
  for (int j = 0; j < sh; ++j) {
    const int y = ys[(size_t)j];
    if (y < 0 || y >= th) continue;
    for (int i = 0; i < sw; ++i) {
      const int x = xs[(size_t)i];
      if (x < 0 || x >= tw) continue;
      const uint8_t *s = source + (size_t)j * s_linesize + (size_t)i * sbpp;
      uint8_t *t = target + (size_t)y * t_linesize + (size_t)x * tbpp;
      t[0] = s[0];
      t[1] = s[1];
      t[2] = s[2];
    }
  }
}

// ------------------------------------------- {SATDecoder,ImageSampler}::InterpolateFrameRectCPU
// One axis of the un-warp for every output position: the reduced sample an exact hit lands on,
// the two reduced samples to blend otherwise, and the blend ratio (indices already offset by
// n / 2 and clamped into the reduced frame).
struct RectAxis {
  std::vector<int> hit, lo, hi;
  std::vector<float> ratio;    // weight of `hi`
  std::vector<uint8_t> exact;  // the forward map of `hit` is this very position
};
inline RectAxis rect_inverse_axis(int n_full, int n_reduced, float center) {
  RectAxis a;
  a.hit.resize((size_t)n_full);
  a.lo.resize((size_t)n_full);
  a.hi.resize((size_t)n_full);
  a.ratio.resize((size_t)n_full);
  a.exact.resize((size_t)n_full);
  const float lambda = rect_lambda(n_full);
  const int c = (int)(center * n_full);
  auto idx = [&](int u) { return std::min(std::max(u + n_reduced / 2, 0), n_reduced - 1); };
  for (int p = 0; p < n_full; ++p) {
    const int d = p - c;
    // log on float (abs(d) / lambda + 1 is a float), pow and the product in double
    int u = (int)std::ceil(0.5 * n_reduced * std::pow(std::log(std::abs(d) / lambda + 1), 0.25)) *
            sgn(d);
    if (std::abs(u) > std::abs(d) || u == 0) u = d;
    const int calc = rect_forward(u, lambda, n_reduced);
    a.hit[(size_t)p] = idx(u);
    a.exact[(size_t)p] = calc == d;
    // the neighbour towards the centre; its forward position keeps the sign of u
    const int un = u + ((p < c) - (p > c));
    const int dmin =
        std::max(std::abs(un),
                 (int)(lambda * (std::exp(std::pow(2.0 * std::abs(un) / n_reduced, 4.0)) - 1))) *
        sgn(u);
    const int pmin = std::min(c + dmin, c + calc), pmax = std::max(c + dmin, c + calc);
    int umin = std::min(u, un), umax = std::max(u, un);
    if (pmin < 0) umin = umax;
    if (pmax >= n_full) umax = umin;
    a.lo[(size_t)p] = idx(umin);
    a.hi[(size_t)p] = idx(umax);
    a.ratio[(size_t)p] =
        pmax == pmin ? 0.0f : clampf((float)(p - pmin) / (pmax - pmin), 0.0f, 1.0f);
  }
  return a;
}
// A pixel is copied when BOTH axes hit a sample; otherwise it blends on both axes (on an axis
// that hits, the ratio is 0 or 1 on that very sample).
inline void interpolate_rect(uint8_t *target, int tw, int th, int t_linesize,
                             const uint8_t *source, int sw, int sh, int s_linesize,
                             float center_x, float center_y) {
  const int tbpp = t_linesize / tw, sbpp = s_linesize / sw;
  const RectAxis ax = rect_inverse_axis(tw, sw, center_x), ay = rect_inverse_axis(th, sh, center_y);
  for (int y = 0; y < th; ++y) {
    uint8_t *trow = target + (size_t)y * t_linesize;
    const uint8_t *top = source + (size_t)ay.lo[(size_t)y] * s_linesize;
    const uint8_t *bot = source + (size_t)ay.hi[(size_t)y] * s_linesize;
    const float yr = ay.ratio[(size_t)y];
    for (int x = 0; x < tw; ++x) {
      uint8_t *t = trow + (size_t)x * tbpp;
      if (ax.exact[(size_t)x] && ay.exact[(size_t)y]) {
        const uint8_t *s =
            source + (size_t)ay.hit[(size_t)y] * s_linesize + (size_t)ax.hit[(size_t)x] * sbpp;
        t[0] = s[0];
        t[1] = s[1];
        t[2] = s[2];
        continue;
      }
      const size_t l = (size_t)ax.lo[(size_t)x] * sbpp, r = (size_t)ax.hi[(size_t)x] * sbpp;
      const float xr = ax.ratio[(size_t)x];
      for (int c = 0; c < 3; ++c) {
        const float left = lerpf((float)top[l + c], (float)bot[l + c], yr);
        const float right = lerpf((float)top[r + c], (float)bot[r + c], yr);
        t[c] = (uint8_t)lerpf(left, right, xr);
      }
    }
  }
}

// ---------------------------------------------------------- ImageSampler::SampleFrameRectCPU
// Point sample of a 4-elements-per-pixel uint32 buffer ("Untested" in the reference,
// src/image_sampler.cc:301); positions outside the source are clamped into it.
inline void sample_rect_point(uint8_t *target, int tw, int th, int t_linesize,
                              const uint32_t *buffer, int sw, int sh, float center_x,
                              float center_y) {
  const int tbpp = t_linesize / tw;
  const float lx = rect_lambda(sw), ly = rect_lambda(sh);
  std::vector<int> xs((size_t)tw), ys((size_t)th);
  for (int i = 0; i < tw; ++i)
    xs[(size_t)i] = std::min(std::max((int)(center_x * sw + rect_forward(i - tw / 2, lx, tw)), 0), sw - 1);
  for (int j = 0; j < th; ++j)
    ys[(size_t)j] = std::min(std::max((int)(center_y * sh + rect_forward(j - th / 2, ly, th)), 0), sh - 1);
  for (int j = 0; j < th; ++j)
    for (int i = 0; i < tw; ++i) {
      const uint32_t *s = buffer + ((size_t)ys[(size_t)j] * sw + xs[(size_t)i]) * 4;
      uint8_t *t = target + (size_t)j * t_linesize + (size_t)i * tbpp;
      t[0] = (uint8_t)s[0];
      t[1] = (uint8_t)s[1];
      t[2] = (uint8_t)s[2];
    }
}

// ------------------------------------------------- ImageSampler::ExpandSampledFrameLogPolarCPU
// source column i is a radius exp(10 * i / sw) (float), row j an angle 2 pi j / sh (double cos /
// sin of a float-times-double argument); the product is rounded to float before the sum.
inline void expand_logpolar(uint8_t *target, int tw, int th, int t_linesize, const uint8_t *source,
                            int sw, int sh, int s_linesize, float center_x, float center_y) {
  const int tbpp = t_linesize / tw, sbpp = s_linesize / sw;
  const float alpha = 1.0f;
  std::vector<float> radius((size_t)sw);
  std::vector<double> cs((size_t)sh), sn((size_t)sh);
  for (int i = 0; i < sw; ++i)
    radius[(size_t)i] = std::exp(10.0f * std::pow((float)i / sw, alpha));
  for (int j = 0; j < sh; ++j) {
    cs[(size_t)j] = std::cos((float)j / sh * 2 * M_PI);
    sn[(size_t)j] = std::sin((float)j / sh * 2 * M_PI);
  }
  // the reference's loop is column-major (i outer): where several source pixels land on one
  // target pixel, the one with the largest i, then the largest j, is written last.  The map is
  // not separable, so the order is reproduced literally.
  for (int i = 0; i < sw; ++i)
    for (int j = 0; j < sh; ++j) {
      const float dx = (float)(radius[(size_t)i] * cs[(size_t)j]);
      const float dy = (float)(radius[(size_t)i] * sn[(size_t)j]);
      const int x = (int)(center_x * tw + dx), y = (int)(center_y * th + dy);
      if (x < 0 || x >= tw || y < 0 || y >= th) continue;
      const uint8_t *s = source + (size_t)j * s_linesize + (size_t)i * sbpp;
      uint8_t *t = target + (size_t)y * t_linesize + (size_t)x * tbpp;
      t[0] = s[0];
      t[1] = s[1];
      t[2] = s[2];
    }
}

// --------------------------------------------------- ImageSampler::InterpolateFrameLogPolarCPU
inline void interpolate_logpolar(uint8_t *target, int tw, int th, int t_linesize,
                                 const uint8_t *source, int sw, int sh, int s_linesize,
                                 float center_x, float center_y) {
  const int tbpp = t_linesize / tw, sbpp = s_linesize / sw;
  const float alpha = 1.0f;
  const int cx = (int)(center_x * tw), cy = (int)(center_y * th);
  std::vector<float> radius((size_t)sw);
  std::vector<double> cs((size_t)sh), sn((size_t)sh);
  for (int i = 0; i < sw; ++i)
    radius[(size_t)i] = std::exp(10.0f * std::pow((float)i / sw, alpha));
  for (int j = 0; j < sh; ++j) {
    cs[(size_t)j] = std::cos((float)j / sh * 2.0f * M_PI);
    sn[(size_t)j] = std::sin((float)j / sh * 2.0f * M_PI);
  }
  for (int y = 0; y < th; ++y) {
    const int dy = y - cy;
    for (int x = 0; x < tw; ++x) {
      const int dx = x - cx;
      // pow(int, float) promotes to double; the chain stays double until it lands in a float
      const float i_f =
          dx == 0 && dy == 0
              ? 0.0f
              : (float)(sw * std::pow(std::log(std::sqrt(std::pow((double)dx, 2.0) +
                                                         std::pow((double)dy, 2.0))) /
                                          10.0f,
                                      (double)(1.0f / alpha)));
      const int i = (int)clampf(std::round(i_f), 0.0f, (float)(sw - 1));
      float j_f;
      if (dx != 0) {
        j_f = (float)((std::atan((float)dy / dx) + M_PI * (dx < 0)) * ((float)sh / (2.0 * M_PI)));
        j_f = (float)std::fmod((double)(j_f + 2 * sh), (double)sh);
      } else {
        j_f = (float)((M_PI_2 + M_PI * (dy < 0)) * (sh / (2.0 * M_PI)));
      }
      const int j = (int)clampf(std::round(j_f), 0.0f, (float)(sh - 1));
      uint8_t *t = target + (size_t)y * t_linesize + (size_t)x * tbpp;
      const int back_x = (int)(center_x * tw + radius[(size_t)i] * cs[(size_t)j]);
      const int back_y = (int)(center_y * th + radius[(size_t)i] * sn[(size_t)j]);
      if (back_x == x && back_y == y) {
        const uint8_t *s = source + (size_t)j * s_linesize + (size_t)i * sbpp;
        t[0] = s[0];
        t[1] = s[1];
        t[2] = s[2];
        continue;
      }
      const int i0 = (int)clampf(std::floor(i_f), 0.0f, (float)(sw - 1));
      const int i1 = (int)clampf(std::ceil(i_f), 0.0f, (float)(sw - 1));
      const int j0 = (((int)std::floor(j_f + sh) % sh) + sh) % sh;
      const int j1 = (((int)std::ceil(j_f + sh) % sh) + sh) % sh;
      const float ir = i_f - std::floor(i_f), jr = j_f - std::floor(j_f);
      const uint8_t *top = source + (size_t)j0 * s_linesize, *bot = source + (size_t)j1 * s_linesize;
      for (int c = 0; c < 3; ++c) {
        const float left = lerpf((float)top[(size_t)i0 * sbpp + c], (float)bot[(size_t)i0 * sbpp + c], jr);
        const float right = lerpf((float)top[(size_t)i1 * sbpp + c], (float)bot[(size_t)i1 * sbpp + c], jr);
        t[c] = (uint8_t)lerpf(left, right, ir);
      }
    }
  }
}

}  // namespace f360cpu
