/*
 * f360_oracle_expand.c -- CPU oracle for the "expand" debug views:
 * every pixel of a reduced frame scattered to the place it was sampled from.
 * TEST INFRASTRUCTURE ONLY, see f360_oracle.h.  PARITY UNPINNED (the reference is unbuildable
 * here); restates the C++ text with its promotion rules:
 *   `using namespace std;` + float arguments select the float overloads of exp / pow,
 *   an unsuffixed 2.0 / 4.0 makes an expression double, `int x = float_expr` truncates.
 */
#include "f360_oracle.h"

#include <math.h>
#include <stdlib.h>

static int sgn(int v) { return (v > 0) - (v < 0); }

/* max((int)abs(u), (int)(lambda * (exp(pow(2.0 * abs(u) / n, 4.0)) - 1))) * sgn(u)
 * src/sat_decoder.cc:583-587 (== src/image_sampler.cc:386-390); lambda is float, the rest double */
static int forward_offset(int u, int n, float lambda) {
  const int a = abs(u);
  const int b = (int)(lambda * (exp(pow(2.0 * a / n, 4.0)) - 1));
  return (a > b ? a : b) * sgn(u);
}

/* SATDecoder::ExpandSampledFrameRectCPU, src/sat_decoder.cc:555-616, and its copy
 * ImageSampler::ExpandSampledFrameRectCPU, src/image_sampler.cc:358-419 */
void f360o_expand_rect(uint8_t *dst, int dst_w, int dst_h, int dst_linesize, const uint8_t *src,
                       int src_w, int src_h, int src_linesize, float cx, float cy) {
  const int sbpp = src_linesize / src_w, dbpp = dst_linesize / dst_w;
  const float lambda_x = dst_w / (expf(1.0f) - 1);
  const float lambda_y = dst_h / (expf(1.0f) - 1);
  for (int i = 0; i < src_w; ++i) {
    for (int j = 0; j < src_h; ++j) {
      const int delta_x = forward_offset(i - src_w / 2, src_w, lambda_x);
      const int delta_y = forward_offset(j - src_h / 2, src_h, lambda_y);
      const int x_pos = (int)(cx * dst_w + delta_x);
      const int y_pos = (int)(cy * dst_h + delta_y);
      if (x_pos >= 0 && x_pos < dst_w && y_pos >= 0 && y_pos < dst_h) {
        const size_t t = (size_t)y_pos * dst_linesize + (size_t)x_pos * dbpp;
        const size_t s = (size_t)j * src_linesize + (size_t)i * sbpp;
        dst[t] = src[s];
        dst[t + 1] = src[s + 1];
        dst[t + 2] = src[s + 2];
      }
    }
  }
}

/* ImageSampler::ExpandSampledFrameLogPolarCPU, src/image_sampler.cc:623-666.  Several source
 * pixels land on one target pixel near the centre; the loops run i outer, j inner, so the last
 * writer -- the largest (i, j) in that order -- wins. */
void f360o_expand_logpolar(uint8_t *dst, int dst_w, int dst_h, int dst_linesize,
                           const uint8_t *src, int src_w, int src_h, int src_linesize, float cx,
                           float cy) {
  const float alpha = 1.0f;
  const int sbpp = src_linesize / src_w, dbpp = dst_linesize / dst_w;
  for (int i = 0; i < src_w; ++i) {
    for (int j = 0; j < src_h; ++j) {
      const float radius = expf(10.0f * powf((float)i / src_w, alpha));
      const float delta_x = (float)(radius * cos((float)j / src_h * 2 * M_PI));
      const float delta_y = (float)(radius * sin((float)j / src_h * 2 * M_PI));
      const int x_pos = (int)(cx * dst_w + delta_x);
      const int y_pos = (int)(cy * dst_h + delta_y);
      if (x_pos >= 0 && x_pos < dst_w && y_pos >= 0 && y_pos < dst_h) {
        const size_t t = (size_t)y_pos * dst_linesize + (size_t)x_pos * dbpp;
        const size_t s = (size_t)j * src_linesize + (size_t)i * sbpp;
        dst[t] = src[s];
        dst[t + 1] = src[s + 1];
        dst[t + 2] = src[s + 2];
      }
    }
  }
}
