// Test harness: the CPU twins THROUGH the drop-in classes (include/f360/*.h), as extern "C"
// entry points for ctypes.  Frame / Codec stand in for AVFrame / AVCodecContext.
#include <cstdint>

#include "f360/image_sampler.h"
#include "f360/sat_decoder.h"
#include "f360/sat_encoder.h"

struct Frame {
  uint8_t *data[8];
  int linesize[8];
  int width, height;
};
struct Codec {
  int width, height;
};
static Frame frame_of(uint8_t *p, int w, int h, int linesize) {
  Frame f = {};
  f.data[0] = p;
  f.linesize[0] = linesize;
  f.width = w;
  f.height = h;
  return f;
}

extern "C" {
void t_encode(uint32_t *table, int w, int h, uint8_t *src, int linesize) {
  Codec c = {w, h};
  Frame f = frame_of(src, w, h, linesize);
  SATEncoder enc;  // default-constructed: the CPU-only object of the reference
  enc.EncodeFrameCPU(table, &c, &f);
}
void t_decode(uint8_t *out, int out_linesize, uint32_t *table, int w, int h) {
  Codec c = {w, h};
  Frame f = frame_of(out, w, h, out_linesize);
  SATDecoder dec;
  dec.DecodeFrameCPU(&f, table, &c);
}
#define TWO_FRAMES                                                                  \
  Frame t = frame_of(target, tw, th, tls), s = frame_of(source, sw, sh, sls)
void t_sd_expand_rect(uint8_t *target, int tw, int th, int tls, uint8_t *source, int sw, int sh,
                      int sls, float cx, float cy) {
  TWO_FRAMES;
  SATDecoder dec;
  dec.ExpandSampledFrameRectCPU(&t, &s, cx, cy);
}
void t_sd_interpolate_rect(uint8_t *target, int tw, int th, int tls, uint8_t *source, int sw,
                           int sh, int sls, float cx, float cy) {
  TWO_FRAMES;
  SATDecoder dec;
  dec.InterpolateFrameRectCPU(&t, &s, cx, cy);
}
void t_is_expand_rect(uint8_t *target, int tw, int th, int tls, uint8_t *source, int sw, int sh,
                      int sls, float cx, float cy) {
  TWO_FRAMES;
  ImageSampler smp;
  smp.ExpandSampledFrameRectCPU(&t, &s, cx, cy);
}
void t_is_interpolate_rect(uint8_t *target, int tw, int th, int tls, uint8_t *source, int sw,
                           int sh, int sls, float cx, float cy) {
  TWO_FRAMES;
  ImageSampler smp;
  smp.InterpolateFrameRectCPU(&t, &s, cx, cy);
}
void t_is_expand_logpolar(uint8_t *target, int tw, int th, int tls, uint8_t *source, int sw,
                          int sh, int sls, float cx, float cy) {
  TWO_FRAMES;
  ImageSampler smp;
  smp.ExpandSampledFrameLogPolarCPU(&t, &s, cx, cy);
}
void t_is_interpolate_logpolar(uint8_t *target, int tw, int th, int tls, uint8_t *source, int sw,
                               int sh, int sls, float cx, float cy) {
  TWO_FRAMES;
  ImageSampler smp;
  smp.InterpolateFrameLogPolarCPU(&t, &s, cx, cy);
}
void t_is_sample_rect(uint8_t *target, int tw, int th, int tls, uint32_t *buffer, int sw, int sh,
                      float cx, float cy) {
  Codec c = {sw, sh};
  Frame t = frame_of(target, tw, th, tls);
  ImageSampler smp;
  smp.SampleFrameRectCPU(&t, buffer, &c, cx, cy);
}
}
