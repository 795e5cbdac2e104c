/*
 * f360_oracle.h -- CPU oracle for the foveated-360 frame-transform hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
 * for this path, its kernels are OpenCL C that cannot execute in this
 * container (0 OpenCL devices) and its host sources need a configure-
 * generated FFmpeg header (libavutil/avconfig.h) that is not vendored, so the
 * reference is unbuildable here.  This file is a plain-C restatement of the
 * kernel text with OpenCL C typing rules; the golden vectors under
 * tests/golden/ were produced by this oracle (tests/golden/make_golden.py).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout).
 *
 * Float model.  OpenCL float builtins (exp/pow/log/cos/sin/atan/asin/atan2) are
 * only specified to a few ulp, so the NVIDIA results the reference saw are not
 * reproducible anywhere else.  The oracle models every float builtin as
 * correctly rounded: evaluate in double, round once to float
 * (F360O_FLOAT_CR).  F360O_FLOAT_LIBM switches to glibc's float routines to
 * measure how sensitive a table is to that choice (tests assert the geometry
 * tables of all benchmark configs are identical under both models).
 */
#ifndef F360_ORACLE_H
#define F360_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { F360O_FLOAT_CR = 0, F360O_FLOAT_LIBM = 1 };
void f360o_set_float_model(int model);
int f360o_get_float_model(void);

/* --- synthetic inputs ------------------------------------- */
/* LCG s = s*1664525 + 1013904223 (mod 2^32), byte = s >> 24, every byte in
 * memory order. */
void f360o_lcg_fill(uint8_t *buf, size_t n, uint32_t seed);
/* 64-bit FNV-1a digest of a byte range. */
uint64_t f360o_fnv1a64(const void *buf, size_t n);

/* --- SATEncoder ----------------------------------------------------------- */
/* src/sat_encoder_encode_kernels.cl:1-20,44-74 ; src/sat_encoder.cc:137-185 */
void f360o_sat_encode(uint32_t *sat, const uint8_t *src, int width, int height,
                      int linesize);

/* --- SATDecoder ----------------------------------------------------------- */
/* src/sat_decoder_sample_rect_kernel.cl:243-295 ; grid is
 * short[(Hr+1)][(Wr+1)][2]. */
void f360o_satdec_grid(int16_t *grid, int out_w, int out_h, int src_w,
                       int src_h);
/* 1-D factors of the same table (x depends only on the column, y on the row).
 * gx has out_w+1 entries, gy has out_h+1. */
void f360o_satdec_grid_axes(int16_t *gx, int16_t *gy, int out_w, int out_h,
                            int src_w, int src_h);
/* src/sat_decoder_sample_rect_kernel.cl:138-241 */
void f360o_satdec_sample_rect(uint8_t *dst, int out_w, int out_h,
                              int out_linesize, const uint32_t *sat, int src_w,
                              int src_h, const int16_t *grid, float cx,
                              float cy);
/* src/sat_decoder_interpolate_kernel.cl:1-152 ; texels are 4-byte uchar3, rows
 * tightly packed (the kernel receives no linesize). Pad byte written as 0. */
void f360o_satdec_interpolate_rect(uint8_t *dst, int out_w, int out_h,
                                   const uint8_t *src, int src_w, int src_h,
                                   float cx, float cy);
/* src/sat_decoder_decode_kernel.cl:1-58 */
void f360o_satdec_decode(uint8_t *dst, int dst_linesize, const uint32_t *sat,
                         int width, int height);

/* --- ImageSampler --------------------------------------------------------- */
/* src/image_sampler_sample_rect_kernel.cl:48-88 ; short[Hr][Wr][2] */
void f360o_is_grid(int16_t *grid, int out_w, int out_h, int src_w, int src_h);
/* src/image_sampler_sample_rect_kernel.cl:1-46 */
void f360o_is_sample_rect(uint8_t *dst, int out_w, int out_h, int out_linesize,
                          const uint8_t *src, int src_w, int src_h,
                          int src_linesize, const int16_t *grid, float cx,
                          float cy);
/* src/image_sampler_sample_logpolar_kernel.cl:5-39 ; short[Hr][Wr][2] */
void f360o_is_logpolar_grid(int16_t *grid, int out_w, int out_h, int src_w,
                            int src_h);
/* src/image_sampler_sample_logpolar_kernel.cl:41-86 */
void f360o_is_sample_logpolar(uint8_t *dst, int out_w, int out_h,
                              int out_linesize, const uint8_t *src, int src_w,
                              int src_h, int src_linesize, const int16_t *grid,
                              float cx, float cy);
/* src/image_sampler_interpolate_kernel.cl:1-81 ; 4-byte texels, tight rows */
void f360o_is_interpolate_logpolar(uint8_t *dst, int out_w, int out_h,
                                   const uint8_t *src, int src_w, int src_h,
                                   float cx, float cy);
/* src/image_sampler_sample_logpolar_kernel.cl:88-142 ; 4-byte texels, tight */
void f360o_is_logpolar_blur(uint8_t *dst, int w, int h, const uint8_t *src);

/* --- Projections ---------------------------------------------------------- */
/* src/projections_program.cl:7-47 ; 4-byte texels, tight rows */
void f360o_gnomonic(uint8_t *dst, int dst_w, int dst_h, const uint8_t *src,
                    int src_w, int src_h, float cx, float cy);

/* --- colour-space step in front of the path (f360_oracle_yuv.c) ----------- */
/* What sws_scale does for src/video_decoder.cc:167-170,222-224 (yuv420p -> RGB0, same size),
 * restated from the vendored FFmpeg 4.2 libswscale: the table-driven C converter or the x86
 * MMX converter (they round differently). */
enum { F360O_YUV_SWS_C = 0, F360O_YUV_SWS_X86 = 1 };
void f360o_yuv_to_rgb_pixel(int model, int Y, int U, int V, uint8_t *rgb);
void f360o_yuv420p_to_rgb0(uint8_t *dst, int dst_linesize, const uint8_t *y_plane,
                           int y_linesize, const uint8_t *u_plane, int u_linesize,
                           const uint8_t *v_plane, int v_linesize, int width, int height,
                           int model);
/* --- colour-space step behind the path (f360_oracle_rgb2yuv.c) ----------- */
/* What sws_scale does for src/video_encoder.cc:380-395 (RGB0 -> yuv420p, same size): the
 * generic scaler of FFmpeg 4.2, C functions (model F360O_YUV_SWS_C) or with the x86 vertical
 * chroma scaler (F360O_YUV_SWS_X86).  Even width and height. */
void f360o_rgb0_to_yuv420p(uint8_t *y_plane, int y_linesize, uint8_t *u_plane, int u_linesize,
                           uint8_t *v_plane, int v_linesize, const uint8_t *src, int src_linesize,
                           int width, int height, int model);
/* the vertical chroma filter initFilter builds for that call: returns the filter size (taps
 * per chroma row), fills filter_out[(height + 1) / 2][size] and pos_out[(height + 1) / 2] when
 * size <= max_size */
int f360o_rgb2yuv_chroma_vfilter(int16_t *filter_out, int32_t *pos_out, int height, int max_size);
/* RY GY BY RU GU BU RV GV BV (utils.c:811-821) */
void f360o_rgb2yuv_coeffs(int32_t *out9);

/* --- "expand" debug views (f360_oracle_expand.c) --------------------------- */
/* src/sat_decoder.cc:555-616 (== src/image_sampler.cc:358-419): reduced frame scattered back to
 * where the log-rectilinear sampler took it from; untouched target pixels keep their value */
void f360o_expand_rect(uint8_t *dst, int dst_w, int dst_h, int dst_linesize, const uint8_t *src,
                       int src_w, int src_h, int src_linesize, float cx, float cy);
/* src/image_sampler.cc:623-666: the log-polar counterpart (last writer in i-outer, j-inner order
 * wins where several source pixels land on one target pixel) */
void f360o_expand_logpolar(uint8_t *dst, int dst_w, int dst_h, int dst_linesize,
                           const uint8_t *src, int src_w, int src_h, int src_linesize, float cx,
                           float cy);

/* --- whole hot path, used by bench.py's cpu_baseline leg ------------------ */
/* SAT encode + log-rectilinear SAT sample of `frames` frames; frame k is an
 * LCG fill with seed seed0+k, gaze is the Lissajous of SURVEY.md 8d(2).
 * Returns a digest over all reduced frames (keeps the work observable). */
uint64_t f360o_pipeline_encode_sample(int frames, int src_w, int src_h,
                                      int out_w, int out_h, uint32_t seed0,
                                      double *seconds_out);
/* the same over frames synthesised by the caller: compute only (bench.py's cpu_baseline) */
uint64_t f360o_pipeline_compute(const uint8_t *frames, int nframes, int first_index, int src_w,
                                int src_h, int out_w, int out_h, double *seconds_out);

#ifdef __cplusplus
}
#endif
#endif
