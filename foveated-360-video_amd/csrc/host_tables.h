// host_tables.h -- geometry tables built once per frame geometry on the host.
//
// Every transcendental expression of the reference kernels that depends on one
// image axis only is hoisted into a 1-D table here, so that the per-pixel HIP
// kernels contain no libm calls and stay HBM-bound.  OpenCL float builtins are
// evaluated as correctly rounded floats (double evaluation, one rounding); see
// DESIGN.md "Float model".
#pragma once

#include <cstdint>
#include <vector>

#include "yuv_device.h"

namespace f360 {

// Constants of libswscale's yuv420p -> RGB converters for the context the reference creates
// (src/video_decoder.cc:167-170): ITU-R 601 matrix, limited-range source, brightness 0,
// contrast = saturation = 1 << 16.  Follows ff_yuv2rgb_c_init_tables
// (include/FFmpeg42/libswscale/yuv2rgb.c:774-855, 32-bit case :968-993).
void build_yuv2rgb_consts(YuvConsts &k);

// SATDecoder create_grid_kernel (src/sat_decoder_sample_rect_kernel.cl:243-295),
// one axis: n_out+1 midpoint offsets.
void build_satdec_grid_axis(std::vector<int16_t> &g, int n_out, int n_src);

// ImageSampler create_grid_kernel (src/image_sampler_sample_rect_kernel.cl:48-88),
// one axis: n_out offsets.
void build_is_grid_axis(std::vector<int16_t> &g, int n_out, int n_src);

// ImageSampler create_logpolar_grid_kernel
// (src/image_sampler_sample_logpolar_kernel.cl:5-39): radius per column and
// cos/sin per row, all float.
void build_logpolar_axes(std::vector<float> &radius, std::vector<float> &cs,
                         std::vector<float> &sn, int out_w, int out_h);

// interpolate_logpolar_kernel (src/image_sampler_interpolate_kernel.cl:46-51):
// float radius per reduced column, double cos/sin per reduced row.
void build_logpolar_inverse_axes(std::vector<float> &radius,
                                 std::vector<double> &cs,
                                 std::vector<double> &sn, int src_w, int src_h);

// interpolate_rect_kernel (src/sat_decoder_interpolate_kernel.cl:43-89), one
// axis.  Entry for pixel offset d (index d + range) holds
//   .u     reduced-buffer coordinate after the :50-55 fallback
//   .dcalc forward map of u in double math (:56-65)
//   .dmin  forward map of the neighbour u + du in float math (:77-89)
//   .du    step towards the gaze centre used for the neighbour (:75-76)
struct InterpAxisEntry {
  int32_t u, dcalc, dmin, du;
};
void build_interp_axis(std::vector<InterpAxisEntry> &t, int range, int n_full,
                       int n_reduced);

// ExpandSampledFrameRectCPU (src/sat_decoder.cc:575-598), one axis: the offset from the gaze
// centre at which reduced pixel i was sampled, max(|u|, (int)(lambda (exp(pow(2.0|u|/n, 4.0)) - 1)))
// sgn(u) with u = i - n_reduced / 2, lambda = n_full / (expf(1) - 1) in float, the rest double.
void build_expand_axis(std::vector<int32_t> &d, int n_reduced, int n_full);
// ExpandSampledFrameLogPolarCPU (src/image_sampler.cc:646-651): radius per reduced column
// (float: expf(10 * powf(i / w, 1))), cos / sin per reduced row (double, of a float angle * 2 * pi)
void build_expand_logpolar_axes(std::vector<float> &radius, std::vector<double> &cs,
                                std::vector<double> &sn, int src_w, int src_h);

}  // namespace f360
