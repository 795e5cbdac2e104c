/*
 * f360.h -- C ABI of the MI355X-native foveated-360 frame-transform engine.
 *
 * This is the drop-in boundary.  It replaces the reference's OpenCLManager +
 * cl_kernel plumbing (src/opencl_manager.{h,cc}) underneath the reference's
 * SATEncoder / SATDecoder / ImageSampler / Projections classes; the C++
 * mirrors of those classes live in the include/f360/ directory and are thin inline
 * wrappers over the functions below.  Plain pointers and sizes only: a
 * "device pointer" is whatever hipMalloc / f360_malloc returned (or
 * torch.Tensor.data_ptr()).
 *
 * Execution model (mirrors the reference): every transform only
 * ENQUEUES work on the context's single in-order HIP stream and returns;
 * ordering between encode -> sample -> interpolate relies on that stream; the
 * caller forces completion with f360_sync() or a blocking f360_memcpy_*.
 * Objects are not thread-safe; use one context (+ decoder/sampler) per thread,
 * exactly like one OpenCLManager per connection thread in the reference.
 *
 * All functions return F360_OK (0) or a negative f360_status; the message of
 * the last failure on the calling thread is available from
 * f360_last_error_string().  There is NO CPU fallback: without a HIP device
 * every entry point that touches the device fails with F360_ERR_NO_DEVICE.
 */
#ifndef F360_H
#define F360_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define F360_VERSION_MAJOR 0
#define F360_VERSION_MINOR 1

typedef enum f360_status {
  F360_OK = 0,
  F360_ERR_INVALID_ARG = -1,
  F360_ERR_NO_DEVICE = -2,
  F360_ERR_HIP = -3,
  F360_ERR_OOM = -4,
  F360_ERR_NOT_INITIALIZED = -5
} f360_status;

typedef struct f360_ctx f360_ctx;                     /* replaces OpenCLManager */
typedef struct f360_sat_decoder f360_sat_decoder;     /* state of SATDecoder    */
typedef struct f360_image_sampler f360_image_sampler; /* state of ImageSampler  */

/* ---- runtime: replaces OpenCLManager (src/opencl_manager.h:8-22,
 *      src/opencl_manager.cc:7-67: platform/device/context/in-order queue) -- */
int f360_version(void);
const char *f360_last_error_string(void);
const char *f360_status_string(int status); /* OpenCLManager::GetCLErrorString */
int f360_device_count(int *count);
/* OpenCLManager::InitializeContext(): device `device`, one new in-order
 * stream owned by the context. */
int f360_ctx_create(int device, f360_ctx **out);
/* Same, but enqueue on a caller-owned hipStream_t (e.g. torch's current
 * stream); the stream is borrowed, not destroyed. */
int f360_ctx_create_on_stream(int device, void *hip_stream, f360_ctx **out);
int f360_ctx_destroy(f360_ctx *ctx);
int f360_ctx_device(const f360_ctx *ctx, int *device);
int f360_ctx_stream(const f360_ctx *ctx, void **hip_stream);
/* clFlush + clFinish (src/video_server.cc:302-303) */
int f360_sync(f360_ctx *ctx);

/* ---- buffers: replaces cl::Buffer / cl::copy
 *      (src/video_server.cc:224-232,298-299,342-345) -------------------------- */
int f360_malloc(f360_ctx *ctx, size_t bytes, void **dptr);
int f360_free(f360_ctx *ctx, void *dptr);
int f360_memset(f360_ctx *ctx, void *dptr, int value, size_t bytes); /* async */
/* blocking copies (cl::copy semantics: returns when the copy is done) */
int f360_memcpy_h2d(f360_ctx *ctx, void *dst_dev, const void *src_host,
                    size_t bytes);
int f360_memcpy_d2h(f360_ctx *ctx, void *dst_host, const void *src_dev,
                    size_t bytes);
/* stream-ordered copies (host memory should be pinned for true asynchrony) */
int f360_memcpy_h2d_async(f360_ctx *ctx, void *dst_dev, const void *src_host,
                          size_t bytes);
int f360_memcpy_d2h_async(f360_ctx *ctx, void *dst_host, const void *src_dev,
                          size_t bytes);
int f360_host_alloc_pinned(size_t bytes, void **hptr);
int f360_host_free_pinned(void *hptr);

/* ---- timing helpers (HIP events on the context's stream; used by bench.py
 *      to time kernels on the stream they are launched on) ------------------ */
typedef struct f360_event f360_event;
int f360_event_create(f360_ctx *ctx, f360_event **ev);
int f360_event_destroy(f360_event *ev);
int f360_event_record(f360_ctx *ctx, f360_event *ev);
int f360_event_elapsed_ms(f360_event *start, f360_event *stop, float *ms);

/* ---- SATEncoder --------------------------------------------------------- */
/* SATEncoder::EncodeFrameGPU (src/sat_encoder.h:39-40, src/sat_encoder.cc
 * :67-135; kernels src/sat_encoder_encode_kernels.cl:1-20,44-74).
 *   sat_dev : uint32[height][width][3] (row stride 3*width elements)
 *   src_dev : packed 8-bit frame, `linesize` bytes per row,
 *             bytes-per-pixel = linesize / width (4 for RGB0, 3 for RGB24)
 * Any width/height >= 1 is accepted (the reference needs multiples of 8). */
int f360_sat_encode(f360_ctx *ctx, uint32_t *sat_dev, const uint8_t *src_dev,
                    int width, int height, int linesize);
/* Optional: allocate the encoder's scratch for a geometry ahead of time so
 * that f360_sat_encode itself never allocates (graph-capture safe). */
int f360_sat_encode_prepare(f360_ctx *ctx, int width, int height);

/* ---- colour-space step in front of the path ----------- */
/* Replaces the CPU sws_scale of VideoDecoder::GetFrame (src/video_decoder.cc:167-170,
 * 222-224: sws_getContext(w, h, yuv420p, w, h, AV_PIX_FMT_RGB0, SWS_BILINEAR) + sws_scale):
 * planar 8-bit YUV 4:2:0 (ITU-R 601, limited range) -> RGB0 with 255 in the fourth byte.
 * FFmpeg 4.2's libswscale serves that call with its unscaled yuv420p->RGB converter; the
 * context option "yuv.model" selects which of its two implementations is reproduced bit
 * for bit: 1 (default) the x86 MMX converter the reference's x86-64 builds run
 * (include/FFmpeg42/libswscale/x86/yuv2rgb_template.c:84-122,424-443), 0 the table-driven C
 * converter (libswscale/yuv2rgb.c:70-81,241-262,774-993).  Height must be even (an odd height
 * takes a different libswscale path); any width. */
int f360_yuv420p_to_rgb0(f360_ctx *ctx, uint8_t *dst_dev, int dst_linesize,
                         const uint8_t *y_dev, const uint8_t *u_dev, const uint8_t *v_dev,
                         int y_linesize, int u_linesize, int v_linesize, int width,
                         int height);
/* The colour-space step BEHIND the path: replaces the CPU sws_scale of VideoEncoder::EncodeFrame
 * in front of NVENC (src/video_encoder.cc:380-395: sws_getContext(w, h, RGB0, w, h,
 * AV_PIX_FMT_YUV420P, SWS_BILINEAR) + sws_scale): RGB0 -> planar 8-bit YUV 4:2:0, ITU-R 601
 * limited range.  FFmpeg 4.2 runs its generic scaler for this pair (rgb32ToY_c /
 * rgb32ToUV_half_c input conversion, include/FFmpeg42/libswscale/input.c:252-346; one-tap
 * horizontal filters; chroma rows halved by the 4-tap bilinear filter of utils.c:331-726);
 * "yuv.model" 1 (default) reproduces x86 builds, whose vertical chroma scaler truncates every
 * tap's product (x86/swscale.c:201-275), 0 the C functions (output.c:380-403).  Even width,
 * even height >= 8.  The fourth byte of the source pixels is ignored. */
int f360_rgb0_to_yuv420p(f360_ctx *ctx, uint8_t *y_dev, uint8_t *u_dev, uint8_t *v_dev,
                         int y_linesize, int u_linesize, int v_linesize,
                         const uint8_t *src_dev, int src_linesize, int width, int height);
/* f360_yuv420p_to_rgb0 + f360_sat_encode in one pass: the table of the RGB0 frame the call
 * above would write, computed from the planes without materialising that frame (the encoder
 * reads 1.5 instead of 4 bytes per pixel, twice).  Not in the reference.  Needs
 * width % 4 == 0, an even height, y_linesize % 4 == 0, even chroma linesizes, planes aligned
 * to 4 / 2 / 2 bytes and the table to 16. */
int f360_sat_encode_yuv420p(f360_ctx *ctx, uint32_t *sat_dev, const uint8_t *y_dev,
                            const uint8_t *u_dev, const uint8_t *v_dev, int y_linesize,
                            int u_linesize, int v_linesize, int width, int height);
/* SATEncoder::EncodeFrameGPU for `count` frames of one geometry at once: table sat_dev[k] of
 * source src_dev[k] (HOST arrays of device pointers), the same bytes as `count` single calls.
 * The frames share each of the encoder's three launches (a grid dimension of their own), which
 * is what small frames need: a 1080p frame alone cannot fill the device and its three launches
 * cost more than its traffic (16 per launch: 2.5 times the throughput).  The call splits into
 * launches of as many frames as stay cache-resident between the encoder's two reads of them
 * (about 180 MB of source, at most f360_sat_encode_batch_max() = 16: one 8K RGB0 frame or four
 * 8K frames from planes, six 3840x1920 frames, sixteen 1080p frames).  Not in the
 * reference, whose encoder handles one frame per call (src/sat_encoder.cc:67-135); a server
 * that holds several decoded frames, or several connections on one GPU, can use it.  The
 * encoder's scratch grows to `count` frames' worth (17 MB per 8K frame). */
int f360_sat_encode_batch(f360_ctx *ctx, int count, uint32_t *const *sat_dev,
                          const uint8_t *const *src_dev, int width, int height, int linesize);
int f360_sat_encode_batch_max(void);
/* Tables for f360_sat_encode_batch, placed for the read-once encoder.  A launch of that encoder
 * writes a group of tables (32 at 8K) at the same time, and the rate at which this device takes
 * those writes depends on what backs the tables -- 77-80 us per 8K frame or 85-94, by allocation,
 * not by address or allocation API (profiles/round4_table_placement.txt); user space can only
 * measure it.  This call allocates `count` tables of width x height x 12 bytes for calls of `count`
 * frames: it draws groups (each allocated while the earlier ones are still held, alternately one
 * allocation per table and one slab), times one encode launch of scratch frames into each, keeps the
 * fastest and gives the rest back (transiently up to six groups more than needed; a few ms).
 * Calls below the read-once encoder's threshold get one plain allocation per table.
 * tables_out[k], k < count (HOST array); release with f360_sat_tables_free.  Blocks.
 * The counterpart of the cl::Buffer allocations of src/video_server.cc:225-232 for callers
 * that keep many frames in flight. */
typedef struct f360_table_pool f360_table_pool;
int f360_sat_tables_alloc(f360_ctx *ctx, int width, int height, int count, uint32_t **tables_out,
                          f360_table_pool **pool_out);
int f360_sat_tables_free(f360_ctx *ctx, f360_table_pool *pool);
/* What was drawn and what was kept (a line for logs); valid until the pool is freed. */
const char *f360_sat_tables_report(const f360_table_pool *pool);
/* The same from planes: f360_sat_encode_yuv420p for `count` frames that share their three
 * linesizes (frames of one decoder do).  HOST arrays of device pointers. */
int f360_sat_encode_yuv420p_batch(f360_ctx *ctx, int count, uint32_t *const *sat_dev,
                                  const uint8_t *const *y_dev, const uint8_t *const *u_dev,
                                  const uint8_t *const *v_dev, int y_linesize, int u_linesize,
                                  int v_linesize, int width, int height);

/* ---- SATDecoder --------------------------------------------------------- */
int f360_satdec_create(f360_ctx *ctx, f360_sat_decoder **out);
int f360_satdec_destroy(f360_sat_decoder *dec);
/* SATDecoder::InitializeGrid (src/sat_decoder.h:48-49, src/sat_decoder.cc
 * :139-174; create_grid_kernel src/sat_decoder_sample_rect_kernel.cl:243-295).
 * No-op when the geometry is unchanged. */
int f360_satdec_initialize_grid(f360_sat_decoder *dec, int target_width,
                                int target_height, int source_width,
                                int source_height);
/* Export the grid in the reference's layout short[(Hr+1)][(Wr+1)][2] (host
 * buffer) -- for parity checks against create_grid_kernel. */
int f360_satdec_export_grid(f360_sat_decoder *dec, int16_t *grid_host);
/* SATDecoder::SampleFrameRectGPU (src/sat_decoder.h:63-66, src/sat_decoder.cc
 * :301-348; sample_rect_kernel src/sat_decoder_sample_rect_kernel.cl:138-241).
 * Writes bytes 0..2 of each 4-byte target pixel; byte 3 and pixels whose box
 * falls outside the frame are left untouched, as in the reference.
 * Initialises the grid on first use (src/sat_decoder.cc:312-317). */
int f360_satdec_sample_rect(f360_sat_decoder *dec, uint8_t *target_dev,
                            int target_width, int target_height,
                            int target_linesize, const uint32_t *sat_dev,
                            int source_width, int source_height, float center_x,
                            float center_y);
/* Several gaze points against one table in one launch (1 <= count <= 16):
 * target k is sampled at (centers_xy[2k], centers_xy[2k+1]).  Not in the
 * reference, which encodes once per connection (src/video_server.cc:62-66,300);
 * listed as the next step in SURVEY.md 8(f)-1: clients that watch the same
 * video share one encode.  `targets_dev` and `centers_xy` are HOST arrays. */
int f360_satdec_sample_rect_batch(f360_sat_decoder *dec,
                                  uint8_t *const *targets_dev, int count,
                                  int target_width, int target_height,
                                  int target_linesize, const uint32_t *sat_dev,
                                  int source_width, int source_height,
                                  const float *centers_xy);
/* The same launch shape for several FRAMES: target k is frame k's table sats_dev[k] sampled at
 * gaze k (any count; launches of 16).  The counterpart of f360_sat_encode_batch.  All three
 * arrays are HOST arrays. */
int f360_satdec_sample_rect_frames(f360_sat_decoder *dec, uint8_t *const *targets_dev,
                                   int count, int target_width, int target_height,
                                   int target_linesize, const uint32_t *const *sats_dev,
                                   int source_width, int source_height,
                                   const float *centers_xy);
/* SATEncoder::EncodeFrameGPU + SATDecoder::SampleFrameRectGPU for `count` frames whose gaze is
 * known before the encode: the reference's offline modes, which read it from a trace
 * (src/run_satlogrectilinear.cc:932-938).  The reference's SERVER is not such a caller: it
 * encodes, sleeps to the tick and only then reads the latest gaze
 * (src/video_server.cc:296-303,324-328,336), so using this call there changes its control flow
 * -- it keeps the two calls.  Table k of source k AND reduced frame k at
 * gaze k, byte for byte what f360_sat_encode_batch followed by f360_satdec_sample_rect_frames
 * write.  The reduced pixels are produced during the encoder's pass, from table rows still in
 * registers, and the tables are not read back: by the read-once encoder's strip owners with
 * enough frames to fill the device (f360_sat_encode_batch's rule: 23 8K frames), by the
 * three-kernel encoder's table writer for 4 .. 22 (RGB0 frames; "fuse.band"); a call of 1 .. 3
 * frames, or one whose sources the encoders' fast paths do not take, IS the two calls.  A call on
 * the three-kernel encoder runs its launch groups on two streams between its first and its last
 * kernel ("sat.pipeline"); from outside it is work enqueued on the context's one in-order stream.
 * All arrays are HOST arrays.  Not in the reference. */
int f360_satdec_encode_sample_frames(f360_sat_decoder *dec, uint8_t *const *targets_dev,
                                     uint32_t *const *sats_dev,
                                     const uint8_t *const *sources_dev, int count,
                                     int target_width, int target_height, int target_linesize,
                                     int source_width, int source_height, int source_linesize,
                                     const float *centers_xy);
/* The same from the decoder's planar YUV 4:2:0 frames (the planes of f360_sat_encode_yuv420p_batch:
 * one linesize per plane kind for all frames): what f360_sat_encode_yuv420p_batch followed by
 * f360_satdec_sample_rect_frames write. */
int f360_satdec_encode_sample_frames_yuv420p(
    f360_sat_decoder *dec, uint8_t *const *targets_dev, uint32_t *const *sats_dev,
    const uint8_t *const *y_dev, const uint8_t *const *u_dev, const uint8_t *const *v_dev,
    int y_linesize, int u_linesize, int v_linesize, int count, int target_width,
    int target_height, int target_linesize, int source_width, int source_height,
    const float *centers_xy);
/* The reduced frames alone: f360_satdec_encode_sample_frames without the tables (a server that
 * only sends the reduced frame never looks at the table again, src/video_server.cc:336-345) --
 * the batched form of f360_satdec_foveate_rect below, same bytes.  With enough RGB0 frames for the
 * read-once encoder its strip owners run without storing their rows; otherwise the single-frame
 * call, frame by frame. */
int f360_satdec_foveate_rect_frames(f360_sat_decoder *dec, uint8_t *const *targets_dev,
                                    const uint8_t *const *sources_dev, int count,
                                    int target_width, int target_height, int target_linesize,
                                    int source_width, int source_height, int source_linesize,
                                    const float *centers_xy);
/* ... and from planar YUV 4:2:0 frames (f360_satdec_foveate_rect_yuv420p for a batch). */
int f360_satdec_foveate_rect_frames_yuv420p(
    f360_sat_decoder *dec, uint8_t *const *targets_dev, const uint8_t *const *y_dev,
    const uint8_t *const *u_dev, const uint8_t *const *v_dev, int y_linesize, int u_linesize,
    int v_linesize, int count, int target_width, int target_height, int target_linesize,
    int source_width, int source_height, const float *centers_xy);
/* Fused SATEncoder::EncodeFrameGPU + SATDecoder::SampleFrameRectGPU for a gaze that
 * is known before the encode (the reference's offline modes read it from a trace,
 * src/run_satlogrectilinear.cc:926-938): frame -> reduced frame, the same bytes as
 * the two calls, without writing or re-reading the table.
 * Not in the reference.  Target semantics are f360_satdec_sample_rect's. */
int f360_satdec_foveate_rect(f360_sat_decoder *dec, uint8_t *target_dev,
                             int target_width, int target_height,
                             int target_linesize, const uint8_t *source_dev,
                             int source_width, int source_height,
                             int source_linesize, float center_x, float center_y);
/* The same from planar YUV 4:2:0 (f360_sat_encode_yuv420p's requirements). */
int f360_satdec_foveate_rect_yuv420p(f360_sat_decoder *dec, uint8_t *target_dev,
                                     int target_width, int target_height,
                                     int target_linesize, const uint8_t *y_dev,
                                     const uint8_t *u_dev, const uint8_t *v_dev,
                                     int y_linesize, int u_linesize, int v_linesize,
                                     int source_width, int source_height, float center_x,
                                     float center_y);
/* SATDecoder::InterpolateFrameRectGPU (src/sat_decoder.h:77-82,
 * src/sat_decoder.cc:887-928; interpolate_rect_kernel
 * src/sat_decoder_interpolate_kernel.cl:1-152).  Like the reference kernel the
 * buffers are 4-byte texels with tightly packed rows; the two linesize
 * arguments are accepted and ignored (src/sat_decoder.cc:902-912 never passes
 * them).  The pad byte of every target texel is written as 0. */
int f360_satdec_interpolate_rect(f360_sat_decoder *dec, uint8_t *target_dev,
                                 int target_width, int target_height,
                                 int target_linesize, const uint8_t *source_dev,
                                 int source_width, int source_height,
                                 int source_linesize, float center_x,
                                 float center_y);
/* SATDecoder::DecodeFrameGPU (src/sat_decoder.h:50-51, src/sat_decoder.cc
 * :176-210; decode_kernel src/sat_decoder_decode_kernel.cl:1-58).  The
 * reference launch uses work_dim 0 and always fails; this one runs. */
int f360_satdec_decode(f360_sat_decoder *dec, uint8_t *target_dev,
                       int target_linesize, const uint32_t *sat_dev, int width,
                       int height);

/* ---- ImageSampler ------------------------------------------------------- */
int f360_is_create(f360_ctx *ctx, f360_image_sampler **out);
int f360_is_destroy(f360_image_sampler *is);
/* ImageSampler::InitializeGrid (src/image_sampler.h:57-58,
 * src/image_sampler.cc:170-201; create_grid_kernel
 * src/image_sampler_sample_rect_kernel.cl:48-88) */
int f360_is_initialize_grid(f360_image_sampler *is, int target_width,
                            int target_height, int source_width,
                            int source_height);
/* ImageSampler::InitializeLogpolarGrid (src/image_sampler.h:59-60,
 * src/image_sampler.cc:203-245; create_logpolar_grid_kernel
 * src/image_sampler_sample_logpolar_kernel.cl:5-39) */
int f360_is_initialize_logpolar_grid(f360_image_sampler *is, int target_width,
                                     int target_height, int source_width,
                                     int source_height);
/* reference layouts short[Hr][Wr][2] (host buffers), for parity checks */
int f360_is_export_grid(f360_image_sampler *is, int16_t *grid_host);
int f360_is_export_logpolar_grid(f360_image_sampler *is, int16_t *grid_host);
/* ImageSampler::SampleFrameRectGPU (src/image_sampler.h:61-65,
 * src/image_sampler.cc:247-299; sample_rect_kernel
 * src/image_sampler_sample_rect_kernel.cl:1-46).  Requires InitializeGrid
 * (the reference's auto-init test `grid_size == -1` never fires,
 * src/image_sampler.cc:261). */
int f360_is_sample_rect(f360_image_sampler *is, uint8_t *target_dev,
                        int target_width, int target_height,
                        int target_linesize, const uint8_t *source_dev,
                        int source_width, int source_height,
                        int source_linesize, float center_x, float center_y);
/* ImageSampler::SampleFrameLogPolarGPU (src/image_sampler.h:75-79,
 * src/image_sampler.cc:577-621; sample_logpolar_kernel
 * src/image_sampler_sample_logpolar_kernel.cl:41-86) */
int f360_is_sample_logpolar(f360_image_sampler *is, uint8_t *target_dev,
                            int target_width, int target_height,
                            int target_linesize, const uint8_t *source_dev,
                            int source_width, int source_height,
                            int source_linesize, float center_x,
                            float center_y);
/* ImageSampler::InterpolateFrameLogPolarGPU (src/image_sampler.h:85-89,
 * src/image_sampler.cc:780-819; interpolate_logpolar_kernel
 * src/image_sampler_interpolate_kernel.cl:1-81).  4-byte texels, tight rows,
 * linesizes ignored like the reference; pad byte written as 0. */
int f360_is_interpolate_logpolar(f360_image_sampler *is, uint8_t *target_dev,
                                 int target_width, int target_height,
                                 int target_linesize, const uint8_t *source_dev,
                                 int source_width, int source_height,
                                 int source_linesize, float center_x,
                                 float center_y);
/* ImageSampler::ApplyLogPolarGaussianBlur (src/image_sampler.h:90-92,
 * src/image_sampler.cc:821-857; logpolar_gaussian_blur_kernel
 * src/image_sampler_sample_logpolar_kernel.cl:88-142) */
int f360_is_logpolar_gaussian_blur(f360_image_sampler *is, uint8_t *target_dev,
                                   int target_width, int target_height,
                                   int target_linesize,
                                   const uint8_t *source_dev);

/* ---- Projections -------------------------------------------------------- */
/* Projections::GnomonicProjection (src/projections.h:31-35,
 * src/projections.cc:51-86; gnomonic_kernel src/projections_program.cl:7-47).
 * Argument ORDER follows the .cc definition (width, then height). */
int f360_gnomonic(f360_ctx *ctx, uint8_t *target_dev, int target_width,
                  int target_height, int target_linesize,
                  const uint8_t *source_dev, int source_width,
                  int source_height, int source_linesize, float center_x,
                  float center_y);

/* ---- "expand" debug views --------------------- */
/* SATDecoder::ExpandSampledFrameRectCPU (src/sat_decoder.cc:555-616) and its copy
 * ImageSampler::ExpandSampledFrameRectCPU (src/image_sampler.cc:358-419), on the device: every
 * pixel of the reduced (source) frame is copied to the target pixel the log-rectilinear sampler
 * took it from; 3 bytes per pixel, bytes per pixel = linesize / width on both sides; target
 * pixels nothing lands on keep their value.  The reference has these on the CPU only. */
int f360_expand_rect(f360_ctx *ctx, uint8_t *dst_dev, int dst_w, int dst_h, int dst_linesize,
                     const uint8_t *src_dev, int src_w, int src_h, int src_linesize,
                     float center_x, float center_y);
/* ImageSampler::ExpandSampledFrameLogPolarCPU (src/image_sampler.cc:623-666): the log-polar
 * counterpart.  Where several source pixels land on one target pixel the reference's loop order
 * (columns outer, rows inner) decides; the result here is that of the sequential loop. */
int f360_expand_logpolar(f360_ctx *ctx, uint8_t *dst_dev, int dst_w, int dst_h,
                         int dst_linesize, const uint8_t *src_dev, int src_w, int src_h,
                         int src_linesize, float center_x, float center_y);

/* ---- host-only geometry tables (no device needed) -------------------------
 * The 1-D factors the kernels read instead of the reference's 2-D grids and
 * per-pixel transcendentals; exported so the host logic can be checked on a
 * machine without a GPU. */
/* n_out+1 midpoint offsets of create_grid_kernel
 * (src/sat_decoder_sample_rect_kernel.cl:243-295), one axis */
int f360_tables_satdec_grid_axis(int16_t *out, int n_out, int n_src);
/* n_out offsets of ImageSampler's create_grid_kernel
 * (src/image_sampler_sample_rect_kernel.cl:48-88), one axis */
int f360_tables_is_grid_axis(int16_t *out, int n_out, int n_src);
/* radius[out_w], cos[out_h], sin[out_h] (float) of create_logpolar_grid_kernel
 * (src/image_sampler_sample_logpolar_kernel.cl:5-39) */
int f360_tables_logpolar_axes(float *radius, float *cs, float *sn, int out_w,
                              int out_h);
/* (2*range+1) x {u, dcalc, dmin, du} (int32) of interpolate_rect_kernel
 * (src/sat_decoder_interpolate_kernel.cl:43-89), one axis, indexed by the
 * pixel offset from the gaze centre + range */
int f360_tables_interp_axis(int32_t *out, int range, int n_full, int n_reduced);
/* 16 int32 constants of the yuv420p -> RGB converters: cy, c0, crv, cbu, cgu, cgv, r0, gu0,
 * gv0, b0 (C tables in closed form) and the 16-bit lanes yCoeff, vrCoeff, ubCoeff, vgCoeff,
 * ugCoeff, yOffset (include/FFmpeg42/libswscale/yuv2rgb.c:774-855) */
int f360_tables_yuv2rgb(int32_t *out16);

/* ---- tuning / introspection (not part of the reference surface) --------- */
/* Selects kernel variants for A/B measurements; key/value documented in
 * DESIGN.md.  Unknown keys return F360_ERR_INVALID_ARG. */
int f360_ctx_set_option(f360_ctx *ctx, const char *key, int value);
int f360_ctx_get_option(const f360_ctx *ctx, const char *key, int *value);

/* Per-kernel timing with HIP events on the context's stream (what bench.py's
 * roofline object is computed from).  f360_ctx_profile_arm(ctx, n) makes the
 * next n transform calls record an event pair around each kernel they launch;
 * f360_ctx_profile_read() waits for the recorded events and returns the
 * accumulated time and launch count of one kernel id since the last reset. */
int f360_kernel_count(void);
const char *f360_kernel_name(int kernel_id);
int f360_ctx_profile_arm(f360_ctx *ctx, int calls);
int f360_ctx_profile_read(f360_ctx *ctx, int kernel_id, double *total_ms,
                          int *launches);
/* frames those launches covered (a batched call's launch covers several) */
int f360_ctx_profile_frames(f360_ctx *ctx, int kernel_id, int *frames);
int f360_ctx_profile_reset(f360_ctx *ctx);
/* Debug: per (frame, strip) unit of the last read-once encoder launch that ran with option
 * "debug.ablate" bit 8 set: {start, end} of the unit's wave in 100 MHz ticks, hand-off waits
 * that took the slow path, polls spent in them; then, for f360_satdec_encode_sample_frames, the
 * unit's helper wave: cycles spent waiting for rows, cycles spent on them, rows, boxes (8 x
 * 64-bit words per unit, launch order).
 * Returns the number of units written (<= max_units), or a negative status. */
int f360_debug_walk_stats(f360_ctx *ctx, unsigned long long *out, int max_units);
/* Strips of the read-once encoder whose wait for a hand-off ran into its bound ("debug.walk_spin"
 * polls, about a tenth of a second) since the last call of this function.  Such a strip finishes
 * alone -- it recomputes the row sums to its left from the source -- so the tables are exact
 * either way; a non-zero count only says that time was lost.  Blocks until the stream is idle. */
int f360_debug_walk_recoveries(f360_ctx *ctx, unsigned *count_out);
/* The same count under a name of its own for production callers (a monitoring loop polls it
 * beside f360_sync: a hand-off that timed out costs time, never a result, and f360_sync does not
 * report it).  Reads and clears the counter; blocks until the stream is idle. */
int f360_ctx_handoff_recoveries(f360_ctx *ctx, unsigned *count_out);
/* Test entries for the index-guarded gnomonic remap (csrc/gn_fast_math.h, option
 * "gnomonic.guard").  _sweep: the largest absolute error of a fast float core against double
 * precision over a device-side sweep -- kind 0: asin over every float in [-1, 1] (n ignored);
 * kind 1: atan2 over n pseudo-random and dense argument pairs -- and the bound the guard assumes
 * for it.  _worklist: how many pixels of the last f360_gnomonic call made with option
 * "debug.ablate" bit 9 set took the exact chain. */
int f360_debug_gn_fast_sweep(f360_ctx *ctx, int kind, unsigned long long n, float *worst_out,
                             float *bound_out);
int f360_debug_gnomonic_worklist(f360_ctx *ctx, unsigned *count_out);

#ifdef __cplusplus
}
#endif
#endif /* F360_H */
