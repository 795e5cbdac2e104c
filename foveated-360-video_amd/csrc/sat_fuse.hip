// sat_fuse.hip -- encode + sample in one pass on the read-once encoder
// (f360_satdec_encode_sample_frames with enough frames to fill the device): the row plan, the
// one-pass instantiations of sat_walk_kernel and the fix-up kernel.
#include "sat_walk.h"
#include "sat_fuse_kernels.h"

using namespace f360::sat;

// Encode + sample in one pass (f360_satdec_encode_sample_frames, sat_decoder.hip).  False: the
// read-once encoder would not take this call, or the frames / grid are outside what the strip
// owners' packed bookkeeping holds; the caller then makes the two calls.
bool f360::sat_encode_sample_applies(const f360_ctx *ctx, int count, int width, int height,
                                     int linesize, int out_w, int out_h, int dst_linesize,
                                     const f360::YuvPlanes *yuv) {
  // the source side: f360_sat_encode_batch's / f360_sat_encode_yuv420p_batch's own rules
  const bool source_ok =
      yuv ? height % 2 == 0 && yuv->y_linesize >= width && yuv->u_linesize >= width / 2 &&
                yuv->v_linesize >= width / 2 && yuv->y_linesize % 4 == 0 &&
                yuv->u_linesize % 2 == 0 && yuv->v_linesize % 2 == 0
          : linesize / width == 4 && linesize % 16 == 0;
  return walk_wanted(ctx, count, width) && source_ok && width % 4 == 0 &&
         (size_t)width * height * 3 < ((size_t)1 << 31) &&
         out_w < 65536 && out_h < 65536 && (width + kStripPx - 1) / kStripPx <= kFixCols / 4 &&
         dst_linesize % 4 == 0 && dst_linesize >= 4 * out_w;
}

// The launches of one call: per launch the row plan, the one-pass strip walker and the fix-up.
int f360::sat_encode_sample_walk(f360_ctx *ctx, int count, uint32_t *const *sats,
                                 const uint8_t *const *srcs, const f360::YuvPlanes *yuvs,
                                 int width, int height, int linesize, const f360::SatFuse &fuse,
                                 bool prof) {
  WalkSetup ws;
  int st = walk_prepare(ctx, count, yuvs, width, height, linesize, &fuse, ws);
  if (st != F360_OK) return st;
  EncodeArgs &a = ws.a;
  f360::SatEncodePlan &p = ctx->enc;
  for (int k0 = 0; k0 < count; k0 += ws.per_launch) {
    const int n = std::min(count - k0, ws.per_launch);
    WalkBatch wb;
    walk_fill_batch(wb, k0, n, sats, srcs, yuvs);
    a.walk_units = n * ws.nstrips;
    WalkFuse wf;
    for (int k = 0; k < kWalkFrames; ++k) {
      const int q = k0 + (k < n ? k : 0);
      wf.dst[k] = fuse.dsts[q];
      wf.cxp[k] = (int)(fuse.centers_xy[2 * q] * (float)width);  // sat_decoder_sample_rect_kernel.cl:176-179
      wf.cyp[k] = (int)(fuse.centers_xy[2 * q + 1] * (float)height);
    }
    wf.gx = fuse.gx;
    wf.gy = fuse.gy;
    wf.rowplan = p.walk_plan.as<uint32_t>();
    wf.plan_stride = ws.plan_stride;
    wf.out_w = fuse.out_w;
    wf.out_h = fuse.out_h;
    wf.dst_linesize = fuse.dst_linesize;
    wf.spix = wf.rowplan + ws.plan_words;
    wf.side = wf.spix + (size_t)ws.per_launch * kSpixWords;
    wf.pmax = ws.pmax;
    wf.side_stride = ws.side_stride;
    wf.lrows_max = (ctx->opt_fuse_force & 2) ? 0 : kFixLrowsWalk;
    wf.force_tail = (ctx->opt_fuse_force & 4) ? 1 : 0;
    wf.band_rows = 0;
    wf.ent = nullptr;
    {
      f360::KernelSpan span(ctx, f360::kWalkFusePlan, prof, n);
      hipLaunchKernelGGL(walk_fuse_plan_kernel<0>, dim3(n), dim3(256), 0, ctx->stream, wf.gy,
                         wf.out_h, width, height, wf.rowplan, ws.plan_stride, wf);
    }
    {
      f360::KernelSpan span(ctx, f360::kSatWalk, prof, n);
      const dim3 fgrid((a.walk_units + kFuseOwners - 1) / kFuseOwners), fblock(128 * kFuseOwners);
      if (ws.yuv_src == kSrcYuvSwsX86)
        hipLaunchKernelGGL((sat_walk_kernel<kSrcYuvSwsX86, 2, true>), fgrid, fblock, 0,
                           ctx->stream, a, wb, wf);
      else if (ws.yuv_src == kSrcYuvSwsC)
        hipLaunchKernelGGL((sat_walk_kernel<kSrcYuvSwsC, 2, true>), fgrid, fblock, 0,
                           ctx->stream, a, wb, wf);
      else
        hipLaunchKernelGGL((sat_walk_kernel<kSrcRgb0, 2, true>), fgrid, fblock, 0, ctx->stream,
                           a, wb, wf);
    }
    {
      f360::KernelSpan span(ctx, f360::kWalkFuseFix, prof, n);
      hipLaunchKernelGGL(walk_fuse_fix_kernel<0>,
                         dim3((wf.out_h * ws.pmax + 255) / 256 +
                                  kFixLrowsWalk * ((wf.out_w + 255) / 256), n),
                         dim3(256), 0, ctx->stream, wb, wf, width, height, linesize,
                         !yuvs ? -1 : ctx->opt_yuv_model == 1 ? 1 : 0, a.yuv, a.k);
    }
  }
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}
