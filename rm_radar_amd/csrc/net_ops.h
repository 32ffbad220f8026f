// net_ops.h -- the non-GEMM layers of YOLOv8 on NHWC f16 views: SPPF max-pools, nearest 2x
// upsample into a concat slice, and the Detect head's DFL + dist2bbox + sigmoid.
#pragma once
#include <hip/hip_fp16.h>

#include "common.h"

namespace rmr {

// SPPF (Ultralytics nn/modules/block.py): y1 = pool5(x), y2 = pool5(y1), y3 = pool5(y2) with
// MaxPool2d(5,1,2).  buf is [N][H][W][cs]; x lives at channels [co, co+C); y1,y2,y3 are written
// at [co+C, co+2C), [co+2C, co+3C), [co+3C, co+4C).  C % 8 == 0.
void launch_sppf_pools(DeviceCtx& ctx, hipStream_t s, __half* buf, int N, int H, int W, int cs,
                       int co, int C);

// nearest-neighbour 2x upsample of src view [N][H][W][C] into dst view [N][2H][2W][C]
void launch_upsample2x(DeviceCtx& ctx, hipStream_t s, const __half* src, int src_cs, int src_co,
                       __half* dst, int dst_cs, int dst_co, int N, int H, int W, int C);

// Detect inference form (Ultralytics nn/modules/head.py): for every anchor of one scale,
// DFL softmax-expectation over 16 bins x 4 sides, dist2bbox -> cx,cy,w,h times stride, class
// sigmoid.  box: f32 [N][H*W][64]; cls: f32 [N][H*W][cls_cs]; out: f32 [N][4+nc][A_total],
// this scale's anchors start at a_off.
void launch_head_decode(DeviceCtx& ctx, hipStream_t s, const float* box, const float* cls,
                        int cls_cs, int nc, float* out, int N, int H, int W, int stride,
                        int a_off, int a_total);

// up to three scales in one launch (the arithmetic of launch_head_decode, bit for bit); cls_cs is common to the scales
void launch_head_decode3(DeviceCtx& ctx, hipStream_t s, int scales, const float* const* box, const float* const* cls, int cls_cs, int nc,
                         float* out, int N, const int* H, const int* W, const int* stride, const int* a_off, int a_total);

// The Detect head's LAST convolutions and its decode in one launch (round 6): per scale the 1x1 convolution 64 -> 64 of the box
// branch (the 4 x 16 distribution logits) and the 1x1 convolution c3 -> nc of the class branch run on MFMAs straight from the
// f16 feature rows (a wave per 16 anchors, both filters in registers), the DFL softmax expectation + dist2bbox + sigmoid run
// on the accumulators, and only the [N][4 + nc][A_total] f32 tensor is written.  The f32 logit tensors (320 B per anchor
// written and read back) and six of the tiniest GEMMs of the network (N = 64 / 16: 44-95 TFLOP/s) disappear.
struct HeadFusedScale {
    const __half* hb;      // box-branch features [N][H*W] rows of cs_b halves, 64 channels from co_b
    const __half* hc;      // class-branch features, kc channels from co_c
    const __half* wb;      // [64][kp_b] f16, K-inner
    const __half* wc;      // [16 ceil(nc / 16)][kp_c]
    const float* bb;       // 64
    const float* bc;       // 16 ceil(nc / 16)
    int H, W, stride, a_off, cs_b, co_b, cs_c, co_c, kp_b, kp_c, kc;
};
// kc in {64, 96, 128, 192, 256}, nc <= 16; false: the caller keeps the separate convolutions + launch_head_decode3
bool head_fused_supported(int kb, int kc, int nc);
void launch_head_fused(DeviceCtx& ctx, hipStream_t s, int scales, const HeadFusedScale* sc, int nc, float* out, int N, int a_total);

}  // namespace rmr
