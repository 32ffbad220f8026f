// postprocess.h -- fused decode + NMS + restore launcher.
#pragma once
#include "common.h"

namespace rmr {

// net_out: DEVICE [n][channels][anchors] f32.  pps_dev: DEVICE rmr_preparam[n].
// scratch: DEVICE, postprocess_scratch_bytes(n, anchors).  out_dev: DEVICE [n][cap].
void launch_postprocess(DeviceCtx& ctx, hipStream_t stream, const float* net_out, int n,
                        int channels, int anchors, int classes, float nms_thresh,
                        float conf_thresh, const rmr_preparam* pps_dev, void* scratch,
                        rmr_detection* out_dev, int* counts_dev, int cap);
size_t postprocess_scratch_bytes(int n, int anchors);
// heads_dev: DEVICE, n * head_rows * sizeof(rmr_detection) + n * sizeof(int): the first head_rows rows of every image,
// then the n counts, contiguous (one D2H copy instead of a strided one)
void launch_gather_heads(hipStream_t stream, const rmr_detection* dets_dev, const int* counts_dev, int cap, int head_rows, int n,
                         void* heads_dev);

void launch_transpose(hipStream_t stream, const float* src, float* dst, int rows, int cols);

}  // namespace rmr
