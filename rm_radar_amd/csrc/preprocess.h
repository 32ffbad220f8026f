// preprocess.h -- fused letterbox (resize + border + BGR->RGB + scale) launcher.
#pragma once
#include "common.h"

namespace rmr {

// One image (or one crop of it) and where it lands in the canvas.
struct LetterboxDesc {
    const uint8_t* src;  // device pointer, BGR u8 HWC
    int src_stride;      // bytes per source row
    int crop_x, crop_y, crop_w, crop_h;
    int rw, rh;          // resized size (detector.cu:394-400: truncated)
    int top, left;       // paste offset (detector.cu:402-405: rounded)
};

enum LetterboxOut { LB_U8_HWC = 0, LB_F32_NCHW = 1, LB_F16_NHWC8 = 2 };

// descs: DEVICE array of n descriptors.  out: device buffer of n canvases.
void launch_letterbox(DeviceCtx& ctx, hipStream_t stream, const LetterboxDesc* descs, int n,
                      int out_w, int out_h, int fill, float scale, LetterboxOut fmt, void* out);

// Host geometry helpers (preparam.h:46-52, detector.cu:394-405).
rmr_preparam make_preparam(int in_w, int in_h, int out_w, int out_h);
void letterbox_geometry(const rmr_preparam& p, int& rw, int& rh, int& top, int& left);

}  // namespace rmr
