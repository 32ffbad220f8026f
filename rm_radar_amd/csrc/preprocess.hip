// preprocess.hip -- one fused launch for the reference's resizeKernel + copyMakeBorderKernel
// + blobKernel (src/detect/detector.cu:40-81, 102-133, 151-171), batched over images/crops.
//
// HBM-bound byte work: one thread per canvas pixel, x fastest so stores coalesce; the four
// bilinear taps of neighbouring threads fall in neighbouring source lines (L2/MALL absorb the
// re-reads).  Arithmetic keeps the reference's intermediate quantisation (SURVEY Appendix A
// Q1-Q4): f32 bilinear in the order tl+tr+bl+br, truncation to u8 in-register, then
// u8 * scale in f32.  Compiled with -ffp-contract=off so the op order is the oracle's.
#include <hip/hip_fp16.h>

#include "preprocess.h"

namespace rmr {

template <int FMT>
__global__ __launch_bounds__(256) void letterbox_kernel(const LetterboxDesc* __restrict__ descs,
                                                        int out_w, int out_h, int fill,
                                                        float scale, void* __restrict__ out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int img = blockIdx.z;
    if (x >= out_w || y >= out_h) return;
    const LetterboxDesc d = descs[img];

    unsigned char px[3];
    letterbox_pixel(d, x, y, fill, px);

    const size_t plane = (size_t)out_w * out_h;
    const size_t pix = (size_t)y * out_w + x;
    if (FMT == LB_U8_HWC) {
        uint8_t* o = (uint8_t*)out + ((size_t)img * plane + pix) * 3;
        o[0] = px[0];
        o[1] = px[1];
        o[2] = px[2];
    } else if (FMT == LB_F32_NCHW) {
        // detector.cu:160-165: dst[c] = src[2-c] * scale
        float* o = (float*)out + (size_t)img * plane * 3 + pix;
        o[0] = (float)px[2] * scale;
        o[plane] = (float)px[1] * scale;
        o[2 * plane] = (float)px[0] * scale;
    } else {
        // network input: f16 NHWC padded to 8 channels (16 B per pixel, one store)
        ((uint4*)out)[(size_t)img * plane + pix] = letterbox_pixel_f16x8(px, scale);
    }
}

void launch_letterbox(DeviceCtx& ctx, hipStream_t stream, const LetterboxDesc* descs, int n,
                      int out_w, int out_h, int fill, float scale, LetterboxOut fmt, void* out) {
    if (n <= 0) return;
    dim3 grid((out_w + 63) / 64, (out_h + 3) / 4, n);
    const double out_bytes =
        (double)n * out_w * out_h * (fmt == LB_U8_HWC ? 3 : fmt == LB_F32_NCHW ? 12 : 16);
    ProfScope ps(ctx.prof, stream, "letterbox", 0, out_bytes + (double)n * out_w * out_h * 3);
    switch (fmt) {
        case LB_U8_HWC:
            letterbox_kernel<LB_U8_HWC><<<grid, 256, 0, stream>>>(descs, out_w, out_h, fill, scale, out);
            break;
        case LB_F32_NCHW:
            letterbox_kernel<LB_F32_NCHW><<<grid, 256, 0, stream>>>(descs, out_w, out_h, fill, scale, out);
            break;
        default:
            letterbox_kernel<LB_F16_NHWC8><<<grid, 256, 0, stream>>>(descs, out_w, out_h, fill, scale, out);
            break;
    }
    RMR_HIP(hipGetLastError());
}

// ---- host geometry --------------------------------------------------------------------

// preparam.h:46-52
rmr_preparam make_preparam(int in_w, int in_h, int out_w, int out_h) {
    rmr_preparam p;
    p.height = (float)in_h;
    p.width = (float)in_w;
    const float rh = (float)out_h / p.height;
    const float rw = (float)out_w / p.width;
    p.ratio = 1.0f / (rh < rw ? rh : rw);
    p.dw = ((float)out_w - roundf(p.width / p.ratio)) * 0.5f;
    p.dh = ((float)out_h - roundf(p.height / p.ratio)) * 0.5f;
    return p;
}

// detector.cu:394-405: float sizes truncate into the kernels' int parameters; the border
// offsets are round(double(d) - 0.1).
void letterbox_geometry(const rmr_preparam& p, int& rw, int& rh, int& top, int& left) {
    const float padding_width = p.width / p.ratio;
    const float padding_height = p.height / p.ratio;
    rw = (int)padding_width;
    rh = (int)padding_height;
    top = (int)round((double)p.dh - 0.1);
    left = (int)round((double)p.dw - 0.1);
}

}  // namespace rmr
