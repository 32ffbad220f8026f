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

    unsigned char px[3] = {(unsigned char)fill, (unsigned char)fill, (unsigned char)fill};
    const int rx = x - d.left;
    const int ry = y - d.top;
    if (rx >= 0 && rx < d.rw && ry >= 0 && ry < d.rh) {
        // detector.cu:53-79
        const float src_y = (float)ry * (float)d.crop_h / (float)d.rh;
        const float src_x = (float)rx * (float)d.crop_w / (float)d.rw;
        const int y_lo = (int)src_y;
        const int y_hi = min(y_lo + 1, d.crop_h - 1);
        const int x_lo = (int)src_x;
        const int x_hi = min(x_lo + 1, d.crop_w - 1);
        const float ly = src_y - (float)y_lo;
        const float lx = src_x - (float)x_lo;
        const float hy = 1.f - ly;
        const float hx = 1.f - lx;
        const uint8_t* r0 = d.src + (size_t)(d.crop_y + y_lo) * d.src_stride + (size_t)d.crop_x * 3;
        const uint8_t* r1 = d.src + (size_t)(d.crop_y + y_hi) * d.src_stride + (size_t)d.crop_x * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float tl = (float)r0[x_lo * 3 + c] * hy * hx;
            const float tr = (float)r0[x_hi * 3 + c] * hy * lx;
            const float bl = (float)r1[x_lo * 3 + c] * ly * hx;
            const float br = (float)r1[x_hi * 3 + c] * ly * lx;
            const float value = tl + tr + bl + br;
            px[c] = (unsigned char)value;
        }
    }

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
        union {
            __half h[8];
            uint4 v;
        } u;
        u.h[0] = __float2half_rn((float)px[2] * scale);
        u.h[1] = __float2half_rn((float)px[1] * scale);
        u.h[2] = __float2half_rn((float)px[0] * scale);
#pragma unroll
        for (int c = 3; c < 8; ++c) u.h[c] = __float2half_rn(0.f);
        ((uint4*)out)[(size_t)img * plane + pix] = u.v;
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
