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

#ifdef __HIPCC__
// One canvas pixel, BGR u8 (detector.cu:53-79 resize, 102-133 border).  Shared by the stand-alone
// letterbox kernel and by the first-layer kernel that samples the source itself (conv_stem.hip):
// both produce the same bytes.  Every translation unit is built with -ffp-contract=off, so the f32
// operation order is the oracle's.
__device__ __forceinline__ void letterbox_pixel(const LetterboxDesc& d, int x, int y, int fill, unsigned char px[3]) {
    px[0] = px[1] = px[2] = (unsigned char)fill;
    const int rx = x - d.left;
    const int ry = y - d.top;
    if (rx >= 0 && rx < d.rw && ry >= 0 && ry < d.rh) {
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
}

// The same pixel with two 12-byte loads in place of twelve byte loads.  The taps x_lo and x_lo + 1 of
// one source row are six consecutive bytes: the three aligned dwords around them are fetched
// through a bounds-checked buffer resource (out-of-range dwords read as zero and never fault; a
// dword holding a legal byte never crosses a page) and shifted into place.  Arithmetic as above.
struct LetterboxSrc {
    __amdgpu_buffer_rsrc_t rsrc;  // base = src rounded down to 4 bytes
    unsigned delta;               // src - base
};

__device__ __forceinline__ LetterboxSrc letterbox_src(const LetterboxDesc& d) {
    const size_t addr = (size_t)d.src;
    const unsigned delta = (unsigned)(addr & 3);
    // every byte the crop may be sampled at lies below this offset from src
    const unsigned legal = (unsigned)(d.crop_y + d.crop_h - 1) * (unsigned)d.src_stride + (unsigned)(d.crop_x + d.crop_w) * 3u;
    LetterboxSrc s;
    s.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(addr - delta), 0, (delta + legal + 3u) & ~3u, 0x00020000);
    s.delta = delta;
    return s;
}

// o0 / o1: byte offsets (from the resource base) of tap x_lo in rows y_lo / y_hi.  Split into the
// loads and the arithmetic so that a caller with several pixels per lane can have all its loads in
// flight before the first blend.
typedef unsigned int lb_u32x3 __attribute__((ext_vector_type(3)));

__device__ __forceinline__ lb_u32x3 letterbox_taps(const LetterboxSrc& s, unsigned o) {
    return __builtin_amdgcn_raw_buffer_load_b96(s.rsrc, o & ~3u, 0, 0);
}

// a / b: letterbox_taps(o0) / (o1); edge: x_hi == x_lo
__device__ __forceinline__ void letterbox_blend(lb_u32x3 a, lb_u32x3 b, unsigned o0, unsigned o1, bool edge, float lx,
                                                float ly, unsigned char px[3]) {
    const float hy = 1.f - ly;
    const float hx = 1.f - lx;
    // t0 / u0: bytes 0..2 = tap x_lo of the two rows; t1 / u1: bytes 0..2 = tap x_hi
    const unsigned t0 = __builtin_amdgcn_alignbyte(a.y, a.x, o0 & 3u), a_hi = __builtin_amdgcn_alignbyte(a.z, a.y, o0 & 3u);
    const unsigned u0 = __builtin_amdgcn_alignbyte(b.y, b.x, o1 & 3u), b_hi = __builtin_amdgcn_alignbyte(b.z, b.y, o1 & 3u);
    const unsigned t1 = edge ? t0 : __builtin_amdgcn_alignbyte(a_hi, t0, 3u);
    const unsigned u1 = edge ? u0 : __builtin_amdgcn_alignbyte(b_hi, u0, 3u);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float tl = (float)((t0 >> (8 * c)) & 255u) * hy * hx;
        const float tr = (float)((t1 >> (8 * c)) & 255u) * hy * lx;
        const float bl = (float)((u0 >> (8 * c)) & 255u) * ly * hx;
        const float br = (float)((u1 >> (8 * c)) & 255u) * ly * lx;
        const float value = tl + tr + bl + br;
        px[c] = (unsigned char)value;
    }
}

__device__ __forceinline__ void letterbox_pixel_wide(const LetterboxSrc& s, const LetterboxDesc& d, int x, int y,
                                                     int fill, unsigned char px[3]) {
    px[0] = px[1] = px[2] = (unsigned char)fill;
    const int rx = x - d.left;
    const int ry = y - d.top;
    if (rx >= 0 && rx < d.rw && ry >= 0 && ry < d.rh) {
        const float src_y = (float)ry * (float)d.crop_h / (float)d.rh;
        const float src_x = (float)rx * (float)d.crop_w / (float)d.rw;
        const int y_lo = (int)src_y;
        const int y_hi = min(y_lo + 1, d.crop_h - 1);
        const int x_lo = (int)src_x;
        const unsigned col = s.delta + (unsigned)(d.crop_x + x_lo) * 3u;
        const unsigned o0 = (unsigned)(d.crop_y + y_lo) * (unsigned)d.src_stride + col;
        const unsigned o1 = (unsigned)(d.crop_y + y_hi) * (unsigned)d.src_stride + col;
        letterbox_blend(letterbox_taps(s, o0), letterbox_taps(s, o1), o0, o1, x_lo + 1 > d.crop_w - 1,
                        src_x - (float)x_lo, src_y - (float)y_lo, px);
    }
}

// The network-input form of a canvas pixel: RGB * scale as f16, padded to 8 channels (16 B).
__device__ __forceinline__ uint4 letterbox_pixel_f16x8(const unsigned char px[3], float scale) {
    union {
        _Float16 h[8];
        uint4 v;
    } u;
    u.v = make_uint4(0, 0, 0, 0);
    u.h[0] = (_Float16)((float)px[2] * scale);  // detector.cu:160-165: dst[c] = src[2-c] * scale
    u.h[1] = (_Float16)((float)px[1] * scale);
    u.h[2] = (_Float16)((float)px[0] * scale);
    return u.v;
}
#endif

// descs: DEVICE array of n descriptors.  out: device buffer of n canvases.
void launch_letterbox(DeviceCtx& ctx, hipStream_t stream, const LetterboxDesc* descs, int n,
                      int out_w, int out_h, int fill, float scale, LetterboxOut fmt, void* out);

// Host geometry helpers (preparam.h:46-52, detector.cu:394-405).
rmr_preparam make_preparam(int in_w, int in_h, int out_w, int out_h);
void letterbox_geometry(const rmr_preparam& p, int& rw, int& rh, int& top, int& left);

}  // namespace rmr
