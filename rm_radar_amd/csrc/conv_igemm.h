// conv_igemm.h -- implicit-GEMM convolution on MFMA (f16 in, f32 accumulate), NHWC.
#pragma once
#include <hip/hip_fp16.h>

#include "common.h"

namespace rmr {

// One conv + bias (+SiLU) (+residual) over strided NHWC views.  Every tensor is addressed as
// base + pixel * cstride + coff, so a layer can read or write a channel slice of a wider
// buffer (C2f / SPPF / FPN concats are never materialised by a copy).
struct ConvArgs {
    const __half* in;
    int in_cs, in_co;  // elements per pixel of the input buffer, first channel consumed
    int N, H, W, Cin;  // Cin: channels consumed, multiple of 8
    int Ho, Wo, KH, KW, stride, pad;
    const __half* wt;   // packed [Cout_pad][Kp], k = (kh*KW + kw)*Cin + ci, zero padded
    const float* bias;  // [Cout_pad]
    __half* out;        // f16 output view (or null when out32 is set)
    float* out32;       // f32 output view
    int out_cs, out_co;
    const __half* res;  // optional residual view, added after the activation
    int res_cs, res_co;
    // optional f32 addend at HALF the output resolution, added before the activation: pixel (y, x)
    // takes pre[(y/2, x/2)].  A 1x1 convolution commutes with nearest-neighbour upsampling, so
    // conv1x1(concat[up2x(A), B]) = act(W_B.B + bias + up2x(W_A.A)): `pre` is W_A.A (conv_igemm and
    // conv_dma only; Ho and Wo even)
    const float* pre;
    int pre_cs;
    // planar channel groups ("slabs", conv_pw only; 0 = off): channel ch of the input lives in slab
    // ch / in_slab_c at pixel pitch in_cs, slabs in_slab_stride BYTES apart; likewise the output.  A
    // C2f keeps its chunks as slabs so that the 3x3 convs between its two 1x1 convs read and write
    // contiguous rows instead of 96..192-byte slices of wide pixels (whole 128-byte lines travel)
    int in_slab_c, out_slab_c;
    unsigned in_slab_stride, out_slab_stride;
    int Cout_pad;       // multiple of the tile's BN
    int K, Kp, M;       // K = KH*KW*Cin, Kp = K rounded up to 64, M = N*Ho*Wo
    int act;            // 1 = SiLU
    unsigned in_bytes;  // bytes addressable from `in` (buffer-load bounds: reads past it return 0)
    unsigned wt_bytes;  // bytes of the packed weights
    // conv_t32 only: the same weights as the LDS images of their (chunk, tap) slices (pack_conv_weights_t32)
    const __half* wt_t32;
    unsigned wt_t32_bytes;
    // conv_w1d only: the Winograd F(2, 3) transformed weights U = g G^T as the LDS images of their 12 (filter row, xi)
    // slices per 32-channel chunk (pack_conv_weights_w1d)
    const __half* wt_w1d;
    unsigned wt_w1d_bytes;
    // conv_wsf only (a fused C2f bottleneck, conv_ws.hip): the SECOND convolution's packed weights and bias; `in` is the
    // bottleneck's input (and its shortcut), `out` the bottleneck's output, the hidden tensor lives in LDS
    const __half* wt2;
    const float* bias2;
    // conv_t32f8 only: the input quantised to e4m3 (rows of in8_cs bytes, zero beyond Cin), the weights as e4m3
    // LDS images with one scale per output channel (pack_conv_weights_t32f8)
    const unsigned char* in8;
    int in8_cs;
    unsigned in8_bytes;
    const unsigned char* wt8;
    unsigned wt8_bytes;
    const float* wscale;
    // optional: the f16 output also as e4m3 rows of out8_cs bytes (the next e4m3 layer's input, written here
    // instead of by a quantiser pass); bytes beyond Cout stay as they are (the planner keeps them zero)
    unsigned char* out8;
    int out8_cs;
    int out8_only;      // 1: the e4m3 rows are the ONLY output (the tensor's single reader is an e4m3 layer): no f16 store
    // split-K (conv_dma only): `split` workgroups share one output tile, each accumulating a
    // contiguous range of K slices; partial tiles meet in splitk_ws and the last arriver (ticket in
    // splitk_cnt, which it resets to 0) reduces them and runs the epilogue.  split <= 1: off.
    int split;
    float* splitk_ws;
    int* splitk_cnt;
    long long* timing;  // debug: per-phase cycle totals of one wave (RMR_CONV_TIMING), else null
    double flops;       // algorithmic FLOPs of this launch (true channel counts), for profiling
};

struct ConvTile {
    int bm, bn, bk;
};

int conv_num_tiles();
ConvTile conv_tile(int id);
// best tile for (M, Cout_pad) on a chip with num_cus compute units
int conv_pick_tile(int M, int cout_pad, int num_cus);
void launch_conv(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int tile);

// LDS-DMA pipelined variant (conv_dma.hip) for layers with Cin % 32 == 0
bool conv_dma_supported(const ConvArgs& a);
int conv_dma_num_tiles();
ConvTile conv_dma_tile(int id);
int conv_dma_pick_tile(int M, int cout_pad, int num_cus);
void launch_conv_dma(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int tile);
// workspace floats / counters a split-K launch of `tile` needs for this layer
size_t conv_dma_splitk_ws_floats(const ConvArgs& a, int tile, int split);
int conv_dma_splitk_tiles(const ConvArgs& a, int tile);
// halo-staged 3x3 / stride-1 variant (conv_halo.hip): the input range is fetched once per 32-channel
// chunk and reused by all nine taps
int conv_halo_num_tiles();
ConvTile conv_halo_tile(int id);
bool conv_halo_supported(const ConvArgs& a, int tile);  // tile < 0: any
void launch_conv_halo(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int tile);
// weights-stationary 3x3 / stride-1 variant for the 48-channel, 160-wide layers (conv_ws.hip): the
// filter stays in registers and each workgroup walks a strip of image rows
int conv_ws_num_variants();
bool conv_ws_supported(const ConvArgs& a, int variant);  // variant < 0: any
void launch_conv_ws(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int variant);
// a whole C2f bottleneck of that shape in one launch: out = in + SiLU(conv2(SiLU(conv1(in)))), both 3x3 / 48 -> 48 on
// 160-wide maps (a.wt / a.bias, a.wt2 / a.bias2); the hidden tensor never leaves LDS.  Bit-identical to the two launches.
int conv_wsf_num_variants();
bool conv_wsf_supported(const ConvArgs& a, int variant);  // variant < 0: any
void launch_conv_wsf(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int variant);
// small-batch variant (conv_direct.hip): operands fetched from L2 in MFMA fragment shape, the K loop
// split across the waves of a workgroup; Cin % 32 == 0
int conv_direct_num_tiles();
ConvTile conv_direct_tile(int id);
bool conv_direct_supported(const ConvArgs& a, int tile);  // tile < 0: any
void launch_conv_direct(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int tile);
// pointwise layers with K = 96 / 192 / 384 and N = 96 / 192 (conv_pw.hip): weights stationary in
// registers, persistent walk over the pixels through an LDS-DMA ring
int conv_pw_num_variants();
bool conv_pw_supported(const ConvArgs& a, int variant);  // variant < 0: any
void launch_conv_pw(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int variant);

// the network's first layer (conv_stem.hip): 3x3 / stride 2, 8 stored -> 48 channels, an HBM stream
bool conv_stem_supported(const ConvArgs& a);
void launch_conv_stem(DeviceCtx& ctx, hipStream_t stream, ConvArgs a);
// the same layer sampling the BGR u8 source frames itself (letterbox fused in): descs is a DEVICE array
// of a.N descriptors, a.in is not read
struct LetterboxDesc;
void launch_conv_stem_letterbox(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, const LetterboxDesc* descs, int fill,
                                float scale);
// the network's second layer (conv_ws_s2.hip): 3x3 / stride 2, 48 -> 96 channels on a 320-wide map,
// weights stationary in registers, input rows de-interleaved by column parity in an LDS ring
int conv_ws_s2_num_variants();
bool conv_ws_s2_supported(const ConvArgs& a, int variant);  // variant < 0: any
void launch_conv_ws_s2(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int variant);
// 3x3 / stride-1 layers with Cin % 32 == 0 on 32x32x16 MFMAs (conv_t32.hip): one 8-wave workgroup per
// CU, fragment reads half a tap ahead of the MFMAs, weights pre-packed as LDS images (a.wt_t32)
int conv_t32_num_tiles();
ConvTile conv_t32_tile(int id);
bool conv_t32_supported(const ConvArgs& a, int tile);  // tile < 0: any
void launch_conv_t32(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int tile);
// split-K (a.split > 1, small batches): one workgroup per (tile, range of 32-channel chunks), partial tiles through a.splitk_ws
int conv_t32_splitk_tiles(const ConvArgs& a, int tile);
size_t conv_t32_splitk_ws_floats(const ConvArgs& a, int tile, int split);
bool conv_t32_splitk_supported(const ConvArgs& a, int tile, int split, int num_cus);
// taps = KH * KW of the layer (9, or 1 for the 1x1 layers conv_g32 runs)
void pack_conv_weights_t32(const __half* packed, int cout_pad, int cin, int Kp, std::vector<__half>& out, int taps = 9);
// 1x1 and 3x3 layers of any stride with Cin % 32 == 0 on the same skeleton (conv_g32.hip): the pixel rows of
// every (tap, chunk) stage are gathered by the DMA next to the stage's weight slice (a.wt_t32)
int conv_g32_num_tiles();
ConvTile conv_g32_tile(int id);
bool conv_g32_supported(const ConvArgs& a, int tile);  // tile < 0: any
void launch_conv_g32(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int tile);
// 3x3 / stride-1 layers through Winograd F(2, 3) along x on the conv_t32 skeleton: a measured experiment of round 3 (slower
// than conv_t32 on every layer, DESIGN.md "Round 3"), NOT part of the product library: tools/experiments/conv_w1d.hip is
// compiled in by `make EXPERIMENTS=1` only; the default build has no such tiles (ids 980.. are rejected)
#ifdef RMR_EXPERIMENTS
int conv_w1d_num_tiles();
ConvTile conv_w1d_tile(int id);
bool conv_w1d_supported(const ConvArgs& a, int tile);  // tile < 0: any
void launch_conv_w1d(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int tile);
void pack_conv_weights_w1d(const __half* packed, int cout_pad, int cin, int Kp, std::vector<__half>& out);
#else
inline int conv_w1d_num_tiles() { return 0; }
inline bool conv_w1d_supported(const ConvArgs&, int) { return false; }
inline void launch_conv_w1d(DeviceCtx&, hipStream_t, ConvArgs, int) {}
inline void pack_conv_weights_w1d(const __half*, int, int, int, std::vector<__half>& out) { out.assign(8, __half()); }
#endif
// small batches (conv_sb.hip, kernel ids kSbBase + variant): small tiles whose whole operand set is in flight at once --
// 3x3 / stride-1 layers in the halo form (even variants), 1x1 and strided 3x3 layers in the gathered form (odd variants);
// Cin % 32 == 0, weights from the conv_t32 LDS images (a.wt_t32)
constexpr int kSbBase = 100000;
int conv_sb_num_variants();
ConvTile conv_sb_tile(int id);
bool conv_sb_supported(const ConvArgs& a, int variant);  // variant < 0: the layer shape only
void launch_conv_sb(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int variant);
// n (2..8) INDEPENDENT layers in one launch of variant `variant` (the Detect head's branches; kernel ids kSbGroupBase + variant on
// the first layer of a group, kGroupedAway on the others): the problem table (conv_sb_group_bytes(n) bytes, built on the host by
// conv_sb_group_build) lives in device memory for as long as launches (or captured graphs) use it
constexpr int kSbGroupBase = 200000;
constexpr int kGroupedAway = 398;
size_t conv_sb_group_bytes(int n);
bool conv_sb_group_supported(const ConvArgs* args, int n, int variant);
int conv_sb_group_build(const ConvArgs* args, int n, int variant, void* host_buf);
void launch_conv_sb_group(DeviceCtx& ctx, hipStream_t stream, const ConvArgs* args, int n, const void* dev_probs, int variant);
// the fp8 form (conv_t32f8.hip): e4m3 operands on v_mfma_scale_f32_32x32x64_f8f6f4, f16 output
int conv_t32f8_num_tiles();
ConvTile conv_t32f8_tile(int id);
bool conv_t32f8_supported(const ConvArgs& a, int tile);  // tile < 0: the layer SHAPE only (no tile geometry)
// plan time: whether at least one tile runs a 3x3 / stride-1 layer of this width on W-wide maps (N divisible by a tile's
// channel count, the halo rows in the tile's LDS slots) -- the planner gives a layer e4m3 operands only then; returns the
// first such tile, -1 if none
int conv_t32f8_first_tile(int cout_pad, int W);
void launch_conv_t32f8(DeviceCtx& ctx, hipStream_t stream, ConvArgs a, int tile);
void launch_quant_f8(DeviceCtx& ctx, hipStream_t stream, const __half* in, int cs, int co, int C, unsigned char* out, int pitch, long npix);
void pack_conv_weights_t32f8(const __half* packed, int cout_pad, int cin, int Kp, std::vector<unsigned char>& out, std::vector<float>& scale);
unsigned char f32_to_e4m3(float x);
float e4m3_to_f32(unsigned char b);
// picks the kernel family and tile for a layer (RMR_CONV=igemm|dma overrides) and launches it
void launch_conv_auto(DeviceCtx& ctx, hipStream_t stream, const ConvArgs& a);

// Packs OIHW f32 weights into the engine's [Cout_pad][Kp] f16 layout (host side).
// cin_pad >= cin: extra input channels get zero weights.
void pack_conv_weights(const float* w_oihw, int cout, int cin, int kh, int kw, int cin_pad,
                       int cout_pad, std::vector<__half>& out, int& K, int& Kp);

}  // namespace rmr
