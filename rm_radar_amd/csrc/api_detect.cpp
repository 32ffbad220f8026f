// api_detect.cpp -- C-ABI entry points (include/rmr.h) for Detector, RobotDetector and the
// single-layer conv hook.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "conv_igemm.h"
#include "api_handles.h"

using namespace rmr;

extern "C" {

// detector.h:87-93 defaults
void rmr_detector_cfg_default(rmr_detector_cfg* c) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->nms_thresh = 0.65f;
    c->conf_thresh = 0.25f;
    c->input_width = 640;
    c->input_height = 640;
    c->input_channels = 3;
    c->max_batch_size = 1;
}

rmr_status rmr_detector_create(const rmr_detector_cfg* cfg, rmr_detector** out) {
    return guarded([&] {
        if (!cfg || !out) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_detector_create: null argument");
        *out = new rmr_detector(*cfg);
    });
}

void rmr_detector_destroy(rmr_detector* det) { delete det; }

rmr_status rmr_detector_detect(rmr_detector* det, const rmr_image* imgs, const int* crops, int n,
                               rmr_detection* out, int* counts, int cap) {
    return guarded([&] {
        if (!det) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_detector_detect: null detector");
        det->impl.detect(imgs, crops, n, out, counts, cap);
    });
}

rmr_status rmr_detector_infer(rmr_detector* det, const rmr_image* imgs, const int* crops, int n, float* net_out,
                              rmr_preparam* pp) {
    return guarded([&] {
        if (!det) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_detector_infer: null detector");
        det->impl.infer(imgs, crops, n, net_out, pp);
    });
}

rmr_status rmr_detector_read_feature(rmr_detector* det, const char* name, int img, float* out, int* dims) {
    return guarded([&] {
        if (!det || !name || !dims) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_detector_read_feature: null argument");
        det->impl.ctx().use();
        if (!det->impl.net().read_feature(det->impl.stream(), name, img, out, dims))
            fail(RMR_ERR_INVALID_ARGUMENT, "rmr_detector_read_feature: no stage named '%s'", name);
    });
}

double rmr_detector_arena_bytes(const rmr_detector* det) { return det ? (double)const_cast<rmr_detector*>(det)->impl.net().arena_bytes() : 0.0; }
int rmr_detector_chunk(const rmr_detector* det) { return det ? const_cast<rmr_detector*>(det)->impl.net().chunk() : 0; }
int rmr_detector_anchors(const rmr_detector* det) { return det ? const_cast<rmr_detector*>(det)->impl.net().anchors() : 0; }
int rmr_detector_channels(const rmr_detector* det) { return det ? const_cast<rmr_detector*>(det)->impl.net().channels() : 0; }
double rmr_detector_flops_per_image(const rmr_detector* det) {
    return det ? const_cast<rmr_detector*>(det)->impl.net().flops_per_image() : 0.0;
}

// detector.h:173-180 defaults
void rmr_robot_detector_cfg_default(rmr_robot_detector_cfg* c) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->iou_thresh = 0.75f;
    c->car_nms_thresh = 0.65f;
    c->car_conf_thresh = 0.25f;
    c->armor_nms_thresh = 0.65f;
    c->armor_conf_thresh = 0.50f;
    c->input_width = 640;
    c->input_height = 640;
    c->input_channels = 3;
    c->max_frames = 1;
}

rmr_status rmr_robot_detector_create(const rmr_robot_detector_cfg* cfg, rmr_robot_detector** out) {
    return guarded([&] {
        if (!cfg || !out) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_robot_detector_create: null argument");
        if (!cfg->car_engine_path || !cfg->armor_engine_path)
            fail(RMR_ERR_INVALID_ARGUMENT, "rmr_robot_detector_create: null engine path");
        *out = new rmr_robot_detector(*cfg);
    });
}

void rmr_robot_detector_destroy(rmr_robot_detector* rd) { delete rd; }
double rmr_robot_detector_arena_bytes(rmr_robot_detector* rd) { return rd ? (double)rd->impl.arena_bytes() : 0.0; }

rmr_status rmr_robot_detector_detect(rmr_robot_detector* rd, const rmr_image* img, rmr_robot* out, int* n_out,
                                     int cap) {
    return guarded([&] {
        if (!rd) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_robot_detector_detect: null detector");
        rd->impl.detect_batch(img, 1, nullptr, 0, out, n_out, cap);
    });
}

rmr_status rmr_robot_detector_detect_batch(rmr_robot_detector* rd, const rmr_image* imgs, int n_frames,
                                           const int* forced_crops, int forced_per_frame, rmr_robot* out,
                                           int* n_out, int cap) {
    return guarded([&] {
        if (!rd) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_robot_detector_detect_batch: null detector");
        rd->impl.detect_batch(imgs, n_frames, forced_crops, forced_per_frame, out, n_out, cap);
    });
}

rmr_status rmr_robot_detector_read_heads(rmr_robot_detector* rd, int stage, int first, int n, float* out, rmr_preparam* pp,
                                         int* n_last) {
    return guarded([&] {
        if (!rd || (stage != 0 && stage != 1)) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_robot_detector_read_heads: bad detector / stage");
        const int m = rd->impl.stage(stage).read_heads(first, n, out, pp);
        if (n_last) *n_last = m;
    });
}

// One layer through the conv engine with host f32 tensors (parity tests of the kernel).
rmr_status rmr_conv2d(int device, const float* x, int n, int h, int w, int cin, const float* wt, const float* bias,
                      int cout, int kh, int kw, int stride, int pad, int silu, const float* residual, float* y,
                      int tile) {
    return guarded([&] {
        if (!x || !wt || !y || n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0)
            fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv2d: bad arguments");
        DeviceCtx& ctx = device_ctx(device);
        const int cin_pad = (cin + 7) / 8 * 8, cout_pad = (cout + 15) / 16 * 16;
        const int ho = (h + 2 * pad - kh) / stride + 1, wo = (w + 2 * pad - kw) / stride + 1;
        if (ho <= 0 || wo <= 0) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv2d: empty output");
        const size_t npx_in = (size_t)n * h * w, npx_out = (size_t)n * ho * wo;
        std::vector<__half> hx(npx_in * cin_pad, __float2half(0.f)), hr;
        for (size_t p = 0; p < npx_in; ++p)
            for (int c = 0; c < cin; ++c) hx[p * cin_pad + c] = __float2half(x[p * cin + c]);
        if (residual) {
            hr.assign(npx_out * cout_pad, __float2half(0.f));
            for (size_t p = 0; p < npx_out; ++p)
                for (int c = 0; c < cout; ++c) hr[p * cout_pad + c] = __float2half(residual[p * cout + c]);
        }
        std::vector<__half> packed;
        ConvArgs a{};
        pack_conv_weights(wt, cout, cin, kh, kw, cin_pad, cout_pad, packed, a.K, a.Kp);
        std::vector<float> b(cout_pad, 0.f);
        if (bias) std::copy(bias, bias + cout, b.begin());
        DevBuf<__half> dx, dw, dr;
        DevBuf<float> db, dy;
        dx.alloc(hx.size());
        dw.alloc(packed.size());
        db.alloc(b.size());
        dy.alloc(npx_out * cout_pad);
        RMR_HIP(hipMemcpyAsync(dx.p, hx.data(), hx.size() * sizeof(__half), hipMemcpyHostToDevice, ctx.stream));
        RMR_HIP(hipMemcpyAsync(dw.p, packed.data(), packed.size() * sizeof(__half), hipMemcpyHostToDevice, ctx.stream));
        RMR_HIP(hipMemcpyAsync(db.p, b.data(), b.size() * sizeof(float), hipMemcpyHostToDevice, ctx.stream));
        if (residual) {
            dr.alloc(hr.size());
            RMR_HIP(hipMemcpyAsync(dr.p, hr.data(), hr.size() * sizeof(__half), hipMemcpyHostToDevice, ctx.stream));
        }
        a.in = dx.p;
        a.in_cs = cin_pad;
        a.N = n;
        a.H = h;
        a.W = w;
        a.Cin = cin_pad;
        a.Ho = ho;
        a.Wo = wo;
        a.KH = kh;
        a.KW = kw;
        a.stride = stride;
        a.pad = pad;
        a.wt = dw.p;
        a.bias = db.p;
        // RMR_CONV2D_OUT16=1 (tests): the layer writes its f16 view, as inside the network -- the kernels' production
        // epilogues (LDS stages, 16-byte stores, in-register shortcut adds) instead of their f32 parity view
        const bool out16 = std::getenv("RMR_CONV2D_OUT16") && std::atoi(std::getenv("RMR_CONV2D_OUT16")) != 0;
        DevBuf<__half> dy16;
        if (out16) {
            dy16.alloc(npx_out * cout_pad);
            a.out = dy16.p;
        } else {
            a.out32 = dy.p;
        }
        a.out_cs = cout_pad;
        if (residual) {
            a.res = dr.p;
            a.res_cs = cout_pad;
        }
        a.Cout_pad = cout_pad;
        a.M = (int)npx_out;
        a.act = silu;
        a.in_bytes = (unsigned)(hx.size() * sizeof(__half));
        a.wt_bytes = (unsigned)(packed.size() * sizeof(__half));
        DevBuf<__half> dw32;
        if (kh == kw && (kh == 3 || kh == 1) && cin_pad % 32 == 0) {
            std::vector<__half> p32;
            pack_conv_weights_t32(packed.data(), cout_pad, cin_pad, a.Kp, p32, kh * kw);
            dw32.alloc(p32.size());
            RMR_HIP(hipMemcpyAsync(dw32.p, p32.data(), p32.size() * sizeof(__half), hipMemcpyHostToDevice, ctx.stream));
            RMR_HIP(hipStreamSynchronize(ctx.stream));  // p32 dies at the end of this block
            a.wt_t32 = dw32.p;
            a.wt_t32_bytes = (unsigned)(p32.size() * sizeof(__half));
        }
        // conv_w1d (ids 980..): the Winograd F(2, 3) transformed weights
        DevBuf<__half> dw1d;
        if (tile >= 980 && tile < 1000 && kh == 3 && kw == 3 && cin_pad % 32 == 0) {
            std::vector<__half> pw;
            pack_conv_weights_w1d(packed.data(), cout_pad, cin_pad, a.Kp, pw);
            dw1d.alloc(pw.size());
            RMR_HIP(hipMemcpyAsync(dw1d.p, pw.data(), pw.size() * sizeof(__half), hipMemcpyHostToDevice, ctx.stream));
            RMR_HIP(hipStreamSynchronize(ctx.stream));
            a.wt_w1d = dw1d.p;
            a.wt_w1d_bytes = (unsigned)(pw.size() * sizeof(__half));
        }
        // conv_t32f8 (ids 900..): e4m3 weights with one scale per output channel, the input quantised on the device
        DevBuf<unsigned char> dw8, dx8;
        DevBuf<float> dws;
        if (tile >= 900 && tile < 950 && kh == 3 && kw == 3) {
            std::vector<unsigned char> p8;
            std::vector<float> ws;
            pack_conv_weights_t32f8(packed.data(), cout_pad, cin_pad, a.Kp, p8, ws);
            dw8.alloc(p8.size());
            dws.alloc(ws.size());
            RMR_HIP(hipMemcpy(dw8.p, p8.data(), p8.size(), hipMemcpyHostToDevice));
            RMR_HIP(hipMemcpy(dws.p, ws.data(), ws.size() * sizeof(float), hipMemcpyHostToDevice));
            const int pitch = (cin_pad + 63) / 64 * 64;
            dx8.alloc(npx_in * pitch);
            launch_quant_f8(ctx, ctx.stream, dx.p, cin_pad, 0, cin_pad, dx8.p, pitch, (long)npx_in);
            a.in8 = dx8.p;
            a.in8_cs = pitch;
            a.in8_bytes = (unsigned)(npx_in * pitch);
            a.wt8 = dw8.p;
            a.wt8_bytes = (unsigned)p8.size();
            a.wscale = dws.p;
        }
        DevBuf<long long> dtiming;
        // per-phase cycle stamps of conv_dma / conv_direct: needs a build with -DRMR_CONV_TIMING_BUILD=1
        const bool want_timing = std::getenv("RMR_CONV_TIMING") != nullptr;
        if (want_timing) {
            dtiming.alloc(8);
            RMR_HIP(hipMemsetAsync(dtiming.p, 0, 64, ctx.stream));
            a.timing = dtiming.p;
        }
        // tile < 0: automatic; 0..99: conv_igemm tile; 100..199: conv_dma tile; 200..299: conv_halo tile; 300..399: conv_ws variant; 400..499: conv_direct tile; 500: conv_stem; 600..699: conv_ws_s2 variant; 700..799: conv_pw variant; 800..899: conv_t32 tile; 900..949: conv_t32f8 tile; 950..979: conv_g32 tile; 980..999: conv_w1d tile
        if (tile < 0) {
            launch_conv_auto(ctx, ctx.stream, a);
        } else if (tile >= kSbBase) {
            // kSbBase + variant: the small-batch family (conv_sb.hip)
            if (!conv_sb_supported(a, tile - kSbBase))
                fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv2d: small-batch variant %d cannot run this layer", tile - kSbBase);
            launch_conv_sb(ctx, ctx.stream, a, tile - kSbBase);
        } else if (tile >= 980 && tile < 1000) {
            const int t = tile - 980;
            if (t >= conv_w1d_num_tiles() || !conv_w1d_supported(a, t))
                fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv2d: w1d tile %d cannot run this layer", t);
            launch_conv_w1d(ctx, ctx.stream, a, t);
        } else if (tile >= 950 && tile < 980) {
            const int t = tile - 950;
            if (t >= conv_g32_num_tiles() || !conv_g32_supported(a, t))
                fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv2d: g32 tile %d cannot run this layer", t);
            launch_conv_g32(ctx, ctx.stream, a, t);
        } else if (tile >= 900 && tile < 950) {
            const int t = tile - 900;
            if (t >= conv_t32f8_num_tiles() || !conv_t32f8_supported(a, t))
                fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv2d: fp8 tile %d cannot run this layer", t);
            launch_conv_t32f8(ctx, ctx.stream, a, t);
        } else if (tile >= 800 && tile < 900) {
            const int t = tile - 800;
            if (t >= conv_t32_num_tiles() || !conv_t32_supported(a, t))
                fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv2d: t32 tile %d cannot run this layer", t);
            launch_conv_t32(ctx, ctx.stream, a, t);
        } else if (tile >= 1000 && tile % 1000 >= 800 && tile % 1000 < 900) {
            // 1000 * split + 800 + t32 tile: split-K conv_t32 through a private workspace
            const int split = tile / 1000, t = tile % 1000 - 800;
            if (t >= conv_t32_num_tiles() || !conv_t32_splitk_supported(a, t, split, ctx.num_cus))
                fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv2d: split-K t32 tile %d cannot run this layer", tile);
            DevBuf<float> ws;
            DevBuf<int> cnt;
            ws.alloc(conv_t32_splitk_ws_floats(a, t, split));
            cnt.alloc(conv_t32_splitk_tiles(a, t));
            RMR_HIP(hipMemsetAsync(cnt.p, 0, cnt.n * sizeof(int), ctx.stream));
            a.split = split;
            a.splitk_ws = ws.p;
            a.splitk_cnt = cnt.p;
            launch_conv_t32(ctx, ctx.stream, a, t);
            launch_conv_t32(ctx, ctx.stream, a, t);  // twice: the counters must re-arm themselves
            RMR_HIP(hipStreamSynchronize(ctx.stream));
        } else if (tile >= 1000) {
            // 1000 * split + 100 + dma tile: split-K through a private workspace
            const int split = tile / 1000, t = tile % 1000 - 100;
            if (t < 0 || t >= conv_dma_num_tiles() || cout_pad % conv_dma_tile(t).bn || !conv_dma_supported(a))
                fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv2d: split-K tile %d cannot run this layer", tile);
            DevBuf<float> ws;
            DevBuf<int> cnt;
            ws.alloc(conv_dma_splitk_ws_floats(a, t, split));
            cnt.alloc(conv_dma_splitk_tiles(a, t));
            RMR_HIP(hipMemsetAsync(cnt.p, 0, cnt.n * sizeof(int), ctx.stream));
            a.split = split;
            a.splitk_ws = ws.p;
            a.splitk_cnt = cnt.p;
            launch_conv_dma(ctx, ctx.stream, a, t);
            launch_conv_dma(ctx, ctx.stream, a, t);  // twice: the counters must re-arm themselves
            RMR_HIP(hipStreamSynchronize(ctx.stream));
        } else if (tile >= 700) {
            if (!conv_pw_supported(a, tile - 700))
                fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv2d: pointwise variant %d cannot run this layer", tile - 700);
            launch_conv_pw(ctx, ctx.stream, a, tile - 700);
        } else if (tile >= 600) {
            if (!conv_ws_s2_supported(a, tile - 600))
                fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv2d: stride-2 weights-stationary variant %d cannot run this layer", tile - 600);
            launch_conv_ws_s2(ctx, ctx.stream, a, tile - 600);
        } else if (tile == 500) {
            if (!conv_stem_supported(a)) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv2d: the stem kernel cannot run this layer");
            launch_conv_stem(ctx, ctx.stream, a);
        } else if (tile >= 400) {
            if (!conv_direct_supported(a, tile - 400))
                fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv2d: direct tile %d cannot run this layer", tile - 400);
            launch_conv_direct(ctx, ctx.stream, a, tile - 400);
        } else if (tile >= 340 && tile < 400) {
            // conv_wsf (340..): a whole bottleneck, out = x + SiLU(conv(SiLU(conv(x)))), here with ONE filter for both
            // convolutions (the test restates exactly that); `residual` is not used -- the shortcut is the input
            a.wt2 = a.wt;
            a.bias2 = a.bias;
            a.res = nullptr;
            if (!conv_wsf_supported(a, tile - 340))
                fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv2d: fused-bottleneck variant %d cannot run this layer", tile - 340);
            launch_conv_wsf(ctx, ctx.stream, a, tile - 340);
        } else if (tile >= 300) {
            if (!conv_ws_supported(a, tile - 300))
                fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv2d: weights-stationary variant %d cannot run this layer", tile - 300);
            launch_conv_ws(ctx, ctx.stream, a, tile - 300);
        } else if (tile >= 200) {
            const int t = tile - 200;
            if (t >= conv_halo_num_tiles() || !conv_halo_supported(a, t))
                fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv2d: halo tile %d cannot run this layer", t);
            launch_conv_halo(ctx, ctx.stream, a, t);
        } else if (tile >= 100) {
            const int t = tile - 100;
            if (t >= conv_dma_num_tiles() || cout_pad % conv_dma_tile(t).bn || !conv_dma_supported(a))
                fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv2d: dma tile %d cannot run this layer", t);
            launch_conv_dma(ctx, ctx.stream, a, t);
        } else {
            if (tile >= conv_num_tiles() || cout_pad % conv_tile(tile).bn)
                fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv2d: tile %d cannot produce %d channels", tile, cout_pad);
            launch_conv(ctx, ctx.stream, a, tile);
        }
        std::vector<float> hy(npx_out * cout_pad);
        if (out16) {
            std::vector<__half> hy16(hy.size());
            RMR_HIP(hipMemcpyAsync(hy16.data(), dy16.p, hy16.size() * sizeof(__half), hipMemcpyDeviceToHost, ctx.stream));
            RMR_HIP(hipStreamSynchronize(ctx.stream));
            for (size_t i = 0; i < hy.size(); ++i) hy[i] = __half2float(hy16[i]);
        } else {
            RMR_HIP(hipMemcpyAsync(hy.data(), dy.p, hy.size() * sizeof(float), hipMemcpyDeviceToHost, ctx.stream));
            RMR_HIP(hipStreamSynchronize(ctx.stream));
        }
        if (want_timing) {
            long long t[8] = {0};
            RMR_HIP(hipMemcpy(t, dtiming.p, 64, hipMemcpyDeviceToHost));
            const double n = t[5] ? (double)t[5] : 1.0;
            fprintf(stderr, "[conv timing] slices %lld | per slice cycles: vmcnt-wait %.0f barrier %.0f issue %.0f ds_read %.0f mfma %.0f\n",
                    t[5], t[0] / n, t[1] / n, t[2] / n, t[3] / n, t[4] / n);
        }
        for (size_t p = 0; p < npx_out; ++p)
            for (int c = 0; c < cout; ++c) y[p * cout + c] = hy[p * cout_pad + c];
    });
}

int rmr_tune_file_version(void) { return rmr::Yolov8::tune_file_version(); }

// Host e4m3 rounding of the weight packer (round to nearest even, OCP e4m3fn), for the CPU tests.
rmr_status rmr_f32_to_e4m3(const float* x, int n, unsigned char* out) {
    return guarded([&] {
        if (!x || !out || n < 0) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_f32_to_e4m3: bad arguments");
        for (int i = 0; i < n; ++i) out[i] = f32_to_e4m3(x[i]);
    });
}

// The device quantiser of the fp8 plan on host data: x[n] (rounded to f16 first, as activations are stored) ->
// e4m3 bytes, n a multiple of 16.  Parity hook for tests (must equal rmr_f32_to_e4m3 of the f16 values).
rmr_status rmr_quant_e4m3(int device, const float* x, int n, unsigned char* out) {
    return guarded([&] {
        if (!x || !out || n <= 0 || n % 16) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_quant_e4m3: n must be a positive multiple of 16");
        DeviceCtx& ctx = device_ctx(device);
        std::vector<__half> h(n);
        for (int i = 0; i < n; ++i) h[i] = __float2half(x[i]);
        DevBuf<__half> dx;
        DevBuf<unsigned char> dq;
        dx.alloc(n);
        dq.alloc(n);
        RMR_HIP(hipMemcpy(dx.p, h.data(), (size_t)n * 2, hipMemcpyHostToDevice));
        launch_quant_f8(ctx, ctx.stream, dx.p, 16, 0, 16, dq.p, 16, n / 16);
        RMR_HIP(hipStreamSynchronize(ctx.stream));
        RMR_HIP(hipMemcpy(out, dq.p, n, hipMemcpyDeviceToHost));
    });
}

// One layer on device-resident f16 data, timed with HIP events: the kernel-development loop (tools/conv_bench.py).
// x: random f16 NHWC (one image's worth replicated), f16 output, optional residual; `tile` as in rmr_conv2d
// (only the tiled families: 0..299, 800..999).  ms_out = mean launch time over `reps` launches.
rmr_status rmr_conv_bench(int device, int n, int h, int w, int cin, int cout, int k, int stride, int residual, int tile,
                          int reps, float* ms_out) {
    return guarded([&] {
        if (!ms_out || n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cin % 8 || cout <= 0 || cout % 16 || reps <= 0)
            fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv_bench: bad arguments");
        DeviceCtx& ctx = device_ctx(device);
        const int pad = k / 2;
        const int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
        const size_t img_in = (size_t)h * w * cin, img_out = (size_t)ho * wo * cout;
        if ((double)n * img_in * 2 > 3.7e9) fail(RMR_ERR_CAPACITY, "rmr_conv_bench: input view larger than 3.7 GB");
        std::vector<__half> hx(img_in), hw;
        unsigned seed = 12345u;
        const auto rnd = [&] {
            seed = seed * 1664525u + 1013904223u;
            return ((seed >> 8) & 0xffff) / 32768.0f - 1.0f;
        };
        // RMR_BENCH_DATA: 0 = uniform [-1, 1) (default), 1 = zeros, 2 = SiLU-like (what the layers of the
        // network see: SiLU of a unit normal-ish value, mostly small, never below -0.28)
        const int mode = std::getenv("RMR_BENCH_DATA") ? std::atoi(std::getenv("RMR_BENCH_DATA")) : 0;
        for (auto& v : hx) {
            float x = rnd();
            if (mode == 1) x = 0.f;
            if (mode == 2) {
                const float g = (rnd() + rnd() + x) * 1.0f;  // roughly normal, sigma ~ 1
                x = g / (1.0f + std::exp(-g));
            }
            v = __float2half(x);
        }
        std::vector<float> wf((size_t)cout * cin * k * k), b(cout);
        const float ws = 1.0f / std::sqrt((float)cin * k * k);
        for (auto& v : wf) v = rnd() * ws;
        for (auto& v : b) v = rnd() * 0.5f;
        ConvArgs a{};
        pack_conv_weights(wf.data(), cout, cin, k, k, cin, cout, hw, a.K, a.Kp);
        DevBuf<__half> dx, dw, dw32, dy, dr;
        DevBuf<float> db;
        dx.alloc(n * img_in);
        dw.alloc(hw.size());
        db.alloc(b.size());
        dy.alloc(n * img_out);
        RMR_HIP(hipMemcpy(dx.p, hx.data(), img_in * 2, hipMemcpyHostToDevice));
        for (int i = 1; i < n; ++i) RMR_HIP(hipMemcpyAsync(dx.p + i * img_in, dx.p, img_in * 2, hipMemcpyDeviceToDevice, ctx.stream));
        RMR_HIP(hipMemcpy(dw.p, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        RMR_HIP(hipMemcpy(db.p, b.data(), b.size() * 4, hipMemcpyHostToDevice));
        if (residual) {
            dr.alloc(n * img_out);
            RMR_HIP(hipMemcpyAsync(dr.p, dx.p, std::min(n * img_out, n * img_in) * 2, hipMemcpyDeviceToDevice, ctx.stream));
            a.res = dr.p;
            a.res_cs = cout;
        }
        if ((k == 3 || k == 1) && cin % 32 == 0) {
            std::vector<__half> p32;
            pack_conv_weights_t32(hw.data(), cout, cin, a.Kp, p32, k * k);
            dw32.alloc(p32.size());
            RMR_HIP(hipMemcpy(dw32.p, p32.data(), p32.size() * 2, hipMemcpyHostToDevice));
            a.wt_t32 = dw32.p;
            a.wt_t32_bytes = (unsigned)(p32.size() * 2);
        }
        DevBuf<__half> dw1d;
        if (k == 3 && tile >= 980 && tile < 1000 && cin % 32 == 0) {
            std::vector<__half> pw;
            pack_conv_weights_w1d(hw.data(), cout, cin, a.Kp, pw);
            dw1d.alloc(pw.size());
            RMR_HIP(hipMemcpy(dw1d.p, pw.data(), pw.size() * 2, hipMemcpyHostToDevice));
            a.wt_w1d = dw1d.p;
            a.wt_w1d_bytes = (unsigned)(pw.size() * 2);
        }
        DevBuf<unsigned char> dw8, dx8;
        DevBuf<float> dws;
        if (k == 3 && tile >= 900 && tile < 950) {
            std::vector<unsigned char> p8;
            std::vector<float> ws;
            pack_conv_weights_t32f8(hw.data(), cout, cin, a.Kp, p8, ws);
            dw8.alloc(p8.size());
            dws.alloc(ws.size());
            RMR_HIP(hipMemcpy(dw8.p, p8.data(), p8.size(), hipMemcpyHostToDevice));
            RMR_HIP(hipMemcpy(dws.p, ws.data(), ws.size() * sizeof(float), hipMemcpyHostToDevice));
            const int pitch = (cin + 63) / 64 * 64;
            dx8.alloc((size_t)n * h * w * pitch);
            launch_quant_f8(ctx, ctx.stream, dx.p, cin, 0, cin, dx8.p, pitch, (long)n * h * w);
            a.in8 = dx8.p;
            a.in8_cs = pitch;
            a.in8_bytes = (unsigned)((size_t)n * h * w * pitch);
            a.wt8 = dw8.p;
            a.wt8_bytes = (unsigned)p8.size();
            a.wscale = dws.p;
        }
        a.in = dx.p;
        a.in_cs = cin;
        a.N = n;
        a.H = h;
        a.W = w;
        a.Cin = cin;
        a.Ho = ho;
        a.Wo = wo;
        a.KH = a.KW = k;
        a.stride = stride;
        a.pad = pad;
        a.wt = dw.p;
        a.bias = db.p;
        a.out = dy.p;
        a.out_cs = cout;
        a.Cout_pad = cout;
        a.M = n * ho * wo;
        a.act = std::getenv("RMR_BENCH_ACT") ? std::atoi(std::getenv("RMR_BENCH_ACT")) : 1;
        a.in_bytes = (unsigned)(n * img_in * 2);
        a.wt_bytes = (unsigned)(hw.size() * 2);
        // 1000 * split + 800 + t32 tile: split-K conv_t32
        DevBuf<float> sk_ws;
        DevBuf<int> sk_cnt;
        if (tile < kSbBase && tile >= 1000 && tile % 1000 >= 800 && tile % 1000 < 900) {
            const int split = tile / 1000, t = tile % 1000 - 800;
            if (t >= conv_t32_num_tiles() || !conv_t32_splitk_supported(a, t, split, ctx.num_cus))
                fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv_bench: split-K t32 tile %d cannot run this layer", tile);
            sk_ws.alloc(conv_t32_splitk_ws_floats(a, t, split));
            sk_cnt.alloc(conv_t32_splitk_tiles(a, t));
            RMR_HIP(hipMemsetAsync(sk_cnt.p, 0, sk_cnt.n * sizeof(int), ctx.stream));
            a.split = split;
            a.splitk_ws = sk_ws.p;
            a.splitk_cnt = sk_cnt.p;
        }
#ifdef RMR_T32_FINISH
        if (tile >= 800 && tile < 900) {   // development build: two 64-bit stamps per workgroup (conv_t32.hip)
            sk_ws.alloc(4 * 2 * ctx.num_cus * 4);
            RMR_HIP(hipMemsetAsync(sk_ws.p, 0, sk_ws.n * sizeof(float), ctx.stream));
            a.splitk_ws = sk_ws.p;
        }
#endif
        const auto launch = [&] {
            if (tile >= kSbBase) {
                if (!conv_sb_supported(a, tile - kSbBase))
                    fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv_bench: small-batch variant %d cannot run this layer", tile - kSbBase);
                launch_conv_sb(ctx, ctx.stream, a, tile - kSbBase);
            } else if (tile >= 1000 && tile % 1000 >= 800 && tile % 1000 < 900) {
                launch_conv_t32(ctx, ctx.stream, a, tile % 1000 - 800);
            } else if (tile >= 980 && tile < 1000) {
                if (tile - 980 >= conv_w1d_num_tiles() || !conv_w1d_supported(a, tile - 980))
                    fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv_bench: w1d tile %d cannot run this layer", tile - 980);
                launch_conv_w1d(ctx, ctx.stream, a, tile - 980);
            } else if (tile >= 950 && tile < 980) {
                if (tile - 950 >= conv_g32_num_tiles() || !conv_g32_supported(a, tile - 950))
                    fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv_bench: g32 tile %d cannot run this layer", tile - 950);
                launch_conv_g32(ctx, ctx.stream, a, tile - 950);
            } else if (tile >= 900 && tile < 950) {
                if (tile - 900 >= conv_t32f8_num_tiles() || !conv_t32f8_supported(a, tile - 900))
                    fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv_bench: fp8 tile %d cannot run this layer", tile - 900);
                launch_conv_t32f8(ctx, ctx.stream, a, tile - 900);
            } else if (tile >= 800 && tile < 900) {
                if (tile - 800 >= conv_t32_num_tiles() || !conv_t32_supported(a, tile - 800))
                    fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv_bench: t32 tile %d cannot run this layer", tile - 800);
                launch_conv_t32(ctx, ctx.stream, a, tile - 800);
            } else if (tile >= 700 && tile < 800) {
                if (tile - 700 >= conv_pw_num_variants() || !conv_pw_supported(a, tile - 700))
                    fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv_bench: pw variant %d cannot run this layer", tile - 700);
                launch_conv_pw(ctx, ctx.stream, a, tile - 700);
            } else if (tile >= 600 && tile < 700) {
                if (tile - 600 >= conv_ws_s2_num_variants() || !conv_ws_s2_supported(a, tile - 600))
                    fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv_bench: ws_s2 variant %d cannot run this layer", tile - 600);
                launch_conv_ws_s2(ctx, ctx.stream, a, tile - 600);
            } else if (tile >= 340 && tile < 400) {   // the fused bottleneck: both convolutions (twice the FLOPs of the shape)
                ConvArgs f = a;
                f.wt2 = f.wt;
                f.bias2 = f.bias;
                f.res = nullptr;
                if (tile - 340 >= conv_wsf_num_variants() || !conv_wsf_supported(f, tile - 340))
                    fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv_bench: fused-bottleneck variant %d cannot run this layer", tile - 340);
                launch_conv_wsf(ctx, ctx.stream, f, tile - 340);
            } else if (tile >= 300 && tile < 400) {
                if (tile - 300 >= conv_ws_num_variants() || !conv_ws_supported(a, tile - 300))
                    fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv_bench: ws variant %d cannot run this layer", tile - 300);
                launch_conv_ws(ctx, ctx.stream, a, tile - 300);
            } else if (tile >= 200 && tile < 300) {
                if (tile - 200 >= conv_halo_num_tiles() || !conv_halo_supported(a, tile - 200))
                    fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv_bench: halo tile %d cannot run this layer", tile - 200);
                launch_conv_halo(ctx, ctx.stream, a, tile - 200);
            } else if (tile >= 100 && tile < 200) {
                if (tile - 100 >= conv_dma_num_tiles() || cout % conv_dma_tile(tile - 100).bn || !conv_dma_supported(a))
                    fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv_bench: dma tile %d cannot run this layer", tile - 100);
                launch_conv_dma(ctx, ctx.stream, a, tile - 100);
            } else if (tile >= 0 && tile < 100) {
                if (tile >= conv_num_tiles() || cout % conv_tile(tile).bn)
                    fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv_bench: tile %d cannot run this layer", tile);
                launch_conv(ctx, ctx.stream, a, tile);
            } else {
                fail(RMR_ERR_INVALID_ARGUMENT, "rmr_conv_bench: kernel id %d is not a tiled family", tile);
            }
        };
        // RMR_BENCH_COLD=<copies>: every launch works on its own replica of the input, the weights and the output (round robin),
        // as the layers of a network do -- back-to-back launches of ONE layer find their operands in the L2 of every XCD, which
        // a batch-1 frame never does (a layer's weights were last read a frame ago)
        const int copies = std::getenv("RMR_BENCH_COLD") ? std::max(1, std::atoi(std::getenv("RMR_BENCH_COLD"))) : 1;
        DevBuf<__half> cx, cw, cw32, cy;
        if (copies > 1) {
            cx.alloc(copies * dx.n), cw.alloc(copies * dw.n), cy.alloc(copies * dy.n);
            if (dw32.p) cw32.alloc(copies * dw32.n);
            for (int c = 0; c < copies; ++c) {
                RMR_HIP(hipMemcpyAsync(cx.p + c * dx.n, dx.p, dx.n * 2, hipMemcpyDeviceToDevice, ctx.stream));
                RMR_HIP(hipMemcpyAsync(cw.p + c * dw.n, dw.p, dw.n * 2, hipMemcpyDeviceToDevice, ctx.stream));
                if (dw32.p) RMR_HIP(hipMemcpyAsync(cw32.p + c * dw32.n, dw32.p, dw32.n * 2, hipMemcpyDeviceToDevice, ctx.stream));
            }
        }
        // development builds (-DRMR_SB_TIMING) with RMR_CONV_TIMING=1: eight 100 MHz stamps per workgroup of the LAST launch
        DevBuf<long long> sb_stamps;
        if (tile >= kSbBase && std::getenv("RMR_CONV_TIMING")) {
            sb_stamps.alloc(16 * 16384);
            RMR_HIP(hipMemsetAsync(sb_stamps.p, 0, sb_stamps.n * 8, ctx.stream));
            a.timing = sb_stamps.p;
        }
        int turn = 0;
        const auto launch_one = launch;
        const auto launch_cold = [&] {
            if (copies > 1) {
                const int c = turn++ % copies;
                a.in = cx.p + c * dx.n, a.wt = cw.p + c * dw.n, a.out = cy.p + c * dy.n;
                if (dw32.p) a.wt_t32 = cw32.p + c * dw32.n;
            }
            launch_one();
        };
        launch_cold();
        launch_cold();
        hipEvent_t e0, e1;
        RMR_HIP(hipEventCreate(&e0));
        RMR_HIP(hipEventCreate(&e1));
        RMR_HIP(hipEventRecord(e0, ctx.stream));
        for (int r = 0; r < reps; ++r) launch_cold();
        RMR_HIP(hipEventRecord(e1, ctx.stream));
        RMR_HIP(hipEventSynchronize(e1));
        float ms = 0;
        RMR_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *ms_out = ms / reps;
        if (sb_stamps.p) {
            std::vector<long long> st(sb_stamps.n);
            RMR_HIP(hipMemcpy(st.data(), sb_stamps.p, st.size() * 8, hipMemcpyDeviceToHost));
            long long t0 = -1;
            size_t wgs = 0;
            for (size_t i = 0; i + 15 < st.size(); i += 16)
                if (st[i]) t0 = t0 < 0 ? st[i] : std::min(t0, st[i]), ++wgs;
            static const char* const what[16] = {"entry", "DMAs issued", "stage 0 landed", "last stage landed", "K loop done", "reduced", "epilogue issued", "stores done",
                                                 "stage 0", "stage 1", "stage 2", "stage 3", "stage 0 waited for", "fragment constants", "epilogue preloads", "DMA tables"};
            for (int k = 0; k < 16 && wgs; ++k) {
                std::vector<double> v;
                for (size_t i = 0; i + 15 < st.size(); i += 16)
                    if (st[i] && st[i + k]) v.push_back((st[i + k] - t0) * 0.01);
                if (v.empty()) continue;
                std::sort(v.begin(), v.end());
                std::fprintf(stderr, "[sb stamps] %-18s min %6.2f p50 %6.2f max %6.2f us (%zu workgroups; launch %.1f us)\n", what[k], v.front(), v[v.size() / 2], v.back(),
                             v.size(), ms / reps * 1e3);
            }
        }
#ifdef RMR_T32_FINISH
        if (tile >= 800 && tile < 900) {   // the last launch's stamps: start / finish of every workgroup, in us from the first start
            std::vector<unsigned long long> st(sk_ws.n / 2);
            RMR_HIP(hipMemcpy(st.data(), sk_ws.p, st.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> b, e;
            unsigned long long t0 = ~0ull;
            for (size_t i = 0; i + 1 < st.size(); i += 2)
                if (st[i + 1]) t0 = std::min(t0, st[i]);
            for (size_t i = 0; i + 1 < st.size(); i += 2)
                if (st[i + 1]) b.push_back((st[i] - t0) * 0.01), e.push_back((st[i + 1] - t0) * 0.01);
            if (!e.empty()) {
                std::sort(e.begin(), e.end());
                std::sort(b.begin(), b.end());
                double mean = 0;
                for (double v : e) mean += v;
                mean /= e.size();
                const auto q = [&](const std::vector<double>& v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
                std::fprintf(stderr, "[t32 finish] %zu workgroups: start p50 %.1f max %.1f us | finish min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f mean %.1f us (launch %.1f us)\n",
                             e.size(), q(b, 0.5), b.back(), e.front(), q(e, 0.1), q(e, 0.5), q(e, 0.9), e.back(), mean, ms / reps * 1e3);
            }
        }
#endif
    });
}

}  // extern "C"
