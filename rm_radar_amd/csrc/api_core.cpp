// api_core.cpp -- C-ABI entry points (include/rmr.h) for geometry, the unit kernels, the
// Locator and the host-side Robot assembly.  Detector entry points live in api_detect.cpp.
#include "api_handles.h"
#include "common.h"
#include "locator.h"
#include "postprocess.h"
#include "preprocess.h"
#include "robot.h"

using namespace rmr;

extern "C" {

const char* rmr_last_error(void) { return last_error().c_str(); }
int rmr_abi_version(void) { return RMR_ABI_VERSION; }
int rmr_device_count(void) { return usable_device_count(); }

// ---- geometry ------------------------------------------------------------------------

rmr_status rmr_preparam_make(int in_w, int in_h, int out_w, int out_h, rmr_preparam* out) {
    return guarded([&] {
        if (!out || in_w <= 0 || in_h <= 0 || out_w <= 0 || out_h <= 0)
            fail(RMR_ERR_INVALID_ARGUMENT, "rmr_preparam_make: sizes must be positive");
        *out = make_preparam(in_w, in_h, out_w, out_h);
    });
}

rmr_status rmr_letterbox_geometry(const rmr_preparam* pp, int* resized_w, int* resized_h, int* top,
                                  int* left) {
    return guarded([&] {
        if (!pp || !resized_w || !resized_h || !top || !left)
            fail(RMR_ERR_INVALID_ARGUMENT, "rmr_letterbox_geometry: null argument");
        letterbox_geometry(*pp, *resized_w, *resized_h, *top, *left);
    });
}

static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (hi < v ? hi : v); }

// detector.cpp:258-268
rmr_status rmr_restore_detection(rmr_detection* d, const rmr_preparam* p) {
    return guarded([&] {
        if (!d || !p) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_restore_detection: null argument");
        d->x = clampf((d->x - p->dw) * p->ratio, 0.0f, p->width);
        d->y = clampf((d->y - p->dh) * p->ratio, 0.0f, p->height);
        d->width = clampf(d->width * p->ratio, 0.0f, p->width - d->x);
        d->height = clampf(d->height * p->ratio, 0.0f, p->height - d->y);
    });
}

// ---- unit kernels --------------------------------------------------------------------

namespace {

// Uploads host images (or passes device ones through) and builds device descriptors.
struct StagedImages {
    std::vector<DevBuf<uint8_t>> owned;
    DevBuf<LetterboxDesc> descs;
    std::vector<LetterboxDesc> host;

    void stage(hipStream_t s, const rmr_image* imgs, const int* crops, int n) {
        host.resize(n);
        for (int i = 0; i < n; ++i) {
            const rmr_image& im = imgs[i];
            if (!im.data || im.width <= 0 || im.height <= 0 || im.stride < im.width * 3)
                fail(RMR_ERR_INVALID_ARGUMENT, "image %d: bad data/size/stride", i);
            const uint8_t* dev = im.data;
            if (im.mem != RMR_MEM_DEVICE) {
                owned.emplace_back();
                owned.back().alloc((size_t)im.stride * im.height);
                RMR_HIP(hipMemcpyAsync(owned.back().p, im.data, (size_t)im.stride * im.height,
                                       hipMemcpyHostToDevice, s));
                dev = owned.back().p;
            }
            LetterboxDesc& d = host[i];
            d.src = dev;
            d.src_stride = im.stride;
            d.crop_x = crops ? crops[4 * i + 0] : 0;
            d.crop_y = crops ? crops[4 * i + 1] : 0;
            d.crop_w = crops ? crops[4 * i + 2] : im.width;
            d.crop_h = crops ? crops[4 * i + 3] : im.height;
            if (d.crop_x < 0 || d.crop_y < 0 || d.crop_w <= 0 || d.crop_h <= 0 ||
                d.crop_x + d.crop_w > im.width || d.crop_y + d.crop_h > im.height)
                fail(RMR_ERR_INVALID_ARGUMENT, "image %d: crop outside the image", i);
        }
    }
    void upload(hipStream_t s) {
        descs.alloc(host.size());
        RMR_HIP(hipMemcpyAsync(descs.p, host.data(), host.size() * sizeof(LetterboxDesc),
                               hipMemcpyHostToDevice, s));
    }
};

}  // namespace

rmr_status rmr_letterbox(int device, const rmr_image* imgs, const int* crops, int n, int resized_w,
                         int resized_h, int top, int left, int out_w, int out_h, int fill,
                         float scale, int fmt, void* out) {
    return guarded([&] {
        if (!imgs || !out || n <= 0 || out_w <= 0 || out_h <= 0)
            fail(RMR_ERR_INVALID_ARGUMENT, "rmr_letterbox: bad arguments");
        if (fmt != RMR_FMT_U8_HWC && fmt != RMR_FMT_F32_NCHW)
            fail(RMR_ERR_INVALID_ARGUMENT, "rmr_letterbox: unknown format %d", fmt);
        DeviceCtx& ctx = device_ctx(device);
        StagedImages st;
        st.stage(ctx.stream, imgs, crops, n);
        for (auto& d : st.host) {
            d.rw = resized_w;
            d.rh = resized_h;
            d.top = top;
            d.left = left;
        }
        st.upload(ctx.stream);
        const size_t elems = (size_t)n * out_w * out_h * 3;
        const size_t bytes = elems * (fmt == RMR_FMT_U8_HWC ? 1 : 4);
        DevBuf<uint8_t> dout;
        dout.alloc(bytes);
        launch_letterbox(ctx, ctx.stream, st.descs.p, n, out_w, out_h, fill, scale,
                         fmt == RMR_FMT_U8_HWC ? LB_U8_HWC : LB_F32_NCHW, dout.p);
        RMR_HIP(hipMemcpyAsync(out, dout.p, bytes, hipMemcpyDeviceToHost, ctx.stream));
        RMR_HIP(hipStreamSynchronize(ctx.stream));
    });
}

rmr_status rmr_preprocess(int device, const rmr_image* imgs, const int* crops, int n, int out_w,
                          int out_h, float* blob, rmr_preparam* pp) {
    return guarded([&] {
        if (!imgs || !blob || n <= 0 || out_w <= 0 || out_h <= 0)
            fail(RMR_ERR_INVALID_ARGUMENT, "rmr_preprocess: bad arguments");
        DeviceCtx& ctx = device_ctx(device);
        StagedImages st;
        st.stage(ctx.stream, imgs, crops, n);
        for (int i = 0; i < n; ++i) {
            LetterboxDesc& d = st.host[i];
            const rmr_preparam p = make_preparam(d.crop_w, d.crop_h, out_w, out_h);
            letterbox_geometry(p, d.rw, d.rh, d.top, d.left);
            if (pp) pp[i] = p;
        }
        st.upload(ctx.stream);
        const size_t bytes = (size_t)n * out_w * out_h * 3 * sizeof(float);
        DevBuf<uint8_t> dout;
        dout.alloc(bytes);
        launch_letterbox(ctx, ctx.stream, st.descs.p, n, out_w, out_h, 128, 1 / 255.f, LB_F32_NCHW, dout.p);
        RMR_HIP(hipMemcpyAsync(blob, dout.p, bytes, hipMemcpyDeviceToHost, ctx.stream));
        RMR_HIP(hipStreamSynchronize(ctx.stream));
    });
}

rmr_status rmr_postprocess(int device, const float* net_out, int n, int channels, int anchors,
                           int classes, float nms_thresh, float conf_thresh,
                           const rmr_preparam* pp, rmr_detection* out, int* counts, int cap) {
    return guarded([&] {
        if (!net_out || !pp || !out || !counts || n <= 0 || anchors <= 0 || classes <= 0 ||
            channels != 4 + classes || cap <= 0)
            fail(RMR_ERR_INVALID_ARGUMENT, "rmr_postprocess: bad arguments");
        DeviceCtx& ctx = device_ctx(device);
        DevBuf<float> dnet;
        DevBuf<rmr_preparam> dpp;
        DevBuf<uint8_t> scratch;
        DevBuf<rmr_detection> dout;
        DevBuf<int> dcnt;
        const size_t nelem = (size_t)n * channels * anchors;
        dnet.alloc(nelem);
        dpp.alloc(n);
        scratch.alloc(postprocess_scratch_bytes(n, anchors));
        dout.alloc((size_t)n * cap);
        dcnt.alloc(n);
        RMR_HIP(hipMemcpyAsync(dnet.p, net_out, nelem * sizeof(float), hipMemcpyHostToDevice, ctx.stream));
        RMR_HIP(hipMemcpyAsync(dpp.p, pp, n * sizeof(rmr_preparam), hipMemcpyHostToDevice, ctx.stream));
        launch_postprocess(ctx, ctx.stream, dnet.p, n, channels, anchors, classes, nms_thresh,
                           conf_thresh, dpp.p, scratch.p, dout.p, dcnt.p, cap);
        RMR_HIP(hipMemcpyAsync(counts, dcnt.p, n * sizeof(int), hipMemcpyDeviceToHost, ctx.stream));
        RMR_HIP(hipMemcpyAsync(out, dout.p, (size_t)n * cap * sizeof(rmr_detection), hipMemcpyDeviceToHost, ctx.stream));
        RMR_HIP(hipStreamSynchronize(ctx.stream));
        for (int i = 0; i < n; ++i)
            if (counts[i] > cap)
                fail(RMR_ERR_CAPACITY, "rmr_postprocess: image %d has %d detections, cap is %d", i, counts[i], cap);
    });
}

rmr_status rmr_transpose(int device, const float* src, float* dst, int rows, int cols) {
    return guarded([&] {
        if (!src || !dst || rows <= 0 || cols <= 0) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_transpose: bad arguments");
        DeviceCtx& ctx = device_ctx(device);
        DevBuf<float> a, b;
        a.alloc((size_t)rows * cols);
        b.alloc((size_t)rows * cols);
        RMR_HIP(hipMemcpyAsync(a.p, src, sizeof(float) * rows * cols, hipMemcpyHostToDevice, ctx.stream));
        launch_transpose(ctx.stream, a.p, b.p, rows, cols);
        RMR_HIP(hipMemcpyAsync(dst, b.p, sizeof(float) * rows * cols, hipMemcpyDeviceToHost, ctx.stream));
        RMR_HIP(hipStreamSynchronize(ctx.stream));
    });
}

// ---- robot assembly (host) -------------------------------------------------------------

rmr_status rmr_robot_set_detection(rmr_robot* r, const rmr_detection* car, const rmr_detection* armors,
                                   int n_armors) {
    return guarded([&] {
        if (!r || !car || (n_armors > 0 && !armors))
            fail(RMR_ERR_INVALID_ARGUMENT, "rmr_robot_set_detection: null argument");
        robot_set_detection(*r, *car, armors, n_armors);
    });
}

float rmr_compute_iou(const float rect_a[4], const float rect_b[4]) { return compute_iou(rect_a, rect_b); }

rmr_status rmr_group_robots(const rmr_robot* in, int n, float iou_thresh, rmr_robot* out, int* n_out) {
    return guarded([&] {
        if ((n > 0 && !in) || !out || !n_out) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_group_robots: null argument");
        auto v = group_robots(in, n, iou_thresh);
        for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
        *n_out = (int)v.size();
    });
}

// ---- Locator ---------------------------------------------------------------------------

// locator.h:59-65 defaults
void rmr_locator_cfg_default(rmr_locator_cfg* c) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->zoom_factor = 0.5f;
    c->queue_size = 3;
    c->min_depth_diff = 500;
    c->max_depth_diff = 4000;
    c->cluster_tolerance = 400;
    c->min_cluster_size = 8;
    c->max_cluster_size = 1000;
    c->max_distance = 29300;
    c->max_points = 262144;
    c->max_foreground = 32768;
    c->max_frames = 1;
}

rmr_status rmr_locator_create(const rmr_locator_cfg* cfg, rmr_locator** out) {
    return guarded([&] {
        if (!cfg || !out) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_locator_create: null argument");
        *out = new rmr_locator(*cfg);
    });
}

void rmr_locator_destroy(rmr_locator* loc) { delete loc; }

#define LOC_CALL(expr)                                                              \
    return guarded([&] {                                                            \
        if (!loc) fail(RMR_ERR_INVALID_ARGUMENT, "%s: null locator", __func__);     \
        expr;                                                                       \
    })

rmr_status rmr_locator_update(rmr_locator* loc, const float* xyz, int n, int stride_bytes, int mem) {
    LOC_CALL(loc->impl.update(xyz, n, stride_bytes, mem));
}
rmr_status rmr_locator_cluster(rmr_locator* loc) { LOC_CALL(loc->impl.cluster()); }
rmr_status rmr_locator_search(rmr_locator* loc, rmr_robot* robots, int n) {
    LOC_CALL(loc->impl.search(robots, n, -1));
}
rmr_status rmr_locator_keep(rmr_locator* loc, int frame) { LOC_CALL(loc->impl.keep(frame)); }
rmr_status rmr_locator_search_kept(rmr_locator* loc, int frame, rmr_robot* robots, int n) {
    LOC_CALL(if (frame < 0) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_locator_search_kept: frame < 0");
             loc->impl.search(robots, n, frame));
}
int rmr_locator_width(const rmr_locator* loc) { return loc ? loc->impl.width() : 0; }
int rmr_locator_height(const rmr_locator* loc) { return loc ? loc->impl.height() : 0; }
rmr_status rmr_locator_read_image(rmr_locator* loc, int which, float* host_out) {
    LOC_CALL(loc->impl.read_image(which, host_out));
}
rmr_status rmr_locator_write_image(rmr_locator* loc, int which, const float* host_in) {
    LOC_CALL(loc->impl.write_image(which, host_in));
}
rmr_status rmr_locator_search_batch(rmr_locator* loc, rmr_robot* robots, const int* counts, int n_frames, int cap) {
    LOC_CALL(loc->impl.search_batch(robots, counts, n_frames, cap));
}
rmr_status rmr_locator_update_cluster_batch(rmr_locator* loc, const float* const* clouds, const int* n_points,
                                            int stride_bytes, int mem, int n_frames) {
    LOC_CALL(if (!clouds || !n_points || n_frames <= 0) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_locator_update_cluster_batch: bad arguments");
             loc->impl.update_cluster_batch(clouds, n_points, stride_bytes, mem, n_frames));
}
rmr_status rmr_locator_state_bytes(const rmr_locator* loc, size_t* bytes) {
    LOC_CALL(if (!bytes) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_locator_state_bytes: null argument");
             *bytes = loc->impl.state_bytes());
}
rmr_status rmr_locator_save_state(rmr_locator* loc, void* host_out, size_t cap) {
    LOC_CALL(loc->impl.save_state(host_out, cap));
}
rmr_status rmr_locator_load_state(rmr_locator* loc, const void* host_in, size_t bytes) {
    LOC_CALL(loc->impl.load_state(host_in, bytes));
}
rmr_status rmr_locator_transform(const rmr_locator* loc, int which, const float in[3], float out[3]) {
    LOC_CALL(loc->impl.transform(which, in, out));
}
rmr_status rmr_locator_zoom(const rmr_locator* loc, const int rect[4], int out[4]) {
    LOC_CALL(loc->impl.zoom(rect, out));
}
rmr_status rmr_locator_foreground(rmr_locator* loc, float* xyz, int* pixel, int* cluster, int cap, int* n) {
    LOC_CALL(if (!n) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_locator_foreground: null n");
             loc->impl.foreground(xyz, pixel, cluster, cap, n));
}
int rmr_locator_num_clusters(rmr_locator* loc) {
    int v = -1;
    (void)guarded([&] {
        if (!loc) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_locator_num_clusters: null locator");
        v = loc->impl.num_clusters();
    });
    return v;
}

// ---- profiling ---------------------------------------------------------------------------

rmr_status rmr_profile_enable(int device, int on) {
    return guarded([&] {
        if (on < 0 || on > 2) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_profile_enable: level %d", on);
        device_ctx(device).prof.on = on;
    });
}
rmr_status rmr_profile_reset(int device) {
    return guarded([&] { device_ctx(device).prof.reset(); });
}
rmr_status rmr_profile_read(int device, rmr_kernel_stat* out, int cap, int* n) {
    return guarded([&] {
        if (!n) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_profile_read: null n");
        Profiler& p = device_ctx(device).prof;
        p.resolve();
        std::lock_guard<std::mutex> lk(p.mu);
        int i = 0;
        for (const auto& kv : p.stats) {
            if (out && i < cap) {
                std::memset(&out[i], 0, sizeof(out[i]));
                std::strncpy(out[i].name, kv.first.c_str(), sizeof(out[i].name) - 1);
                out[i].launches = kv.second.launches;
                out[i].total_ms = kv.second.total_ms;
                out[i].flops = kv.second.flops;
                out[i].bytes = kv.second.bytes;
            }
            ++i;
        }
        *n = i;
    });
}

}  // extern "C"
