// api_pipeline.cpp -- throughput mode of the whole path in one call: what SampleRadar::runOnce does
// per frame (samples/sample_radar.h:106-127: update + cluster on one thread while detect runs on
// another, join, search), over a batch of frames of one camera / LiDAR stream -- or of several streams that
// share the GPU and its detector, each with its own Locator.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <functional>
#include <thread>
#include <vector>

#include "api_handles.h"
#include "common.h"

using namespace rmr;

// n_streams streams, frames stream-major: stream s owns frames [s * per, (s + 1) * per), per = n_frames / n_streams
static void run_streams(rmr_robot_detector* rd, rmr_locator* const* locs, int n_streams, const rmr_image* imgs,
                        const float* const* clouds, const int* n_points, int stride_bytes, int mem, int n_frames,
                        const int* forced_crops, int forced_per_frame, rmr_robot* out, int* n_out, int cap) {
    if (!rd || !locs || !imgs || !clouds || !n_points || !out || !n_out || n_frames <= 0 || cap <= 0 || n_streams <= 0)
        fail(RMR_ERR_INVALID_ARGUMENT, "rmr_pipeline_run: bad arguments");
    if (n_frames % n_streams) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_pipeline_run_streams: %d frames do not divide into %d streams", n_frames, n_streams);
    for (int s = 0; s < n_streams; ++s)
        if (!locs[s]) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_pipeline_run: null locator");
    const int per = n_frames / n_streams;
    // RMR_STEP_TIMING=1: host-side timeline of a call on stderr (where the time between two steps goes)
    static const bool timing = std::getenv("RMR_STEP_TIMING") != nullptr;
    static std::chrono::steady_clock::time_point last_exit;
    const auto t_in = std::chrono::steady_clock::now();
    const auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count();
    };
    std::chrono::steady_clock::time_point t_cars = t_in, t_cars_done = t_in, t_det = t_in, t_search = t_in, t_car_enq = t_in, t_loc_enq = t_in;
    // threads A: a Locator carries temporal state, so the frames of a stream go in order, one helper thread per
    // stream (each Locator enqueues on its own HIP stream); a frame's foreground list is kept in slot f of its
    // stream for the batched search
    std::vector<std::exception_ptr> locate_error(n_streams);
    auto locate_stream = [&](int s) {
        try {
            // update + cluster + keep(f) of the stream's frames, the cluster stage batched over them (same results)
            locs[s]->impl.update_cluster_batch(clouds + (size_t)s * per, n_points + (size_t)s * per, stride_bytes, mem, per);
        } catch (...) {
            locate_error[s] = std::current_exception();
        }
    };
    std::vector<std::thread> locate;
    if (n_frames > 1)
        for (int s = 0; s < n_streams; ++s) locate.emplace_back(locate_stream, s);
    // (one frame: a few launches, cheaper to enqueue on this thread than to start one -- but not in FRONT of the detector:
    // detect_batch runs it once the car stage is in flight, so its ~45 us of host time travel under the car network)
    const std::function<void()> locate_one_frame = [&] {
        if (timing) t_car_enq = std::chrono::steady_clock::now();
        locate_stream(0);
        if (timing) t_loc_enq = std::chrono::steady_clock::now();
    };
    const auto join_all = [&] {
        for (auto& t : locate)
            if (t.joinable()) t.join();
    };
    const auto any_locate_error = [&]() -> std::exception_ptr {
        for (auto& e : locate_error)
            if (e) return e;
        return nullptr;
    };
    // thread B (the caller): two-stage detect over all frames of all streams.  Once the car boxes are known and
    // the armor stage is in flight, the searches are enqueued behind the locate work -- they need the boxes
    // only -- so they run under the armor stage instead of after it, and so does the host time of enqueueing them.
    const int stride = cap;
    std::vector<rmr_robot> car_robots((size_t)n_frames * stride);
    std::vector<int> car_counts(n_frames, 0), car_index((size_t)n_frames * cap, -1);
    bool searching = false;
    auto after_cars = [&](const std::vector<std::vector<rmr_detection>>& cars) {
        t_cars = std::chrono::steady_clock::now();
        join_all();
        if (any_locate_error()) return;
        for (int f = 0; f < n_frames; ++f) {
            car_counts[f] = std::min((int)cars[f].size(), stride);
            for (int i = 0; i < car_counts[f]; ++i) {
                rmr_robot& r = car_robots[(size_t)f * stride + i];
                r.rect[0] = cars[f][i].x, r.rect[1] = cars[f][i].y, r.rect[2] = cars[f][i].width, r.rect[3] = cars[f][i].height;
            }
        }
        for (int s = 0; s < n_streams; ++s)
            locs[s]->impl.search_batch_begin(car_robots.data() + (size_t)s * per * stride, car_counts.data() + s * per, per, stride);
        searching = true;
        t_cars_done = std::chrono::steady_clock::now();
    };
    std::exception_ptr detect_error;
    try {
        rd->impl.detect_batch(imgs, n_frames, forced_crops, forced_per_frame, out, n_out, cap, after_cars, car_index.data(),
                              n_frames > 1 ? std::function<void()>() : locate_one_frame);
    } catch (...) {
        detect_error = std::current_exception();
    }
    t_det = std::chrono::steady_clock::now();
    join_all();
    if (detect_error) std::rethrow_exception(detect_error);
    if (auto e = any_locate_error()) std::rethrow_exception(e);
    if (searching)
        for (int s = 0; s < n_streams; ++s)
            locs[s]->impl.search_batch_end(car_robots.data() + (size_t)s * per * stride, car_counts.data() + s * per, per, stride);
    t_search = std::chrono::steady_clock::now();
    for (int f = 0; f < n_frames; ++f)
        for (int i = 0; i < std::min(n_out[f], cap); ++i) {
            const int c = car_index[(size_t)f * cap + i];
            if (c < 0 || c >= car_counts[f]) continue;  // a car beyond the caller's cap: not searched
            const rmr_robot& src = car_robots[(size_t)f * stride + c];
            rmr_robot& dst = out[(size_t)f * cap + i];
            if (src.has_location) dst.has_location = 1, std::copy(src.location, src.location + 3, dst.location);
        }
    if (timing) {
        const auto t_out = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[rmr step] since last exit %ld us | entry -> cars known, armor stage enqueued %ld | after_cars %ld | rest of the armor stage + assembly %ld | search end %ld | merge %ld",
                     us(last_exit, t_in), us(t_in, t_cars), us(t_cars, t_cars_done), us(t_cars_done, t_det), us(t_det, t_search), us(t_search, t_out));
        if (n_frames == 1)   // one frame: the locator's enqueue runs inside the detector's call, under the car stage
            std::fprintf(stderr, " | of the first: frame staged + car stage enqueued %ld, locator enqueued %ld, wait + heads + armor enqueue %ld", us(t_in, t_car_enq),
                         us(t_car_enq, t_loc_enq), us(t_loc_enq, t_cars));
        std::fprintf(stderr, "\n");
        last_exit = t_out;
    }
}

extern "C" rmr_status rmr_pipeline_run_batch(rmr_robot_detector* rd, rmr_locator* loc, const rmr_image* imgs,
                                             const float* const* clouds, const int* n_points, int stride_bytes,
                                             int mem, int n_frames, const int* forced_crops, int forced_per_frame,
                                             rmr_robot* out, int* n_out, int cap) {
    return guarded([&] {
        run_streams(rd, &loc, 1, imgs, clouds, n_points, stride_bytes, mem, n_frames, forced_crops, forced_per_frame, out, n_out, cap);
    });
}

// Several camera / LiDAR streams on one GPU: one detector batch over the frames of all of them (detection is
// stateless), one Locator per stream (its background image and depth ring are that stream's history).
extern "C" rmr_status rmr_pipeline_run_streams(rmr_robot_detector* rd, rmr_locator* const* locs, int n_streams,
                                               const rmr_image* imgs, const float* const* clouds, const int* n_points,
                                               int stride_bytes, int mem, int n_frames, const int* forced_crops,
                                               int forced_per_frame, rmr_robot* out, int* n_out, int cap) {
    return guarded([&] {
        run_streams(rd, locs, n_streams, imgs, clouds, n_points, stride_bytes, mem, n_frames, forced_crops, forced_per_frame, out, n_out, cap);
    });
}
