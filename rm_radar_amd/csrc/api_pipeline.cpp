// api_pipeline.cpp -- throughput mode of the whole path in one call: what SampleRadar::runOnce does
// per frame (samples/sample_radar.h:106-127: update + cluster on one thread while detect runs on
// another, join, search), over a batch of frames of ONE camera / LiDAR stream.
#include <exception>
#include <thread>

#include "api_handles.h"
#include "common.h"

using namespace rmr;

extern "C" rmr_status rmr_pipeline_run_batch(rmr_robot_detector* rd, rmr_locator* loc, const rmr_image* imgs,
                                             const float* const* clouds, const int* n_points, int stride_bytes,
                                             int mem, int n_frames, const int* forced_crops, int forced_per_frame,
                                             rmr_robot* out, int* n_out, int cap) {
    return guarded([&] {
        if (!rd || !loc || !imgs || !clouds || !n_points || !out || !n_out || n_frames <= 0 || cap <= 0)
            fail(RMR_ERR_INVALID_ARGUMENT, "rmr_pipeline_run_batch: bad arguments");
        // thread A: the Locator carries temporal state, so its frames go in stream order; each
        // frame's foreground list is kept in slot f for the batched search
        std::exception_ptr locate_error;
        auto locate_all = [&] {
            try {
                for (int f = 0; f < n_frames; ++f) {
                    loc->impl.update(clouds[f], n_points[f], stride_bytes, mem);
                    loc->impl.cluster();
                    loc->impl.keep(f);
                }
            } catch (...) {
                locate_error = std::current_exception();
            }
        };
        std::thread locate;
        if (n_frames > 1)
            locate = std::thread(locate_all);
        else
            locate_all();  // one frame: a few launches, cheaper to enqueue here than to start a thread
        // thread B (the caller): two-stage detect over all frames.  As soon as the car boxes are
        // known the search is enqueued behind the locate work -- it needs the boxes only -- so it runs
        // under the armor stage instead of after it.
        const int stride = cap;
        std::vector<rmr_robot> car_robots((size_t)n_frames * stride);
        std::vector<int> car_counts(n_frames, 0), car_index((size_t)n_frames * cap, -1);
        bool searching = false;
        auto after_cars = [&](const std::vector<std::vector<rmr_detection>>& cars) {
            if (locate.joinable()) locate.join();
            if (locate_error) return;
            for (int f = 0; f < n_frames; ++f) {
                car_counts[f] = std::min((int)cars[f].size(), stride);
                for (int i = 0; i < car_counts[f]; ++i) {
                    rmr_robot& r = car_robots[(size_t)f * stride + i];
                    r.rect[0] = cars[f][i].x, r.rect[1] = cars[f][i].y, r.rect[2] = cars[f][i].width, r.rect[3] = cars[f][i].height;
                }
            }
            loc->impl.search_batch_begin(car_robots.data(), car_counts.data(), n_frames, stride);
            searching = true;
        };
        std::exception_ptr detect_error;
        try {
            rd->impl.detect_batch(imgs, n_frames, forced_crops, forced_per_frame, out, n_out, cap, after_cars,
                                  car_index.data());
        } catch (...) {
            detect_error = std::current_exception();
        }
        if (locate.joinable()) locate.join();
        if (detect_error) std::rethrow_exception(detect_error);
        if (locate_error) std::rethrow_exception(locate_error);
        if (searching) loc->impl.search_batch_end(car_robots.data(), car_counts.data(), n_frames, stride);
        for (int f = 0; f < n_frames; ++f)
            for (int i = 0; i < std::min(n_out[f], cap); ++i) {
                const int c = car_index[(size_t)f * cap + i];
                if (c < 0 || c >= car_counts[f]) continue;  // a car beyond the caller's cap: not searched
                const rmr_robot& src = car_robots[(size_t)f * stride + c];
                rmr_robot& dst = out[(size_t)f * cap + i];
                if (src.has_location) dst.has_location = 1, std::copy(src.location, src.location + 3, dst.location);
            }
    });
}
