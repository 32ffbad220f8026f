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
        std::thread locate([&] {
            try {
                for (int f = 0; f < n_frames; ++f) {
                    loc->impl.update(clouds[f], n_points[f], stride_bytes, mem);
                    loc->impl.cluster();
                    loc->impl.keep(f);
                }
            } catch (...) {
                locate_error = std::current_exception();
            }
        });
        // thread B (the caller): two-stage detect over all frames
        std::exception_ptr detect_error;
        try {
            rd->impl.detect_batch(imgs, n_frames, forced_crops, forced_per_frame, out, n_out, cap);
        } catch (...) {
            detect_error = std::current_exception();
        }
        locate.join();
        if (detect_error) std::rethrow_exception(detect_error);
        if (locate_error) std::rethrow_exception(locate_error);
        loc->impl.search_batch(out, n_out, n_frames, cap);
    });
}
