// tracker.h -- radar::Tracker / radar::TrackState (src/track/tracker.h:23-54, track.h:26) over the
// C-ABI.  Host code, as in the reference; the filters, the auction and the bookkeeping live in
// librmr.so (rm_radar_amd/csrc/tracker.cpp).
#pragma once
#include <chrono>
#include <stdexcept>
#include <vector>

#include "../rmr.h"
#include "robot.h"

namespace radar {

class Tracker {
   public:
    Tracker(const Point3f& observation_noise, int class_num, int init_thresh = 4, int miss_thresh = 10,
            float max_acceleration = 2.0f, float acceleration_correlation_time = 1.0f, float distance_weight = 0.40f,
            float feature_weight = 0.60f, int max_iter = 100, float distance_thresh = 0.8f) {
        rmr_tracker_cfg cfg;
        rmr_tracker_cfg_default(&cfg);
        cfg.observation_noise[0] = observation_noise.x, cfg.observation_noise[1] = observation_noise.y;
        cfg.observation_noise[2] = observation_noise.z;
        cfg.class_num = class_num, cfg.init_thresh = init_thresh, cfg.miss_thresh = miss_thresh;
        cfg.max_acceleration = max_acceleration, cfg.acceleration_correlation_time = acceleration_correlation_time;
        cfg.distance_weight = distance_weight, cfg.feature_weight = feature_weight;
        cfg.max_iter = max_iter, cfg.distance_thresh = distance_thresh;
        if (rmr_tracker_create(&cfg, &h_) != RMR_OK) throw std::invalid_argument(rmr_last_error());
    }
    ~Tracker() { rmr_tracker_destroy(h_); }
    Tracker(const Tracker&) = delete;
    Tracker& operator=(const Tracker&) = delete;

    // tracker.cpp:126-220
    void update(std::vector<Robot>& robots, const std::chrono::high_resolution_clock::time_point& timestamp) {
        std::vector<rmr_robot> c(robots.size());
        for (size_t i = 0; i < robots.size(); ++i) c[i] = robots[i].toC();
        const int64_t ns = std::chrono::duration_cast<std::chrono::nanoseconds>(timestamp.time_since_epoch()).count();
        if (rmr_tracker_update(h_, c.data(), (int)c.size(), ns) != RMR_OK) throw std::runtime_error(rmr_last_error());
        for (size_t i = 0; i < robots.size(); ++i) robots[i].fromC(c[i]);
    }

    rmr_tracker* handle() const noexcept { return h_; }

   private:
    rmr_tracker* h_ = nullptr;
};

}  // namespace radar
