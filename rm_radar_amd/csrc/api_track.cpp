// api_track.cpp -- C-ABI entry points of the tracker stage (include/rmr.h, "Tracker").
#include "tracker.h"

using namespace rmr;
using namespace rmr::track;

struct rmr_kalman {
    Kalman kf;
};
struct rmr_singer {
    Singer s;
};
struct rmr_tracker {
    Tracker t;
};

static void copy_state(const Mat& x, const Mat& P, float* xo, float* Po) {
    if (xo) std::copy(x.v.begin(), x.v.end(), xo);
    if (Po) std::copy(P.v.begin(), P.v.end(), Po);
}

extern "C" {

rmr_status rmr_kalman_create(int n, int m, const float* x0, const float* P0, const float* F, const float* Q,
                             const float* H, const float* R, rmr_kalman** out) {
    return guarded([&] {
        if (!out || !x0 || !P0 || !R || n <= 0 || m <= 0 || n > 64 || m > 64)
            fail(RMR_ERR_INVALID_ARGUMENT, "rmr_kalman_create: bad arguments");
        if ((F != nullptr) != (Q != nullptr) || (F != nullptr) != (H != nullptr))
            fail(RMR_ERR_INVALID_ARGUMENT, "rmr_kalman_create: F, Q and H come together or not at all");
        *out = new rmr_kalman{Kalman(n, m, x0, P0, F, Q, H, R)};
    });
}
void rmr_kalman_destroy(rmr_kalman* kf) { delete kf; }
rmr_status rmr_kalman_predict(rmr_kalman* kf) {
    return guarded([&] {
        if (!kf) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_kalman_predict: null filter");
        kf->kf.predict();
    });
}
rmr_status rmr_kalman_update(rmr_kalman* kf, const float* z) {
    return guarded([&] {
        if (!kf || !z) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_kalman_update: null argument");
        kf->kf.update(z);
    });
}
rmr_status rmr_kalman_predict_ekf(rmr_kalman* kf, const float* F, const float* Q) {
    return guarded([&] {
        if (!kf || !F || !Q) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_kalman_predict_ekf: null argument");
        kf->kf.predict_with(F, Q);
    });
}
rmr_status rmr_kalman_update_ekf(rmr_kalman* kf, const float* z, const float* hx, const float* H) {
    return guarded([&] {
        if (!kf || !z || !hx || !H) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_kalman_update_ekf: null argument");
        kf->kf.update_with(z, hx, H);
    });
}
rmr_status rmr_kalman_state(const rmr_kalman* kf, float* x, float* P) {
    return guarded([&] {
        if (!kf) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_kalman_state: null filter");
        copy_state(kf->kf.state(), kf->kf.covariance(), x, P);
    });
}

rmr_status rmr_singer_create(const float* x0, const float* P0, float max_a, float tau, const float* R,
                             rmr_singer** out) {
    return guarded([&] {
        if (!out || !x0 || !P0 || !R) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_singer_create: null argument");
        if (!(tau > 0.f)) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_singer_create: tau must be positive");
        *out = new rmr_singer{Singer(x0, P0, max_a, tau, R)};
    });
}
void rmr_singer_destroy(rmr_singer* s) { delete s; }
rmr_status rmr_singer_predict(rmr_singer* s, float dt) {
    return guarded([&] {
        if (!s) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_singer_predict: null filter");
        s->s.predict(dt);
    });
}
rmr_status rmr_singer_update(rmr_singer* s, const float* z) {
    return guarded([&] {
        if (!s || !z) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_singer_update: null argument");
        s->s.update(z);
    });
}
rmr_status rmr_singer_state(const rmr_singer* s, float* x, float* P) {
    return guarded([&] {
        if (!s) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_singer_state: null filter");
        copy_state(s->s.state(), s->s.covariance(), x, P);
    });
}

rmr_status rmr_auction(const float* values, int agents, int tasks, int max_iter, int* assignment) {
    return guarded([&] {
        if (agents < 0 || tasks < 0 || (agents > 0 && !assignment) || ((size_t)agents * tasks > 0 && !values))
            fail(RMR_ERR_INVALID_ARGUMENT, "rmr_auction: bad arguments");
        const std::vector<int> a = auction(values, agents, tasks, max_iter);
        std::copy(a.begin(), a.end(), assignment);
    });
}

rmr_status rmr_robot_feature(const rmr_robot* r, int class_num, float* out) {
    return guarded([&] {
        if (!r || !out || class_num <= 0) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_robot_feature: bad arguments");
        robot_feature(*r, class_num, out);
    });
}

void rmr_tracker_cfg_default(rmr_tracker_cfg* cfg) {
    if (!cfg) return;
    *cfg = rmr_tracker_cfg{};
    cfg->observation_noise[0] = cfg->observation_noise[1] = cfg->observation_noise[2] = 0.1f;
    cfg->class_num = 12;
    cfg->init_thresh = 4;
    cfg->miss_thresh = 10;
    cfg->max_acceleration = 2.0f;
    cfg->acceleration_correlation_time = 1.0f;
    cfg->distance_weight = 0.40f;
    cfg->feature_weight = 0.60f;
    cfg->max_iter = 100;
    cfg->distance_thresh = 0.8f;
}

rmr_status rmr_tracker_create(const rmr_tracker_cfg* cfg, rmr_tracker** out) {
    return guarded([&] {
        if (!cfg || !out) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_tracker_create: null argument");
        if (!(cfg->acceleration_correlation_time > 0.f))
            fail(RMR_ERR_INVALID_ARGUMENT, "rmr_tracker_create: acceleration_correlation_time must be positive");
        TrackerCfg c{};
        std::copy(cfg->observation_noise, cfg->observation_noise + 3, c.observation_noise);
        c.class_num = cfg->class_num, c.init_thresh = cfg->init_thresh, c.miss_thresh = cfg->miss_thresh;
        c.max_acceleration = cfg->max_acceleration;
        c.acceleration_correlation_time = cfg->acceleration_correlation_time;
        c.distance_weight = cfg->distance_weight, c.feature_weight = cfg->feature_weight;
        c.max_iter = cfg->max_iter, c.distance_thresh = cfg->distance_thresh;
        *out = new rmr_tracker{Tracker(c)};
    });
}
void rmr_tracker_destroy(rmr_tracker* t) { delete t; }

rmr_status rmr_tracker_update(rmr_tracker* t, rmr_robot* robots, int n, int64_t timestamp_ns) {
    return guarded([&] {
        if (!t) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_tracker_update: null tracker");
        t->t.update(robots, n, timestamp_ns);
    });
}

rmr_status rmr_tracker_tracks(const rmr_tracker* t, rmr_track_info* out, int cap, int* n) {
    return guarded([&] {
        if (!t || !n || cap < 0 || (cap > 0 && !out)) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_tracker_tracks: bad arguments");
        const auto& tr = t->t.tracks();
        *n = (int)tr.size();
        for (int i = 0; i < std::min(cap, *n); ++i) {
            rmr_track_info& o = out[i];
            o.id = tr[i].id, o.state = tr[i].state, o.label = tr[i].label();
            o.init_count = tr[i].init_count, o.miss_count = tr[i].miss_count;
            tr[i].location(o.location);
            std::copy(tr[i].filter.state().v.begin(), tr[i].filter.state().v.end(), o.state_vector);
        }
    });
}

}  // extern "C"
