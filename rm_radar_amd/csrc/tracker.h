// tracker.h -- the downstream tracker: Kalman / extended Kalman filter, the Singer
// acceleration model, the auction assignment and the track bookkeeping of radar::Tracker
// (src/track/*, src/robot/robot.cpp:81-122).  Host code only: at most a few dozen 9-state filters
// per frame -- there is nothing here for a GPU to do.
#pragma once
#include <cstdint>
#include <vector>

#include "common.h"

namespace rmr {
namespace track {

// Row-major float matrix with run-time size (n <= 16 in every use here).
struct Mat {
    int rows = 0, cols = 0;
    std::vector<float> v;
    Mat() = default;
    Mat(int r, int c) : rows(r), cols(c), v((size_t)r * c, 0.f) {}
    Mat(int r, int c, const float* src) : rows(r), cols(c), v(src, src + (size_t)r * c) {}
    float& operator()(int r, int c) { return v[(size_t)r * cols + c]; }
    float operator()(int r, int c) const { return v[(size_t)r * cols + c]; }
    static Mat identity(int n);
};
Mat mul(const Mat& a, const Mat& b);
Mat mul_bt(const Mat& a, const Mat& b);  // a * b^T
Mat add(const Mat& a, const Mat& b);
Mat sub(const Mat& a, const Mat& b);
Mat inverse(const Mat& a);               // closed form up to 3x3, Gauss-Jordan above

// KalmanFilter / ExtendedKalmanFilter (kalman_filter.h:77-296): one object serves both -- the
// extended filter is the same algebra with the transition / observation supplied per call.
class Kalman {
   public:
    Kalman(int n, int m, const float* x0, const float* P0, const float* F, const float* Q, const float* H,
           const float* R);
    int n() const { return n_; }
    int m() const { return m_; }
    void predict();                                                  // kalman_filter.h:116-121
    void update(const float* z);                                     // kalman_filter.h:129-152
    void predict_with(const float* F, const float* Q);               // kalman_filter.h:221-232
    void update_with(const float* z, const float* hx, const float* H);  // kalman_filter.h:243-248, 274-293
    const Mat& state() const { return x_; }
    const Mat& covariance() const { return P_; }

   private:
    void correct(const Mat& residual);
    int n_, m_;
    Mat x_, P_, F_, Q_, H_, R_;
    bool has_model_;
};

// SingerEKF (singer.h:33-132): state [x vx ax y vy ay z vz az], measurement [x y z]
class Singer {
   public:
    Singer(const float* x0, const float* P0, float max_a, float tau, const float* R);
    void predict(float dt);
    void update(const float* z);
    const Mat& state() const { return kf_.state(); }
    const Mat& covariance() const { return kf_.covariance(); }

   private:
    Kalman kf_;
    float max_a_, tau_;
};

// auction(value_matrix, max_iter) (auction.h:49-127): rows = agents, cols = tasks, -1 = unmatched
std::vector<int> auction(const float* values, int agents, int tasks, int max_iter);

// Robot::feature (robot.cpp:102-122)
void robot_feature(const rmr_robot& r, int class_num, float* out);

enum TrackState { kNone = 0, kTentative = 1, kConfirmed = 2, kDeleted = 3 };  // rmr.h RMR_TRACK_*

// Track (track.h:36-197) with the Features ring (features.h:30-209) reduced to what it is used
// for: per-class sums over everything pushed so far.
struct Track {
    Track(const float loc[3], const std::vector<float>& feature, int64_t t_ns, int id, float max_a, float tau,
          const float noise[3]);
    void predict(int64_t t_ns);
    void update(const float loc[3], const std::vector<float>& feature);
    int label() const;                       // features.h:178-183
    std::vector<float> feature() const;      // features.h:190-199
    void location(float out[3]) const;       // track.h:170-173
    std::vector<float> sums;
    int64_t t_ns;
    int id, init_count = 0, miss_count = 0;
    TrackState state = kTentative;
    Singer filter;
};

struct TrackerCfg {
    float observation_noise[3];
    int class_num, init_thresh, miss_thresh;
    float max_acceleration, acceleration_correlation_time, distance_weight, feature_weight;
    int max_iter;
    float distance_thresh;
};

// Tracker (tracker.h:23-54, tracker.cpp:85-220)
class Tracker {
   public:
    explicit Tracker(const TrackerCfg& cfg);
    void update(rmr_robot* robots, int n, int64_t t_ns);
    const std::vector<Track>& tracks() const { return tracks_; }
    float cost(const Track& t, const rmr_robot& r) const;  // tracker.cpp:85-119

   private:
    TrackerCfg cfg_;
    std::vector<Track> tracks_;
    int latest_id_ = 0;
};

}  // namespace track
}  // namespace rmr
