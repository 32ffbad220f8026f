// tracker.cpp -- see tracker.h.  Arithmetic is f32 throughout, as in the reference (Eigen float
// matrices); where the reference's expressions promote to double (std::pow with an int exponent,
// the nanosecond -> second conversion) the same promotions are made here.
#include "tracker.h"

#include <algorithm>
#include <cmath>
#include <limits>

namespace rmr {
namespace track {

Mat Mat::identity(int n) {
    Mat m(n, n);
    for (int i = 0; i < n; ++i) m(i, i) = 1.f;
    return m;
}

Mat mul(const Mat& a, const Mat& b) {
    Mat c(a.rows, b.cols);
    for (int i = 0; i < a.rows; ++i)
        for (int j = 0; j < b.cols; ++j) {
            float s = 0.f;
            for (int k = 0; k < a.cols; ++k) s += a(i, k) * b(k, j);
            c(i, j) = s;
        }
    return c;
}

Mat mul_bt(const Mat& a, const Mat& b) {
    Mat c(a.rows, b.rows);
    for (int i = 0; i < a.rows; ++i)
        for (int j = 0; j < b.rows; ++j) {
            float s = 0.f;
            for (int k = 0; k < a.cols; ++k) s += a(i, k) * b(j, k);
            c(i, j) = s;
        }
    return c;
}

Mat add(const Mat& a, const Mat& b) {
    Mat c(a.rows, a.cols);
    for (size_t i = 0; i < c.v.size(); ++i) c.v[i] = a.v[i] + b.v[i];
    return c;
}

Mat sub(const Mat& a, const Mat& b) {
    Mat c(a.rows, a.cols);
    for (size_t i = 0; i < c.v.size(); ++i) c.v[i] = a.v[i] - b.v[i];
    return c;
}

Mat inverse(const Mat& a) {
    const int n = a.rows;
    if (n != a.cols || n == 0) fail(RMR_ERR_LOGIC, "inverse: matrix is %d x %d", a.rows, a.cols);
    Mat r(n, n);
    if (n == 1) {
        r(0, 0) = 1.f / a(0, 0);
        return r;
    }
    if (n == 2) {
        const float inv = 1.f / (a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0));
        r(0, 0) = a(1, 1) * inv, r(0, 1) = -a(0, 1) * inv;
        r(1, 0) = -a(1, 0) * inv, r(1, 1) = a(0, 0) * inv;
        return r;
    }
    if (n == 3) {  // adjugate / determinant
        const float c00 = a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1);
        const float c01 = a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2);
        const float c02 = a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0);
        const float inv = 1.f / (a(0, 0) * c00 + a(0, 1) * c01 + a(0, 2) * c02);
        r(0, 0) = c00 * inv, r(1, 0) = c01 * inv, r(2, 0) = c02 * inv;
        r(0, 1) = (a(0, 2) * a(2, 1) - a(0, 1) * a(2, 2)) * inv;
        r(1, 1) = (a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0)) * inv;
        r(2, 1) = (a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1)) * inv;
        r(0, 2) = (a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1)) * inv;
        r(1, 2) = (a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2)) * inv;
        r(2, 2) = (a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0)) * inv;
        return r;
    }
    Mat w = a;
    r = Mat::identity(n);
    for (int c = 0; c < n; ++c) {  // Gauss-Jordan with partial pivoting
        int p = c;
        for (int i = c + 1; i < n; ++i)
            if (std::fabs(w(i, c)) > std::fabs(w(p, c))) p = i;
        if (p != c)
            for (int j = 0; j < n; ++j) std::swap(w(c, j), w(p, j)), std::swap(r(c, j), r(p, j));
        const float inv = 1.f / w(c, c);
        for (int j = 0; j < n; ++j) w(c, j) *= inv, r(c, j) *= inv;
        for (int i = 0; i < n; ++i) {
            if (i == c) continue;
            const float f = w(i, c);
            if (f == 0.f) continue;
            for (int j = 0; j < n; ++j) w(i, j) -= f * w(c, j), r(i, j) -= f * r(c, j);
        }
    }
    return r;
}

// ------------------------------------------------------------------------------ Kalman

Kalman::Kalman(int n, int m, const float* x0, const float* P0, const float* F, const float* Q, const float* H,
               const float* R)
    : n_(n), m_(m), x_(n, 1, x0), P_(n, n, P0), R_(m, m, R), has_model_(F && Q && H) {
    if (has_model_) F_ = Mat(n, n, F), Q_ = Mat(n, n, Q), H_ = Mat(m, n, H);
}

void Kalman::predict() {
    if (F_.rows != n_ || Q_.rows != n_) fail(RMR_ERR_LOGIC, "Kalman::predict: no transition model set");
    x_ = mul(F_, x_);
    P_ = add(mul_bt(mul(F_, P_), F_), Q_);
}

void Kalman::correct(const Mat& residual) {
    const Mat PHt = mul_bt(P_, H_);
    const Mat S = add(mul(H_, PHt), R_);
    const Mat K = mul(PHt, inverse(S));
    x_ = add(x_, mul(K, residual));
    P_ = mul(sub(Mat::identity(n_), mul(K, H_)), P_);
}

void Kalman::update(const float* z) {
    if (H_.rows != m_) fail(RMR_ERR_LOGIC, "Kalman::update: no observation model set");
    correct(sub(Mat(m_, 1, z), mul(H_, x_)));
}

void Kalman::predict_with(const float* F, const float* Q) {
    F_ = Mat(n_, n_, F);
    Q_ = Mat(n_, n_, Q);
    predict();
}

void Kalman::update_with(const float* z, const float* hx, const float* H) {
    H_ = Mat(m_, n_, H);
    correct(sub(Mat(m_, 1, z), Mat(m_, 1, hx)));
}

// ------------------------------------------------------------------------------ Singer model

Singer::Singer(const float* x0, const float* P0, float max_a, float tau, const float* R)
    : kf_(9, 3, x0, P0, nullptr, nullptr, nullptr, R), max_a_(max_a), tau_(tau) {}

void Singer::predict(float dt) {
    Mat F = Mat::identity(9), Q(9, 9);
    const float decay = std::exp(-dt / tau_);
    const float q = (float)std::pow((double)max_a_, 2);
    for (int i = 0; i < 3; ++i) {
        const int b = 3 * i;
        F(b, b + 1) = dt;
        F(b, b + 2) = dt * dt / 2;
        F(b + 1, b + 2) = dt;
        F(b + 2, b + 2) = decay;
        // singer.h:108-122; std::pow(float, int) is evaluated in double
        Q(b, b) = (float)(std::pow((double)dt, 3) / 3);
        Q(b + 1, b) = Q(b, b + 1) = (float)(std::pow((double)dt, 2) / 2);
        Q(b + 2, b) = Q(b, b + 2) = dt / 2;
        Q(b + 1, b + 1) = dt;
        Q(b + 2, b + 1) = Q(b + 1, b + 2) = 1 - decay;
        Q(b + 2, b + 2) = (1 - std::exp(-2 * dt / tau_)) / 2;
    }
    for (float& e : Q.v) e *= q;
    kf_.predict_with(F.v.data(), Q.v.data());
}

void Singer::update(const float* z) {
    float H[27] = {0}, hx[3];
    const Mat& x = kf_.state();
    for (int i = 0; i < 3; ++i) hx[i] = x(3 * i, 0), H[i * 9 + 3 * i] = 1.f;
    kf_.update_with(z, hx, H);
}

// ------------------------------------------------------------------------------ auction

// Each unassigned agent takes the task with the largest value - price, raises that task's price
// by the margin it saw and evicts the previous holder (auction.h:83-118).  With more agents than
// tasks zero-valued virtual tasks square the problem; they come back as -1.
std::vector<int> auction(const float* values, int agents, int tasks, int max_iter) {
    const int real_tasks = tasks;
    const int total = agents > tasks ? agents : tasks;
    std::vector<float> val((size_t)agents * total, 0.f);
    for (int a = 0; a < agents; ++a)
        for (int t = 0; t < real_tasks; ++t) val[(size_t)a * total + t] = values[(size_t)a * real_tasks + t];
    std::vector<float> price(total, 0.f);
    std::vector<int> owner_of(agents, -1);
    for (int it = 0; it < max_iter; ++it) {
        int settled = 0;
        for (int v : owner_of) settled += v >= 0 && v <= real_tasks;  // auction.h:72-74, '<=' as written
        if (settled >= agents) break;
        bool changed = false;
        for (int a = 0; a < agents; ++a) {
            if (owner_of[a] != -1) continue;
            int best = -1;
            float margin = -std::numeric_limits<float>::infinity();
            for (int t = 0; t < total; ++t) {
                const float m = val[(size_t)a * total + t] - price[t];
                if (m > margin) margin = m, best = t;
            }
            if (best < 0) continue;
            price[best] += margin;
            for (int o = 0; o < agents; ++o)
                if (owner_of[o] == best) {
                    owner_of[o] = -1;
                    break;
                }
            owner_of[a] = best;
            changed = true;
        }
        if (!changed) break;
    }
    for (int& v : owner_of)
        if (v >= real_tasks) v = -1;
    return owner_of;
}

// ------------------------------------------------------------------------------ Robot / Track

void robot_feature(const rmr_robot& r, int class_num, float* out) {
    std::fill(out, out + class_num, 0.f);
    if (r.n_armors <= 0) return;  // not detected
    for (int i = 0; i < r.n_armors; ++i) {
        const int l = (int)r.armors[i].label;
        if (l < 0 || l >= class_num) fail(RMR_ERR_INVALID_ARGUMENT, "robot feature: armor label %d outside [0, %d)", l, class_num);
        out[l] += r.armors[i].confidence;
    }
    float sum = 0.f;
    for (int i = 0; i < class_num; ++i) sum += out[i];
    if (sum == 0.f) return;
    for (int i = 0; i < class_num; ++i) out[i] /= sum;
}

static std::vector<float> initial_state(const float loc[3]) { return {loc[0], 0, 0, loc[1], 0, 0, loc[2], 0, 0}; }
static std::vector<float> scaled_identity(int n, float s) {
    std::vector<float> m((size_t)n * n, 0.f);
    for (int i = 0; i < n; ++i) m[(size_t)i * n + i] = s;
    return m;
}
static std::vector<float> diagonal3(const float d[3]) { return {d[0], 0, 0, 0, d[1], 0, 0, 0, d[2]}; }

Track::Track(const float loc[3], const std::vector<float>& feature, int64_t t, int track_id, float max_a, float tau,
             const float noise[3])
    : sums(feature), t_ns(t), id(track_id),
      filter(initial_state(loc).data(), scaled_identity(9, 0.1f).data(), max_a, tau, diagonal3(noise).data()) {}

void Track::predict(int64_t t) {
    const float dt = (float)((double)(float)(t - t_ns) * 1e-9);  // track.h:112-117
    filter.predict(dt);
    t_ns = t;
}

void Track::update(const float loc[3], const std::vector<float>& feature) {
    if (feature.size() != sums.size()) fail(RMR_ERR_LOGIC, "Track::update: feature size changed");
    for (size_t i = 0; i < sums.size(); ++i) sums[i] += feature[i];
    filter.update(loc);
}

int Track::label() const {
    int best = 0;
    for (size_t i = 1; i < sums.size(); ++i)
        if (sums[i] > sums[best]) best = (int)i;
    return best;
}

std::vector<float> Track::feature() const {
    float total = 0.f;
    for (float s : sums) total += s;
    std::vector<float> f(sums.size(), 0.f);
    if (total != 0.f)
        for (size_t i = 0; i < sums.size(); ++i) f[i] = sums[i] / total;
    return f;
}

void Track::location(float out[3]) const {
    const Mat& x = filter.state();
    out[0] = x(0, 0), out[1] = x(3, 0), out[2] = x(6, 0);
}

// ------------------------------------------------------------------------------ Tracker

Tracker::Tracker(const TrackerCfg& cfg) : cfg_(cfg) {
    if (cfg.class_num <= 0 || cfg.class_num > 256) fail(RMR_ERR_INVALID_ARGUMENT, "Tracker: class_num %d", cfg.class_num);
}

static float distance3(const float a[3], const float b[3]) {
    return std::sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
}

float Tracker::cost(const Track& t, const rmr_robot& r) const {
    const bool located = r.has_location != 0, detected = r.n_armors > 0;
    if (!located && !detected) return 0.f;
    float d_score = 0.f;
    if (located) {
        float tl[3];
        t.location(tl);
        const float d = distance3(r.location, tl);
        d_score = d < cfg_.distance_thresh ? 1.f : d < 2 * cfg_.distance_thresh ? -d / cfg_.distance_thresh + 2.f : 0.f;
    }
    std::vector<float> fr(cfg_.class_num);
    robot_feature(r, cfg_.class_num, fr.data());
    const std::vector<float> ft = t.feature();
    float dot = 0.f, nr = 0.f, nt = 0.f;
    for (int i = 0; i < cfg_.class_num; ++i) dot += fr[i] * ft[i], nr += fr[i] * fr[i], nt += ft[i] * ft[i];
    const float denom = std::sqrt(nr) * std::sqrt(nt);
    float f_score = 0.f;
    if (denom != 0.f) f_score = (dot / denom + 1.f) / 2.f;
    return d_score * cfg_.distance_weight + f_score * cfg_.feature_weight;
}

// Robot::setTrack (robot.cpp:81-94)
static void set_track(rmr_robot& r, const Track& t) {
    r.track_state = t.state;
    float loc[3];
    t.location(loc);
    if (t.state == kConfirmed) {
        r.has_label = 1, r.label = t.label();
        r.has_location = 1, std::copy(loc, loc + 3, r.location);
    } else {
        if (!r.has_label) r.has_label = 1, r.label = t.label();
        if (!r.has_location) r.has_location = 1, std::copy(loc, loc + 3, r.location);
    }
}

void Tracker::update(rmr_robot* robots, int n, int64_t t_ns) {
    if (n < 0 || (n > 0 && !robots)) fail(RMR_ERR_INVALID_ARGUMENT, "Tracker::update: bad robot array");
    for (Track& t : tracks_) t.predict(t_ns);

    const int nt = (int)tracks_.size();
    std::vector<float> value((size_t)n * std::max(nt, 1), 0.f);
    for (int r = 0; r < n; ++r)
        for (int t = 0; t < nt; ++t) value[(size_t)r * nt + t] = cost(tracks_[t], robots[r]);
    const std::vector<int> match = auction(value.data(), n, nt, cfg_.max_iter);

    std::vector<int> fresh;                 // robots that may start a track
    std::vector<char> matched(nt, 0);
    std::vector<float> feat(cfg_.class_num);
    for (int r = 0; r < n; ++r) {
        rmr_robot& rb = robots[r];
        const int t = match[r];
        if (!rb.has_location || t < 0) {
            fresh.push_back(r);
            continue;
        }
        Track& tr = tracks_[t];
        float tl[3];
        tr.location(tl);
        // the auction hands every agent something: an assignment that is both far away and of
        // another label is no association (tracker.cpp:156-166)
        if (distance3(rb.location, tl) > 2 * cfg_.distance_thresh && (rb.has_label ? rb.label : -1) != tr.label()) {
            fresh.push_back(r);
            continue;
        }
        robot_feature(rb, cfg_.class_num, feat.data());
        tr.update(rb.location, feat);
        if (tr.state == kTentative && ++tr.init_count >= cfg_.init_thresh) tr.state = kConfirmed;
        tr.miss_count = 0;
        set_track(rb, tr);
        matched[t] = 1;
    }

    for (int t = 0; t < nt; ++t) {
        if (matched[t]) continue;
        Track& tr = tracks_[t];
        if (tr.state == kTentative) {
            tr.state = kDeleted;
        } else if (tr.state == kConfirmed && ++tr.miss_count >= cfg_.miss_thresh) {
            tr.state = kDeleted;
        }
    }
    tracks_.erase(std::remove_if(tracks_.begin(), tracks_.end(), [](const Track& t) { return t.state == kDeleted; }),
                  tracks_.end());

    for (int r : fresh) {
        rmr_robot& rb = robots[r];
        if (rb.n_armors > 0 && rb.has_location) {
            robot_feature(rb, cfg_.class_num, feat.data());
            tracks_.emplace_back(rb.location, feat, t_ns, latest_id_++, cfg_.max_acceleration,
                                 cfg_.acceleration_correlation_time, cfg_.observation_noise);
            set_track(rb, tracks_.back());
        }
    }
}

}  // namespace track
}  // namespace rmr
