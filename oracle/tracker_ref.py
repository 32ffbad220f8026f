"""CPU restatement (numpy, f32) of the reference's tracker stage -- TEST INFRASTRUCTURE ONLY.

Follows src/track/kalman_filter.h, singer.h, auction.h, features.h, track.h, tracker.cpp and
Robot::feature / Robot::setTrack (src/robot/robot.cpp:81-122) line by line; each function cites
the lines it restates.  Pinned to the reference's own known-answer tests (tests/test_tracker.py):
the filterpy constants of test/track/kf_test.cpp:77-85 and ekf_test.cpp:109-116, the Singer
convergence tests of singer_test.cpp and the auction cases of auction_test.cpp.

Eigen evaluates fixed-size products and row sums in an unspecified (vectorised) order, so results
are defined to a few ulp only; the product is compared with a tolerance.
"""
import numpy as np

F32 = np.float32
NOT_MATCHED = -1
TENTATIVE, CONFIRMED, DELETED = 1, 2, 3


def _m(a):
    return np.asarray(a, F32)


class Kalman:
    """KalmanFilter (kalman_filter.h:77-166) and ExtendedKalmanFilter (:182-296)."""

    def __init__(self, x0, P0, R, F=None, Q=None, H=None):
        self.x, self.P, self.R = _m(x0).copy(), _m(P0).copy(), _m(R)
        self.F = None if F is None else _m(F)
        self.Q = None if Q is None else _m(Q)
        self.H = None if H is None else _m(H)

    def predict(self, F=None, Q=None):  # :116-121 / :221-232
        if F is not None:
            self.F, self.Q = _m(F), _m(Q)
        self.x = (self.F @ self.x).astype(F32)
        self.P = ((self.F @ self.P).astype(F32) @ self.F.T + self.Q).astype(F32)

    def update(self, z, hx=None, H=None):  # :129-152 / :243-248, 274-293
        if H is not None:
            self.H = _m(H)
        pred = (self.H @ self.x).astype(F32) if hx is None else _m(hx)
        res = (_m(z) - pred).astype(F32)
        S = ((self.H @ self.P).astype(F32) @ self.H.T + self.R).astype(F32)
        K = ((self.P @ self.H.T).astype(F32) @ np.linalg.inv(S.astype(np.float64)).astype(F32)).astype(F32)
        self.x = (self.x + K @ res).astype(F32)
        n = len(self.x)
        self.P = ((np.eye(n, dtype=F32) - K @ self.H).astype(F32) @ self.P).astype(F32)


class SingerEKF:
    """singer.h:33-132"""

    def __init__(self, x0, P0, max_a, tau, R):
        self.kf = Kalman(x0, P0, R)
        self.max_a, self.tau = F32(max_a), F32(tau)

    def predict(self, dt):
        dt = F32(dt)
        F = np.eye(9, dtype=F32)
        Q = np.zeros((9, 9), F32)
        decay = F32(np.exp(F32(-dt / self.tau)))
        for i in range(3):  # singer.h:94-122; std::pow(float, int) is evaluated in double
            b = 3 * i
            F[b, b + 1] = dt
            F[b, b + 2] = F32(F32(dt * dt) / F32(2))
            F[b + 1, b + 2] = dt
            F[b + 2, b + 2] = decay
            Q[b, b] = F32(float(dt) ** 3 / 3)
            Q[b + 1, b] = Q[b, b + 1] = F32(float(dt) ** 2 / 2)
            Q[b + 2, b] = Q[b, b + 2] = F32(dt / F32(2))
            Q[b + 1, b + 1] = dt
            Q[b + 2, b + 1] = Q[b + 1, b + 2] = F32(F32(1) - decay)
            Q[b + 2, b + 2] = F32((F32(1) - F32(np.exp(F32(F32(-2) * dt / self.tau)))) / F32(2))
        Q = (Q * F32(float(self.max_a) ** 2)).astype(F32)
        self.kf.predict(F, Q)

    def update(self, z):
        H = np.zeros((3, 9), F32)
        for i in range(3):
            H[i, 3 * i] = 1
        self.kf.update(z, hx=self.kf.x[0::3].copy(), H=H)

    @property
    def state(self):
        return self.kf.x


def auction(values, max_iter):
    """auction.h:49-127"""
    values = _m(values)
    agents, tasks = values.shape
    real = tasks
    if agents > tasks:
        ext = np.zeros((agents, agents), F32)
        ext[:, :tasks] = values
        values, tasks = ext, agents
    prices = np.zeros(tasks, F32)
    assign = [NOT_MATCHED] * agents
    it = 0
    while it < max_iter:
        if sum(1 for v in assign if 0 <= v <= real) >= agents:  # '<=' as written (auction.h:72-74)
            break
        changed = False
        for a in range(agents):
            if assign[a] != NOT_MATCHED:
                continue
            best, best_v = NOT_MATCHED, F32(-np.inf)
            for t in range(tasks):
                v = F32(values[a, t] - prices[t])
                if v > best_v:
                    best_v, best = v, t
            if best != NOT_MATCHED:
                prices[best] = F32(prices[best] + best_v)
                for o in range(agents):
                    if assign[o] == best:
                        assign[o] = NOT_MATCHED
                        break
                assign[a] = best
                changed = True
        if not changed:
            break
        it += 1
    return [NOT_MATCHED if v >= real else v for v in assign]


class Robot:
    """The slice of radar::Robot the tracker reads and writes (robot.h:53-164)."""

    def __init__(self, armors=None, location=None, label=None):
        self.armors = armors  # list of (label, confidence) or None: isDetected()
        self.location = None if location is None else tuple(F32(v) for v in location)  # metres; isLocated()
        self.label = label
        self.track_state = None

    def feature(self, class_num):  # robot.cpp:102-122
        f = np.zeros(class_num, F32)
        if not self.armors:
            return f
        for lab, conf in self.armors:
            f[int(lab)] = F32(f[int(lab)] + F32(conf))
        s = F32(f.sum(dtype=F32))
        return f if s == 0 else (f / s).astype(F32)

    def set_track(self, track):  # robot.cpp:81-94
        self.track_state = track.state
        if track.state == CONFIRMED:
            self.label, self.location = track.label(), track.location()
        else:
            if self.label is None:
                self.label = track.label()
            if self.location is None:
                self.location = track.location()


class Track:
    """track.h:36-197 with features.h:30-209 (columns are only ever summed)."""

    def __init__(self, location, feature, t_ns, track_id, max_acc, tau, noise):
        self.columns = [feature.copy()]
        self.t_ns, self.id = t_ns, track_id
        self.init_count = self.miss_count = 0
        self.state = TENTATIVE
        x0 = [location[0], 0, 0, location[1], 0, 0, location[2], 0, 0]
        self.filter = SingerEKF(x0, np.eye(9, dtype=F32) * F32(0.1), max_acc, tau, np.diag(_m(noise)))

    def predict(self, t_ns):  # track.h:107-121
        dt = F32(float(F32(t_ns - self.t_ns)) * 1e-9)
        self.filter.predict(dt)
        self.t_ns = t_ns

    def update(self, location, feature):  # track.h:128-136
        self.columns.append(feature.copy())
        self.filter.update(_m(location))

    def _sums(self):
        s = np.zeros_like(self.columns[0])
        for c in self.columns:
            s = (s + c).astype(F32)
        return s

    def label(self):  # features.h:178-183
        return int(np.argmax(self._sums()))

    def feature(self):  # features.h:190-199
        s = self._sums()
        tot = F32(s.sum(dtype=F32))
        return np.zeros_like(s) if tot == 0 else (s / tot).astype(F32)

    def location(self):  # track.h:170-173
        x = self.filter.state
        return (F32(x[0]), F32(x[3]), F32(x[6]))


class Tracker:
    """tracker.h:23-54, tracker.cpp:85-220"""

    def __init__(self, observation_noise, class_num, init_thresh=4, miss_thresh=10, max_acceleration=2.0,
                 acceleration_correlation_time=1.0, distance_weight=0.40, feature_weight=0.60, max_iter=100,
                 distance_thresh=0.8):
        self.noise, self.class_num = observation_noise, class_num
        self.init_thresh, self.miss_thresh = init_thresh, miss_thresh
        self.max_acc, self.tau = max_acceleration, acceleration_correlation_time
        self.wd, self.wf = F32(distance_weight), F32(feature_weight)
        self.max_iter, self.dthr = max_iter, F32(distance_thresh)
        self.tracks, self.latest_id = [], 0

    @staticmethod
    def distance(a, b):  # tracker.cpp:68-73
        d = [F32(F32(a[i]) - F32(b[i])) for i in range(3)]
        return F32(np.sqrt(F32(F32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])))

    def cost(self, track, robot):  # tracker.cpp:85-119
        if robot.location is None and not robot.armors:
            return F32(0)
        if robot.location is None:
            ds = F32(0)
        else:
            d = self.distance(robot.location, track.location())
            ds = F32(1) if d < self.dthr else (F32(-d / self.dthr + F32(2)) if d < F32(2) * self.dthr else F32(0))
        fr, ft = robot.feature(self.class_num), track.feature()
        denom = F32(np.sqrt(F32(np.dot(fr, fr))) * np.sqrt(F32(np.dot(ft, ft))))
        fs = F32(0) if denom == 0 else F32((F32(np.dot(fr, ft)) / denom + F32(1)) / F32(2))
        return F32(ds * self.wd + fs * self.wf)

    def update(self, robots, t_ns):  # tracker.cpp:126-220
        for t in self.tracks:
            t.predict(t_ns)
        cost = np.zeros((len(robots), len(self.tracks)), F32)
        for r, rb in enumerate(robots):
            for t, tr in enumerate(self.tracks):
                cost[r, t] = self.cost(tr, rb)
        match = auction(cost, self.max_iter) if len(robots) else []
        unmatched, matched = [], []
        for r, rb in enumerate(robots):
            if rb.location is None:
                unmatched.append(r)
                continue
            t = match[r]
            if t == NOT_MATCHED:
                unmatched.append(r)
                continue
            tr = self.tracks[t]
            lab = rb.label if rb.label is not None else -1
            if self.distance(rb.location, tr.location()) > F32(2) * self.dthr and lab != tr.label():
                unmatched.append(r)
                continue
            tr.update(rb.location, rb.feature(self.class_num))
            if tr.state == TENTATIVE:
                tr.init_count += 1
                if tr.init_count >= self.init_thresh:
                    tr.state = CONFIRMED
            tr.miss_count = 0
            rb.set_track(tr)
            matched.append(t)
        for i, tr in enumerate(self.tracks):
            if i in matched:
                continue
            if tr.state == TENTATIVE:
                tr.state = DELETED
            elif tr.state == CONFIRMED:
                tr.miss_count += 1
                if tr.miss_count >= self.miss_thresh:
                    tr.state = DELETED
        self.tracks = [t for t in self.tracks if t.state != DELETED]
        for r in unmatched:
            rb = robots[r]
            if rb.armors and rb.location is not None:
                tr = Track(rb.location, rb.feature(self.class_num), t_ns, self.latest_id, self.max_acc, self.tau,
                           self.noise)
                self.latest_id += 1
                rb.set_track(tr)
                self.tracks.append(tr)
