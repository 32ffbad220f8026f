"""Tracker stage: the C++ implementation behind the C-ABI (rm_radar_amd.Tracker,
KalmanFilter, SingerEKF, auction) against the reference's own known answers
(test/track/{kf,ekf,singer,auction}_test.cpp) and against the numpy restatement
(oracle/tracker_ref.py) on seeded scenarios.  CPU only: the tracker is host code in the reference
and here."""
import numpy as np
import pytest

import oracle.tracker_ref as REF


@pytest.fixture(scope="module")
def rmr():
    import __graft_entry__ as g
    g.build()
    import rm_radar_amd
    return rm_radar_amd


F = np.array([[1, 0, 1, 0], [0, 1, 0, 1], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
H = np.array([[1, 0, 0, 0], [0, 1, 0, 0]], np.float32)
# filterpy results quoted by the reference (kf_test.cpp:77-85, ekf_test.cpp:109-116)
X_FILTERPY = np.array([0.47727273, 0.47727273, 0.22727273, 0.22727273], np.float32)
P_FILTERPY = np.array([[0.09545455, 0, 0.04545455, 0], [0, 0.09545455, 0, 0.04545455],
                       [0.04545455, 0, 0.64545455, 0], [0, 0.04545455, 0, 0.64545455]], np.float32)


def _is_approx(a, b, prec):  # Eigen isApprox: ||a - b|| <= prec * min(||a||, ||b||)
    return np.linalg.norm(a - b) <= prec * min(np.linalg.norm(a), np.linalg.norm(b))


def test_kalman_filter_matches_filterpy_constants(rmr):
    mk = dict(F=F, Q=0.1 * np.eye(4), H=H)
    for kf in (rmr.KalmanFilter(np.zeros(4), np.eye(4), 0.1 * np.eye(2), **mk),
               REF.Kalman(np.zeros(4), np.eye(4), 0.1 * np.eye(2), **mk)):
        kf.predict()
        kf.update([0.5, 0.5])
        x = kf.state if hasattr(kf, "state") else kf.x
        P = kf.covariance if hasattr(kf, "covariance") else kf.P
        assert _is_approx(x, X_FILTERPY, 1e-5) and _is_approx(P, P_FILTERPY, 1e-5)


def test_extended_kalman_filter_matches_filterpy_constants(rmr):
    # ekf_test.cpp: transition and observation functions are evaluated by the caller
    for kf in (rmr.KalmanFilter(np.zeros(4), np.eye(4), 0.1 * np.eye(2)), REF.Kalman(np.zeros(4), np.eye(4), 0.1 * np.eye(2))):
        kf.predict(F, 0.1 * np.eye(4, dtype=np.float32))
        x = kf.state if hasattr(kf, "state") else kf.x
        kf.update([0.5, 0.5], hx=x[:2].copy(), H=H)
        x = kf.state if hasattr(kf, "state") else kf.x
        P = kf.covariance if hasattr(kf, "covariance") else kf.P
        assert _is_approx(x, X_FILTERPY, 1e-5) and _is_approx(P, P_FILTERPY, 1e-5)


def test_kalman_bad_arguments(rmr):
    with pytest.raises(rmr.InvalidArgument):
        rmr.KalmanFilter(np.zeros(4), np.eye(3), 0.1 * np.eye(2))
    kf = rmr.KalmanFilter(np.zeros(4), np.eye(4), 0.1 * np.eye(2))
    with pytest.raises(rmr.RmrError):
        kf.predict()  # an extended filter has no stored transition model


def _singer(rmr, which):
    args = (np.zeros(9), 0.5 * np.eye(9), 2.0, 1.0, 0.2 * np.eye(3))  # singer_test.cpp:17-31
    return rmr.SingerEKF(*args) if which == "product" else REF.SingerEKF(*args)


@pytest.mark.parametrize("which", ["product", "oracle"])
def test_singer_known_motions(rmr, which):
    # singer_test.cpp: a fixed target, uniform motion, uniformly accelerated motion
    f = _singer(rmr, which)
    for _ in range(10):
        f.predict(1.0)
        f.update([10, 20, 30])
    assert _is_approx(f.state[0::3], np.array([10, 20, 30], np.float32), 1e-1)

    f = _singer(rmr, which)
    for i in range(10):
        f.predict(1.0)
        f.update([10 + 2 * i, 20 + 4 * i, 30 + 6 * i])
    s = f.state
    assert _is_approx(s[0::3], np.array([28, 56, 84], np.float32), 1e-1)
    assert _is_approx(s[1::3], np.array([2, 4, 6], np.float32), 1e-1)
    assert np.all(np.abs(s[2::3]) < 1e-1)

    f = _singer(rmr, which)
    acc = np.array([0.0, 0.5, 1.0])
    for i in range(10):
        f.predict(1.0)
        f.update(np.array([10, 20, 30]) + np.array([2, 4, 6]) * i + 0.5 * acc * i * i)
    s = f.state
    assert _is_approx(s[0::3], (np.array([10, 20, 30]) + np.array([2, 4, 6]) * 9 + 0.5 * acc * 81).astype(np.float32), 1e-1)
    assert _is_approx(s[1::3], (np.array([2, 4, 6]) + acc * 9).astype(np.float32), 1e-1)


def test_singer_matches_oracle_step_by_step(rmr):
    rng = np.random.default_rng(0)
    a, b = _singer(rmr, "product"), _singer(rmr, "oracle")
    for i in range(40):
        dt = float(rng.uniform(0.02, 0.3))
        z = rng.normal(0, 5, 3).astype(np.float32)
        a.predict(dt), b.predict(dt)
        a.update(z), b.update(z)
        assert np.allclose(a.state, b.state, rtol=2e-4, atol=2e-4), i
        assert np.allclose(a.covariance, b.kf.P, rtol=2e-4, atol=2e-5), i


def test_auction_known_answers(rmr):
    sq = np.arange(1, 10, dtype=np.float32).reshape(3, 3)
    for fn in (lambda v, it: list(rmr.auction(v, it)), REF.auction):
        assert list(fn(sq, 100)) == [2, 1, 0]                                  # auction_test.cpp:10-22
        r = fn(np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9], [1, 4, 7]], np.float32), 100)  # more agents than tasks
        assert len(r) == 4 and all(t in r for t in range(3))
        r = fn(np.arange(1, 13, dtype=np.float32).reshape(3, 4), 100)           # more tasks than agents
        assert len(r) == 3 and all(t != -1 for t in r)
        assert list(fn(sq, 0)) == [-1, -1, -1]                                  # zero iterations


def test_auction_matches_oracle_on_random_matrices(rmr):
    rng = np.random.default_rng(1)
    for _ in range(200):
        a, t = int(rng.integers(0, 8)), int(rng.integers(0, 8))
        v = rng.uniform(0, 1, (a, t)).astype(np.float32)
        if rng.random() < 0.3:
            v = np.round(v * 4) / 4  # ties
        it = int(rng.choice([0, 1, 3, 100]))
        want = REF.auction(v, it) if a else []
        assert list(rmr.auction(v, it)) == want, (v, it)


def test_robot_feature(rmr):
    arm = np.array([(0, 0, 1, 1, 3, 0.5), (0, 0, 1, 1, 7, 0.25), (0, 0, 1, 1, 3, 0.25)], rmr.DET_DTYPE)
    r = rmr.Robot(armors=arm, label=3, confidence=0.75)
    assert np.allclose(r.feature(12), np.eye(12)[3] * 0.75 + np.eye(12)[7] * 0.25)
    assert not rmr.Robot().feature(12).any()                     # not detected
    with pytest.raises(rmr.InvalidArgument):
        r.feature(5)                                             # label 7 outside 5 classes


# ------------------------------------------------------------------------- scenarios

def _scenario(seed, frames=60, n_targets=4, class_num=12):
    """Per frame: a list of (armors or None, location or None) observations of moving targets,
    with dropouts, unlocated / undetected robots, clutter and label noise."""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(-6, 6, (n_targets, 3)).astype(np.float64)
    vel = rng.uniform(-1.5, 1.5, (n_targets, 3))
    labels = rng.permutation(class_num)[:n_targets]
    t_ns, out = 1_000_000_000, []
    for f in range(frames):
        dt = float(rng.choice([0.05, 0.1, 0.1, 0.2]))
        t_ns += int(dt * 1e9)
        pos += vel * dt
        obs = []
        for k in range(n_targets):
            if rng.random() < 0.15:
                continue  # missed entirely
            loc = None if rng.random() < 0.1 else tuple((pos[k] + rng.normal(0, 0.05, 3)).astype(np.float32))
            arm = None
            if rng.random() > 0.15:
                lab = labels[k] if rng.random() > 0.1 else int(rng.integers(0, class_num))
                arm = [(int(lab), float(np.float32(rng.uniform(0.5, 1.0)))) for _ in range(int(rng.integers(1, 4)))]
            obs.append((arm, loc))
        if rng.random() < 0.2:  # clutter
            obs.append(([(int(rng.integers(0, class_num)), 0.6)], tuple(rng.uniform(-6, 6, 3).astype(np.float32))))
        order = rng.permutation(len(obs))
        out.append((t_ns, [obs[i] for i in order]))
    return out


def _to_product(rmr, obs):
    robots = []
    for arm, loc in obs:
        a = None
        if arm:
            a = np.array([(0, 0, 10, 10, l, c) for l, c in arm], rmr.DET_DTYPE)
        # Robot::setDetection picks the label with the largest summed confidence (robot.cpp:41-74)
        lab = conf = None
        if arm:
            sums = {}
            for l, c in arm:
                sums[l] = np.float32(sums.get(l, np.float32(0)) + np.float32(c))
            lab = max(sorted(sums), key=lambda k: sums[k])
            conf = float(sums[lab])
        robots.append(rmr.Robot(armors=a, label=lab, confidence=conf, location=loc))
    return robots


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_tracker_matches_oracle_on_scenarios(rmr, seed):
    kw = dict(init_thresh=3, miss_thresh=4) if seed % 2 else {}
    prod = rmr.Tracker((0.1, 0.1, 0.1), 12, **kw)
    ref = REF.Tracker((0.1, 0.1, 0.1), 12, **kw)
    confirmed_seen = deleted_seen = 0
    for t_ns, obs in _scenario(seed):
        pr = _to_product(rmr, obs)
        rr = [REF.Robot(armors=a, location=l, label=p.label) for (a, l), p in zip(obs, pr)]
        n_before = len(ref.tracks)
        prod.update(pr, t_ns)
        ref.update(rr, t_ns)
        # robots: same track state, label, location
        for p, r in zip(pr, rr):
            assert p.track_state == r.track_state
            assert p.label == r.label
            assert (p.location is None) == (r.location is None)
            if p.location is not None:
                assert np.allclose(p.location, r.location, rtol=1e-4, atol=1e-4)
        # tracks: same ids, states, counters, labels; filter state within tolerance
        pt = prod.tracks()
        assert [t["id"] for t in pt] == [t.id for t in ref.tracks]
        for a, b in zip(pt, ref.tracks):
            assert (a["state"], a["label"], a["init_count"], a["miss_count"]) == (b.state, b.label(), b.init_count, b.miss_count)
            assert np.allclose(a["state_vector"], b.filter.state, rtol=5e-4, atol=5e-4)
        confirmed_seen += sum(t.state == REF.CONFIRMED for t in ref.tracks)
        deleted_seen += max(0, n_before - len([t for t in ref.tracks if t.id < ref.latest_id - 0]))
    assert confirmed_seen > 0 and ref.latest_id > 4  # the scenario exercised confirmation and track churn


def test_tracker_edge_cases(rmr):
    tr = rmr.Tracker((0.1, 0.1, 0.1), 12)
    assert tr.update([], 1_000_000_000) == [] and tr.tracks() == []          # no robots, no tracks
    arm = np.array([(0, 0, 1, 1, 2, 0.9)], rmr.DET_DTYPE)
    # detected but not located: no track is started; located but not detected: neither
    rs = tr.update([rmr.Robot(armors=arm, label=2, confidence=0.9), rmr.Robot(location=(1.0, 2.0, 3.0))], 1_100_000_000)
    assert tr.tracks() == [] and all(r.track_state is None for r in rs)
    # detected and located: tentative track 0; confirmed after init_thresh = 4 matched updates
    states = []
    for i in range(6):
        r = rmr.Robot(armors=arm, label=2, confidence=0.9, location=(1.0 + 0.01 * i, 2.0, 3.0))
        tr.update([r], 1_200_000_000 + i * 100_000_000)
        states.append(r.track_state)
    assert states == [rmr.TRACK_TENTATIVE] * 4 + [rmr.TRACK_CONFIRMED] * 2
    t = tr.tracks()
    assert len(t) == 1 and t[0]["id"] == 0 and t[0]["label"] == 2 and t[0]["state"] == rmr.TRACK_CONFIRMED
    # a confirmed track fills in the location of a matched, label-compatible robot and survives
    # miss_thresh - 1 = 9 empty frames, then is deleted
    for i in range(9):
        tr.update([], 2_000_000_000 + i * 100_000_000)
        assert len(tr.tracks()) == 1 and tr.tracks()[0]["miss_count"] == i + 1
    tr.update([], 3_000_000_000)
    assert tr.tracks() == []
    # a tentative track that misses once is dropped at once
    tr.update([rmr.Robot(armors=arm, label=2, confidence=0.9, location=(0.0, 0.0, 0.0))], 3_100_000_000)
    assert len(tr.tracks()) == 1 and tr.tracks()[0]["id"] == 1
    tr.update([], 3_200_000_000)
    assert tr.tracks() == []
    with pytest.raises(rmr.InvalidArgument):
        rmr.Tracker((0.1, 0.1, 0.1), 0)
