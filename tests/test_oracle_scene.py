"""CPU-only checks of the oracle on the synthetic scenes: the clustering partition is
cross-checked against scipy (cKDTree radius graph + connected components), and the scenes are
shown to exercise foreground / cluster / search (SURVEY.md 8c "independent oracles")."""
import numpy as np
import pytest
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components
from scipy.spatial import cKDTree

import scenes


def _scipy_partition(xyz, tol, lo, hi):
    n = len(xyz)
    d = xyz[:, None, :] - xyz[None, :, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]).astype(np.float32)
    adj = d2 < np.float32(tol) * np.float32(tol)
    ncomp, lab = connected_components(coo_matrix(adj), directed=False)
    sizes = np.bincount(lab, minlength=ncomp)
    roots = np.array([np.nonzero(lab == c)[0][0] for c in range(ncomp)])
    valid = [c for c in range(ncomp) if lo <= sizes[c] <= hi]
    valid.sort(key=lambda c: (-sizes[c], roots[c]))
    ids = {c: i for i, c in enumerate(valid)}
    return np.array([ids.get(c, -1) for c in lab], np.int32), len(valid)


@pytest.mark.parametrize("seed", [0, 1])
def test_cluster_partition_matches_scipy(oracle, seed):
    clouds, rects = scenes.scene(seed, 30000, (640, 640))
    loc = oracle.Locator(640, 640, scenes.K640, scenes.SAMPLE_L2C, np.eye(4, dtype=np.float32))
    seen_fg = seen_loc = 0
    for f, c in enumerate(clouds):
        loc.update(c)
        loc.cluster()
        xyz, pix, cid = loc.foreground()
        if len(xyz):
            want, ncl = _scipy_partition(xyz, 400.0, 8, 1000)
            assert ncl == loc.num_clusters
            assert np.array_equal(cid, want)
            # the kd-tree neighbour graph agrees with the brute-force f32 graph
            pairs = cKDTree(xyz.astype(np.float64)).query_pairs(400.0 - 1e-3)
            for i, j in list(pairs)[:2000]:
                assert cid[i] == cid[j]
        seen_fg += len(xyz)
        seen_loc += sum(loc.search(r) is not None for r in rects[f])
    assert seen_fg > 100 and seen_loc >= 4


def test_assets_clouds_are_plumbing_only(oracle):
    # SURVEY 8c feasibility finding: without background.pcd the sample clouds give few fg pixels
    import os
    data = np.load(os.path.join(os.path.dirname(__file__), "golden", "assets_clouds.npz"))
    loc = oracle.Locator(2592, 2048, scenes.SAMPLE_K, scenes.SAMPLE_L2C, scenes.SAMPLE_W2C)
    fg = []
    for i in range(10):
        loc.update(data[f"cloud{i}"].astype(np.float32))
        loc.cluster()
        fg.append(len(loc.foreground()[0]))
    assert (loc.wz, loc.hz) == (1296, 1024)
    assert max(fg) < 200


def test_grouping_and_vote(oracle):
    # robot.cpp:41-74 / detector.cpp:427-454 semantics on hand-built cases
    D = oracle.DET_DTYPE
    car = (100, 50, 200, 100, 0, 0.9)
    armors = np.array([(10, 10, 20, 10, 3, 0.6), (40, 10, 20, 10, 5, 0.5), (70, 12, 20, 10, 3, 0.3),
                       (90, 40, 20, 10, 5, 0.4)], D)
    r = oracle.make_robot(car, armors)
    assert r.has_label and r.label == 3  # sums 0.9 vs 0.9: tie -> lowest label
    assert abs(r.confidence - np.float32(np.float32(0.6) + np.float32(0.3)) / 2) < 1e-7
    assert r.armors[0].x == 110 and r.armors[0].y == 60
    r_none = oracle.make_robot(car, np.zeros(0, D))
    assert not r_none.has_label
    # same label, overlapping (IoU > thr) -> newcomer dropped; non-overlapping -> higher conf kept
    a = oracle.make_robot((100, 50, 200, 100, 0, 0.9), np.array([(1, 1, 5, 5, 2, 0.6)], D))
    b = oracle.make_robot((102, 51, 200, 100, 0, 0.9), np.array([(1, 1, 5, 5, 2, 0.9)], D))
    c = oracle.make_robot((900, 500, 100, 100, 0, 0.9), np.array([(1, 1, 5, 5, 2, 0.95)], D))
    d = oracle.make_robot((10, 10, 50, 50, 0, 0.9), np.array([(1, 1, 5, 5, 1, 0.7)], D))
    out = oracle.group_robots([a, r_none, b, d, c], 0.75)
    assert [o.has_label for o in out] == [0, 1, 1]
    assert [o.label for o in out[1:]] == [1, 2]
    assert out[2].rect[0] == 900  # c replaced a (higher confidence, IoU 0); b was dropped
    # rounding used by Robot::rect(): half to even
    assert oracle.rect_round((0.5, 1.5, 2.5, -0.5)) == (0, 2, 2, 0)
    assert oracle.crop_rect((10.9, 20.1, 30.99, 40.5, 0, 1)) == (10, 20, 30, 40)
