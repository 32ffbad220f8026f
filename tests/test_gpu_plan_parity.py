"""The TIMED configuration's network, under the kernel plan it is timed with, against the torch oracle.

bench.py, tools/round_profile.sh and the PMC passes run under the committed plans (profiles/plans/*.tune, RMR_PLAN): the
exact kernel per (layer, batch).  The other network tests run whatever the autotuner picks on the test box, so a wrong
entry in a committed plan would be timed and never compared with anything ("pinned-plan hole").  Here the
plan files are applied exactly as bench.apply_plan applies them, on bench's own packs (seed 1 / 2, cls_bias -6), and the
heads at the plan's own batch sizes -- car 64 and 1, armor 256 and 4 (fp8: car 64 / 256, armor 256 / 4) -- are held to
oracle/yolov8_ref.py: the f16 plan with test_gpu_network._check_head's tolerances against the f16-emulating oracle
(2 px / 1e-2, mean 0.25 px), the fp8 plan with the absolute bar of round 5 (no further from the fp8 oracle than a second
exact implementation of the plan is, x 1.15).  The TensorRT engine these layers replace: src/detect/detector.h:122.

A plan entry that names a kernel which cannot run its layer must fail loudly (never fall back to tuning):
test_a_corrupted_plan_entry_is_an_error.  One that names ANOTHER correct kernel gives another f32 summation order and is
caught by test_the_plan_is_what_runs (the heads under the plan differ bit-wise from the heads under that edited plan)."""
import os
import shutil
import sys

import numpy as np
import pytest

import netutil

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _Args:
    def __init__(self, dtype):
        self.dtype, self.plan = dtype, "auto"


@pytest.fixture(scope="module")
def rmr():
    import rm_radar_amd as r
    assert r.device_count() >= 1
    return r


@pytest.fixture(scope="module")
def images():
    return [netutil.test_image(1), netutil.test_image(2, 810, 1080), netutil.test_image(3, 1280, 720)]


@pytest.fixture(scope="module")
def bench_packs(tmp_path_factory):
    """bench.py's packs: W.make_synthetic_pack(seed=1 / 2, cls_bias=-6.0)"""
    from rm_radar_amd import weights as W
    d = tmp_path_factory.mktemp("bench_packs")
    return (W.make_synthetic_pack(str(d / "car.rmrw"), "m", 1, seed=1, cls_bias=-6.0),
            W.make_synthetic_pack(str(d / "armor.rmrw"), "m", 12, seed=2, cls_bias=-6.0))


@pytest.fixture()
def pinned(monkeypatch):
    """bench.apply_plan on a pair of packs; RMR_PLAN is restored afterwards (apply_plan writes os.environ)."""
    import bench
    monkeypatch.setenv("RMR_PLAN", "")   # monkeypatch records the old value and restores it at teardown

    def apply(dtype, packs):
        plan = bench.apply_plan(_Args(dtype), packs)
        assert plan is not None, f"no committed plan for {dtype}"
        assert os.environ.get("RMR_PLAN") == "1"
        return plan
    yield apply
    for k in ("RMR_PLAN",):
        os.environ.pop(k, None)


def _blobs(oracle, images):
    return np.stack([oracle.preprocess(im)[0] for im in images])


def _check_head(got, want, box_tol, score_tol):
    assert got.shape == want.shape
    assert np.abs(got[:, :4] - want[:, :4]).max() <= box_tol, np.abs(got[:, :4] - want[:, :4]).max()
    assert np.abs(got[:, :4] - want[:, :4]).mean() <= 0.25
    assert np.abs(got[:, 4:] - want[:, 4:]).max() <= score_tol, np.abs(got[:, 4:] - want[:, 4:]).max()


# (stage, classes, batch sizes of the plan)
F16_CASES = [("car", 1, 64), ("car", 1, 1), ("armor", 12, 256), ("armor", 12, 4)]


@pytest.mark.parametrize("which,nc,n", F16_CASES)
def test_f16_plan_heads_match_the_oracle(rmr, oracle, bench_packs, images, pinned, which, nc, n):
    from oracle import yolov8_ref as R
    pinned("f16", bench_packs)
    pack = bench_packs[0 if which == "car" else 1]
    det = rmr.Detector(pack, nc, (1920, 1080), n, conf_thresh=0.25 if nc == 1 else 0.5)
    got, _ = det.infer([images[i % 3] for i in range(n)])
    det.close()
    want = R.load(pack, True).forward(_blobs(oracle, images))   # f16-emulating oracle
    assert np.isfinite(got).all()
    for i in range(n):
        _check_head(got[i:i + 1], want[i % 3:i % 3 + 1], 2.0, 1e-2)
    for i in range(3, n):   # slots with the same image ran through the same kernels and tiles
        assert np.array_equal(got[i], got[i % 3]), f"slot {i} differs from slot {i % 3}"


FP8_CASES = [("car", 1, 64), ("car", 1, 256), ("armor", 12, 256), ("armor", 12, 4)]


@pytest.mark.parametrize("which,nc,n", FP8_CASES)
def test_fp8_plan_heads_hold_the_bar_of_round_5(rmr, oracle, bench_packs, images, pinned, which, nc, n):
    from oracle import yolov8_ref as R
    pinned("fp8", bench_packs)
    pack = bench_packs[0 if which == "car" else 1]
    det = rmr.Detector(pack, nc, (1920, 1080), n, conf_thresh=0.25 if nc == 1 else 0.5, precision="fp8")
    got, _ = det.infer([images[i % 3] for i in range(n)])
    det.close()
    blobs = _blobs(oracle, images)
    want8 = R.load(pack, fp8=True).forward(blobs)
    jit = R.load(pack, fp8=True, jitter=2.0 ** -22, jitter_seed=1).forward(blobs)
    want16 = R.load(pack, True).forward(blobs)
    floor_b, floor_s = np.abs(jit[:, :4] - want8[:, :4]).mean(), np.abs(jit[:, 4:] - want8[:, 4:]).mean()
    m = min(n, 3)
    eng_b, eng_s = np.abs(got[:m, :4] - want8[:m, :4]).mean(), np.abs(got[:m, 4:] - want8[:m, 4:]).mean()
    print(f"fp8 plan, {which} at {n}: engine vs fp8 oracle {eng_b:.3f} px / {eng_s:.5f}; two exact implementations {floor_b:.3f} / {floor_s:.5f}")
    assert np.isfinite(got).all()
    assert eng_b <= 1.15 * floor_b + 0.02 and eng_s <= 1.15 * floor_s + 1e-5
    assert np.abs(got[:m] - want16[:m]).max() > 0.05   # the e4m3 layers did run
    for i in range(3, n):
        assert np.array_equal(got[i], got[i % 3]), f"slot {i} differs from slot {i % 3}"


def _edit_plan(path, pick, new_choice):
    """Rewrite the first entry `pick(op, n, choice)` accepts to `new_choice`; returns (op, n, old choice)."""
    lines = open(path).read().splitlines()
    for i, l in enumerate(lines[1:], 1):
        op, n, c = (int(v) for v in l.split())
        if pick(op, n, c):
            lines[i] = f"{op} {n} {new_choice}"
            open(path, "w").write("\n".join(lines) + "\n")
            return op, n, c
    raise AssertionError("no such entry in the plan")


def test_a_corrupted_plan_entry_is_an_error(rmr, bench_packs, images, pinned, tmp_path):
    """A 3x3 layer's entry pointed at a conv_pw (1x1-only) kernel id: the pinned detector must refuse, not re-tune."""
    packs = tuple(shutil.copyfile(p, str(tmp_path / os.path.basename(p))) for p in bench_packs)
    pinned("f16", packs)
    # 800..899 = conv_t32 tiles (3x3 / stride-1 layers only); 700.. = conv_pw, which runs 1x1 layers only
    op, n, c = _edit_plan(packs[0] + ".tune", lambda op, n, c: n == 64 and 800 <= c < 900, 700)
    with pytest.raises(rmr.RmrError):
        det = rmr.Detector(packs[0], 1, (1920, 1080), 64)
        try:
            det.infer([images[0]] * 64)
        finally:
            det.close()


def test_the_plan_is_what_runs(rmr, bench_packs, images, pinned, tmp_path):
    """The same four-crop armor batch under the committed plan and under the plan with ONE layer moved to a kernel of another
    family: the heads must move somewhere (another f32 summation order reached the output), i.e. the entries of the file are
    what is launched -- and both stay on the oracle.  At four images the plan runs its 3x3 layers on conv_sb (100000 + v: K
    shared by the waves of a tile, partial sums met in LDS); conv_t32 (800 + tile) sums a value's K in one accumulator.  Not
    every pair of kernels differs in the bits on every layer (conv_halo and conv_t32 were seen to agree bit for bit on a
    256-image layer), so entries are tried in op order until the heads move; a kernel that cannot run the layer it is given is
    an RmrError (never a silent fallback), which is also what test_a_corrupted_plan_entry_is_an_error holds."""
    from oracle import yolov8_ref as R
    import oracle as O
    packs = tuple(shutil.copyfile(p, str(tmp_path / os.path.basename(p))) for p in bench_packs)
    pinned("f16", packs)
    n = 4
    batch = [images[i % 3] for i in range(n)]
    det = rmr.Detector(packs[1], 12, (1920, 1080), n, conf_thresh=0.5)
    a, _ = det.infer(batch)
    det.close()
    plan_text = open(packs[1] + ".tune").read()
    entries = [tuple(int(v) for v in l.split()) for l in plan_text.splitlines()[1:]]
    sb = [(op, c) for op, nn, c in entries if nn == n and 100000 <= c < 200000]
    assert sb, "the plan runs no layer of a four-image batch on conv_sb"
    moved, tried = None, 0
    for op, c in sb[:16]:
        for cand in (812, 810, 806):
            open(packs[1] + ".tune", "w").write(plan_text)
            _edit_plan(packs[1] + ".tune", lambda o, n_, c_: o == op and n_ == n, cand)
            det = rmr.Detector(packs[1], 12, (1920, 1080), n, conf_thresh=0.5)
            try:
                b, _ = det.infer(batch)
            except rmr.RmrError:
                continue
            finally:
                det.close()
            tried += 1
            if not np.array_equal(a, b):
                moved = (op, c, cand, b)
                break
        if moved:
            break
    assert moved, f"{tried} substitutions ran and none moved a bit of the heads: the plan file is not what is launched"
    op, c, cand, b = moved
    want = R.load(packs[1], True).forward(_blobs(O, images))
    for i in range(3):
        _check_head(b[i:i + 1], want[i:i + 1], 2.0, 1e-2)
    print(f"layer {op} at {n} images: kernel {c} -> {cand}; heads differ in {int((a != b).sum())} values")
