"""The committed kernel plans (profiles/plans/*.tune) and the tools around them -- CPU side.

bench.py, tools/round_profile.sh and tools/pmc_refresh.sh run under these files (RMR_PLAN), so that the driver's line, the
rocprofv3 kernel stats and the PMC traffic describe the same launches.  A plan of another file version is ignored by the
library (bench.py then autotunes and says so): this test makes a stale committed plan a red test instead of a silent
fallback.  The parsers of tools/make_plan.py / tools/pmc_traffic.py are checked on hand-written logs."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLANS = os.path.join(ROOT, "profiles", "plans")


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("name", ["yolov8m_car_f16", "yolov8m_armor_f16", "yolov8m_car_fp8", "yolov8m_armor_fp8"])
def test_committed_plan_is_of_this_library_version_and_consistent(name):
    from rm_radar_amd import _lib
    version = _lib.lib().rmr_tune_file_version()
    lines = open(os.path.join(PLANS, name + ".tune")).read().splitlines()
    head = lines[0].split()
    assert head[0] == "rmr-tune" and int(head[1]) == version, f"{name}: plan written for file version {head[1]}, library reads {version}"
    assert (int(head[3]), int(head[4])) == (640, 640) and head[7].startswith("gfx950")
    entries = [tuple(int(v) for v in l.split()) for l in lines[1:]]
    assert len(entries) == len(set((op, n) for op, n, _ in entries)), "duplicate (layer, batch) entries"
    sizes = sorted({n for _, n, _ in entries})
    want = {"car_f16": [1, 64], "armor_f16": [4, 256], "car_fp8": [1, 64, 256], "armor_fp8": [4, 256]}[name.split("_", 1)[1]]
    assert sizes == want                       # the bench's chunks, the batch-1 latency leg, configs[4]
    by = {(op, n): c for op, n, c in entries}
    per_size = {n: sum(1 for _, m, _ in entries if m == n) for n in sizes}
    assert len(set(per_size.values())) == 1    # every layer at every batch size
    for (op, n), c in by.items():
        if c == 399:                           # "done by the layer before": only behind a fused bottleneck (340..)
            assert 340 <= by.get((op - 1, n), -1) < 399, (op, n)
        if 340 <= c < 398:
            assert by.get((op + 1, n)) == 399, (op, n)
        if c == 398:                           # "done by the group's launch": behind a grouped launch (200000..) a few layers back
            k = op - 1
            while by.get((k, n)) == 398:
                k -= 1
            assert by.get((k, n), -1) >= 200000 and op - k < 8, (op, n)
        if c >= 200000:                        # a grouped launch carries at least one more layer
            assert by.get((op + 1, n)) == 398, (op, n)


def test_make_plan_parses_the_tuners_log(tmp_path):
    mp = _load("make_plan")
    log = tmp_path / "tune.log"
    log.write_text(
        "some other stderr line\n"
        "tune M409600 N192 K1728 k3 s1: 200:350.1 206:340.0 800:312.4 806:305.9 810:301.2 [810:300.4] [806:302.0] [800:311.0]  -> 810 (300.4 us)\n"
        "tune M6553600 N48 K432 k3 s1: 300:442.0 312:397.5  -> 312 (397.5 us)\n"
        "fuse layers 3 + 4 at 256 images: two launches 696.3 us, conv_wsf variant 1 678.8 us -> fused\n"
        "tune M6400 N192 K1728 k3 s1: 215:20.1 2812:21.5  -> 215 (20.1 us)\n")
    t = mp.parse_tuning(str(log))
    assert [c for c, _, _ in t] == [810, 312, 215]
    assert t[0][1][810] == 300.4 and t[0][1][806] == 302.0 and t[0][1][206] == 340.0   # run-off times replace first-pass times
    assert t[2][1] == {215: 20.1, 2812: 21.5}                                             # split-K ids are plain integers
    assert mp.parse_fusions(str(log)) == {3: (4, 1)}


def test_make_plan_reads_per_launch_times_in_order(tmp_path):
    mp = _load("make_plan")
    order = tmp_path / "order.txt"
    rows = []
    for rep in range(2):
        rows += [f"2 |conv n256 M26214400 N48 K72 k3 s2 stem+letterbox|1|1|{0.9 + rep}",      # not in the plan: skipped
                 f"2 armor|conv n256 M6553600 N96 K432 k3 s2 v0|5|7|{1.0 - 0.1 * rep:.3f}",
                 f"1 |loc_scatter|0|0|0.01",                                                    # level 1: another profile
                 f"2 armor|conv n256 M6553600 N96 K96 k1 s1 p0|5|7|{0.5 + 0.1 * rep:.3f}"]
    order.write_text("\n".join(rows) + "\n")
    ms, names = mp.per_op_times(str(order), 2)
    assert ms == [0.9, 0.5] and names[0].endswith("v0") and names[1].endswith("p0")        # minimum over the forwards, op order
    with pytest.raises(AssertionError):
        mp.per_op_times(str(order), 3)


def test_pmc_traffic_finds_the_steps_launch_sequence(tmp_path):
    ns = vars(_load("pmc_traffic"))
    order = tmp_path / "launch_order.txt"
    seq = ["conv a", "conv b", "conv b", "conv c"]
    lines = [f"2 car|{n}|10|20|0.1" for n in seq * 3] + ["1 |postprocess|0|0|0.01"]
    order.write_text("\n".join(lines) + "\n")
    period = ns["step_order"](str(order))
    assert [r[1] for r in period] == seq and period[0][2:] == (10.0, 20.0)


def test_pin_plan_takes_every_entry_from_one_batch_size(tmp_path):
    """a group's launch (200000 + v) and its members' 398 markers must come from the same tuned batch, split-K ids
    lose their split, and small-batch kernels are not pinned for batches beyond the one they were tuned at."""
    import rm_radar_amd as rmr
    tune = tmp_path / "a.tune"
    tune.write_text("rmr-tune 17 4 640 640 1 256 gfx950\n"
                    "0 1 100040\n1 1 200009\n2 1 398\n3 1 2806\n"
                    "0 4 806\n1 4 810\n2 4 812\n3 4 3810\n"
                    "0 8 806\n1 8 810\n")                      # 8 images: only half the layers tuned -> not the reference size
    plan = rmr.pin_plan(str(tune), str(tmp_path / "a.plan"), (1, 2, 4))
    rows = [tuple(int(v) for v in l.split()) for l in open(plan).read().splitlines()[1:]]
    assert rows == [(0, 1, 806), (0, 2, 806), (0, 4, 806), (1, 1, 810), (1, 2, 810), (1, 4, 810),
                    (2, 1, 812), (2, 2, 812), (2, 4, 812), (3, 1, 810), (3, 2, 810), (3, 4, 810)]
    small = tmp_path / "b.tune"
    small.write_text("rmr-tune 17 3 640 640 1 256 gfx950\n0 1 100040\n1 1 200009\n2 1 398\n")
    rows = open(rmr.pin_plan(str(small), str(tmp_path / "b.plan"), (1,))).read().splitlines()[1:]
    assert rows == ["0 1 100040", "1 1 200009", "2 1 398"]   # the group stays whole
    with pytest.raises(ValueError):
        rmr.pin_plan(str(small), str(tmp_path / "c.plan"), (1, 64))
