"""The headline step itself, checked: bench.py's inputs (BASELINE configs[2]: 64 synthetic 640x640 frames + 30k-point
clouds, K = 4 injected crops per frame) through the one native call the bench times (rmr_pipeline_run_batch ->
update_cluster_batch over 64 frames -> two-stage detect at 64 / 256 images -> search_batch), compared frame by frame with
the CPU oracle (tests/step_parity.py)."""
import os

import numpy as np
import pytest

import scenes
import step_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bench():
    import bench as B
    return B


@pytest.fixture(scope="module")
def step_inputs(bench):
    args = bench.parse([])   # the driver's defaults: batch 64, crops 4, 30k points, 640x640
    assert (args.batch, args.crops, args.points, bench.frame_size(args)) == (64, 4, 30000, (640, 640))
    images, clouds, rects = bench.make_inputs(args, 0)
    return args, images, clouds, rects


@pytest.fixture(scope="module")
def packs(tmp_path_factory):
    from rm_radar_amd import weights as W
    d = tmp_path_factory.mktemp("step_packs")
    car, armor = str(d / "car.rmrw"), str(d / "armor.rmrw")
    W.make_synthetic_pack(car, "m", 1, seed=1, cls_bias=-6.0)      # bench.py's packs
    W.make_synthetic_pack(armor, "m", 12, seed=2, cls_bias=-6.0)
    return car, armor


def _run_steps(bench, oracle, step_inputs, packs, n_steps, device_inputs, assembly_bytes=1 << 28, **det_kw):
    import torch

    import rm_radar_amd as rmr
    args, images, clouds, rects = step_inputs
    B, K, size = args.batch, args.crops, bench.frame_size(args)
    rdet = rmr.RobotDetector(packs[0], packs[1], size, 12, max_cars=K, opt_cars=K, max_frames=B, **det_kw)
    loc = rmr.Locator(size[0], size[1], bench.intrinsic(args), scenes.SAMPLE_L2C, np.eye(4, dtype=np.float32), max_frames=B)
    if device_inputs:   # as the bench: frames and clouds resident in HBM, descriptors marshalled once
        d_images, d_clouds = torch.from_numpy(images).cuda(), torch.from_numpy(clouds).cuda()
        frames = rmr.FrameBatch([d_images[f] for f in range(B)], [d_clouds[f] for f in range(B)])
    else:
        frames = rmr.FrameBatch(list(images), list(clouds))
    forced = np.ascontiguousarray(rects, np.int32)
    cpu = step_parity.oracle_locator(oracle, size, bench.intrinsic(args), scenes.SAMPLE_L2C)
    stats = []
    for _ in range(n_steps):
        robots, counts = rmr.run_batch(rdet, loc, frames, None, forced)
        stats.append(step_parity.check_step(oracle, rmr, rdet, cpu, robots, counts, clouds, forced,
                                            armor_conf=det_kw.get("armor_conf_thresh", 0.5), max_head_bytes=assembly_bytes))
        if len(stats) == 1:   # the step's own network against the torch oracle: one car head, three armor heads
            stats[0]["network"] = step_parity.check_network(oracle, rdet, images, forced, packs, det_kw.get("precision", "f16"))
    rdet.close()
    loc.close()
    return stats


def test_headline_step_matches_oracle(bench, oracle, step_inputs, packs):
    """Three consecutive steps of the bench's own workload on ONE stream (the Locator's background and depth ring carry
    over from step to step, as in the timed loop): located XYZ / presence of every robot of every frame against an
    oracle.Locator fed the same 3 x 64 clouds in order, robot assembly against the oracle on the step's own armor heads."""
    stats = _run_steps(bench, oracle, step_inputs, packs, 3, device_inputs=True)
    for s in stats:
        assert s["frames"] == 64 and s["assembly_frames"] == 64
        assert s["robots"] >= 64
        assert s["max_xyz_err_m"] <= 1e-3
    # frames 2.. of the first step carry robots in front of the background: nearly every robot that survives the per-label
    # grouping (about 100 of the 256 crops: the synthetic armor network votes for few labels) is located
    assert stats[0]["located"] >= 64 and stats[1]["located"] >= 64
    assert all(s["labelled"] >= 32 and s["armors"] >= s["labelled"] for s in stats)


def test_headline_step_under_the_committed_plan(bench, oracle, step_inputs, packs, monkeypatch):
    """The step as the driver times it: bench.apply_plan pins profiles/plans/yolov8m_{car,armor}_f16.tune, and what
    bench.step_parity_leg reports as parity.network_checked is computed here -- locate, assembly AND one car + three armor
    heads of the step against the torch oracle, under the plan's kernels (64-image car chunk, 256-image armor chunk)."""
    import shutil
    monkeypatch.setenv("RMR_PLAN", "")
    d = os.path.dirname(packs[0])
    mine = tuple(shutil.copyfile(p, os.path.join(d, "planned_" + os.path.basename(p))) for p in packs)
    args = step_inputs[0]
    try:
        assert bench.apply_plan(args, mine) is not None and os.environ.get("RMR_PLAN") == "1"
        s = _run_steps(bench, oracle, step_inputs, mine, 1, device_inputs=True)[0]
    finally:
        os.environ.pop("RMR_PLAN", None)
    assert s["frames"] == 64 and s["assembly_frames"] == 64 and s["max_xyz_err_m"] <= 1e-3
    assert s["network"]["car_slots"] == [0] and len(s["network"]["armor_slots"]) == 3
    assert s["network"]["max_box_err_px"] <= 2.0 and s["network"]["max_score_err"] <= 1e-2


def test_headline_step_with_labels(bench, oracle, step_inputs, packs):
    """The same step from HOST-resident frames and clouds (the staging path of detector.cu:388) and with a lower armor
    threshold, placed inside the score range of the step's heads: several times more armors survive decode + NMS, more
    robots carry labels and more same-label robots of a frame are grouped -- still bit-equal to the oracle's assembly on
    the GPU's own heads."""
    import rm_radar_amd as rmr
    args, images, clouds, rects = step_inputs
    # where the threshold has to be: the scores of a few crops of the step
    probe = rmr.Detector(packs[1], 12, (640, 640), 8)
    heads, _ = probe.infer([images[f] for f in range(8)], crops=[tuple(int(v) for v in rects[f, 0]) for f in range(8)])
    probe.close()
    best = np.sort(heads[:, 4:].max(1).reshape(-1))
    t = min(0.35, float(best[-400]))   # a few hundred candidate anchors per 8 crops
    assert 0.0 < t
    stats = _run_steps(bench, oracle, step_inputs, packs, 1, device_inputs=False, armor_conf_thresh=t)
    s = stats[0]
    assert s["assembly_frames"] == 64 and s["labelled"] >= 16 and s["armors"] >= s["labelled"]


def test_headline_step_locates_every_injected_robot(bench, oracle, step_inputs, packs):
    """with the default threshold about 100 of a step's 256 injected robots survive the per-label grouping
    (the reference keeps ONE robot per label and frame, detector.cpp:427-454), so search_batch was checked on 40 % of them.
    With an armor threshold no score reaches, no robot gets a label, grouping keeps all four of every frame, and every one
    of the 256 rects is searched for and compared with the oracle (presence identical, XYZ <= 1e-3 m)."""
    stats = _run_steps(bench, oracle, step_inputs, packs, 2, device_inputs=True, armor_conf_thresh=0.9999)
    for s in stats:
        assert s["frames"] == 64 and s["robots"] == 256 and s["labelled"] == 0 and s["assembly_frames"] == 64
        assert s["max_xyz_err_m"] <= 1e-3
    assert stats[1]["located"] >= 200     # the second step's 256 robots stand in front of a settled background


def test_config3_step_matches_oracle(bench, oracle, packs):
    """BASELINE configs[3]'s frame shape through the same native call (VERDICT r03 missing #2): 1920 x 1080 frames + 100 k-point
    clouds, `bench.py --config 3 --batch 8` -> rmr_pipeline_run_batch -> step_parity.check_step.  The first layer samples
    1920 x 1080 sources (crops included), the Locator runs on 960 x 540 depth images with the 66-degree intrinsic of
    bench.intrinsic; located XYZ <= 1e-3 m with identical presence (locate.cpp:276-311), robot assembly bit-equal to the
    oracle on the step's own armor heads (detector.cpp:413-455).  Two steps: the second sees the first's background."""
    args = bench.parse(["--config", "3", "--batch", "8"])
    assert (args.points, bench.frame_size(args)) == (100000, (1920, 1080))
    images, clouds, rects = bench.make_inputs(args, 0)
    assert images.shape == (8, 1080, 1920, 3) and clouds.shape == (8, 100000, 4)
    stats = _run_steps(bench, oracle, (args, images, clouds, rects), packs, 2, device_inputs=True)
    for s in stats:
        assert s["frames"] == 8 and s["assembly_frames"] == 8
        assert s["robots"] >= 8
        assert s["max_xyz_err_m"] <= 1e-3
    assert stats[0]["located"] >= 6 and stats[1]["located"] >= 8   # frames 0-1 of step one are background-only clouds


def test_two_ranks_through_the_whole_bench_flow_on_one_gpu():
    """`bench.py --gpus 2 --share-gpu`: the N > 1 control flow of the bench's main() for real -- the launcher starts two ranks
    under torch.distributed.run, each owns its own streams' frames, detectors and Locator, the timed region is bracketed by
    barriers, the clock is the slower rank's, the robot records travel through the C-ABI communicator (FILE transport here,
    RCCL on a real multi-GPU node), and the legs only rank 0 runs (stage breakdown, upload ring) must not call a collective.
    Both ranks compute on GPU 0, so the line carries "share_gpu": true and its rate says nothing about scaling."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-latency", "--no-parity", "--seconds", "0", "--profile-steps", "1"],
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["share_gpu"] is True and line["ranks_seen"] == [0, 1]
    assert len(line["per_rank_frames_per_s"]) == 2 and line["value"] > 0
    assert "FILE transport" in line["gather"], line["gather"]
    assert line["stage_ms_per_step"]["network, armor stage"] > 0        # rank 0's own leg ran to the end
    assert line["value"] <= sum(line["per_rank_frames_per_s"]) * 1.001    # whole-job rate from the slower rank's clock


def test_config1_bench_line_is_parity_checked():
    """`bench.py --config 1` (BASELINE configs[1]: batch 1 on the reference sample's 2592 x 2048 frames + its clouds from host memory,
    sample calibration, background update first -- samples/main.cpp:12-22,87): the whole mode end to end with a short latency
    sample.  The line must carry its own parity verdict -- locate and robot assembly of the three frames against the oracle and
    the network heads of the first against the torch oracle -- and the split of the frame into staging + H2D and the rest."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "1", "--latency-frames", "10", "60", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert "configs[1]" in line["config"]["workload"] and "2592x2048" in line["config"]["workload"]
    assert line["parity_checked"] is True, line["parity"]
    assert line["parity"]["frames"] == 3 and line["parity"]["assembly_frames"] == 3 and line["parity"]["network_checked"] is True
    assert line["parity"]["max_xyz_err_m"] <= 1e-3 and line["parity"]["network"]["max_box_err_px"] <= 2.0
    assert line["parity"]["located"] >= 1                       # the injected returns in front of the background are found
    assert 0.5 < line["p50_ms_batch1"] < 20 and line["p99_ms_batch1"] >= line["p50_ms_batch1"]
    split = line["stage_split_ms_per_frame"]
    assert 0.0 <= split["h2d_fraction_of_frame"] < 0.6
    assert split["kernels (events around every launch, inputs in HBM)"]["network, armor stage"] > 0
