"""Parity of ONE throughput-mode step as the bench runs it (rmr_pipeline_run_batch over a batch of frames of one
stream with injected crops) against the CPU oracle.  Test infrastructure: used by tests/test_gpu_bench_step.py and by
bench.py's cpu_baseline leg (which reports `parity_checked`); never by the product path.

What is compared, per frame f of the step:
  * locate (src/locate/locate.cpp:158-326): an oracle.Locator fed the same clouds in the same order; every robot the GPU
    returned is searched for with its own rect -- presence identical, XYZ within 1e-3 m (north_star);
  * robot assembly (src/detect/detector.cu:522-582, detector.cpp:258-268, 427-454, robot.cpp:41-74): the armor network's
    head tensors of this very step are read back from the GPU (rmr_robot_detector_read_heads) and pushed through the
    oracle's decode + NMS + restore, Robot construction and per-label grouping -- rects, labels, confidences and armor boxes
    of the GPU's robots must equal that bit for bit (the network itself is held to the torch oracle in test_gpu_network.py;
    here the question is whether the step, at this batch size, assembled what its own network produced).
"""
import numpy as np


def oracle_locator(oracle, size, intrinsic, l2c, w2c=None):
    return oracle.Locator(size[0], size[1], intrinsic, l2c, np.eye(4, dtype=np.float32) if w2c is None else w2c)


def check_step(oracle, rmr, rdet, cpu_loc, robots_c, counts, clouds, rects, armor_conf=0.5, armor_nms=0.65, iou_thresh=0.75,
               classes=12, check_assembly=True, max_head_bytes=1 << 28):
    """robots_c / counts: what rmr.run_batch returned for this step; cpu_loc: the oracle Locator holding the stream's
    history up to (not including) this step -- it is advanced over `clouds` here.  rects: int [n_frames, K, 4] injected crops.
    Returns counters; raises AssertionError at the first difference."""
    n_frames = len(counts)
    cap = rdet.max_cars
    K = rects.shape[1]
    got = [[rmr.Robot.from_c(robots_c[f * cap + i]) for i in range(int(counts[f]))] for f in range(n_frames)]
    stat = {"frames": n_frames, "robots": 0, "located": 0, "not_located": 0, "max_xyz_err_m": 0.0, "labelled": 0,
            "armors": 0, "assembly_frames": 0}

    # ---- locate
    for f in range(n_frames):
        cpu_loc.update(clouds[f])
        cpu_loc.cluster()
        forced = {tuple(float(v) for v in r) for r in rects[f]}
        seen = set()
        assert 1 <= len(got[f]) <= K or K == 0, f"frame {f}: {len(got[f])} robots from {K} crops"
        for g in got[f]:
            assert g.rect in forced, f"frame {f}: robot rect {g.rect} is not an injected crop"
            assert g.rect not in seen, f"frame {f}: rect {g.rect} twice"
            seen.add(g.rect)
            want = cpu_loc.search(g.rect)
            assert (want is None) == (g.location is None), \
                f"frame {f} rect {g.rect}: oracle {'not ' if want is None else ''}located, GPU {'not ' if g.location is None else ''}located"
            stat["robots"] += 1
            if want is None:
                stat["not_located"] += 1
                continue
            err = float(np.max(np.abs(np.asarray(g.location, np.float64) - want.astype(np.float64))))
            assert err <= 1e-3, f"frame {f} rect {g.rect}: GPU {g.location} oracle {tuple(want)} ({err:.2e} m)"
            stat["located"] += 1
            stat["max_xyz_err_m"] = max(stat["max_xyz_err_m"], err)

    # ---- robot assembly from the GPU's own armor heads
    if check_assembly and K > 0:
        per_img = (4 + classes) * 8400 * 4
        frames_per_read = max(1, int(max_head_bytes // (per_img * K)))
        for f0 in range(0, n_frames, frames_per_read):
            f1 = min(n_frames, f0 + frames_per_read)
            heads, pps = rdet.read_heads(1, f0 * K, (f1 - f0) * K)   # every injected crop is non-empty: slot = f * K + k
            for f in range(f0, f1):
                robots = []
                for k in range(K):
                    i = (f - f0) * K + k
                    r = rects[f, k]
                    assert (pps[i].width, pps[i].height) == (float(r[2]), float(r[3])), "armor batch slot order"
                    armors = oracle.postprocess(heads[i], classes, armor_nms, armor_conf, oracle.PreParam(*pps[i].astuple()))
                    car = (float(r[0]), float(r[1]), float(r[2]), float(r[3]), 0.0, 1.0)
                    robots.append(oracle.make_robot(car, armors))
                want = oracle.group_robots(robots, iou_thresh)
                assert len(want) == len(got[f]), f"frame {f}: oracle groups {len(want)} robots, GPU {len(got[f])}"
                for w, g in zip(want, got[f]):
                    assert tuple(w.rect) == g.rect, f"frame {f}: robot order / rect {tuple(w.rect)} vs {g.rect}"
                    assert (w.label if w.has_label else None) == g.label, f"frame {f} rect {g.rect}: label"
                    if w.n_armors > 0:
                        assert g.armors is not None and len(g.armors) == w.n_armors
                        assert np.float32(w.confidence) == np.float32(g.confidence), f"frame {f} rect {g.rect}: confidence"
                        wa = np.array([(a.x, a.y, a.width, a.height, a.label, a.confidence) for a in w.armors[: w.n_armors]],
                                      g.armors.dtype)
                        assert wa.tobytes() == g.armors.tobytes(), f"frame {f} rect {g.rect}: armor boxes"
                        stat["armors"] += int(w.n_armors)
                        stat["labelled"] += 1
                    else:
                        assert g.armors is None
                stat["assembly_frames"] += 1
    return stat


def check_network(oracle, rdet, images, rects, packs, dtype="f16", car_slots=(0,), armor_slots=None, box_tol=2.0, score_tol=1e-2):
    """The NETWORK of the step that was just run (under whatever kernel plan the caller pinned), held to the torch oracle
    (oracle/yolov8_ref.py, the restatement of the engine behind src/detect/detector.h:122): a few head slots of the car batch
    and of the armor batch are read back (rmr_robot_detector_read_heads) and compared with the oracle's forward of the same
    letterboxed pixels on the same pack.  f16 plan: the f16-emulating oracle, boxes within `box_tol` px (mean 0.25), scores
    within `score_tol` -- test_gpu_network._check_head's bars.  fp8 plan: the fp8-emulating oracle, no further than 1.15 x the
    distance between two exact implementations of the plan (the oracle against itself with every convolution result moved by
    2^-22), the absolute bar of round 5.  images: the step's frames (u8 HWC), rects: int [n_frames, K, 4] injected crops.
    Returns counters; raises AssertionError at the first slot out of tolerance."""
    from oracle import yolov8_ref as R
    n_frames, K = len(images), rects.shape[1]
    if armor_slots is None:
        armor_slots = sorted({0, (n_frames * K) // 2 + 1, n_frames * K - 1}) if K > 0 else ()
    stat = {"car_slots": list(car_slots), "armor_slots": list(armor_slots), "max_box_err_px": 0.0, "max_score_err": 0.0}
    for stage, pack, slots in ((0, packs[0], car_slots), (1, packs[1], armor_slots)):
        if not len(slots):
            continue
        blobs = []
        for s in slots:
            if stage == 0:
                blobs.append(oracle.preprocess(np.asarray(images[s]))[0])
            else:
                f, k = divmod(s, K)
                blobs.append(oracle.preprocess(np.asarray(images[f]), crop=tuple(int(v) for v in rects[f, k]))[0])
        blobs = np.stack(blobs)
        got = np.concatenate([rdet.read_heads(stage, s, 1)[0] for s in slots])
        assert np.isfinite(got).all(), f"stage {stage}: non-finite head values"
        if dtype == "f16":
            want = R.load(pack, True).forward(blobs)
            for i, s in enumerate(slots):
                box, score = np.abs(got[i, :4] - want[i, :4]), np.abs(got[i, 4:] - want[i, 4:])
                assert box.max() <= box_tol and box.mean() <= 0.25 and score.max() <= score_tol, \
                    f"stage {stage} slot {s}: boxes {box.max():.3f} px (mean {box.mean():.3f}), scores {score.max():.5f} from the f16 oracle"
                stat["max_box_err_px"] = max(stat["max_box_err_px"], float(box.max()))
                stat["max_score_err"] = max(stat["max_score_err"], float(score.max()))
        else:
            want = R.load(pack, fp8=True).forward(blobs)
            jit = R.load(pack, fp8=True, jitter=2.0 ** -22, jitter_seed=1).forward(blobs)
            floor_b, floor_s = np.abs(jit[:, :4] - want[:, :4]).mean(), np.abs(jit[:, 4:] - want[:, 4:]).mean()
            eng_b, eng_s = np.abs(got[:, :4] - want[:, :4]).mean(), np.abs(got[:, 4:] - want[:, 4:]).mean()
            assert eng_b <= 1.15 * floor_b + 0.02 and eng_s <= 1.15 * floor_s + 1e-5, \
                f"stage {stage}: engine {eng_b:.3f} px / {eng_s:.5f} from the fp8 oracle, two exact implementations {floor_b:.3f} / {floor_s:.5f}"
            stat["max_box_err_px"] = max(stat["max_box_err_px"], float(eng_b))
            stat["max_score_err"] = max(stat["max_score_err"], float(eng_s))
            stat["bar"] = "mean distance from the fp8 oracle <= 1.15 x that of a second exact implementation"
    return stat
