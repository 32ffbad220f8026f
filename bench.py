#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X: frames/sec of the per-frame
detect + locate hot path (640x640 BGR frame + 30k-point cloud), synthetic data.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one batch of BATCH frames (default 64, BASELINE configs[2]: "Batch=64 synthetic
640x640 frames + 30k-pt clouds on one MI355X (throughput mode)") through the whole hot path:
  locate   per frame, in stream order (the Locator is stateful): update -> cluster -> keep
  detect   car YOLOv8m over the 64 frames, armor YOLOv8m over 64*CROPS crops, decode+NMS
  search   per frame: robots located against that frame's kept clusters
  gather   (N > 1) one RCCL all-gather of the fixed-size robot records
Inputs are resident in HBM before the timed region.  Weights are seeded synthetic YOLOv8m
(the reference's car.onnx / armor.onnx are absent); with those the car stage yields arbitrary
boxes, so CROPS fixed crop rects per frame are injected for the armor stage -- the car stage still
runs in full (forward + decode + NMS).  Multi-GPU: streams shard across ranks (weak scaling: every
rank processes its own BATCH frames per step), no data-path collective.

Measurement layout (one process):
  1. untimed: tuning call + W warm-up steps;
  2. HEADLINE: exactly K steps, no profiling events anywhere, barrier + synchronize on both sides -> value;
  3. steady state: further un-profiled steps until --seconds have been timed in total (the driver's 20 steps
     are under a second; SURVEY 8d asks for >= 10 s) -> steady_state;
  4. roofline: a few more steps with HIP events around every convolution launch on the stream it runs on,
     named per GEMM shape -> roofline (dominant kernel family) and layer_roofline (every layer against
     max(FLOPs / MFMA peak, algorithmic bytes / HBM peak));
  5. H2D: the step's inputs (frames + clouds) copied from pinned host memory, timed on their own ->
     h2d_ms_per_step and value_incl_h2d (never `value`: inputs are resident in HBM in the timed region);
  6. batch-1 latency with host inputs; the CPU baseline on a bounded sample.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

os.environ.setdefault("RMR_PROFILE_LAYERS", "1")  # per-GEMM-shape names in the profiled loop (read when librmr loads)

F16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense f16/bf16 MFMA
HBM_PEAK_TBS = 8.0              # MI355X_MICROARCH.md: 8 TB/s (6.3 achievable)
# what a chip-filling loop of nothing but v_mfma_f32_32x32x16_f16 delivers on random f16 operands before the
# power limit clocks it down (tools/microbench/mfma_power.hip, profiles/r02_mfma_power.txt); zeros: 2400
F16_POWER_LIMITED_TFLOPS = 1650.0
FP8_DENSE_PEAK_TFLOPS = 5000.0  # MI355X_MICROARCH.md: ~5 PF dense fp8 MFMA (the e4m3 layers of the fp8 plan run on this pipe)

# the tag at the end of a per-layer profile name ("conv n256 M409600 N192 K1728 k3 s1 g10") -> the kernel template it is an
# instantiation of; tag + number = one instantiation = one symbol in a rocprofv3 kernel trace
KERNEL_OF_TAG = {"g": "conv_t32_kernel (csrc/conv_t32.hip)", "m": "conv_g32_kernel (csrc/conv_g32.hip)",
                 "p": "conv_pw_kernel (csrc/conv_pw.hip)", "w": "conv_ws_kernel (csrc/conv_ws.hip)",
                 "v": "conv_ws_s2_kernel (csrc/conv_ws_s2.hip)", "d": "conv_dma_kernel (csrc/conv_dma.hip)",
                 "t": "conv_igemm_kernel (csrc/conv_igemm.hip)", "h": "conv_halo_kernel (csrc/conv_halo.hip)",
                 "x": "conv_direct_kernel (csrc/conv_direct.hip)", "f": "conv_t32f8_kernel (csrc/conv_t32f8.hip)",
                 "s": "conv_stem_kernel (csrc/conv_stem.hip)", "b": "conv_wsf_kernel (csrc/conv_ws.hip: a fused C2f bottleneck)",
                 "y": "head_fused_kernel (csrc/net_ops.hip: the Detect head's last 1x1 convolutions + DFL decode)"}


def instantiation_of(layer_name):
    """'conv n256 M409600 N192 K1728 k3 s1 g10' -> 'g10' ('stem+letterbox' -> 'stem'; split-K 'd44/2' -> 'd44')"""
    tag = layer_name.split()[-1]
    return "stem" if tag.startswith("stem") else tag.split("/")[0]


def source_hash():
    """Hash of the kernel sources: profiles/*_pmc_conv_traffic.json records the one it was measured on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rm_radar_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".cpp", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--crops", type=int, default=4)   # K: kOptBatchSize, sample_radar.h:34
    ap.add_argument("--points", type=int, default=30000)
    ap.add_argument("--size", type=int, default=640, help="frame width (and height unless --height is given)")
    ap.add_argument("--height", type=int, default=0, help="frame height, e.g. --size 1920 --height 1080 for configs[3]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--latency-frames", type=int, nargs=2, default=(200, 2000), metavar=("WARMUP", "TIMED"),
                    help="batch-1 latency leg: untimed and timed frames (SURVEY 8d: 200 + 2000)")
    ap.add_argument("--cpu-frames", type=int, default=1, help="cpu_baseline: timed frames per worker")
    ap.add_argument("--cpu-workers", type=int, default=0, help="cpu_baseline: frames in flight (0 = cores / 4)")
    ap.add_argument("--launch-order", default="", help="write the enqueue order of the profiled launches (layer, FLOPs, bytes) to this "
                    "file (RMR_PROFILE_ORDER): what tools/pmc_traffic.py maps the dispatches of a rocprofv3 --pmc pass with")
    ap.add_argument("--no-parity", action="store_true", help="skip the step-parity leg (parity_checked: false)")
    ap.add_argument("--plan", default="auto", help="auto: run under the committed pinned plan (profiles/plans/) when it fits "
                    "this build and workload, else autotune; tune: always autotune; <dir>: plans from that directory")
    ap.add_argument("--no-profile", action="store_true", help="skip the profiled loop (no roofline objects)")
    ap.add_argument("--seconds", type=float, default=10.0, help="steady state: keep stepping until this much time is measured")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--gather", choices=("abi", "torch"), default="abi",
                    help="N > 1: the robot-record all-gather through the library's C-ABI communicator (rmr_comm_*, RCCL) "
                         "or through torch.distributed (RCCL as well)")
    ap.add_argument("--streams", type=int, default=1, help="camera / LiDAR streams per GPU: the batch divides into this many "
                    "streams, each with its own Locator state (background image + depth ring in HBM), one detector batch over all")
    ap.add_argument("--dtype", choices=("f16", "fp8"), default="f16",
                    help="fp8 = BASELINE configs[4]: e4m3 weights and activations in the 3x3 layers of backbone and neck")
    ap.add_argument("--config", type=int, default=2, help="BASELINE configs index: 1 = batch 1 on the reference sample's 2592x2048 "
                    "frames + 10k-point clouds from host memory (latency); 2 = 640x640 + 30k points; 3 = one 1920x1080 "
                    "stream + 100k-point clouds per GPU; 4 = the fp8 plan at 256 frames per step")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST HOOK (tests/test_gpu_bench_step.py): every rank computes on GPU 0, the process group is gloo and the "
                         "robot-record exchange the C-ABI FILE transport -- the whole N > 1 control flow of main() (shards per rank, "
                         "barriers, max-over-ranks clock, rank-0-only legs) on a one-GPU box; everything except RCCL itself.  The "
                         "line says \"share_gpu\": true: ranks that share a chip do not measure scaling")
    ap.add_argument("--stub-step", action="store_true",
                    help="TEST HOOK (tests/test_bench_launcher.py): the launcher, process group (gloo), C-ABI communicator (FILE "
                         "transport), barriers, max-over-ranks clock and JSON assembly run for real on CPU; the GPU step is replaced "
                         "by a stand-in that only fabricates robot records.  The line says \"stub\": true and measures nothing")
    args = ap.parse_args(argv)
    if args.config == 1:   # the reference sample's frames: 2592 x 2048 (samples/main.cpp:12), one frame per call
        args.size, args.height, args.batch = 2592, 2048, 3
    if args.config == 3:
        args.size, args.height, args.points = 1920, 1080, 100000
    if args.config == 4:   # fp8-MFMA weights, batch = 256
        args.dtype, args.batch = "fp8", 256
    return args


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start N ranks of this same script, one per GPU, under
    torch.distributed.run -- the command form the contract gives for N > 1 -- and hand back its exit code.  A box
    with fewer than N GPUs is an error, never a silent one-rank run."""
    import subprocess
    if not args.stub_step and not args.share_gpu:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs, this node has {have}; "
                             "refusing to run fewer ranks than asked for\n")
            return 3
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def frame_size(args):
    return (args.size, args.height or args.size)


def intrinsic(args):
    """640x640: the survey's K; other sizes: the same 66-degree horizontal field of view."""
    import scenes
    w, h = frame_size(args)
    if (w, h) == (640, 640):
        return scenes.K640
    if (w, h) == scenes.SAMPLE_SIZE:
        return scenes.SAMPLE_K     # samples/main.cpp:13-14
    f = 416.0 * w / 640.0
    return np.array([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1]], np.float32)


def crop_rects(rng, n_frames, k, size):
    W, H = size
    out = np.zeros((n_frames, k, 4), np.int32)
    for f in range(n_frames):
        for i in range(k):
            w = int(rng.integers(W // 8, W // 3))
            h = int(rng.integers(H // 8, H // 3))
            out[f, i] = (int(rng.integers(0, W - w)), int(rng.integers(0, H - h)), w, h)
    return out


def make_inputs(args, rank):
    """BATCH frames of one camera/LiDAR stream: images + clouds + robot rects (numpy)."""
    import scenes
    size = frame_size(args)
    rng = np.random.default_rng(100 + rank)
    rects = crop_rects(rng, args.batch, args.crops, size)
    images = np.stack([scenes.synthetic_image(rank * 10000 + f, size) for f in range(args.batch)])
    clouds = np.zeros((args.batch, args.points, 4), np.float32)
    crng = np.random.default_rng(200 + rank)
    for f in range(args.batch):
        robots = [(tuple(float(v) for v in r), float(crng.uniform(1000, 3000)), int(crng.integers(40, 400)))
                  for r in rects[f]] if f >= 2 else []
        clouds[f] = scenes.make_cloud(crng, args.points, intrinsic(args), scenes.SAMPLE_L2C, size, robots)
    return images, clouds, rects


def step_parity_leg(args, rmr, rdet, frames, forced, clouds, local, images=None, packs=None):
    """Part of the cpu_baseline leg (the only place bench.py touches the oracle, and only as the checker): ONE more step
    of the timed configuration on a fresh Locator, compared with the CPU oracle frame by frame -- located XYZ / presence
    of every robot (<= 1e-3 m) and the robot assembly on the step's own armor heads (bit-exact); tests/step_parity.py,
    the same check tests/test_gpu_bench_step.py runs.  Outside every timed region."""
    import oracle
    import scenes
    import step_parity
    size = frame_size(args)
    loc = rmr.Locator(size[0], size[1], intrinsic(args), scenes.SAMPLE_L2C, np.eye(4, dtype=np.float32), device=local,
                      max_frames=args.batch)
    try:
        robots, counts = rmr.run_batch(rdet, loc, frames, None, forced)
        cpu = step_parity.oracle_locator(oracle, size, intrinsic(args), scenes.SAMPLE_L2C)
        stat = step_parity.check_step(oracle, rmr, rdet, cpu, robots, counts, clouds, forced)
        stat["network_checked"] = False
        if images is not None and packs is not None:
            # the NETWORK of this very step, under the plan it was timed with: one car head and three armor heads against
            # the torch oracle (f16-emulating; fp8: the round-5 bar) -- a wrong entry in a committed plan turns this red
            net = step_parity.check_network(oracle, rdet, images, forced, packs, args.dtype)
            stat["network_checked"] = True
            stat["network"] = net
        return True, {k: (round(v, 9) if isinstance(v, float) else v) for k, v in stat.items()}
    except AssertionError as e:
        return False, {"error": str(e)[:300]}
    except Exception as e:  # noqa: BLE001 -- a broken checker must not cost the bench line
        return False, {"error": f"checker failed: {type(e).__name__}: {e}"[:300]}
    finally:
        loc.close()


def cpu_baseline(args, packs, images, clouds, rects):
    """The same frames on the host CPU: C oracle (pre / decode+NMS / locate) + PyTorch-CPU fp32 YOLOv8m (oneDNN) standing in
    for ONNX-Runtime-CPU + PCL, which this image lacks.  SURVEY 8d: frames in parallel -- W workers (threads: torch's CPU
    ops and the ctypes oracle both release the GIL), each with its own Locator stream and its own share of the frames,
    torch intra-op threads = cores / W per worker; cores = what this process may run on (sched_getaffinity).  W = cores / 4
    by default (at most one worker per frame of the batch), one frame per worker: about 40 s on the GPU box's 256 cores."""
    from concurrent.futures import ThreadPoolExecutor

    import torch

    import oracle
    import scenes
    from oracle import yolov8_ref as R
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # measured on the GPU box's 256 cores (frames/s at one frame per worker): 16 x 16 threads 0.61, 32 x 8 0.86, 64 x 4 1.59 --
    # the more frames in flight the better oneDNN's batch-1 / batch-4 convolutions use the cores: cores / 4 workers
    workers = max(1, min(args.cpu_workers or max(1, cores // 4), cores, args.batch))
    threads = max(1, cores // workers)
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(threads)
    car, armor = R.load(packs[0]), R.load(packs[1])
    locs = [oracle.Locator(*frame_size(args), intrinsic(args), scenes.SAMPLE_L2C, np.eye(4, dtype=np.float32)) for _ in range(workers)]

    def one_frame(w, f):
        img, loc = images[f], locs[w]
        loc.update(clouds[f])
        loc.cluster()
        blob, p = oracle.preprocess(img)
        oracle.postprocess(car.forward(blob[None])[0], 1, 0.65, 0.25, p)
        if len(rects[f]):
            blobs, pps = zip(*[oracle.preprocess(img, crop=tuple(int(v) for v in r)) for r in rects[f]])
            outs = armor.forward(np.stack(blobs))
            for o, pc in zip(outs, pps):
                oracle.postprocess(o, 12, 0.65, 0.5, pc)
        for r in rects[f]:
            loc.search(tuple(float(v) for v in r))

    # ONE executor for the warm-up and the timed pass (the timed pass's threads must be the warmed ones).  Untimed
    # warm-up: worker 0 runs one whole frame alone (oneDNN's primitive creation / JIT for the two shapes -- the primitive cache
    # is process-wide), every other worker one tiny convolution (the spin-up of its own intra-op team).  Round 5 warmed with a
    # whole frame on EVERY worker: 38 s of the driver's 95 s for nothing the timed pass needs.
    per_worker = max(1, args.cpu_frames)

    import threading
    all_here = threading.Barrier(workers)   # every pool thread exists and takes exactly one warm-up task

    def warm(w):
        torch.set_num_threads(threads)   # the intra-op team of THIS calling thread
        all_here.wait()
        if w == 0:
            one_frame(0, 0)
        else:
            torch.nn.functional.conv2d(torch.zeros(1, 8, 32, 32), torch.zeros(8, 8, 3, 3))

    def work(w):
        torch.set_num_threads(threads)
        for i in range(per_worker):
            one_frame(w, (workers + w * per_worker + i) % args.batch)

    with ThreadPoolExecutor(workers) as ex:
        t0 = time.perf_counter()
        list(ex.map(warm, range(workers)))
        warm_dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        list(ex.map(work, range(workers)))
        dt = time.perf_counter() - t0
    torch.set_num_threads(prev_threads)
    n = workers * per_worker
    return {"value": n / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "workers": workers, "threads_per_worker": threads,
            "sample": f"{n} frame(s) of the same workload ({workers} workers x {per_worker} frame(s), {threads} torch intra-op "
                      f"threads each; per frame 1 car + {args.crops} armor YOLOv8m forwards in PyTorch-CPU fp32, C oracle "
                      f"pre/post/locate on the worker's own Locator stream), {dt:.1f} s, after an untimed warm-up on the same "
                      f"threads (one whole frame on worker 0, a tiny convolution on the others: {warm_dt:.1f} s)"}


def stage_split(all_stats, scale=1.0):
    """rmr.profile(...).read(by_stage=True) -> ms per stage of the hot path.  Stages run on different streams and
    overlap (locate under detect), so the figures add up to more than the wall time."""
    stage_ms = {"first layer + letterbox sampling (car + armor)": 0.0, "network, car stage": 0.0, "network, armor stage": 0.0,
                "head decode": 0.0, "box decode + NMS + restore": 0.0, "locate: update (scatter + diff)": 0.0,
                "locate: cluster": 0.0, "locate: search": 0.0, "other": 0.0}
    for name, v in all_stats.items():
        # "car|conv n64 ..." / "armor|conv n256 ...": the stage comes from the library (RobotDetector tags what it enqueues),
        # not from the image count -- at batch 256 both stages launch 256-image shapes
        stage, k = name.split("|", 1) if "|" in name else ("", name)
        ms = v["total_ms"] * scale
        if "stem" in k or k == "letterbox":
            stage_ms["first layer + letterbox sampling (car + armor)"] += ms
        elif k.startswith("conv ") or k in ("sppf_pools", "upsample2x", "quant_f8"):
            stage_ms["network, car stage" if stage == "car" else "network, armor stage" if stage == "armor" else "other"] += ms
        elif k == "head_decode":
            stage_ms["head decode"] += ms
        elif k == "postprocess":
            stage_ms["box decode + NMS + restore"] += ms
        elif k in ("loc_scatter", "loc_diff"):
            stage_ms["locate: update (scatter + diff)"] += ms
        elif k == "loc_cluster":
            stage_ms["locate: cluster"] += ms
        elif k == "loc_search":
            stage_ms["locate: search"] += ms
        else:
            stage_ms["other"] += ms
    return {k: round(v, 4) for k, v in stage_ms.items()}


PLAN_DIR = os.path.join(ROOT, "profiles", "plans")


def plan_files(args, plan_dir=None, which=("car", "armor")):
    """The committed pinned plans of a precision: (car, armor) tuning files written by tools/make_plan.py on an MI355X.  They
    hold the kernel of every layer at the batch sizes of configs[2] / [3] (64-image car chunk, 256-image armor chunk) and of the
    batch-1 latency leg (1 / 4 images) -- for the f16 plan and the fp8 plan (configs[4]: 256 / 256)."""
    d = plan_dir or PLAN_DIR
    return tuple(os.path.join(d, f"yolov8m_{w}_{args.dtype}.tune") for w in which)


def apply_plan(args, packs, which=("car", "armor")):
    """Runs under the committed plan so that the driver's line, tools/round_profile.sh's kernel stats and the PMC passes all
    describe the SAME launches: the plan files become the packs' tuning caches and RMR_PLAN=1 pins
    them (nothing is timed, a missing entry is an error -> main() falls back to autotuning and says so)."""
    os.environ.pop("RMR_PLAN", None)
    if args.plan == "tune":
        return None
    import shutil
    files = plan_files(args, None if args.plan == "auto" else args.plan, which)
    if not all(os.path.exists(f) for f in files):
        return None
    for f, pk in zip(files, packs):
        shutil.copyfile(f, pk + ".tune")
    os.environ["RMR_PLAN"] = "1"
    return [os.path.relpath(f, ROOT) for f in files]


class Ranks:
    """What every rank of a run shares, GPU or stub: the process group the launcher's environment describes, the
    C-ABI communicator of the robot-record exchange (include/rmr.h rmr_comm_*: RCCL on GPUs, the FILE transport
    on CPU), the barrier and the max-over-ranks clock of the contract."""

    def __init__(self, args, backend, transport, device=None, compute_on_gpu=None):
        import torch
        import torch.distributed as dist
        from rm_radar_amd import dist as rd
        self.torch, self.dist, self.rd = torch, dist, rd
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.nccl = backend == "nccl"                 # collectives on device tensors over RCCL (else gloo on host tensors)
        self.gpu = self.nccl if compute_on_gpu is None else compute_on_gpu   # the step runs on a GPU: synchronise it around the clock
        if getattr(args, "share_gpu", False):
            self.local = 0
        # under torch.distributed.run (RANK set) the process group is always created, also for one
        # rank, so the RCCL path is exercised by every launcher-driven run
        self.use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            kw = {"device_id": torch.device("cuda", self.local)} if self.nccl else {}
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
        self.dev = torch.device("cuda", self.local) if self.nccl else torch.device("cpu")
        self.comm, self.ranks_seen = None, [0]
        self.gather_via = "none (one rank, no process group)"
        if self.use_dist:
            self.gather_via = f"torch.distributed all_gather_into_tensor ({'RCCL' if self.nccl else 'gloo'})"
            self.ranks_seen = self._probe_torch()
            if args.gather == "abi":
                # the 128-byte id travels over the torch.distributed group that exists anyway for the barrier and
                # the clock.  A failure to set the communicator up falls back to torch.distributed and says so.
                try:
                    ids = [rd.Comm.unique_id(transport) if self.rank == 0 else None]
                    dist.broadcast_object_list(ids, src=0)
                    self.comm = rd.Comm(transport, self.rank, self.world, ids[0], device=self.local)
                    probe = self.comm.all_gather_records(np.full((1, 1, rd.RECORD_WORDS), self.rank + 1, np.int32))
                    seen = [int(probe[r, 0, 0, 0]) - 1 for r in range(probe.shape[0])]
                    assert seen == list(range(self.world)), seen
                    self.ranks_seen = seen
                    self.gather_via = ("rmr_comm_all_gather_records (C-ABI, " +
                                       ("RCCL ncclAllGather" if transport == "rccl" else "FILE transport") + ")")
                except Exception as e:  # noqa: BLE001
                    self.comm = None
                    self.gather_via += f" [C-ABI communicator unavailable: {type(e).__name__}: {e}]"
                    # the first real N > 1 run must be diagnosable from its log: the library's message carries the IPC mode and
                    # RCCL's switches; the launcher's view of the same environment goes beside it
                    env_keys = ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "NCCL_SOCKET_IFNAME", "MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE", "LOCAL_RANK")
                    print(f"[bench] rank {self.rank}: C-ABI communicator ({transport}) unavailable, falling back to torch.distributed: "
                          f"{type(e).__name__}: {e} | " + " ".join(f"{k}={os.environ.get(k, '(unset)')}" for k in env_keys), file=sys.stderr, flush=True)
                # every rank must take the same road from here on: one rank gathering through the C-ABI communicator while
                # another fell back to torch.distributed would hang both
                ok = torch.tensor([1 if self.comm is not None else 0], dtype=torch.int32, device=self.dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0 and self.comm is not None:
                    self.comm.close()
                    self.comm = None
                    self.gather_via = (f"torch.distributed all_gather_into_tensor ({'RCCL' if self.nccl else 'gloo'}) "
                                       "[C-ABI communicator unavailable on another rank]")

    def _probe_torch(self):
        t = self.torch.full((1,), self.rank, dtype=self.torch.int32, device=self.dev)
        out = self.torch.empty((self.world,), dtype=self.torch.int32, device=self.dev)
        self.dist.all_gather_into_tensor(out, t)
        return [int(v) for v in out.cpu()]

    def gather(self, block):
        """block: this rank's int32 [frames, cap, 12] records (numpy) -> [world, frames, cap, 12] (torch, CPU or device)."""
        torch = self.torch
        if self.comm is not None:
            return torch.from_numpy(self.comm.all_gather_records(block))
        if self.use_dist:
            return self.rd.all_gather_records(torch.from_numpy(block).to(self.dev), force=True)
        return torch.from_numpy(block)[None]

    def sync(self):
        if self.gpu:
            self.torch.cuda.synchronize()
        if self.use_dist:
            self.dist.barrier()
            if self.gpu:
                self.torch.cuda.synchronize()

    def timed(self, step, n):
        """EXACTLY n steps between two barrier + synchronize pairs; returns (max over ranks, every rank's own time)."""
        self.sync()
        t0 = time.perf_counter()
        last = None
        for _ in range(n):
            last = step()
        self.sync()
        dt = time.perf_counter() - t0
        per_rank = [dt]
        if self.use_dist:
            t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.dev)
            out = self.torch.empty((self.world,), dtype=self.torch.float64, device=self.dev)
            self.dist.all_gather_into_tensor(out, t)
            per_rank = [float(v) for v in out.cpu()]
        return max(per_rank), per_rank, last

    def close(self):
        if self.comm is not None:
            self.comm.close()
        if self.use_dist:
            self.dist.barrier()
            self.dist.destroy_process_group()


def main_stub(args):
    """--stub-step: everything around the step for real (launcher -> ranks -> gloo group -> FILE-transport communicator ->
    barriers -> max-over-ranks clock -> one JSON line from rank 0), the step itself a stand-in.  No GPU, no librmr compute."""
    from rm_radar_amd import dist as rd
    R = Ranks(args, "gloo", "file")
    B, cap = args.batch, max(args.crops, 1)
    rng = np.random.default_rng(R.rank)

    def step():
        time.sleep(0.002 * (1 + R.rank))            # ranks deliberately unequal: the clock must be the slowest rank's
        block = np.zeros((B, cap, rd.RECORD_WORDS), np.int32)
        block[:, :, 9] = 4                          # valid slots
        block[:, :, 10] = R.rank                    # stream id = rank (stream s lives on rank s % world)
        block[:, :, 11] = np.arange(B)[:, None]
        block[:, :, 0] = rng.integers(0, 640, (B, cap))
        return R.gather(block)

    for _ in range(args.warmup):
        step()
    dt, per_rank, block = R.timed(step, args.steps)
    streams = sorted({int(v) for v in block.reshape(-1, rd.RECORD_WORDS)[:, 10]})
    if R.rank == 0:
        print(json.dumps({
            "stub": True, "metric": "NOT A MEASUREMENT: bench.py --stub-step (launcher / harness test)", "value": B * args.steps * R.world / dt,
            "unit": "stub frames/s", "n_gpus": R.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": None, "data": "none",
            "config": {"workload": "stub", "frames_per_step_per_gpu": B}, "ranks_seen": R.ranks_seen, "gather": R.gather_via,
            "per_rank_frames_per_s": [round(B * args.steps / t, 2) for t in per_rank],
            "gathered_shape": list(block.shape), "streams_in_gathered_list": streams}))
    R.close()


def main_config1(args):
    """BASELINE configs[1]: the full pipeline at batch 1 on the reference sample's inputs -- three of its frames at their own
    size (assets/images/{0,4,9}.jpg, 2592 x 2048 BGR u8, committed re-encoded under tests/golden/assets_images) with their
    sample clouds (assets/clouds/N.pcd, 10 k points, tests/golden/assets_clouds.npz), the calibration of samples/main.cpp:12-22
    and the sample's call order (main.cpp:87 background update first, then runOnce per frame, sample_radar.h:106-127).  Every
    timed call takes the frame and the cloud from HOST memory: the 15.9 MB frame is staged and copied inside the clock, as
    the reference's preprocess does (detector.cu:388-399).  Seeded synthetic weights (the reference ships no models), K = 4
    injected crops per frame for the armor stage, a few hundred injected LiDAR returns per crop so that the
    search has something to find.  Legs: (a) host inputs -> p50 / p99 (the metric); (b) the same frames resident in HBM ->
    what staging + H2D cost; (c) the bare pinned H2D of one frame; (d) events around every launch -> first layer (samples
    the 15.9 MB source) against the rest; (e) parity: the sequence against the CPU oracle, network heads included."""
    import torch

    import rm_radar_amd as rmr
    import scenes
    from rm_radar_amd import assets
    from rm_radar_amd import weights as W
    local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    gold = os.path.join(ROOT, "tests", "golden")
    ids = (0, 4, 9)
    frames = [np.ascontiguousarray(assets.read_image(os.path.join(gold, "assets_images", f"full_{i}.jpg"))) for i in ids]
    size, Kc = scenes.SAMPLE_SIZE, scenes.SAMPLE_K
    assert all(f.shape == (size[1], size[0], 3) and f.dtype == np.uint8 for f in frames)
    data = np.load(os.path.join(gold, "assets_clouds.npz"))
    rng = np.random.default_rng(9)
    background = scenes.make_cloud(rng, 60000, Kc, scenes.SAMPLE_L2C, size)
    K = max(args.crops, 1)
    rects = crop_rects(rng, len(frames), K, size)
    clouds = []
    for f, i in enumerate(ids):
        asset = np.zeros((10000, 4), np.float32)
        asset[:, :3] = data[f"cloud{i}"]
        spec = [(tuple(float(v) for v in r), 2000.0, 300) for r in rects[f]]
        extra = scenes.make_cloud(rng, sum(sp[2] for sp in spec), Kc, scenes.SAMPLE_L2C, size, spec, zero_frac=0, far_frac=0)
        clouds.append(np.ascontiguousarray(np.concatenate([asset, extra])))
    n_pts = clouds[0].shape[0]

    pack_dir = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rmr_packs")
    os.makedirs(pack_dir, exist_ok=True)
    packs = (os.path.join(pack_dir, "car_r0.rmrw"), os.path.join(pack_dir, "armor_r0.rmrw"))
    W.make_synthetic_pack(packs[0], "m", 1, seed=1, cls_bias=-6.0)
    W.make_synthetic_pack(packs[1], "m", 12, seed=2, cls_bias=-6.0)
    plan = apply_plan(args, packs)

    def make():
        rd = rmr.RobotDetector(packs[0], packs[1], size, 12, max_cars=K, opt_cars=K, device=local, max_frames=1, precision=args.dtype)
        lc = rmr.Locator(size[0], size[1], Kc, scenes.SAMPLE_L2C, scenes.SAMPLE_W2C, device=local, max_frames=1)
        lc.update(background)   # main.cpp:87
        return rd, lc
    r1, l1 = make()
    fc = [np.ascontiguousarray(np.asarray(rects[f], np.int32).reshape(1, -1, 4)) for f in range(len(frames))]
    fb_host = [rmr.FrameBatch([frames[f]], [clouds[f]]) for f in range(len(frames))]
    plan_note = "pinned (RMR_PLAN): " + " + ".join(plan) if plan else "autotuned on this box"
    try:
        rmr.run_batch(r1, l1, fb_host[0], None, fc[0])
    except rmr.RmrError as e:
        if "pinned plan" not in str(e):
            raise
        r1.close(), l1.close()
        os.environ.pop("RMR_PLAN", None)
        r1, l1 = make()
        plan_note = "autotuned on this box (the committed plan has no entries for these batch sizes)"
        rmr.run_batch(r1, l1, fb_host[0], None, fc[0])

    def timed(fbs, n_warm, n_timed):
        lat = []
        for i in range(n_warm + n_timed):
            f = i % len(fbs)
            t0 = time.perf_counter()
            rmr.run_batch(r1, l1, fbs[f], None, fc[f])
            lat.append((time.perf_counter() - t0) * 1e3)
        return np.array(lat[n_warm:])
    n_warm, n_timed = args.latency_frames
    # (a) host inputs: the configuration as specified
    lat = timed(fb_host, n_warm, n_timed)
    # (b) the same frames and clouds resident in HBM
    d_frames = [torch.from_numpy(f).to(dev) for f in frames]
    d_clouds = [torch.from_numpy(c).to(dev) for c in clouds]
    fb_dev = [rmr.FrameBatch([d_frames[f]], [d_clouds[f]]) for f in range(len(frames))]
    lat_dev = timed(fb_dev, max(20, n_warm // 4), max(200, n_timed // 4))
    # (c) the bare H2D of one frame from pinned memory
    p_img = rmr.PinnedArray(frames[0].shape, np.uint8, device=local)
    p_img.a[...] = frames[0]
    ring = rmr.UploadRing(1, frames[0].nbytes + 4096, device=local)
    ring.begin(0, [p_img.a]); ring.wait(0)
    t0 = time.perf_counter()
    for _ in range(20):
        ring.begin(0, [p_img.a]); ring.wait(0)
    h2d_ms = (time.perf_counter() - t0) / 20 * 1e3
    ring.close(); p_img.close()
    # (d) events around every launch, device-resident inputs, 30 frames
    stage_ms = None
    if not args.no_profile:
        with rmr.profile(local) as prof:
            for i in range(30):
                rmr.run_batch(r1, l1, fb_dev[i % len(fb_dev)], None, fc[i % len(fb_dev)])
            torch.cuda.synchronize()
            stage_ms = stage_split(prof.read(by_stage=True), 1.0 / 30)
    # (e) parity: a fresh stream (background first), the three frames in order, against the CPU oracle
    parity_checked, parity = False, {"skipped": "--no-parity"}
    if not args.no_parity:
        import oracle
        import step_parity
        l1.close()
        l1 = rmr.Locator(size[0], size[1], Kc, scenes.SAMPLE_L2C, scenes.SAMPLE_W2C, device=local, max_frames=1)
        l1.update(background)
        cpu = oracle.Locator(size[0], size[1], Kc, scenes.SAMPLE_L2C, scenes.SAMPLE_W2C)
        cpu.update(background)
        try:
            tot = {"frames": 0, "robots": 0, "located": 0, "max_xyz_err_m": 0.0, "assembly_frames": 0}
            for f in range(len(frames)):
                robots, counts = rmr.run_batch(r1, l1, fb_host[f], None, fc[f])
                st = step_parity.check_step(oracle, rmr, r1, cpu, robots, counts, clouds[f:f + 1], fc[f])
                for k in ("frames", "robots", "located", "assembly_frames"):
                    tot[k] += st[k]
                tot["max_xyz_err_m"] = max(tot["max_xyz_err_m"], st["max_xyz_err_m"])
                if f == 0:
                    tot["network"] = step_parity.check_network(oracle, r1, frames[:1], fc[0], packs, args.dtype,
                                                               armor_slots=tuple(range(min(K, 3))))
                    tot["network_checked"] = True
            parity_checked, parity = True, {k: (round(v, 9) if isinstance(v, float) else v) for k, v in tot.items()}
        except AssertionError as e:
            parity_checked, parity = False, {"error": str(e)[:300]}
    r1.close(), l1.close()
    flops_frame = W.flops_per_image("m", 1) + K * W.flops_per_image("m", 12)
    p50, p99 = float(np.percentile(lat, 50)), float(np.percentile(lat, 99))
    p50d = float(np.percentile(lat_dev, 50))
    result = {
        "metric": "frames/sec detect+locate at batch=1 on the reference sample's 2592x2048 frames + 10k-pt clouds; p50 ms/frame",
        "value": 1e3 / float(lat.mean()), "unit": "frames/s", "n_gpus": 1, "steps": int(n_timed), "warmup": int(n_warm),
        "ms_per_step": float(lat.mean()), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "the reference's sample frames and clouds (committed fixtures), synthetic weights",
        "config": {"workload": f"configs[1]: batch=1, {len(frames)} of the reference's sample frames ({size[0]}x{size[1]} BGR u8, 15.9 MB each) "
                               f"+ their sample clouds ({n_pts} points: 10000 of the asset + {n_pts - 10000} injected returns) FROM HOST MEMORY per call, "
                               f"sample calibration, background update first; car YOLOv8m + {K} injected armor crops/frame, {args.dtype} MFMA",
                   "gflop_per_frame": round(flops_frame / 1e9, 3), "kernel_plan": plan_note},
        "p50_ms_batch1": round(p50, 3), "p99_ms_batch1": round(p99, 3), "max_ms": round(float(lat.max()), 3),
        "latency_sample": {"warmup_frames": int(n_warm), "timed_frames": int(n_timed)},
        "inputs_resident_in_hbm": {"p50_ms": round(p50d, 3), "p99_ms": round(float(np.percentile(lat_dev, 99)), 3), "frames": int(len(lat_dev))},
        "stage_split_ms_per_frame": {
            "staging + H2D of frame and cloud (p50 host inputs - p50 HBM inputs)": round(p50 - p50d, 3),
            "bare pinned H2D of one 15.9 MB frame": round(h2d_ms, 3),
            "h2d_fraction_of_frame": round((p50 - p50d) / p50, 4),
            "kernels (events around every launch, inputs in HBM)": stage_ms},
        "parity_checked": parity_checked, "parity": parity,
    }
    if not args.no_cpu_baseline:
        args.crops = K
        result["cpu_baseline"] = cpu_baseline(args, packs, np.stack(frames), np.stack(clouds), rects)
    else:
        result["cpu_baseline"] = None
    print(json.dumps(result))


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    if args.gpus != int(os.environ.get("WORLD_SIZE", "1")):
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: launch as "
                         f"`python bench.py --gpus {args.gpus}` or under torch.distributed.run --nproc-per-node {args.gpus}\n")
        sys.exit(3)
    if args.stub_step:
        return main_stub(args)
    if args.config == 1:
        return main_config1(args)
    if args.launch_order:
        if os.path.exists(args.launch_order):
            os.remove(args.launch_order)
        os.environ["RMR_PROFILE_ORDER"] = args.launch_order   # read by the library's profiler at its first launch
    import torch

    import rm_radar_amd as rmr
    import scenes
    from rm_radar_amd import dist as rd
    from rm_radar_amd import weights as W

    R = Ranks(args, "gloo", "file", compute_on_gpu=True) if args.share_gpu else Ranks(args, "nccl", "rccl")
    world, rank, local, use_dist = R.world, R.rank, R.local, R.use_dist
    dev = torch.device("cuda", local)   # where the step's inputs live (R.dev is where the collectives' tensors live)
    torch.cuda.set_device(local)

    pack_dir = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rmr_packs")
    os.makedirs(pack_dir, exist_ok=True)
    packs = (os.path.join(pack_dir, f"car_r{rank}.rmrw"), os.path.join(pack_dir, f"armor_r{rank}.rmrw"))
    W.make_synthetic_pack(packs[0], "m", 1, seed=1, cls_bias=-6.0)
    W.make_synthetic_pack(packs[1], "m", 12, seed=2, cls_bias=-6.0)

    images, clouds, rects = make_inputs(args, rank)
    size = frame_size(args)
    d_images = torch.from_numpy(images).to(dev)
    d_clouds = torch.from_numpy(clouds).to(dev)
    img_list = [d_images[f] for f in range(args.batch)]

    B, K = args.batch, args.crops
    plan = apply_plan(args, packs)

    def make_rdet(max_frames=B):
        return rmr.RobotDetector(packs[0], packs[1], size, 12, max_cars=max(K, 1), opt_cars=max(K, 1),
                                 device=local, max_frames=max_frames, precision=args.dtype)
    rdet = make_rdet()
    S = max(1, args.streams)
    if B % S:
        raise SystemExit(f"--batch {B} does not divide into --streams {S}")
    locs = [rmr.Locator(size[0], size[1], intrinsic(args), scenes.SAMPLE_L2C, np.eye(4, dtype=np.float32),
                        device=local, max_frames=B // S) for _ in range(S)]
    loc = locs[0] if S == 1 else locs
    cap = rdet.max_cars
    arena_gib = rdet.arena_bytes() / 2 ** 30
    flops_frame = W.flops_per_image("m", 1) + K * W.flops_per_image("m", 12)

    import ctypes as C
    from rm_radar_amd import _lib
    phases = {"detect_locate_search": 0.0, "pack_gather": 0.0}

    cloud_list = [d_clouds[f] for f in range(B)]
    # the frames stay in the same HBM buffers from step to step: their descriptors (64 rmr_image, the
    # cloud pointer table) are marshalled once, as a C++ host would keep them, not rebuilt by 64 x 2
    # Python attribute round trips inside every step
    frames_fb = rmr.FrameBatch(img_list, cloud_list)
    forced = np.ascontiguousarray(np.asarray(rects, np.int32).reshape(B, -1, 4))

    def step():
        # one native call in the reference's order (sample_radar.h:106-127): update + cluster of
        # the 64 frames on a helper thread while detect runs, join, then one batched search
        t0 = time.perf_counter()
        robots, counts = rmr.run_batch(rdet, loc, frames_fb, None, forced)
        t3 = time.perf_counter()
        block = R.gather(rd.pack_records(robots, counts, cap, rank, cap))
        t4 = time.perf_counter()
        phases["detect_locate_search"] += t3 - t0
        phases["pack_gather"] += t4 - t3
        return block, counts

    sync_all = R.sync

    # untimed: the first call autotunes every layer for the two batch sizes (the analogue of the
    # reference's TensorRT engine build, detector.cpp:177-243) -- kept apart from the W warm-up steps
    # so that --warmup 0 cannot put it inside the timed region
    plan_note = "autotuned on this box (--plan tune)" if args.plan == "tune" else "autotuned on this box (no committed plan for this precision)"
    if plan:
        try:
            step()
            plan_note = f"pinned: {plan[0]} + {plan[1]} (RMR_PLAN)"
        except rmr.RmrError as e:
            if "pinned plan" not in str(e):
                raise
            # the committed plan does not cover this build / workload (kernel set changed, another batch size): tune here
            rdet.close()
            os.environ.pop("RMR_PLAN", None)
            for pk in packs:
                if os.path.exists(pk + ".tune"):
                    os.remove(pk + ".tune")
            rdet = make_rdet()
            plan = None
            plan_note = "autotuned on this box (the committed plan does not cover this build or these batch sizes)"
            step()
    else:
        step()
    for _ in range(args.warmup):
        step()
    sync_all()
    for k in phases:
        phases[k] = 0.0
    # ---- 2. headline: exactly K steps, nothing profiled -------------------------------------------------
    dt, per_rank_dt, (block, counts) = R.timed(step, args.steps)
    headline_phases = {k: v for k, v in phases.items()}

    # ---- 3. steady state: the same loop until --seconds are on the clock (all ranks run the same count)
    steady = None
    if args.seconds > dt:
        extra = max(1, int((args.seconds - dt) / (dt / args.steps) + 0.5))
        dt2, _, _ = R.timed(step, extra)
        steady = {"steps": args.steps + extra, "seconds": round(dt + dt2, 3),
                  "value": round(B * (args.steps + extra) * world / (dt + dt2), 2), "unit": "frames/s"}

    # ---- 4. roofline: events around the convolution launches only, per GEMM shape ------------------------
    stats = {}
    if not args.no_profile:
        with rmr.profile(local, flops_only=True) as prof:
            for _ in range(args.profile_steps):
                step()
            sync_all()
            stats = prof.read()

    # ---- 4b. per-stage breakdown: one more step with events around EVERY launch.  Stages run on
    # different streams and overlap (locate under detect), so the figures add up to more than a step.
    stage_ms = None
    if not args.no_profile and rank == 0:
        with rmr.profile(local) as prof:
            # rank 0 alone: the native call without the gather (a collective here would wait for ranks that are not in this leg)
            rmr.run_batch(rdet, loc, frames_fb, None, forced)
            torch.cuda.synchronize()
            all_stats = prof.read(by_stage=True)
        stage_ms = stage_split(all_stats)

    # ---- 5. host inputs: the step's frames + clouds over PCIe.  (a) the copy on its own; (b) the loop a capture host
    # would run: step i + 1's inputs travel from page-locked buffers into the other slot of an upload ring
    # (rmr_upload_*, its own copy stream) while step i computes -> value_incl_h2d.  Never `value`.
    h2d_ms, incl_h2d = None, None
    if rank == 0 and S == 1:
        p_img, p_cld = rmr.PinnedArray(images.shape, np.uint8, device=local), rmr.PinnedArray(clouds.shape, np.float32, device=local)
        p_img.a[...] = images
        p_cld.a[...] = clouds
        ring = rmr.UploadRing(2, images.nbytes + clouds.nbytes + 4096, device=local)
        fbs = []
        for slot in range(2):   # the slots' addresses never change: descriptors marshalled once per slot
            di, dc = ring.begin(slot, [p_img.a, p_cld.a])
            ring.wait(slot)
            fbs.append(rmr.FrameBatch([di[f] for f in range(B)], [dc[f] for f in range(B)]))
        t0 = time.perf_counter()
        for _ in range(5):
            ring.begin(0, [p_img.a, p_cld.a])
            ring.wait(0)
        h2d_ms = (time.perf_counter() - t0) / 5 * 1e3
        n_h = max(args.steps, 10)

        def host_step(i):
            ring.begin((i + 1) % 2, [p_img.a, p_cld.a])   # next step's inputs start travelling ...
            ring.wait(i % 2)                               # ... this step's have landed long ago
            rmr.run_batch(rdet, loc, fbs[i % 2], None, forced)
        ring.begin(0, [p_img.a, p_cld.a])
        for i in range(2):
            host_step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(2, 2 + n_h):
            host_step(i)
        ring.wait(n_h % 2)
        torch.cuda.synchronize()
        incl_h2d = {"steps": n_h, "value": round(B * n_h / (time.perf_counter() - t0), 2), "unit": "frames/s",
                    "how": "inputs of step i+1 copied from pinned host buffers on the upload ring's copy stream while step i runs"}
        ring.close()
        del fbs
        p_img.close()
        p_cld.close()

    n_located = int(((block.view(-1, 12)[:, 9] & 2) != 0).sum().item()) if block.numel() else 0
    # HBM traffic per conv launch: PMC counters cannot be read from inside this process; the figure comes from
    # rocprofv3 --pmc passes over this same command (tools/round_profile.sh) and is only as fresh as the kernel
    # sources it was measured on: the file records their hash, a mismatch is reported as stale
    traffic, traffic_src, traffic_stale, dom_traffic, dom_symbol, pm = None, None, None, None, None, None
    for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_conv_traffic.json")), reverse=True):
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", name)))
            if B == 64 and K == 4 and size == (640, 640) and args.dtype == "f16":
                traffic, traffic_src = pm["traffic_bytes_per_launch"], "profiles/" + name
                dom_traffic, dom_symbol = pm.get("dominant_traffic_bytes_per_launch"), pm.get("dominant_kernel")
                traffic_stale = pm.get("source_hash") != source_hash()
            else:
                pm = None
            break
        except (OSError, KeyError, ValueError):
            pm = None
            continue
    result = None
    if rank == 0:
        frames = B * args.steps * world
        convs = {k: v for k, v in stats.items() if v["flops"] > 0}
        conv = {"total_ms": sum(v["total_ms"] for v in convs.values()), "flops": sum(v["flops"] for v in convs.values()),
                "bytes": sum(v["bytes"] for v in convs.values()), "launches": sum(v["launches"] for v in convs.values())}
        ach = conv["flops"] / (conv["total_ms"] * 1e-3) / 1e12 if conv["total_ms"] > 0 else 0.0
        psteps = max(args.profile_steps, 1)
        # the e4m3 layers of the fp8 plan run on the 5 PF pipe: the peak of a mixed set of launches is FLOP-weighted
        # (the time the set needs when every launch runs at its own pipe's peak)
        def peak_of(vs):
            f8 = sum(v["flops"] for k, v in vs if instantiation_of(k).startswith("f"))
            f16 = sum(v["flops"] for k, v in vs) - f8
            return (f8 + f16) / (f16 / F16_DENSE_PEAK_TFLOPS + f8 / FP8_DENSE_PEAK_TFLOPS) if f8 + f16 > 0 else F16_DENSE_PEAK_TFLOPS
        family_peak = peak_of(convs.items())
        # the dominant kernel = the template instantiation (one symbol of a kernel trace) with the most time in a step
        inst = {}
        for k, v in convs.items():
            e = inst.setdefault(instantiation_of(k), {"total_ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0, "layers": []})
            for f in ("total_ms", "flops", "bytes", "launches"):
                e[f] += v[f]
            e["layers"].append(k)
        dom_tag, dom = max(inst.items(), key=lambda kv: kv[1]["total_ms"]) if inst else ("-", None)
        dom_ach = dom["flops"] / (dom["total_ms"] * 1e-3) / 1e12 if dom and dom["total_ms"] > 0 else 0.0
        dom_peak = FP8_DENSE_PEAK_TFLOPS if dom_tag.startswith("f") else F16_DENSE_PEAK_TFLOPS
        # traffic of THIS instantiation from the per-layer PMC attribution (tools/pmc_traffic.py, the passes run under the same
        # pinned plan): like for like when the PMC file saw as many launches of it per step as this run did
        traffic_mix = None
        if pm and dom and (pm.get("by_instantiation") or {}).get(dom_tag):
            e = pm["by_instantiation"][dom_tag]
            dom_traffic, dom_symbol = e["traffic_bytes_per_launch"], f"instantiation {dom_tag} (per-layer attribution)"
            traffic_mix = {"launches_per_step_in_pmc_passes": e["launches_per_step"],
                           "launches_per_step_in_this_run": dom["launches"] / psteps,
                           "same_launches": e["launches_per_step"] == dom["launches"] / psteps,
                           "algorithmic_bytes_per_launch_in_pmc_passes": e["algorithmic_bytes_per_launch"],
                           "traffic_over_algorithmic": e["traffic_over_algorithmic"]}
        # every layer against its own bound: the time the chip needs at the MFMA peak or at the HBM peak,
        # whichever is larger; summed over the step
        def pk(k):
            return (FP8_DENSE_PEAK_TFLOPS if instantiation_of(k).startswith("f") else F16_DENSE_PEAK_TFLOPS) * 1e12
        bound_ms = sum(max(v["flops"] / pk(k), v["bytes"] / (HBM_PEAK_TBS * 1e12)) * 1e3 for k, v in convs.items())
        top = sorted(convs.items(), key=lambda kv: -kv[1]["total_ms"])[:8]
        layer_roofline = {
            "peaks": {"mfma_tflops": F16_DENSE_PEAK_TFLOPS, "mfma_fp8_tflops": FP8_DENSE_PEAK_TFLOPS, "hbm_tbs": HBM_PEAK_TBS},
            "bound_ms_per_step": round(bound_ms / psteps, 3), "measured_ms_per_step": round(conv["total_ms"] / psteps, 3),
            "frac": round(bound_ms / conv["total_ms"], 4) if conv["total_ms"] > 0 else None,
            "layers": len(convs),
            "top": [{"layer": k, "launches_per_step": v["launches"] / psteps, "ms_per_step": round(v["total_ms"] / psteps, 3),
                     "tflops": round(v["flops"] / v["total_ms"] / 1e9, 1), "gbs": round(v["bytes"] / v["total_ms"] / 1e6),
                     "bound": "mfma" if v["flops"] / pk(k) >= v["bytes"] / HBM_PEAK_TBS / 1e12 else "hbm"}
                    for k, v in top]}
        result = {
            "metric": "frames/sec detect+locate (640x640 + 30k-pt cloud)",
            "value": frames / dt,
            "unit": "frames/s",
            "n_gpus": world,
            **({"share_gpu": True} if args.share_gpu else {}),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16" if args.dtype == "f16" else "fp8 (e4m3 operands in the 3x3 layers of backbone and neck: ~61 % of the FLOPs; f16 elsewhere, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"{'configs[4]' if args.dtype == 'fp8' else 'configs[2]' if size == (640, 640) else 'configs[3]: one stream per GPU,'}: batch={B} synthetic {size[0]}x{size[1]} frames + "
                                   f"{args.points}-pt clouds per step per GPU, car YOLOv8m + {K} injected "
                                   f"armor crops/frame (YOLOv8m, nc=12), seeded synthetic weights, {args.dtype} MFMA",
                       "frames_per_step_per_gpu": B, "crops_per_frame": K, "points_per_cloud": args.points,
                       "streams_per_gpu": S, "gflop_per_frame": round(flops_frame / 1e9, 3),
                       "activation_arena_gib": round(arena_gib, 2), "kernel_plan": plan_note},
            # the dominant kernel of a step: ONE template instantiation (one symbol of the rocprofv3 kernel trace)
            "roofline": {"bound": "mfma",
                         "kernel": f"{KERNEL_OF_TAG.get(dom_tag[0], 'conv')} instantiation {dom_tag}: the kernel with the most time in a step "
                                   f"({100 * dom['total_ms'] / conv['total_ms']:.1f} % of the convolution time)" if dom else None,
                         "achieved": round(dom_ach, 2), "peak": dom_peak, "unit": "TFLOP/s", "frac": round(dom_ach / dom_peak, 4),
                         # PMC counters cannot be read from inside this process: from the rocprofv3 --pmc passes over this same
                         # command (tools/round_profile.sh), for the symbol with the most time in that trace
                         "traffic": dom_traffic, "traffic_kernel_symbol": dom_symbol, "traffic_source": traffic_src,
                         "traffic_stale": traffic_stale, "traffic_mix": traffic_mix,
                         "launches_per_step": dom["launches"] / psteps if dom else 0,
                         "avg_launch_ms": round(dom["total_ms"] / max(dom["launches"], 1), 5) if dom else None,
                         "algorithmic_gflop_per_launch": round(dom["flops"] / max(dom["launches"], 1) / 1e9, 4) if dom else None,
                         "algorithmic_bytes_per_launch": round(dom["bytes"] / max(dom["launches"], 1)) if dom else None,
                         "layers": sorted(dom["layers"]) if dom else [],
                         "measured_in": f"{psteps} profiled step(s) after the headline loop (HIP events on the detector's stream)"},
            # every convolution launch of a step together (all instantiations: what rounds 1-2 reported as `roofline`)
            "roofline_all_conv_launches": {
                         "bound": "mfma", "kernel": "conv_* (every convolution launch of a step)", "achieved": round(ach, 2),
                         "peak": round(family_peak, 1), "unit": "TFLOP/s",
                         "frac": round(ach / family_peak, 4),
                         "power_limited_peak": F16_POWER_LIMITED_TFLOPS if args.dtype == "f16" else None,
                         "frac_of_power_limited_peak": round(ach / F16_POWER_LIMITED_TFLOPS, 4) if args.dtype == "f16" else None,
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                         "algorithmic_bytes_per_launch": round(conv["bytes"] / max(conv["launches"], 1)),
                         "launches_per_step": conv["launches"] / psteps,
                         "avg_launch_ms": round(conv["total_ms"] / max(conv["launches"], 1), 5),
                         "algorithmic_gflop_per_launch": round(conv["flops"] / max(conv["launches"], 1) / 1e9, 4),
                         "by_instantiation": {t: {"ms_per_step": round(e["total_ms"] / psteps, 3), "launches_per_step": e["launches"] / psteps,
                                                  "tflops": round(e["flops"] / e["total_ms"] / 1e9, 1) if e["total_ms"] > 0 else 0.0}
                                              for t, e in sorted(inst.items(), key=lambda kv: -kv[1]["total_ms"])[:12]}},
            "layer_roofline": layer_roofline,
            "stage_ms_per_step": stage_ms,
            "steady_state": steady,
            "h2d_ms_per_step": None if h2d_ms is None else round(h2d_ms, 3),
            "value_incl_h2d": None if incl_h2d is None else incl_h2d["value"],
            "host_input_loop": incl_h2d,
            "value_incl_h2d_not_overlapped": None if h2d_ms is None else round(B * world / (dt / args.steps + h2d_ms * 1e-3), 2),
            "end_to_end_tflops": round(flops_frame * frames / dt / 1e12, 2),
            "host_phase_ms_per_step": {k: round(v / args.steps * 1e3, 2) for k, v in headline_phases.items()},
            "gather": R.gather_via,
            "ranks_seen": R.ranks_seen,
            "per_rank_frames_per_s": [round(B * args.steps / t, 2) for t in per_rank_dt],
            "located_last_step": n_located,
        }

    # ---- the timed configuration checked against the oracle (part of the cpu_baseline leg; needs rdet alive) ----
    if rank == 0 and world == 1 and not args.no_parity and S == 1:
        result["parity_checked"], result["parity"] = step_parity_leg(args, rmr, rdet, frames_fb, forced, clouds, local, images, packs)
    elif rank == 0:
        result["parity_checked"], result["parity"] = False, {"skipped": "runs with one rank and one stream, unless --no-parity"}

    # ---- batch-1 latency (p50), host inputs: H2D inside the timed region ----
    if rank == 0 and not args.no_latency:
        rdet.close()
        for l in locs:
            l.close()
        r1 = make_rdet(1)
        l1 = rmr.Locator(size[0], size[1], intrinsic(args), scenes.SAMPLE_L2C, np.eye(4, dtype=np.float32), device=local,
                         max_frames=1)
        result["latency_kernel_plan"] = "pinned (RMR_PLAN)" if os.environ.get("RMR_PLAN") else "autotuned on this box"
        try:
            rmr.run_batch(r1, l1, [images[0]], [clouds[0]], [rects[0]])
        except rmr.RmrError as e:   # the committed plan has no entries for 1 / K images: tune them here
            if "pinned plan" not in str(e):
                raise
            r1.close()
            os.environ.pop("RMR_PLAN", None)
            r1 = make_rdet(1)
            result["latency_kernel_plan"] = "autotuned on this box (the committed plan has no batch-1 entries)"
        # the ctypes pointer tables of every frame are marshalled once (what a C++ host holds anyway: rmr.FrameBatch); pixels and
        # points stay in host memory and go to the device inside the timed call
        fb1 = [rmr.FrameBatch([images[f]], [clouds[f]]) for f in range(B)]
        fc1 = [np.ascontiguousarray(np.asarray(rects[f], np.int32).reshape(1, -1, 4)) for f in range(B)]
        lat = []
        n_warm, n_timed = args.latency_frames
        for i in range(n_warm + n_timed):  # 200 warm-up + 2000 timed frames
            f = i % B
            t0 = time.perf_counter()
            rmr.run_batch(r1, l1, fb1[f], None, fc1[f])  # the same native call, one frame
            lat.append((time.perf_counter() - t0) * 1e3)
        lat = np.array(lat[n_warm:])
        result["latency_sample"] = {"warmup_frames": n_warm, "timed_frames": n_timed,
                                    "max_ms": round(float(lat.max()), 3), "p999_ms": round(float(np.percentile(lat, 99.9)), 3)}
        result["p50_ms_batch1"] = round(float(np.percentile(lat, 50)), 3)
        result["p99_ms_batch1"] = round(float(np.percentile(lat, 99)), 3)
        r1.close()
        l1.close()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, packs, images, clouds, rects)
    if rank == 0:
        if "cpu_baseline" not in result:
            result["cpu_baseline"] = None
        print(json.dumps(result))
    R.close()


if __name__ == "__main__":
    main()
