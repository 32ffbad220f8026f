"""bench.py's own launcher on CPU: `python bench.py --gpus 2` must start TWO ranks (torch.distributed.run, one
process per GPU), build the process group, exchange robot records through the C-ABI communicator and report
n_gpus == 2 with both ranks seen -- and must refuse, loudly, to run fewer ranks than asked for.  The GPU step is
replaced by bench.py's --stub-step stand-in (gloo + the FILE transport); nothing here is a measurement."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=e, timeout=timeout)


def test_gpus_2_starts_two_ranks_and_gathers():
    p = _run(["--gpus", "2", "--stub-step", "--steps", "4", "--warmup", "1", "--batch", "8"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout          # ONE line, from rank 0
    r = json.loads(lines[0])
    assert r["stub"] is True and r["n_gpus"] == 2 and r["steps"] == 4
    assert r["ranks_seen"] == [0, 1]
    assert "rmr_comm_all_gather_records" in r["gather"]
    assert r["gathered_shape"] == [2, 8, 4, 12] and r["streams_in_gathered_list"] == [0, 1]
    assert len(r["per_rank_frames_per_s"]) == 2
    # whole-job value = frames of ALL ranks / the slowest rank's time
    assert abs(r["value"] - 2 * 8 * 4 / (r["ms_per_step"] * 4e-3)) < 1e-6 * r["value"]
    assert r["value"] <= sum(r["per_rank_frames_per_s"]) * 1.0001


def test_gpus_2_without_two_gpus_is_an_error():
    import torch
    if torch.cuda.device_count() >= 2:
        return
    p = _run(["--gpus", "2", "--steps", "1"])
    assert p.returncode != 0
    assert "needs 2 visible GPUs" in p.stderr and not p.stdout.strip()


def test_world_size_must_match_gpus():
    p = _run(["--gpus", "2", "--stub-step"], env={"RANK": "0", "WORLD_SIZE": "1"})
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr


def test_a_failed_communicator_is_diagnosable_from_the_log():
    # a failure of the C-ABI communicator's set-up (on a GPU node: ncclCommInitRank) must not end the run -- every rank falls
    # back to torch.distributed together -- and must leave in the log what a diagnosis needs: the IPC mode, RCCL's switches
    p = _run(["--gpus", "2", "--stub-step", "--steps", "2", "--warmup", "0", "--batch", "4"], env={"RMR_COMM_INJECT_FAILURE": "1"})
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == 2 and r["ranks_seen"] == [0, 1]
    assert "torch.distributed" in r["gather"] and "unavailable" in r["gather"]
    for needle in ("falling back to torch.distributed", "HSA_ENABLE_IPC_MODE_LEGACY=", "NCCL_DEBUG=", "rank 0 of 2", "rank 1 of 2"):
        assert needle in p.stderr, (needle, p.stderr[-1500:])
