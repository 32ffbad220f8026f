"""The N>1 path on CPU: two gloo processes, stream sharding + the robot-record all-gather."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import ctypes as C

    import torch
    import torch.distributed as dist

    from rm_radar_amd import _lib, dist as rd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        streams = rd.streams_of_rank(5, rank, world)
        n_frames, cap = 3, 4
        robots = (_lib.Robot * (n_frames * cap))()
        counts = np.array([2, 0, 4], np.int32)
        for f in range(n_frames):
            for i in range(counts[f]):
                r = robots[f * cap + i]
                r.rect[:] = [rank * 100 + f, i, 10 + f, 20 + i]
                r.has_label = int(i % 2 == 0)
                r.label = 3 + i
                r.confidence = 0.5 + 0.1 * i
                r.has_location = int(i % 2 == 1)
                r.location[:] = [1.0 + rank, 2.0 + f, 3.0 + i]
        block = torch.from_numpy(rd.pack_records(robots, counts, cap, streams[0], cap))
        allb = rd.all_gather_records(block)
        q.put((rank, streams, allb.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_stream_sharding():
    from rm_radar_amd import dist as rd
    assert rd.streams_of_rank(8, 3, 8) == [3]
    assert rd.streams_of_rank(5, 0, 2) == [0, 2, 4]
    assert sorted(sum((rd.streams_of_rank(11, r, 4) for r in range(4)), [])) == list(range(11))


def test_all_gather_records_world2():
    import torch.multiprocessing as mp
    from rm_radar_amd import dist as rd
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda t: t[0])
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]
    a, b = res[0][2], res[1][2]
    assert a.shape == (2, 3, 4, 12) and np.array_equal(a, b)  # every rank holds the same list
    recs = rd.unpack_records(a)
    assert len(recs) == 2 * (2 + 0 + 4)
    by_rank = {0: [r for r in recs if r["stream_id"] == 0], 1: [r for r in recs if r["stream_id"] == 1]}
    assert len(by_rank[0]) == 6 and len(by_rank[1]) == 6
    r = by_rank[1][1]  # rank 1, frame 0, robot 1
    assert r["rect"] == (100.0, 1.0, 10.0, 21.0) and r["label"] is None
    assert r["location"] == (2.0, 2.0, 4.0) and r["frame_id"] == 0
    r = by_rank[0][2]  # rank 0, frame 2, robot 0
    assert r["label"] == 3 and r["location"] is None and r["frame_id"] == 2
    assert abs(r["confidence"] - 0.5) < 1e-7


def test_single_process_gather_is_identity():
    import torch
    from rm_radar_amd import dist as rd
    x = torch.arange(24, dtype=torch.int32).reshape(1, 2, 12)
    assert torch.equal(rd.all_gather_records(x)[0], x)
