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


def _build_gather_exe():
    import subprocess
    import __graft_entry__ as g
    g.build()
    libdir = os.path.join(ROOT, "rm_radar_amd", "_build")
    exe = os.path.join(libdir, "gather_ranks")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "gather_ranks.cpp"), "-L", libdir, "-lrmr",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_cpp_hosts_gather_through_the_c_abi_world2(tmp_path):
    """Two C++ processes (tests/cpp/gather_ranks.cpp) drive the multi-GPU boundary -- stream assignment,
    rmr_pack_robot_records, rmr_comm_all_gather_records -- over the file transport (no GPU on this box; on a
    node with GPUs the same program runs with transport 0 = RCCL).  Both ranks must print the same table: the
    robots of all five streams, for three rounds."""
    import subprocess
    exe = _build_gather_exe()
    idf = str(tmp_path / "comm.id")
    env = dict(os.environ, TMPDIR=str(tmp_path))
    procs = [subprocess.Popen([exe, "1", idf, str(r), "2", "5", "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, env=env) for r in range(2)]
    outs = [p.communicate(timeout=120) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so + se
    tables = [[l for l in so.splitlines() if l.startswith("round")] for so, _ in outs]
    assert tables[0] == tables[1]
    # streams 0, 2, 4 live on rank 0 and 1, 3 on rank 1; 3 robots per stream per round
    assert len(tables[0]) == 3 * 5 * 3
    streams = sorted({int(l.split()[3]) for l in tables[0]})
    assert streams == [0, 1, 2, 3, 4]
    assert any("stream 3 frame 1 rect 301 0 12 20 label 6 loc 4 3 3" in l for l in tables[0])


def test_python_comm_mirror_matches_numpy_packing(tmp_path):
    """rm_radar_amd.dist.Comm over the file transport with one rank, and rmr_pack_robot_records against the
    numpy packer."""
    sys.path.insert(0, ROOT)
    from rm_radar_amd import _lib, dist as rd
    os.environ["TMPDIR"] = str(tmp_path)
    n_frames, cap = 3, 4
    robots = (_lib.Robot * (n_frames * cap))()
    counts = np.array([2, 0, 4], np.int32)
    for f in range(n_frames):
        for i in range(counts[f]):
            r = robots[f * cap + i]
            r.rect[:] = [f, i, 10 + f, 20 + i]
            r.has_label = int(i % 2 == 0)
            r.label = 3 + i
            r.confidence = 0.5 + 0.1 * i
            r.has_location = int(i % 2 == 1)
            r.location[:] = [1.0, 2.0 + f, 3.0 + i]
    a = rd.pack_records(robots, counts, cap, 7, cap)
    b = rd.pack_records_abi(robots, counts, cap, 7, cap)
    valid = (a[..., 9] & 4) != 0
    assert np.array_equal(valid, (b[..., 9] & 4) != 0)
    for w in (0, 1, 2, 3, 8, 9, 10, 11):   # rect, label, flags, stream, frame of the filled slots
        assert np.array_equal(a[..., w][valid], b[..., w][valid])
    lab, locd = (a[..., 9] & 1) != 0, (a[..., 9] & 2) != 0
    assert np.array_equal(a[..., 7][lab], b[..., 7][lab]) and np.array_equal(a[..., 4:7][locd], b[..., 4:7][locd])
    comm = rd.Comm("file", 0, 1, rd.Comm.unique_id("file"))
    assert np.array_equal(comm.all_gather_records(b)[0], b)
    comm.close()
    assert _lib.lib().rmr_stream_owner(5, 4) == 1


def test_file_transport_second_communicator_never_reads_stale_records(tmp_path):
    """Two communicators, one after the other, on ONE directory (a rank restarted after a crash, a caller-supplied
    path): the files of the first (left behind: it is never closed before the second runs) must not be read by the
    second, whose sequence numbers start at 0 again.  A clean close of every communicator removes the directory."""
    import threading
    sys.path.insert(0, ROOT)
    from rm_radar_amd import dist as rd
    os.environ["TMPDIR"] = str(tmp_path)
    uid = rd.Comm.unique_id("file")
    path = uid.split(b"\0")[0].decode()

    def session(tag, rounds, delay_rank1):
        comms, res = [None, None], [None, None]

        def run(rank):
            if rank == 1 and delay_rank1:
                import time
                time.sleep(0.3)      # rank 0 is already polling for rank 1's file when rank 1 arrives
            comms[rank] = rd.Comm("file", rank, 2, uid)
            got = []
            for k in range(rounds):
                mine = np.full((2, rd.RECORD_WORDS), tag * 1000 + k * 10 + rank, np.int32)
                got.append(comms[rank].all_gather_records(mine))
            res[rank] = got
        th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join(60)
        for rank in range(2):
            for k in range(rounds):
                for src in range(2):
                    assert (res[rank][k][src] == tag * 1000 + k * 10 + src).all(), (tag, rank, k, src)
        return comms

    first = session(1, 3, False)          # "crashed": its last files stay in the directory
    assert any(n.startswith("e0.") for n in os.listdir(path))
    second = session(2, 3, True)
    assert any(n.startswith("e1.") for n in os.listdir(path))

    def close_all(comms):
        th = [threading.Thread(target=c.close) for c in comms]
        for t in th:
            t.start()
        for t in th:
            t.join(30)
    close_all(second)
    assert os.path.isdir(path) and not any(n.startswith("e1.") for n in os.listdir(path))
    close_all(first)
    assert not os.path.exists(path)


def test_file_transport_epoch_is_collective_after_a_clean_close(tmp_path):
    """ADVICE r03 (api_comm.cpp): rank 1 leaves a cleanly closed communicator before rank 0 has swept the directory and
    at once joins the next one on the same path.  With per-rank epochs rank 1 (its old marker still there) and rank 0 (after
    its sweep) chose different epochs and the gather timed out; the epoch is now handed out by rank 0 in the join
    handshake, so both sessions and a third one after them exchange the right records."""
    import threading
    import time
    sys.path.insert(0, ROOT)
    from rm_radar_amd import dist as rd
    path = str(tmp_path / "shared_dir")     # a caller-supplied directory: survives its communicators
    os.makedirs(path)
    uid = path.encode()
    res = {}

    def rank_main(rank):
        for session in range(3):
            comm = rd.Comm("file", rank, 2, uid)
            mine = np.full((2, rd.RECORD_WORDS), session * 100 + rank, np.int32)
            got = comm.all_gather_records(mine)
            res[(session, rank)] = [int(got[src].flat[0]) for src in range(2)]
            if rank == 0:
                time.sleep(0.05)           # rank 1 is through its close and into the next create before rank 0 sweeps
            comm.close()
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(90)
    assert not any(t.is_alive() for t in th)
    for session in range(3):
        for rank in range(2):
            assert res[(session, rank)] == [session * 100, session * 100 + 1], (session, rank, res)
    assert not any(n.startswith(("hello.", "welcome.")) for n in os.listdir(path))


def test_file_transport_restarted_rank_gets_the_current_epoch(tmp_path):
    """A rank that crashed before the exchange leaves a hello file with a dead token; its restart publishes a new token
    and rank 0 answers THAT one (a welcome for the dead token is ignored)."""
    import threading
    sys.path.insert(0, ROOT)
    from rm_radar_amd import dist as rd
    path = str(tmp_path / "d")
    os.makedirs(path)
    open(os.path.join(path, "hello.1"), "w").write("12345")            # the crashed process's hello
    open(os.path.join(path, "welcome.1"), "w").write("999 7")          # a stale answer to an even older one
    open(os.path.join(path, "e4.3.1"), "wb").write(b"x" * 96)          # a record file an earlier epoch left behind
    out = [None, None]

    def run(rank):
        c = rd.Comm("file", rank, 2, path.encode())
        out[rank] = c.all_gather_records(np.full((2, rd.RECORD_WORDS), 40 + rank, np.int32))
        c.close()
    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    th[0].start()
    import time
    time.sleep(0.2)                                                     # rank 0 has already answered the dead token
    th[1].start()
    for t in th:
        t.join(60)
    for rank in range(2):
        assert [int(out[rank][s].flat[0]) for s in range(2)] == [40, 41]
    assert os.path.exists(os.path.join(path, "e4.3.1"))                 # epoch 5 never touched epoch 4's files
