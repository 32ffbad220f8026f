"""Stream-sharded multi-GPU layer (no reference counterpart: the reference is
single-GPU, detector.cpp:61).

Frames shard by STREAM: rank r owns the camera/LiDAR streams {s : s % world == r} together with
their Locator state (background image + depth ring are temporal state, so a stream never
migrates).  Weights are replicated.  The data path needs no collective; the only exchange is one
all-gather per batch of the final robot list as fixed-size records (rmr_robot_record, 48 B), so
every rank ends with the world-frame robots of all streams.  One process per GPU,
torch.distributed backend "nccl" (= RCCL over xGMI); the message is a few KB per rank, i.e.
latency-bound, so records of a whole batch of frames travel in one call.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

# numpy mirror of rmr_robot (include/rmr.h) for vectorised packing
ROBOT_DTYPE = np.dtype([("rect", np.float32, 4), ("has_label", np.int32), ("label", np.int32),
                        ("confidence", np.float32), ("n_armors", np.int32),
                        ("armors", np.float32, (_lib.MAX_ARMORS, 6)), ("has_location", np.int32),
                        ("location", np.float32, 3), ("track_state", np.int32)])
assert ROBOT_DTYPE.itemsize == C.sizeof(_lib.Robot)

# rmr_robot_record as 12 x 4-byte words: rect[4], location[3], confidence, label, flags,
# stream_id, frame_id
RECORD_WORDS = 12
assert C.sizeof(_lib.RobotRecord) == RECORD_WORDS * 4


def streams_of_rank(n_streams: int, rank: int, world: int):
    """Stream -> GPU assignment: stream s lives on rank s % world."""
    return [s for s in range(n_streams) if s % world == rank]


def pack_records(robots_c, counts, cap: int, stream_id: int, max_per_frame: int) -> np.ndarray:
    """ctypes rmr_robot[n_frames*cap] + counts -> int32 [n_frames, max_per_frame, 12] block
    (float fields bit-cast: the wire format is raw rmr_robot_record words), zero padded; slot
    validity is flags bit2."""
    n_frames = len(counts)
    arr = np.frombuffer(robots_c, dtype=ROBOT_DTYPE, count=n_frames * cap).reshape(n_frames, cap)
    oi = np.zeros((n_frames, max_per_frame, RECORD_WORDS), np.int32)
    out = oi.view(np.float32)
    m = min(cap, max_per_frame)
    valid = np.arange(m)[None, :] < np.minimum(counts, m)[:, None]
    a = arr[:, :m]
    out[:, :m, 0:4] = a["rect"]
    out[:, :m, 4:7] = a["location"]
    out[:, :m, 7] = a["confidence"]
    oi[:, :m, 8] = np.where(a["has_label"] != 0, a["label"], -1)
    oi[:, :m, 9] = (a["has_label"] != 0) * 1 + (a["has_location"] != 0) * 2 + 4
    oi[:, :m, 10] = stream_id
    oi[:, :m, 11] = np.arange(n_frames)[:, None]
    oi[:, :m][~valid] = 0
    return oi


def unpack_records(block: np.ndarray):
    """[..., 12] int32 block -> list of dicts for the valid slots."""
    bi = np.ascontiguousarray(block, np.int32).reshape(-1, RECORD_WORDS)
    b = bi.view(np.float32)
    res = []
    for r, ri in zip(b, bi):
        if not (ri[9] & 4):
            continue
        res.append(dict(rect=tuple(float(v) for v in r[0:4]),
                        location=tuple(float(v) for v in r[4:7]) if ri[9] & 2 else None,
                        confidence=float(r[7]) if ri[9] & 1 else None,
                        label=int(ri[8]) if ri[9] & 1 else None,
                        stream_id=int(ri[10]), frame_id=int(ri[11])))
    return res


def all_gather_records(block, group=None, force=False):
    """block: torch tensor [n_frames, max_per_frame, 12] on this rank's device (or CPU for gloo).
    Returns [world, n_frames, max_per_frame, 12].  One collective per batch of frames."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return block.unsqueeze(0)
    if dist.get_world_size(group) == 1 and not force:
        return block.unsqueeze(0)
    world = dist.get_world_size(group)
    block = block.contiguous()
    out = torch.empty((world * block.shape[0],) + tuple(block.shape[1:]), dtype=block.dtype,
                      device=block.device)
    dist.all_gather_into_tensor(out, block, group=group)
    return out.view((world,) + tuple(block.shape))


class Comm:
    """The C-ABI communicator (include/rmr.h: rmr_comm_*): what a C++ host would use, mirrored for Python hosts.
    transport: "rccl" (ncclAllGather over xGMI on `device`) or "file" (a shared directory; no GPU needed).
    `unique_id()` on rank 0, the 128 bytes to every rank by any channel, then `Comm(...)` on all ranks."""
    TRANSPORTS = {"rccl": 0, "file": 1}

    @staticmethod
    def unique_id(transport="rccl") -> bytes:
        buf = C.create_string_buffer(128)
        _lib.check(_lib.lib().rmr_comm_unique_id(Comm.TRANSPORTS[transport], buf))
        return buf.raw

    def __init__(self, transport, rank, world, uid: bytes, device=0):
        self.rank, self.world = rank, world
        self._h = C.c_void_p()
        _lib.check(_lib.lib().rmr_comm_create(Comm.TRANSPORTS[transport], device, rank, world,
                                              C.create_string_buffer(uid, 128), C.byref(self._h)))

    def all_gather_records(self, block: np.ndarray) -> np.ndarray:
        """block: int32 [..., 12] records of this rank -> [world, ..., 12] on every rank."""
        block = np.ascontiguousarray(block, np.int32)
        n = block.size // RECORD_WORDS
        out = np.empty((self.world,) + block.shape, np.int32)
        _lib.check(_lib.lib().rmr_comm_all_gather_records(self._h, block.ctypes.data, n, out.ctypes.data))
        return out

    def close(self):
        if self._h:
            _lib.lib().rmr_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_records_abi(robots_c, counts, cap: int, stream_id: int, max_per_frame: int) -> np.ndarray:
    """pack_records through the C-ABI (rmr_pack_robot_records): the same [n_frames, max_per_frame, 12] block."""
    counts = np.ascontiguousarray(counts, np.int32)
    out = np.zeros((len(counts), max_per_frame, RECORD_WORDS), np.int32)
    _lib.check(_lib.lib().rmr_pack_robot_records(C.addressof(robots_c), _lib.ip(counts), len(counts), cap, stream_id,
                                                 max_per_frame, out.ctypes.data))
    return out
