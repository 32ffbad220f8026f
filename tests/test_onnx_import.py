"""ONNX initialiser -> weight pack converter (rm_radar_amd/onnx_import.py): round trips through a
hand-encoded ONNX protobuf and through a file written by google.protobuf's own encoder from run-time
declared onnx.proto3 messages (tests/onnx_writer.py; the onnx / torch.onnx writers are not usable in this
image: both need the `onnx` package), and the imported pack through the Detector on the GPU."""
import os
import struct

import numpy as np
import pytest

from rm_radar_amd import onnx_import as OI
from rm_radar_amd import weights as W


def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(fn, payload):
    return _varint((fn << 3) | 2) + _varint(len(payload)) + payload


def _tensor_proto(name, arr, mode):
    msg = b""
    if mode == "packed_dims":
        msg += _ld(1, b"".join(_varint(d) for d in arr.shape))
    else:
        msg += b"".join(_varint((1 << 3) | 0) + _varint(d) for d in arr.shape)
    if mode == "f16_raw":
        msg += _varint((2 << 3) | 0) + _varint(10) + _ld(9, arr.astype("<f2").tobytes())
    elif mode == "float_data":
        msg += _varint((2 << 3) | 0) + _varint(1) + _ld(4, arr.astype("<f4").tobytes())
    else:
        msg += _varint((2 << 3) | 0) + _varint(1) + _ld(9, arr.astype("<f4").tobytes())
    msg += _ld(8, name.encode())
    return msg


def _onnx_bytes(tensors, mode="raw"):
    graph = _ld(2, b"main_graph")
    # an int64 initialiser and a node must be skipped by the reader
    graph += _ld(5, _varint((1 << 3) | 0) + _varint(2) + _varint((2 << 3) | 0) + _varint(7) + _ld(8, b"shape_const") + _ld(9, struct.pack("<2q", 1, 2)))
    graph += _ld(1, _ld(4, b"Conv"))
    for name, arr in tensors.items():
        graph += _ld(5, _tensor_proto(name, arr, mode))
    return _varint((1 << 3) | 0) + _varint(8) + _ld(2, b"pytorch") + _ld(7, graph)


@pytest.mark.parametrize("mode", ["raw", "float_data", "packed_dims"])
def test_round_trip_f32(tmp_path, mode):
    t = W.synthesize("m", 12, seed=5)
    onnx = tmp_path / "armor.onnx"
    onnx.write_bytes(_onnx_bytes(t, mode))
    scale, nc = OI.onnx_to_pack(str(onnx), str(tmp_path / "armor.rmrw"))
    assert (scale, nc) == ("m", 12)
    got, meta = W.load_pack(str(tmp_path / "armor.rmrw"))
    assert meta["scale"] == "m" and meta["nc"] == 12
    assert list(got) == list(t)
    for k in t:
        assert np.array_equal(got[k], t[k]), k


def test_round_trip_f16_and_scale_s(tmp_path):
    t = W.synthesize("s", 1, seed=6)
    onnx = tmp_path / "car.onnx"
    onnx.write_bytes(_onnx_bytes(t, "f16_raw"))
    assert OI.onnx_to_pack(str(onnx), str(tmp_path / "car.rmrw")) == ("s", 1)
    got, _ = W.load_pack(str(tmp_path / "car.rmrw"))
    for k in t:
        assert np.array_equal(got[k], t[k].astype(np.float16).astype(np.float32)), k


def test_ensure_pack_builds_from_sibling_onnx(tmp_path):
    t = W.synthesize("m", 1, seed=7)
    (tmp_path / "car.onnx").write_bytes(_onnx_bytes(t))
    p = OI.ensure_pack(str(tmp_path / "car.rmrw"))  # detector.cpp:74-99: engine missing -> build it
    assert os.path.exists(p)
    mtime = os.path.getmtime(p)
    assert OI.ensure_pack(p) == p and os.path.getmtime(p) == mtime  # cached: not rebuilt
    assert OI.ensure_pack(str(tmp_path / "nothing.rmrw")).endswith("nothing.rmrw")  # left for the ctor to reject


def test_rejects_renamed_or_incomplete_graphs(tmp_path):
    t = W.synthesize("m", 1, seed=8)
    renamed = {f"onnx::Conv_{i}": v for i, v in enumerate(t.values())}
    (tmp_path / "a.onnx").write_bytes(_onnx_bytes(renamed))
    with pytest.raises(ValueError, match="module names"):
        OI.onnx_to_pack(str(tmp_path / "a.onnx"), str(tmp_path / "a.rmrw"))
    partial = dict(t)
    del partial["model.4.cv2.conv.bias"]
    (tmp_path / "b.onnx").write_bytes(_onnx_bytes(partial))
    with pytest.raises(ValueError, match="missing tensor"):
        OI.onnx_to_pack(str(tmp_path / "b.onnx"), str(tmp_path / "b.rmrw"))
    (tmp_path / "c.onnx").write_bytes(b"\x08\x08")
    with pytest.raises(ValueError, match="no graph"):
        OI.onnx_to_pack(str(tmp_path / "c.onnx"), str(tmp_path / "c.rmrw"))


def test_round_trip_through_protobuf_written_file(tmp_path):
    """A second, independent writer: google.protobuf serialises the messages (packed repeated fields, its own
    field order), initialisers alternate between raw f32, float_data and FLOAT16 raw, Conv nodes and int64
    shape constants sit between them."""
    import onnx_writer
    t = W.synthesize("m", 12, seed=15)
    path = onnx_writer.write_yolov8_onnx(str(tmp_path / "armor.onnx"), t)
    assert OI.onnx_to_pack(path, str(tmp_path / "armor.rmrw")) == ("m", 12)
    got, _ = W.load_pack(str(tmp_path / "armor.rmrw"))
    assert list(got) == list(t)
    for i, k in enumerate(t):
        want = t[k].astype(np.float16).astype(np.float32) if i % 3 == 2 else t[k]
        assert np.array_equal(got[k], want), k


@pytest.mark.gpu
def test_imported_onnx_runs_through_the_detector(tmp_path):
    """detector.cpp:74-99 / 177-243: no engine on disk, an ONNX file next to it -> the engine is built from
    the ONNX and the detector runs on it.  Here: armor.onnx written by the protobuf writer (raw f32 only),
    Detector('armor.rmrw') builds the pack from it, and its network output equals -- bit for bit -- the
    output on a pack saved directly from the same tensors, and the CPU oracle within the f16 tolerance."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import netutil
    import onnx_writer
    import oracle
    import rm_radar_amd as rmr
    from oracle import yolov8_ref as R
    images = [netutil.test_image(1), netutil.test_image(2, 810, 1080)]
    direct = netutil.tuned_pack(str(tmp_path / "direct.rmrw"), 12, 31, 0.5, 0.01, images)
    tensors, _ = W.load_pack(direct)
    onnx_writer.write_yolov8_onnx(str(tmp_path / "armor.onnx"), tensors, forms=("raw",))
    assert not os.path.exists(tmp_path / "armor.rmrw")
    det = rmr.Detector(str(tmp_path / "armor.rmrw"), 12, (1920, 1080), 2)   # builds the pack from armor.onnx
    assert os.path.exists(tmp_path / "armor.rmrw")
    got, _ = det.infer(images)
    det.close()
    # same kernels for both packs: the second detector runs the first one's plan
    os.replace(str(tmp_path / "armor.rmrw.tune"), str(tmp_path / "direct.rmrw.tune"))
    ref = rmr.Detector(direct, 12, (1920, 1080), 2)
    want, _ = ref.infer(images)
    ref.close()
    assert got.tobytes() == want.tobytes()
    blobs = np.stack([oracle.preprocess(im)[0] for im in images])
    cpu = R.load(str(tmp_path / "armor.rmrw"), True).forward(blobs)
    assert np.abs(got[:, :4] - cpu[:, :4]).max() <= 2.0 and np.abs(got[:, 4:] - cpu[:, 4:]).max() <= 1e-2
