"""Asset readers of the headless sample (samples/main.cpp:24-77 uses cv::imread and
pcl::io::loadPCDFile; neither library exists here): a PCD reader for the public PCD v0.7 format
(ASCII and uncompressed binary, fields x y z in any position) and an image reader (JPEG/PNG through
PIL when it is installed, `.npy` arrays always)."""
from __future__ import annotations

import os

import numpy as np

_NP = {("F", 4): "<f4", ("F", 8): "<f8", ("I", 1): "<i1", ("I", 2): "<i2", ("I", 4): "<i4", ("I", 8): "<i8",
       ("U", 1): "<u1", ("U", 2): "<u2", ("U", 4): "<u4", ("U", 8): "<u8"}


def read_pcd(path) -> np.ndarray:
    """-> [n, 3] float32 (x, y, z) of a .pcd file."""
    with open(path, "rb") as f:
        raw = f.read()
    hdr, pos = {}, 0
    while True:
        end = raw.find(b"\n", pos)
        if end < 0:
            raise ValueError(f"{path}: PCD header has no DATA line")
        line = raw[pos:end].decode("ascii", "replace").strip()
        pos = end + 1
        if not line or line.startswith("#"):
            continue
        key, _, val = line.partition(" ")
        hdr[key.upper()] = val.split()
        if key.upper() == "DATA":
            break
    fields = hdr.get("FIELDS") or hdr.get("COLUMNS")
    if not fields or not all(k in fields for k in ("x", "y", "z")):
        raise ValueError(f"{path}: PCD needs x, y and z fields")
    sizes = [int(v) for v in hdr.get("SIZE", ["4"] * len(fields))]
    types = hdr.get("TYPE", ["F"] * len(fields))
    counts = [int(v) for v in hdr.get("COUNT", ["1"] * len(fields))]
    n = int(hdr["POINTS"][0]) if "POINTS" in hdr else int(hdr["WIDTH"][0]) * int(hdr.get("HEIGHT", ["1"])[0])
    kind = hdr["DATA"][0].lower()
    cols = np.cumsum([0] + counts)
    if kind == "ascii":
        vals = np.array(raw[pos:].split(), dtype=np.float64)
        if vals.size < n * cols[-1]:
            raise ValueError(f"{path}: {vals.size} values for {n} points of {cols[-1]} columns")
        tab = vals[: n * cols[-1]].reshape(n, cols[-1])
        return np.stack([tab[:, cols[fields.index(k)]] for k in ("x", "y", "z")], 1).astype(np.float32)
    if kind == "binary":
        dt = np.dtype([(f"{name}{i}", _NP[(t.upper(), s)], (c,)) for i, (name, t, s, c) in
                       enumerate(zip(fields, types, sizes, counts))])
        if len(raw) - pos < n * dt.itemsize:
            raise ValueError(f"{path}: truncated binary PCD")
        tab = np.frombuffer(raw, dt, n, pos)
        return np.stack([tab[f"{k}{fields.index(k)}"][:, 0] for k in ("x", "y", "z")], 1).astype(np.float32)
    raise ValueError(f"{path}: PCD DATA '{kind}' is not supported (ascii and binary are)")


def write_pcd(path, xyz, binary=False) -> None:
    """Writes [n, 3] points as PCD v0.7 (tests, fixtures)."""
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\n"
            f"COUNT 1 1 1\nWIDTH {len(xyz)}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {len(xyz)}\n"
            f"DATA {'binary' if binary else 'ascii'}\n")
    with open(path, "wb") as f:
        f.write(head.encode())
        if binary:
            f.write(xyz.astype("<f4").tobytes())
        else:
            f.write("".join(f"{a:.9g} {b:.9g} {c:.9g}\n" for a, b, c in xyz).encode())


def read_image(path) -> np.ndarray:
    """-> HxWx3 uint8 BGR (what cv::imread returns)."""
    if str(path).endswith(".npy"):
        img = np.load(path)
    else:
        try:
            from PIL import Image
        except ImportError as e:  # pragma: no cover
            raise RuntimeError(f"{path}: reading JPEG/PNG needs PIL; convert the frame to .npy") from e
        img = np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1]
    img = np.ascontiguousarray(img)
    if img.ndim != 3 or img.shape[2] != 3 or img.dtype != np.uint8:
        raise ValueError(f"{path}: expected an HxWx3 uint8 image")
    return img


def find_frame(folder, index, exts):
    for e in exts:
        p = os.path.join(folder, f"{index}{e}")
        if os.path.exists(p):
            return p
    return None
