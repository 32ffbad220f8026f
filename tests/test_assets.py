"""Asset readers of the headless sample (rm_radar_amd/assets.py): PCD v0.7 ASCII / binary and the
frame reader, against the reference's sample clouds (tests/golden/assets_clouds.npz)."""
import os

import numpy as np
import pytest

from rm_radar_amd import assets

GOLD = os.path.join(os.path.dirname(__file__), "golden", "assets_clouds.npz")


@pytest.mark.parametrize("binary", [False, True])
def test_pcd_round_trip_of_a_reference_cloud(tmp_path, binary):
    cloud = np.load(GOLD)["cloud3"].astype(np.float32)  # 10 000 points, integer millimetres
    p = tmp_path / "3.pcd"
    assets.write_pcd(p, cloud, binary=binary)
    got = assets.read_pcd(p)
    assert got.dtype == np.float32 and got.shape == (10000, 3) and np.array_equal(got, cloud)


def test_pcd_with_extra_fields_and_comments(tmp_path):
    # x y z need not be the first fields; intensity / counts of 2 are skipped (ASCII and binary)
    pts = np.array([[1.5, -2, 3], [4, 5.25, -6]], np.float32)
    ascii_pcd = ("# comment\nVERSION 0.7\nFIELDS intensity x y z normal\nSIZE 4 4 4 4 4\nTYPE F F F F F\nCOUNT 1 1 1 1 2\n"
                 "WIDTH 2\nHEIGHT 1\nPOINTS 2\nDATA ascii\n0.5 1.5 -2 3 0 1\n0.25 4 5.25 -6 1 0\n")
    (tmp_path / "a.pcd").write_text(ascii_pcd)
    assert np.array_equal(assets.read_pcd(tmp_path / "a.pcd"), pts)
    head = ("VERSION 0.7\nFIELDS id x y z\nSIZE 2 4 4 4\nTYPE U F F F\nCOUNT 1 1 1 1\nWIDTH 2\nHEIGHT 1\nPOINTS 2\nDATA binary\n")
    rec = np.zeros(2, np.dtype([("id", "<u2"), ("x", "<f4"), ("y", "<f4"), ("z", "<f4")]))
    rec["id"], rec["x"], rec["y"], rec["z"] = [7, 8], pts[:, 0], pts[:, 1], pts[:, 2]
    (tmp_path / "b.pcd").write_bytes(head.encode() + rec.tobytes())
    assert np.array_equal(assets.read_pcd(tmp_path / "b.pcd"), pts)


def test_pcd_errors(tmp_path):
    (tmp_path / "c.pcd").write_text("VERSION 0.7\nFIELDS x y\nSIZE 4 4\nTYPE F F\nCOUNT 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA ascii\n1 2\n")
    with pytest.raises(ValueError, match="x, y and z"):
        assets.read_pcd(tmp_path / "c.pcd")
    (tmp_path / "d.pcd").write_text("VERSION 0.7\nFIELDS x y z\nPOINTS 1\nDATA binary_compressed\n")
    with pytest.raises(ValueError, match="not supported"):
        assets.read_pcd(tmp_path / "d.pcd")
    (tmp_path / "e.pcd").write_text("VERSION 0.7\nFIELDS x y z\nPOINTS 3\nDATA ascii\n1 2 3\n")
    with pytest.raises(ValueError, match="values for 3 points"):
        assets.read_pcd(tmp_path / "e.pcd")


def test_image_reader(tmp_path):
    img = np.random.default_rng(0).integers(0, 256, (6, 9, 3), dtype=np.uint8)
    np.save(tmp_path / "0.npy", img)
    assert np.array_equal(assets.read_image(str(tmp_path / "0.npy")), img)
    pil = pytest.importorskip("PIL.Image")
    pil.fromarray(img[:, :, ::-1]).save(tmp_path / "1.png")  # files hold RGB; the reader returns BGR like cv::imread
    assert np.array_equal(assets.read_image(str(tmp_path / "1.png")), img)
    assert assets.find_frame(str(tmp_path), 1, (".jpg", ".png", ".npy")).endswith("1.png")
    assert assets.find_frame(str(tmp_path), 5, (".jpg",)) is None
    np.save(tmp_path / "2.npy", np.zeros((4, 4), np.uint8))
    with pytest.raises(ValueError):
        assets.read_image(str(tmp_path / "2.npy"))
