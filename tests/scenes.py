"""Seeded synthetic LiDAR scenes: background = wall + ground seen through every
image column at 6-29 m, foreground = boxes 1-3 m in front of it inside known image rects;
integer-millimetre coordinates, ~1.5 % exact-zero points and ~10 % beyond max_distance."""
import numpy as np

# samples/main.cpp:12-22 (calibration constants of the reference's sample)
SAMPLE_SIZE = (2592, 2048)
SAMPLE_K = np.array([[1685.51538398561, 0, 1278.99324114319],
                     [0, 1685.26471848220, 1037.21273138299],
                     [0, 0, 1]], np.float32)
SAMPLE_L2C = np.array([[0, -1, 0, 0.85443], [0, 0, -1, -37.6845], [1, 0, 0, 12.2631],
                       [0, 0, 0, 1]], np.float32)
SAMPLE_W2C = np.array([[0.05975021, 0.99807031, 0.01689906, -7179.65399136],
                       [0.28962566, -0.00113262, -0.95713933, -4671.34956587],
                       [-0.9552732, 0.06208368, -0.28913445, 28286.8920291],
                       [0, 0, 0, 1]], np.float32)
# calibration for 640x640 synthetic frames
K640 = np.array([[416, 0, 320], [0, 416, 320], [0, 0, 1]], np.float32)


def _backproject(K, L2C, u, v, d):
    """pixel (u,v) at camera depth d (mm) -> lidar frame, float64"""
    Kinv = np.linalg.inv(K.astype(np.float64))
    cam = (Kinv @ np.stack([u, v, np.ones_like(u)])) * d
    C2L = np.linalg.inv(L2C.astype(np.float64))
    return (C2L[:3, :3] @ cam + C2L[:3, 3:4]).T


def background_depth(u, v, width, height):
    """smooth background: far wall at the top, ground plane coming closer towards the bottom"""
    t = v / height
    return 29000.0 - 22000.0 * t ** 1.5 + 400.0 * np.sin(u / width * 9.0)


def make_cloud(rng, n, K, L2C, size, robots=(), zero_frac=0.015, far_frac=0.10):
    """robots: list of (rect=(x,y,w,h) in image px, gap_mm, n_points)"""
    width, height = size
    n_zero = int(n * zero_frac)
    n_far = int(n * far_frac)
    n_rob = sum(r[2] for r in robots)
    n_bg = n - n_zero - n_far - n_rob
    u = rng.uniform(0, width, n_bg)
    v = rng.uniform(0, height, n_bg)
    pts = [_backproject(K, L2C, u, v, background_depth(u, v, width, height))]
    for (x, y, w, h), gap, k in robots:
        ur = rng.uniform(x + 0.15 * w, x + 0.85 * w, k)
        vr = rng.uniform(y + 0.15 * h, y + 0.85 * h, k)
        d = background_depth(ur, vr, width, height) - gap + rng.uniform(-150, 150, k)
        pts.append(_backproject(K, L2C, ur, vr, d))
    far = rng.uniform(0, 1, (n_far, 3)) * [5000, 8000, 3000] + [29400, -4000, -1000]
    pts.append(far)
    pts.append(np.zeros((n_zero, 3)))
    cloud = np.rint(np.concatenate(pts)).astype(np.float32)
    rng.shuffle(cloud)
    out = np.zeros((n, 4), np.float32)  # pcl::PointXYZ: x y z pad (16 B)
    out[:, :3] = cloud
    return out


def scene(seed, n_points=30000, size=(640, 640), K=K640, L2C=SAMPLE_L2C, n_frames=6, n_robots=4):
    """-> (clouds[n_frames], rects per frame).  Frame 0.. are background only until the ring
    and background image are warm; robots appear from frame 2 on and drift."""
    rng = np.random.default_rng(seed)
    width, height = size
    base = []
    for _ in range(n_robots):
        w = rng.uniform(0.08, 0.2) * width
        h = rng.uniform(0.08, 0.2) * height
        x = rng.uniform(0.02 * width, 0.98 * width - w)
        y = rng.uniform(0.25 * height, 0.98 * height - h)
        base.append([x, y, w, h, rng.uniform(1000, 3000), int(rng.integers(40, 400))])
    clouds, rects = [], []
    for f in range(n_frames):
        robots = []
        if f >= 2:
            for b in base:
                dx = (f - 2) * 0.01 * width
                x = min(b[0] + dx, width - b[2] - 1)
                robots.append(((x, b[1], b[2], b[3]), b[4], b[5]))
        clouds.append(make_cloud(rng, n_points, K, L2C, size, robots))
        rects.append([r[0] for r in robots])
    return clouds, rects


def synthetic_image(seed, size=(640, 640)):
    """default_rng(1234 + frame) uniform u8 image, HxWx3 BGR"""
    w, h = size
    return np.random.default_rng(1234 + seed).integers(0, 256, (h, w, 3), dtype=np.uint8)
