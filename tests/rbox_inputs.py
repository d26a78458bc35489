"""Seeded synthetic box sets for the rotated NMS / IoU tests (numpy, no torch RNG dependence)."""
import numpy as np


def nms_boxes(n, seed=0, extent=74.88, clutter=True):
    """Waymo-like BEV boxes [x1,y1,x2,y2,ry] + scores (SURVEY.md §8d config 5 stand-in): car/ped/cyc
    sizes (hv_pointpillars_secfpn_waymo.py:51-55) scattered in [-extent, extent]^2; with `clutter`
    detections come in clusters of jittered duplicates, as a dense head produces them."""
    rng = np.random.default_rng(seed)
    sizes = np.array([[4.73, 2.08], [0.91, 0.84], [1.81, 0.84]], np.float32)
    if clutter:
        nc = max(1, n // 8)
        cx = rng.uniform(-extent, extent, nc); cy = rng.uniform(-extent, extent, nc)
        cls = rng.integers(0, 3, nc); yaw = rng.uniform(-np.pi, np.pi, nc)
        idx = rng.integers(0, nc, n)
        x = cx[idx] + rng.normal(0, 0.3, n); y = cy[idx] + rng.normal(0, 0.3, n)
        wl = sizes[cls[idx]] * rng.uniform(0.9, 1.1, (n, 2))
        r = yaw[idx] + rng.normal(0, 0.1, n)
    else:
        x = rng.uniform(-extent, extent, n); y = rng.uniform(-extent, extent, n)
        wl = sizes[rng.integers(0, 3, n)] * rng.uniform(0.9, 1.1, (n, 2))
        r = rng.uniform(-np.pi, np.pi, n)
    boxes = np.stack([x - wl[:, 0] / 2, y - wl[:, 1] / 2, x + wl[:, 0] / 2, y + wl[:, 1] / 2, r], -1)
    scores = rng.uniform(0, 1, n)
    return boxes.astype(np.float32), scores.astype(np.float32)


def eval_boxes(n, seed=0, spread=12.0):
    rng = np.random.default_rng(seed)
    return np.stack([rng.uniform(0, spread, n), rng.uniform(0, spread, n), rng.uniform(-3, 1, n),
                     rng.uniform(0.5, 2.5, n), rng.uniform(0.5, 4.5, n), rng.uniform(0.5, 2.0, n),
                     rng.uniform(-np.pi, np.pi, n)], -1).astype(np.float32)
