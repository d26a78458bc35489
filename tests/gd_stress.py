"""Stress families for the Gaussian-distance losses (shared by tests/test_gpu_gd_loss.py, tests/test_hostmath.py and
tests/golden/fuzz_device_math_vs_reference.py): KITTI-range pairs pushed out of the comfortable regime one way at a time."""
import numpy as np

FAMILIES = ('hugedim', 'tinydim', 'aspect', 'farcentre', 'fardist', 'bigyaw', 'negdim', 'yaw90', 'square')
# ill-conditioned by construction (dims over 12 decades, or clamped to 1e-7 thickness): isolated rows may exceed 3x the
# fp32 yardstick
ILL_CONDITIONED = ('hugedim', 'tinydim', 'negdim')


def stress_pairs(n, kind, seed=0):
    rng = np.random.default_rng(seed)
    t = np.stack([rng.uniform(0, 70, n), rng.uniform(-40, 40, n), rng.uniform(-3, 1, n), rng.uniform(0.5, 2.5, n),
                  rng.uniform(0.5, 4.5, n), rng.uniform(0.5, 2, n), rng.uniform(-3.14, 3.14, n)], -1)
    p = t + rng.normal(0, 1, (n, 7)) * np.array([0.3, 0.3, 0.1, 0.1, 0.1, 0.1, 0.1])
    if kind == 'hugedim':       # boxes up to kilometres
        p[:, 3:6] *= 10 ** rng.uniform(0, 3, (n, 3)); t[:, 3:6] *= 10 ** rng.uniform(0, 3, (n, 3))
    elif kind == 'tinydim':     # down to and below the 1e-7 clamp
        p[:, 3:6] *= 10 ** rng.uniform(-9, 0, (n, 3)); t[:, 3:6] *= 10 ** rng.uniform(-9, 0, (n, 3))
    elif kind == 'aspect':      # 1 : 1000 footprints, crossed
        p[:, 3] *= 1e3; t[:, 4] *= 1e3
    elif kind == 'farcentre':   # both boxes 1e2 .. 1e6 m from the origin
        off = 10 ** rng.uniform(2, 6, (n, 1)); p[:, :3] += off; t[:, :3] += off
    elif kind == 'fardist':     # boxes 10 .. 1e8 m apart
        p[:, :3] += 10 ** rng.uniform(1, 8, (n, 3))
    elif kind == 'bigyaw':      # yaws of a few hundred radians
        p[:, 6] += rng.uniform(-300, 300, n); t[:, 6] += rng.uniform(-300, 300, n)
    elif kind == 'negdim':      # negative sizes (clamped to 1e-7)
        p[::3, 3:6] *= -1; t[::5, 4] *= -1
    elif kind == 'yaw90':       # within 1e-4 rad of a quarter-turn multiple
        t[:, 6] = p[:, 6] + np.pi / 2 * rng.integers(-2, 3, n) + rng.normal(0, 1e-4, n)
    elif kind == 'square':      # square footprints (the yaw is unobservable)
        p[:, 4] = p[:, 3]; t[:, 4] = t[:, 3] * (1 + rng.normal(0, 1e-6, n))
    else:
        raise ValueError(kind)
    return p.astype(np.float32), t.astype(np.float32)
