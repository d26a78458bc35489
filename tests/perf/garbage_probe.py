"""Probe: garbage inputs (NaN, inf, zeros, negative / huge dims, huge yaw) — finite/NaN pattern of the HIP kernel vs the
fp64 oracle.  Not a parity requirement (the reference yields NaN/inf soup here too); documents the behaviour."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import mmdet3d_gaussian_amd as amd, oracle
rng = np.random.default_rng(0)
n = 64
t = np.stack([rng.uniform(0, 70, n), rng.uniform(-40, 40, n), rng.uniform(-3, 1, n), rng.uniform(.5, 2.5, n),
              rng.uniform(.5, 4.5, n), rng.uniform(.5, 2, n), rng.uniform(-3, 3, n)], -1).astype(np.float32)
p = (t + rng.normal(0, 0.2, (n, 7))).astype(np.float32)
specials = [np.nan, np.inf, -np.inf, 0.0, -1.0, 1e-30, 1e30, 1e8, -1e8]
k = 0
for col in range(7):
    for v in specials:
        if k < n: p[k, col] = v; k += 1
p[60, 6] = 1e6; p[61, 6] = 12345.678; t[62, 3] = 0.0; t[63, 5] = np.nan
for lt in ('gwd3d', 'kld3d', 'bd3d', 'jd3d', 'kld3d_symmax', 'kfiou3d'):
    fun = 'expm1' if lt == 'kfiou3d' else 'log1p'
    with np.errstate(all='ignore'):
        ref = oracle.gd_loss(p, t, oracle.make_params(lt, fun=fun), dtype=np.float32)   # fp32 oracle: same overflow behaviour
    pp = torch.from_numpy(p).cuda().requires_grad_(True)
    out = amd.GDLoss(lt, fun=fun, reduction='none')(pp, torch.from_numpy(t).cuda())
    out.sum().backward()
    got = out.detach().cpu().numpy(); gg = pp.grad.cpu().numpy()
    fin_ref = np.isfinite(ref['loss']); fin_got = np.isfinite(got)
    both = fin_ref & fin_got
    rel = np.abs(got[both] - ref['loss'][both]) / (1 + np.abs(ref['loss'][both]))
    gfin_ref = np.isfinite(ref['grad_pred']).all(-1); gfin_got = np.isfinite(gg).all(-1)
    print(f'{lt:13s} loss finite: ref {fin_ref.sum()} got {fin_got.sum()} pattern-mismatch rows {np.nonzero(fin_ref != fin_got)[0].tolist()} '
          f'max rel err on common-finite {rel.max() if rel.size else 0:.2e}; grad finite rows: ref {gfin_ref.sum()} got {gfin_got.sum()} '
          f'mismatch {np.nonzero(gfin_ref != gfin_got)[0].tolist()}')
    # what IS a requirement (tests/golden/gd_nonfinite.npz pins it against the real reference): a NaN input row gives a NaN loss
    assert np.isnan(got[np.isnan(p).any(-1) | np.isnan(t).any(-1)]).all() or lt == 'kfiou3d', lt
    bad = np.nonzero(both)[0][rel > 1e-4]
    for r in bad[:6]:
        print('    row', r, 'pred', p[r], 'got', got[r], 'ref32', ref['loss'][r])
