"""§8f-2: the torch statement of the CenterPoint yaw coder (oracle/coder_torch.py, the checker of the device coder) vs
golden outputs of the REAL reference classes, and the head-level oracle (decode + GD loss + chain rule) vs the real
reference composition.  CPU only."""
import os

import numpy as np
import pytest
import torch

import oracle
from mmdet3d_gaussian_amd.coders import DeltaXYZWLHRBBoxCoder
from oracle import coder_torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'coder_center.npz')
CASES = (('gwd3d', dict(fun='log1p', tau=0.0)), ('bd3d', dict(fun='log1p', tau=1.0)), ('kld3d', dict(fun='none', tau=0.0)))


def _cfg(g):
    return dict(pc_range=g['cfg_pc_range'].tolist(), out_size_factor=int(g['cfg_out_size_factor']),
                voxel_size=g['cfg_voxel_size'].tolist(), norm_bbox=True)


def test_center_coder_statement_bit_exact_with_reference():
    g = np.load(GOLD)
    locs, pred, anno = torch.from_numpy(g['locs']), torch.from_numpy(g['pred']), torch.from_numpy(g['anno'])
    np.testing.assert_array_equal(coder_torch.center_decode(locs, pred, correct_yaw=False, **_cfg(g)).numpy(), g['decode_noyaw32'])
    np.testing.assert_array_equal(coder_torch.center_decode(locs, pred, correct_yaw=True, **_cfg(g)).numpy(), g['decode_yaw32'])
    enc = coder_torch.center_encode(anno)
    np.testing.assert_array_equal(enc[..., :7].numpy(), g['enc7'])
    assert enc.shape[-1] == 11


@pytest.mark.parametrize('lt,kw', CASES)
def test_decoded_oracle_matches_reference_head_slice(lt, kw):
    g = np.load(GOLD)
    pred = g['pred'].reshape(-1, 11).astype(np.float64)
    anno = g['anno'].reshape(-1, 9).astype(np.float64)
    locs = g['locs'].reshape(-1, 2).astype(np.float64)
    avg = float(g['avg_factor'])
    r = oracle.gd_loss_decoded(pred[:, :7], anno[:, :7], oracle.make_params(lt, **kw), oracle.PRO_CENTER, locs,
                               scale=5.0 / avg, norm_bbox=True, out_size_factor=4, voxel_size=(0.2, 0.2),
                               pc_range=(-51.2, -51.2))
    assert abs(r['loss_sum'] - float(g[lt + '.loss64'])) < 1e-12
    gref = g[lt + '.gpred64'].reshape(-1, 11)
    np.testing.assert_allclose(r['grad_pred'], gref[:, :7], rtol=1e-10, atol=1e-14)
    assert np.abs(gref[:, 7:]).max() == 0.0        # dir / velocity channels do not reach the GD loss


def test_anchor_delta_coder_roundtrip_and_oracle_consistency():
    """mmdet3d's anchor coder is third-party and unpinned: check the restated formulas for self-consistency
    (decode(encode(x)) == x) and the decoded oracle against decode-then-plain-oracle with finite differences."""
    rng = np.random.default_rng(0)
    n = 200
    anchors = np.stack([rng.uniform(0, 70, n), rng.uniform(-40, 40, n), rng.uniform(-2, 0, n), rng.uniform(.6, 2, n),
                        rng.uniform(.8, 4, n), rng.uniform(1.4, 1.8, n), rng.choice([0, np.pi / 2], n)], -1)
    gt = anchors + rng.normal(0, 0.2, (n, 7))
    gt[:, 3:6] = np.abs(gt[:, 3:6]) + 0.3
    ta, tg = torch.from_numpy(anchors), torch.from_numpy(gt)
    enc = DeltaXYZWLHRBBoxCoder.encode(ta, tg)
    np.testing.assert_allclose(DeltaXYZWLHRBBoxCoder.decode(ta, enc).numpy(), gt, rtol=1e-12, atol=1e-12)
    pred_enc = enc.numpy() + rng.normal(0, 0.05, (n, 7))
    prm = oracle.make_params('kld3d', fun='log1p', tau=1.0)
    r = oracle.gd_loss_decoded(pred_enc, enc.numpy(), prm, oracle.PRO_ANCHOR_DELTA, anchors)
    dec_p = DeltaXYZWLHRBBoxCoder.decode(ta, torch.from_numpy(pred_enc)).numpy()
    plain = oracle.gd_loss(dec_p, gt, prm)
    np.testing.assert_allclose(r['loss'], plain['loss'], rtol=1e-10)
    # chain rule: central finite differences on the encoded prediction
    eps = 1e-6
    for k in range(7):
        d = np.zeros((n, 7)); d[:, k] = eps
        lp = oracle.gd_loss_decoded(pred_enc + d, enc.numpy(), prm, oracle.PRO_ANCHOR_DELTA, anchors)['loss']
        lm = oracle.gd_loss_decoded(pred_enc - d, enc.numpy(), prm, oracle.PRO_ANCHOR_DELTA, anchors)['loss']
        np.testing.assert_allclose((lp - lm) / (2 * eps), r['grad_pred'][:, k], rtol=2e-5, atol=1e-7)


POINT_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'coder_point.npz')


def test_point_coder_statement_bit_exact_with_reference():
    """oracle/coder_torch.py point_decode / center_encode vs the REAL PointBBoxYawCoder (tests/golden/coder_point.npz): values bit
    for bit in fp32 and fp64, autograd gradients wrt preds likewise."""
    g = np.load(POINT_GOLD)
    for dtype, t in ((torch.float32, '32'), (torch.float64, '64')):
        priors = torch.from_numpy(g['priors']).to(dtype)
        up = torch.from_numpy(g['up']).to(dtype)
        for cy, tag in ((False, 'noyaw'), (True, 'yaw')):
            p = torch.from_numpy(g['preds']).to(dtype).requires_grad_(True)
            d = coder_torch.point_decode(priors, p, correct_yaw=cy)
            (d * up).sum().backward()
            np.testing.assert_array_equal(d.detach().numpy(), g[f'decode_{tag}{t}'])
            np.testing.assert_array_equal(p.grad.numpy(), g[f'gpreds_{tag}{t}'])
        np.testing.assert_array_equal(coder_torch.center_encode(torch.from_numpy(g['boxes']).to(dtype)).numpy(), g[f'encode{t}'])
    assert g['decode_yaw32'].shape[-1] == 10 and g['encode32'].shape[-1] == 11
    assert 0.4 < (g['decode_yaw32'][..., 3] != g['decode_noyaw32'][..., 3]).mean() < 0.7      # odd quarter turns are exercised
