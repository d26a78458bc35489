"""The round-3 CPU restatements that have no reference fixture (third-party pieces are absent): hand-computed cases and
independent closed forms for the target-assignment and heat-map-loss oracles.  CPU only."""
import math

import numpy as np
import torch

from oracle import center_targets_torch as ct
from oracle import heat_focal_torch as hf

CFG = dict(grid_size=[32, 32, 1], point_cloud_range=[0.0, 0.0, -5.0, 32.0, 32.0, 3.0], voxel_size=[1.0, 1.0, 8], out_size_factor=2,
           gaussian_overlap=0.1, min_radius=2)


def radius64(h, w, mo):
    """the CornerNet three-case radius in plain float64"""
    r1 = ((h + w) + math.sqrt((h + w) ** 2 - 4 * w * h * (1 - mo) / (1 + mo))) / 2
    r2 = (2 * (h + w) + math.sqrt(4 * (h + w) ** 2 - 16 * (1 - mo) * w * h)) / 2
    b3 = -2 * mo * (h + w)
    r3 = (b3 + math.sqrt(b3 ** 2 - 16 * mo * (mo - 1) * w * h)) / 2
    return min(r1, r2, r3)


def test_gaussian_radius_matches_the_closed_form():
    for h, w, mo in ((10.0, 10.0, 0.5), (3.2, 1.1, 0.1), (24.0, 6.0, 0.3), (0.7, 0.4, 0.1)):
        got = float(ct.gaussian_radius((torch.tensor(h), torch.tensor(w)), mo))
        assert abs(got - radius64(h, w, mo)) <= 1e-5 * max(1.0, radius64(h, w, mo))


def test_draw_heatmap_gaussian_peak_symmetry_and_clipping():
    hm = torch.zeros(16, 20)
    ct.draw_heatmap_gaussian(hm, (5, 7), 3)
    assert float(hm[7, 5]) == 1.0 and float(hm.max()) == 1.0
    sigma = 7 / 6
    assert abs(float(hm[7, 6]) - math.exp(-1 / (2 * sigma * sigma))) < 1e-7 and float(hm[7, 6]) == float(hm[7, 4]) == float(hm[8, 5])
    assert float(hm[7, 9]) == 0.0 and float(hm[11, 5]) == 0.0 and int((hm > 0).sum()) == 49
    ct.draw_heatmap_gaussian(hm, (0, 15), 3)                     # corner: the window is clipped, nothing wraps around
    assert float(hm[15, 0]) == 1.0 and int((hm[12:, :4] > 0).sum()) == 16 and float(hm[15, 19]) == 0.0
    before = hm.clone()
    ct.draw_heatmap_gaussian(hm, (5, 7), 1)                      # a smaller Gaussian on the same cell: element-wise max
    assert torch.equal(hm, before)


def test_get_targets_order_cells_and_validity_on_a_hand_case():
    # two tasks: labels {0} and {1, 2}.  Sample 0 holds boxes in label order 2, 0, 1, 2 (+ an ignored -1); sample 1 one box of label 1.
    b0 = torch.tensor([[3.0, 5.0, 0, 2, 2, 1, 0.1, 0, 0], [8.2, 9.9, 0, 4, 2, 1, 0.2, 0, 0], [30.0, 1.0, 0, 2, 6, 1, 0.3, 0, 0],
                       [13.0, 13.0, 0, 2, 2, 1, 0.4, 0, 0], [4.0, 4.0, 0, 2, 2, 1, 0.5, 0, 0]])
    l0 = torch.tensor([2, 0, 1, 2, -1])
    b1 = torch.tensor([[-0.5, 2.0, 0, 2, 2, 1, 0.6, 0, 0], [33.0, 2.0, 0, 2, 2, 1, 0.7, 0, 0], [5.0, 5.0, 0, 0.0, 2, 1, 0.8, 0, 0]])
    l1 = torch.tensor([1, 1, 0])
    hm, an, pi = ct.get_targets([b0, b1], [l0, l1], [1, 2], CFG)
    assert [tuple(h.shape) for h in hm] == [(2, 1, 16, 16), (2, 2, 16, 16)]
    # task 0: the one label-0 box of sample 0 (cell (4, 4)); sample 1's label-0 box has zero width: dropped
    assert pi[0].tolist() == [[0, 4, 4]] and [round(v, 1) for v in an[0][:, 6].tolist()] == [0.2]
    # task 1, sample 0: class 1 first (box 2 -> cell (15, 0)), then class 2 in index order (boxes 0 and 3);
    # sample 1: x = -0.5 truncates to cell 0 and stays, x = 33 falls off the map
    assert pi[1].tolist() == [[0, 15, 0], [0, 1, 2], [0, 6, 6], [1, 0, 1]]
    assert [round(v, 1) for v in an[1][:, 6].tolist()] == [0.3, 0.1, 0.4, 0.6]
    assert float(hm[1][0, 0, 0, 15]) == 1.0 and float(hm[1][0, 1, 2, 1]) == 1.0 and float(hm[1][1, 0, 1, 0]) == 1.0
    assert float(hm[0][1].max()) == 0.0 and float(hm[1][0, 1, 0, 15]) == 0.0


def test_heatmap_loss_against_the_closed_form():
    x = torch.tensor([[-1.0, 0.3], [2.0, -12.0]], dtype=torch.float64).reshape(1, 1, 2, 2).requires_grad_(True)
    t = torch.tensor([[1.0, 0.5], [0.0, 1.0]], dtype=torch.float64).reshape(1, 1, 2, 2)
    loss, npos = hf.heatmap_loss(x, t, alpha=2.0, gamma=4.0, loss_weight=3.0)
    loss.backward()
    s = [1 / (1 + math.exp(1.0)), 1 / (1 + math.exp(-0.3)), 1 / (1 + math.exp(-2.0)), 1e-4]     # the last one is clipped from 6e-6
    want = (-math.log(s[0] + 1e-12) * (1 - s[0]) ** 2 - math.log(1 - s[1] + 1e-12) * s[1] ** 2 * 0.5 ** 4
            - math.log(1 - s[2] + 1e-12) * s[2] ** 2 - math.log(s[3] + 1e-12) * (1 - s[3]) ** 2)
    assert npos == 2.0 and abs(loss.item() - 3.0 * want / 2.0) < 1e-12
    assert float(x.grad.reshape(-1)[3]) == 0.0 and float(x.grad.reshape(-1)[0]) < 0.0 < float(x.grad.reshape(-1)[2])
    eps = 1e-6                                                   # numerical derivative at a cell away from the clamp
    xp = x.detach().clone()
    xp.reshape(-1)[1] += eps
    lp, _ = hf.heatmap_loss(xp, t, 2.0, 4.0, 3.0)
    assert abs((lp.item() - loss.item()) / eps - float(x.grad.reshape(-1)[1])) < 1e-5
    assert hf.heatmap_loss(x.detach(), torch.zeros_like(t))[1] == 0.0          # no positive cell: avg_factor max(0, 1) = 1
