#!/usr/bin/env python3
"""Golden vectors for PointBBoxYawCoder FROM THE REAL REFERENCE CLASS (build container only):
    python3 -B tests/golden/make_golden_coder_point.py
Imports /root/reference/mmdet3d_gaussian/core/bbox/coders/point_bbox_yaw_coders.py through the loader of make_golden_coder.py
(only mmdet's BaseBBoxCoder / BBOX_CODERS are stubbed) and records encode, decode with and without correct_yaw in fp32 and fp64,
and autograd gradients wrt preds for a fixed upstream gradient.  The (sin, cos) channels point k quarter turns (k = -3..3) away
from the yaw channel, 0.05 rad at least from a decision boundary.  Writes tests/golden/coder_point.npz (data only)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_coder import load_reference_coders  # noqa: E402


def main():
    torch.set_num_threads(1)
    coder = load_reference_coders().PointBBoxYawCoder()
    assert coder.code_size == 9
    g = torch.Generator().manual_seed(23)
    B, K = 3, 200
    priors = torch.cat([torch.rand(B, K, 2, generator=g) * 100 - 50, torch.rand(B, K, 1, generator=g) * 3 + 0.25], -1)
    preds = torch.randn(B, K, 12, generator=g) * torch.tensor([.5, .5, 1., .4, .4, .4, 1.5, 1., 1., .3, .3, .3])
    k = torch.randint(-3, 4, (B, K), generator=g).float()
    ang = preds[..., 6] + k * (np.pi / 2) + (torch.rand(B, K, generator=g) * 2 - 1) * (np.pi / 4 - 0.05)
    amp = torch.rand(B, K, generator=g) + 0.5
    preds[..., 7], preds[..., 8] = ang.sin() * amp, ang.cos() * amp
    boxes = torch.cat([torch.randn(B, K, 3, generator=g) * 10, torch.rand(B, K, 3, generator=g) * 3 + 0.3,
                       (torch.rand(B, K, 1, generator=g) * 2 - 1) * 3.14159, torch.randn(B, K, 2, generator=g)], -1)
    up = torch.randn(B, K, 10, generator=g)
    out = dict(priors=priors.numpy(), preds=preds.numpy(), boxes=boxes.numpy(), up=up.numpy(), encode32=coder.encode(boxes).numpy(),
               encode64=coder.encode(boxes.double()).numpy())
    for cy, tag in ((False, 'noyaw'), (True, 'yaw')):
        for dtype, t in ((torch.float32, '32'), (torch.float64, '64')):
            p = preds.to(dtype).clone().requires_grad_(True)
            d = coder.decode(priors.to(dtype), p, correct_yaw=cy)
            (d * up.to(dtype)).sum().backward()
            out[f'decode_{tag}{t}'] = d.detach().numpy()
            out[f'gpreds_{tag}{t}'] = p.grad.numpy()
    swapped = (out['decode_yaw32'][..., 3] != out['decode_noyaw32'][..., 3]).mean()
    np.savez_compressed(os.path.join(HERE, 'coder_point.npz'), **out)
    print('coder_point.npz', os.path.getsize(os.path.join(HERE, 'coder_point.npz')), 'bytes; swapped fraction', float(swapped))


if __name__ == '__main__':
    main()
