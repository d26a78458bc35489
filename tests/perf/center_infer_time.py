#!/usr/bin/env python3
"""CenterPoint inference slice (head maps -> detections), nuScenes geometry: 6 tasks (classes 1,2,2,1,2,2), 128 x 128 maps,
max_per_img 500, rotate NMS (thr 0.2, pre 1000, post 83) — BASELINE configs[4]'s NMS with the steps the reference runs around it
(gd_centerpoint_head.py:218-361).  Prints JSON lines, us per call (host + device, synchronised):
  ours          : center_head_get_bboxes (one selection launch, batched NMS, one merge launch, one read-back)
  eager         : the reference's op sequence on the GPU (oracle/center_infer_torch.py's statement with torch ops on device tensors)
                  with THIS package's nms_gpu per sample — what a user gets who swaps only the NMS op
  eager_batched : the same with nms_gpu_multi over all (task, sample) groups
Asserts that ours and the eager flow return the same number of detections per sample and equal boxes.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel table."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

import mmdet3d_gaussian_amd as amd  # noqa: E402
from oracle import center_infer_torch as cit  # noqa: E402
from oracle import coder_torch  # noqa: E402
from test_gpu_center_infer import NUS, NUS_TEST, make_tasks, midgap  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, it, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6


def eager(tasks, cfg, test_cfg, classes, batched):
    """get_bboxes as the reference writes it, on device tensors; NMS through this package"""
    K, thr, rng = test_cfg['max_per_img'], test_cfg['score_threshold'], test_cfg['post_center_limit_range']
    rets = []
    groups = []
    for pd in tasks:
        B = pd['heatmap'].shape[0]
        heat = pd['heatmap'].sigmoid()
        scores, clses, locs, preds = cit.select_best(heat, cit.reconstruct(pd, 'yaw'), K)
        boxes = coder_torch.center_decode(locs, preds, cfg['pc_range'], cfg['out_size_factor'], cfg['voxel_size'], True, True)
        mask = cit.center_mask(scores, boxes, thr, rng)
        per = [(boxes[i][mask[i]], scores[i][mask[i]], clses[i][mask[i]]) for i in range(B)]
        groups.append(per)
    if batched:
        flat = [g for per in groups for g in per]
        keeps = amd.nms_gpu_multi([cit.bev_xyxyr(b) for b, _, _ in flat], [s for _, s, _ in flat], test_cfg['nms_thr'],
                                  pre_max_size=test_cfg['pre_max_size'], post_max_size=test_cfg['post_max_size'])
    it = 0
    for per in groups:
        ret_task = []
        for bx, sc, lb in per:
            if batched:
                keep = keeps[it]
                it += 1
            elif sc.numel() > 0:
                keep = amd.nms_gpu(cit.bev_xyxyr(bx), sc, thresh=test_cfg['nms_thr'], pre_max_size=test_cfg['pre_max_size'],
                                   post_max_size=test_cfg['post_max_size'])
            else:
                keep = []
            ret_task.append(dict(bboxes=bx[keep], scores=sc[keep], labels=lb[keep]))
        rets.append(ret_task)
    out = []
    for i in range(len(rets[0])):
        bboxes = torch.cat([r[i]['bboxes'] for r in rets])
        bboxes[:, 2] = bboxes[:, 2] - bboxes[:, 5] * 0.5
        flag, labels = 0, []
        for j, nc in enumerate(classes):
            labels.append((rets[j][i]['labels'] + flag).int())
            flag += nc
        out.append([bboxes, torch.cat([r[i]['scores'] for r in rets]), torch.cat(labels)])
    return out


def main():
    classes = [1, 2, 2, 1, 2, 2]
    coder = amd.CenterPointBBoxYawCoder(**NUS)
    for B in (1, 4):
        g = torch.Generator().manual_seed(3)
        cpu = make_tasks(g, B, 128, 128, classes, 'yaw')
        tasks = [{k: v.to(dev) for k, v in pd.items()} for pd in cpu]
        for name, thr in (('config threshold 0.1 (every candidate passes)', NUS_TEST['score_threshold']),
                          ('threshold at the 40 % quantile of the candidates', midgap(cpu, 500, 0.4))):
            cfg = dict(NUS_TEST, score_threshold=thr)
            a = amd.extras.center_head_get_bboxes(tasks, coder, cfg, classes)
            b = eager(tasks, NUS, cfg, classes, False)
            c = eager(tasks, NUS, cfg, classes, True)
            for x, y, z in zip(a, b, c):
                assert x[0].shape == y[0].shape == z[0].shape, (x[0].shape, y[0].shape, z[0].shape)
                torch.testing.assert_close(x[0], y[0], rtol=1e-5, atol=1e-5)
                assert torch.equal(x[2], y[2]) and torch.equal(y[0], z[0])
            us_a = timeit(lambda: amd.extras.center_head_get_bboxes(tasks, coder, cfg, classes), 50)
            us_b = timeit(lambda: eager(tasks, NUS, cfg, classes, False), 10, warm=2)
            us_c = timeit(lambda: eager(tasks, NUS, cfg, classes, True), 10, warm=2)
            # nothing read back (padded=True): the call pipelines, and the slice replays as a hipGraph
            us_p = timeit(lambda: amd.extras.center_head_get_bboxes(tasks, coder, cfg, classes, padded=True), 100)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                amd.extras.center_head_get_bboxes(tasks, coder, cfg, classes, padded=True)
            us_g = timeit(graph.replay, 200)
            print(json.dumps(dict(what=f'get_bboxes, 6 tasks x batch {B}, 128x128, K=500; {name}', detections=[int(x[0].shape[0]) for x in a],
                                  ours_us=round(us_a, 1), ours_padded_no_readback_us=round(us_p, 1), ours_padded_as_a_hipgraph_us=round(us_g, 1),
                                  eager_us=round(us_b, 1), eager_batched_nms_us=round(us_c, 1))), flush=True)
    # the selection alone against two torch.topk + gathers (select_best), one task
    for shape, K in (((4, 2, 128, 128), 500), ((1, 3, 468, 468), 4096)):
        g = torch.Generator().manual_seed(4)
        heat = torch.rand(shape, generator=g).to(dev)
        pred = torch.randn(shape[0], 11, shape[2], shape[3], generator=g).to(dev)
        s1 = amd.extras.select_best(heat, pred, K)
        s2 = cit.select_best(heat, pred, K)
        assert torch.equal(s1[0], s2[0])
        us_a = timeit(lambda: amd.extras.select_best(heat, pred, K), 50)
        us_b = timeit(lambda: cit.select_best(heat, pred, K), 20)
        print(json.dumps(dict(what=f'select_best {shape}, K={K}', ours_us=round(us_a, 1), torch_ops_us=round(us_b, 1))), flush=True)


if __name__ == '__main__':
    main()
