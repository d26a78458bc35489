#!/usr/bin/env python3
"""PVRCNNBboxHead.get_bboxes (pvrcnn_bbox_head.py:352-480) on the GPU, us per call (synchronised): 100 and 512 rois per sample, batch 4,
3 classes, the reference's own thresholds (nms_thr 0.01, score_thr 0.1).
ours  = pvrcnn_head_get_bboxes (decode launch, one batched class NMS for all samples, one read-back)
eager = the reference's statements (oracle/pvrcnn_torch.py's) on device tensors with THIS package's nms_gpu per (sample, class), incl. the
        reference's `roi_batch_id.max().item()`.  Asserts equal detections first."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

import mmdet3d_gaussian_amd as amd  # noqa: E402
from oracle import pvrcnn_torch as ORA  # noqa: E402
from test_gpu_pvrcnn_infer import make  # noqa: E402

dev = torch.device('cuda:0')
# test_cfg.rcnn of configs/kitti/hv_pvrcnn_secfpn_4x4_80e_kitti-3d-3class.py:259-262 (rpn nms_post = 100 rois per sample, :253)
CFG = dict(use_rotate_nms=True, nms_thr=[0.01] * 3, score_thr=[0.1] * 3)


def timeit(fn, it, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6


def eager(rois, cls_score, bbox_pred, class_labels, class_pred):
    bid = rois[..., 0]
    B = int(bid.max().item() + 1)
    rcnn = ORA.decode_rois(rois, bbox_pred)
    out = []
    for b in range(B):
        m = bid == b
        boxes, probs = rcnn[m], class_pred[b]
        for_nms = ORA.xywhr2xyxyr(boxes[:, [0, 1, 3, 4, 6]])
        sel = []
        for k in range(probs.shape[1]):
            keep = probs[:, k] >= CFG['score_thr'][k]
            if keep.int().sum() > 0:
                idx = keep.nonzero(as_tuple=False).view(-1)
                s = amd.nms_gpu(for_nms[keep], probs[keep, k], CFG['nms_thr'][k])
                if s.shape[0]:
                    sel.append(idx[s])
        sel = torch.cat(sel) if sel else []
        out.append((boxes[sel], cls_score[m].view(-1)[sel], class_labels[b][sel]))
    return out


def main():
    for R in (100, 512):
        rois, cls_score, bbox_pred, class_labels, class_pred = make(4, R, 3, seed=R, shuffle=False)
        rois, cls_score, bbox_pred = rois.to(dev), cls_score.to(dev), bbox_pred.to(dev)
        class_labels, class_pred = [l.to(dev) for l in class_labels], [p.to(dev) for p in class_pred]
        ours = lambda: amd.extras.pvrcnn_head_get_bboxes(rois, cls_score, bbox_pred, class_labels, class_pred, CFG)      # noqa: E731
        ref = lambda: eager(rois, cls_score, bbox_pred, class_labels, class_pred)                                  # noqa: E731
        a, e = ours(), ref()
        for (ab, as_, al), (eb, es, el) in zip(a, e):
            assert ab.shape == eb.shape and torch.allclose(ab, eb, rtol=1e-5, atol=1e-4) and torch.equal(al, el)
        print(json.dumps(dict(what=f'PVRCNNBboxHead.get_bboxes, batch 4 x {R} rois, 3 classes', detections=[int(x[0].shape[0]) for x in a],
                              ours_us=round(timeit(ours, 50), 1), reference_ops_on_gpu_us=round(timeit(ref, 10), 1))), flush=True)


if __name__ == '__main__':
    main()
