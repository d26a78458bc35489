"""The bbox coders on either side of the loss.

* ``CenterPointBBoxCoderRev`` — centerpoint_bbox_coders.py:7-112: ``select_best`` (one selection launch, csrc/center_infer.hip),
  ``decode`` (rot = atan2(sin, cos); csrc/coders.hip), ``encode`` (target generation, a handful of torch ops).
* ``CenterPointBBoxYawCoder`` — the call surface of
  /root/reference/mmdet3d_gaussian/core/bbox/coders/centerpoint_bbox_yaw_coders.py:8-56 (``encode``, ``decode(locs, preds,
  correct_yaw=True)``; constructor of centerpoint_bbox_coders.py:7-21) on top of the device kernels of csrc/coders.hip:
  one launch per call instead of ~15 elementwise ops, differentiable wrt ``preds`` (own backward kernel).  GPU tensors
  only; the elementwise torch statement that pins it lives with the test infrastructure, outside this package.
* ``PointBBoxYawCoder`` — point_bbox_yaw_coders.py:7-52 (priors = point + scale; registered by the reference, used by none of its
  shipped heads or configs): ``encode`` shares the yaw coder's kernel, ``decode(priors, preds, correct_yaw=True)`` has its own
  forward and backward kernels.
* ``DeltaXYZWLHRBBoxCoder`` — mmdet3d (third party, absent, unpinned); restated from its published formulas; the
  reference calls it at models/dense_heads/gd_anchor3d_head.py:133-136.

In TRAINING the decode that feeds GDLoss runs from neither: it is fused into the loss kernel's prologue
(head_loss.py -> gd3d_loss_fused_decoded).  These serve inference-time decoding and target encoding.
"""
import ctypes

import torch

from . import _lib


def _coder_struct(c):
    p = _lib.Prologue()
    p.kind = 2
    p.norm_bbox = int(bool(c.norm_bbox))
    p.aux = None
    p.out_size_factor = float(c.out_size_factor)
    p.voxel_size = (ctypes.c_float * 2)(float(c.voxel_size[0]), float(c.voxel_size[1]))
    p.pc_range = (ctypes.c_float * 2)(float(c.pc_range[0]), float(c.pc_range[1]))
    return p


def _rows32(t, cols):
    t = t.reshape(-1, cols)
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


class _CenterDecode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, preds, locs, cs, correct_yaw):
        lib = _lib.load()
        lead, c = preds.shape[:-1], preds.shape[-1]
        p2, l2 = _rows32(preds, c), _rows32(locs, 2)
        n = p2.shape[0]
        if l2.shape[0] != n:
            raise RuntimeError(f'locs {tuple(locs.shape)} and preds {tuple(preds.shape)} describe different box counts')
        co = 7 + max(c - 9, 0)
        out = torch.empty((n, co), dtype=torch.float32, device=preds.device)
        need = ctx.needs_input_grad[0]
        parity = torch.empty(n, dtype=torch.int32, device=preds.device) if (need and correct_yaw) else None
        with torch.cuda.device(preds.device):
            rc = lib.coder_center_decode(cs, l2.data_ptr(), p2.data_ptr(), n, c, int(bool(correct_yaw)), out.data_ptr(),
                                         None if parity is None else parity.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, 'coder_center_decode')
        ctx.cs, ctx.meta = cs, (n, c, lead, preds.dtype)
        if need:
            ctx.save_for_backward(out, parity)
        res = out.reshape(lead + (co,))
        return res if preds.dtype == torch.float32 else res.to(preds.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        out, parity = ctx.saved_tensors
        n, c, lead, dtype = ctx.meta
        go = _rows32(grad_out, out.shape[1])
        gp = torch.empty((n, c), dtype=torch.float32, device=go.device)
        with torch.cuda.device(go.device):
            rc = lib.coder_center_decode_backward(ctx.cs, go.data_ptr(), out.data_ptr(),
                                                  None if parity is None else parity.data_ptr(), n, c, gp.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, 'coder_center_decode_backward')
        gp = gp.reshape(lead + (c,))
        return (gp if dtype == torch.float32 else gp.to(dtype)), None, None, None


class CenterPointBBoxCoderRev:
    """centerpoint_bbox_coders.py:7-112 (`CenterPointBBoxCoderRev`): constructor, `select_best`, `encode`, `decode`."""
    infer_kind = 'rev'

    def __init__(self, pc_range, out_size_factor, voxel_size, code_size=9, norm_bbox=True):
        self.pc_range = pc_range
        self.out_size_factor = out_size_factor
        self.voxel_size = voxel_size
        self.code_size = code_size
        self.norm_bbox = norm_bbox

    def select_best(self, scores, preds, topk):
        """scores (B,C,H,W) after the sigmoid, preds (B,N,H,W) -> scores (B,K), classes, locs (x, y), preds (B,K,N)."""
        from .center_infer import select_best
        return select_best(scores, preds, topk)

    def encode(self, target_boxes):
        """(..., 7+k) metric boxes -> (..., 8+k) regression targets: cell-relative xy, z, (log) dims, sin, cos, others.
        Target generation (get_targets), off the hot path: plain tensor ops on whatever device the boxes live on."""
        cell = [(target_boxes[..., k] - self.pc_range[k]) / self.voxel_size[k] / self.out_size_factor for k in (0, 1)]
        frac = [c - c.floor() for c in cell]
        dims = target_boxes[..., 3:6].log() if self.norm_bbox else target_boxes[..., 3:6]
        yaw = target_boxes[..., 6]
        head = torch.stack((frac[0], frac[1], target_boxes[..., 2]), dim=-1)
        return torch.cat((head, dims, torch.stack((yaw.sin(), yaw.cos()), dim=-1), target_boxes[..., 7:]), dim=-1)

    def decode(self, locs, preds):
        """locs (..., 2) cells, preds (..., N) raw [dx, dy, z, dims x3, sin, cos, others] -> (..., N-1) metric boxes."""
        if not preds.is_cuda:
            raise RuntimeError('CenterPointBBoxCoderRev: the MI355X implementation has no CPU path')
        lib = _lib.load()
        lead, c = preds.shape[:-1], preds.shape[-1]
        p2, l2 = _rows32(preds.detach(), c), _rows32(locs.to(preds.device), 2)
        out = torch.empty((p2.shape[0], c - 1), dtype=torch.float32, device=p2.device)
        with torch.cuda.device(p2.device):
            rc = lib.coder_center_decode(_coder_struct(self), l2.data_ptr(), p2.data_ptr(), p2.shape[0], c, 2, out.data_ptr(), None,
                                         torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, 'coder_center_decode')
        out = out.reshape(lead + (c - 1,))
        return out if preds.dtype == torch.float32 else out.to(preds.dtype)


class CenterPointBBoxYawCoder(CenterPointBBoxCoderRev):
    """centerpoint_bbox_yaw_coders.py:8-56; constructor and `select_best` of the base class."""
    infer_kind = 'yaw'

    def encode(self, target_boxes):
        """(..., 7+k) boxes [x,y,z,w,l,h,yaw, others] -> (..., 9+k): the first 7 as they are, sin yaw, cos yaw, others."""
        if not target_boxes.is_cuda:
            raise RuntimeError('CenterPointBBoxYawCoder: the MI355X implementation has no CPU path')
        lib = _lib.load()
        lead, c = target_boxes.shape[:-1], target_boxes.shape[-1]
        b2 = _rows32(target_boxes.detach(), c)
        out = torch.empty((b2.shape[0], c + 2), dtype=torch.float32, device=b2.device)
        with torch.cuda.device(b2.device):
            rc = lib.coder_center_encode(b2.data_ptr(), b2.shape[0], c, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, 'coder_center_encode')
        out = out.reshape(lead + (c + 2,))
        return out if target_boxes.dtype == torch.float32 else out.to(target_boxes.dtype)

    def decode(self, locs, preds, correct_yaw=True):
        """locs (..., 2) cell coordinates, preds (..., N) raw head outputs -> (..., N-2) metric boxes
        [x, y, z, dims, yaw, others]; with correct_yaw the yaw is snapped to the quarter turn the (sin, cos) channels
        point at and w / l are swapped on odd turns (ref :40-50)."""
        if not preds.is_cuda:
            raise RuntimeError('CenterPointBBoxYawCoder: the MI355X implementation has no CPU path')
        return _CenterDecode.apply(preds, locs.to(preds.device), _coder_struct(self), bool(correct_yaw))


class _PointDecode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, preds, priors, correct_yaw):
        lib = _lib.load()
        lead, c = preds.shape[:-1], preds.shape[-1]
        p2, q2 = _rows32(preds, c), _rows32(priors, 3)
        n = p2.shape[0]
        if q2.shape[0] != n:
            raise RuntimeError(f'priors {tuple(priors.shape)} and preds {tuple(preds.shape)} describe different box counts')
        co = 7 + max(c - 9, 0)
        out = torch.empty((n, co), dtype=torch.float32, device=preds.device)
        need = ctx.needs_input_grad[0]
        parity = torch.empty(n, dtype=torch.int32, device=preds.device) if (need and correct_yaw) else None
        with torch.cuda.device(preds.device):
            rc = lib.coder_point_decode(q2.data_ptr(), p2.data_ptr(), n, c, int(bool(correct_yaw)), out.data_ptr(),
                                        None if parity is None else parity.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, 'coder_point_decode')
        ctx.meta = (n, c, lead, preds.dtype)
        if need:
            ctx.save_for_backward(out, parity, q2)
        res = out.reshape(lead + (co,))
        return res if preds.dtype == torch.float32 else res.to(preds.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        out, parity, q2 = ctx.saved_tensors
        n, c, lead, dtype = ctx.meta
        go = _rows32(grad_out, out.shape[1])
        gp = torch.empty((n, c), dtype=torch.float32, device=go.device)
        with torch.cuda.device(go.device):
            rc = lib.coder_point_decode_backward(q2.data_ptr(), go.data_ptr(), out.data_ptr(), None if parity is None else parity.data_ptr(),
                                                 n, c, gp.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, 'coder_point_decode_backward')
        gp = gp.reshape(lead + (c,))
        return (gp if dtype == torch.float32 else gp.to(dtype)), None, None


class PointBBoxYawCoder:
    """point_bbox_yaw_coders.py:7-52 (`PointBBoxYawCoder`): `code_size`, `encode`, `decode(priors, preds, correct_yaw=True)`."""

    @property
    def code_size(self):
        return 9

    encode = CenterPointBBoxYawCoder.encode          # the same statements (:12-16 == centerpoint_bbox_yaw_coders.py:11-16)

    def decode(self, priors, preds, correct_yaw=True):
        """priors (..., 3) [x, y, scale] of the points, preds (..., N) raw outputs [dx, dy, z, log dims, yaw, sin, cos, others]
        -> (..., N-2) boxes [dx scale + x, dy scale + y, z, exp(dims) (w, l times scale), yaw, others]; correct_yaw as in the
        CenterPoint yaw coder (:38-48).  Differentiable wrt preds (the priors are data: no gradient is produced for them)."""
        if not preds.is_cuda:
            raise RuntimeError('PointBBoxYawCoder: the MI355X implementation has no CPU path')
        if priors.requires_grad:
            raise RuntimeError('PointBBoxYawCoder.decode: priors that require grad are not supported (gradient flows to preds only)')
        return _PointDecode.apply(preds, priors.to(preds.device), bool(correct_yaw))


class DeltaXYZWLHRBBoxCoder:
    """mmdet3d anchor-delta coder for (x, y, z, w, l, h, r) boxes (code_size 7)."""

    def __init__(self, code_size=7):
        self.box_dim = code_size

    @staticmethod
    def encode(src_boxes, dst_boxes):
        xa, ya, za, wa, la, ha, ra = src_boxes[..., :7].unbind(-1)
        xg, yg, zg, wg, lg, hg, rg = dst_boxes[..., :7].unbind(-1)
        za = za + ha / 2
        zg = zg + hg / 2
        diagonal = torch.sqrt(la ** 2 + wa ** 2)
        return torch.stack(((xg - xa) / diagonal, (yg - ya) / diagonal, (zg - za) / ha, torch.log(wg / wa),
                            torch.log(lg / la), torch.log(hg / ha), rg - ra), dim=-1)

    @staticmethod
    def decode(anchors, deltas):
        xa, ya, za, wa, la, ha, ra = anchors[..., :7].unbind(-1)
        xt, yt, zt, wt, lt, ht, rt = deltas[..., :7].unbind(-1)
        za = za + ha / 2
        diagonal = torch.sqrt(la ** 2 + wa ** 2)
        hg = torch.exp(ht) * ha
        return torch.stack((xt * diagonal + xa, yt * diagonal + ya, zt * ha + za - hg / 2, torch.exp(wt) * wa,
                            torch.exp(lt) * la, hg, rt + ra), dim=-1)
