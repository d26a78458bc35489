"""Dynamic point-to-voxel scatter ops — host mirror of /root/reference/mmdet3d_gaussian/ops/voxel/scatter.py
(`scatter_index` :10-26, `scatter_reduce` :29-72, `Scatter` :75-144) on top of the HIP kernels of
csrc/voxel_scatter.hip.

`scatter_index` is ATen-level plumbing in the reference as well (masked_fill + unique_dim,
src/scatter_points_cuda.cu:221-251) and is restated with the same torch ops.  `scatter_reduce` keeps the reference's
signature; the grouping of points by voxel (a stable argsort of the map) that the atomic-free kernels need is computed
once per `Scatter` and reused by every reduce call on it.
"""
import torch
from torch import nn
from torch.autograd import Function

from . import _lib

REDUCE = {'sum': 0, 'mean': 1, 'max': 2}  # reduce_t, src/voxelization.h:4
_raw_stream = torch._C._cuda_getCurrentRawStream
_get_device = torch._C._cuda_getDevice
_set_device = torch._C._cuda_setDevice


class _on_device:
    """Minimal device guard (raw accessors: no Python-level bookkeeping on the per-call path)."""

    def __init__(self, dev):
        self.idx = dev.index

    def __enter__(self):
        self.prev = _get_device()
        if self.prev != self.idx:
            _set_device(self.idx)
        return _raw_stream(self.idx)

    def __exit__(self, *exc):
        if self.prev != self.idx:
            _set_device(self.prev)
        return False


def scatter_index(coors):
    """coors (N, ndim) int -> (voxel_coors (M, ndim), point2voxel_map (N,) int32 with -1 for dropped points,
    voxel_points_count (M,) int32).  Voxels are the sorted unique rows; a point with ANY negative coordinate is
    dropped (scatter_points_cuda.cu:236-246)."""
    if coors.size(0) == 0:
        return (coors.clone().detach(), coors.new_empty((0,), dtype=torch.int32),
                coors.new_empty((0,), dtype=torch.int32))
    clean = coors.masked_fill(coors.lt(0).any(-1, True), -1)
    out_coors, coors_map, reduce_count = torch.unique(clean, dim=0, sorted=True, return_inverse=True,
                                                      return_counts=True)
    if bool(out_coors[0, 0].lt(0)):
        out_coors = out_coors[1:]
        reduce_count = reduce_count[1:]
        coors_map = coors_map - 1
    return out_coors, coors_map.to(torch.int32), reduce_count.to(torch.int32)


def group_points(point2voxel_map, voxel_points_count):
    """(order, seg): point ids grouped by voxel in ascending id, and the (M+1) segment bounds into `order`."""
    order = torch.sort(point2voxel_map, stable=True)[1].to(torch.int32)
    n_invalid = int(point2voxel_map.numel()) - int(voxel_points_count.sum())   # leading -1 entries
    seg = torch.zeros(voxel_points_count.numel() + 1, dtype=torch.int32, device=point2voxel_map.device)
    seg[1:] = torch.cumsum(voxel_points_count, 0)
    seg += n_invalid
    return order.contiguous(), seg.contiguous()


class _ScatterReduce(Function):
    @staticmethod
    def forward(ctx, feats, point2voxel_map, voxel_points_count, reduce_type='max', grouping=None):
        if not feats.is_cuda:
            raise RuntimeError('scatter_reduce: the MI355X implementation has no CPU path '
                               '(neither has the reference: voxelization.h:46)')
        lib = _lib.load()
        if feats.size(0) == 0:
            return feats.clone().detach()
        feats32 = feats.contiguous() if feats.dtype == torch.float32 else feats.float().contiguous()
        n, c = feats32.shape
        if point2voxel_map.numel() != n or point2voxel_map.dtype != torch.int32 or \
                voxel_points_count.dtype != torch.int32:
            raise RuntimeError('scatter_reduce: point2voxel_map must be an int32 tensor with one entry per point and '
                               'voxel_points_count int32 (as scatter_index returns them)')
        v = voxel_points_count.numel()
        order, seg = grouping if grouping is not None else group_points(point2voxel_map, voxel_points_count)
        out = torch.empty((v, c), dtype=torch.float32, device=feats.device)
        red = REDUCE[reduce_type]
        argmax = torch.empty((v, c), dtype=torch.int32, device=feats.device) if red == 2 else None
        with _on_device(feats.device) as stream:
            rc = lib.vox_scatter_reduce(feats32.data_ptr(), order.data_ptr(), seg.data_ptr(), n, c, v, red,
                                        out.data_ptr(), None if argmax is None else argmax.data_ptr(), stream)
        _lib.check(rc, 'vox_scatter_reduce')
        ctx.red, ctx.shape, ctx.in_dtype = red, (n, c, v), feats.dtype
        ctx.save_for_backward(point2voxel_map.contiguous(), voxel_points_count.contiguous(), argmax, order, seg)
        ctx.mark_non_differentiable(point2voxel_map, voxel_points_count)
        return out if feats.dtype == torch.float32 else out.to(feats.dtype)

    @staticmethod
    def backward(ctx, grad_voxel_feats):
        lib = _lib.load()
        pmap, count, argmax, order, seg = ctx.saved_tensors
        n, c, v = ctx.shape
        g = grad_voxel_feats.contiguous().float()
        grad_feats = torch.empty((n, c), dtype=torch.float32, device=g.device)
        am = None if argmax is None else argmax.data_ptr()
        with _on_device(g.device) as stream:
            if c % 4 == 0 and c <= 256 and g.data_ptr() % 16 == 0:
                # voxel order: every gradient row is read once and streamed to its points (half the HBM traffic)
                rc = lib.vox_scatter_backward_grouped(g.data_ptr(), order.data_ptr(), seg.data_ptr(), am, n, c, v, ctx.red,
                                                      grad_feats.data_ptr(), stream)
            else:
                rc = lib.vox_scatter_backward(g.data_ptr(), pmap.data_ptr(), count.data_ptr(), am, n, c, v, ctx.red,
                                              grad_feats.data_ptr(), stream)
        _lib.check(rc, 'vox_scatter_backward')
        if ctx.in_dtype != torch.float32:
            grad_feats = grad_feats.to(ctx.in_dtype)
        return grad_feats, None, None, None, None


def scatter_reduce(feats, point2voxel_map, voxel_points_count, reduce_type='max', grouping=None):
    """feats (N,C) -> (M,C): max | mean | sum of the rows that share a voxel (reference signature + optional
    precomputed `grouping` = group_points(...))."""
    assert reduce_type in REDUCE, f'do not support reduce type {reduce_type}'
    return _ScatterReduce.apply(feats, point2voxel_map, voxel_points_count, reduce_type, grouping)


class Scatter(object):
    """Reference `Scatter` (scatter.py:75-144): voxelise once, reduce / map back many times."""

    def __init__(self, coors):
        self._pts_coors = coors
        self._grouping = None
        if coors.numel() == 0:
            self._batch_size = None if coors.size(-1) == 3 else 1
            self.voxel_coors = coors.clone().detach()
            self.pts_voxel_maps = coors.new_empty((0,), dtype=torch.int32)
            self.voxel_pts_counts = coors.new_empty((0,), dtype=torch.int32)
            return
        if coors.size(-1) == 3:
            self._batch_size = None
            voxel_coors, pts_voxel_maps, voxel_pts_counts = scatter_index(coors.contiguous())
        else:
            batch_size = coors[:, 0].max().item() + 1
            self._batch_size = batch_size
            previous_voxels = 0
            pts_voxel_maps = coors.new_full((coors.size(0),), -1, dtype=torch.int32)
            voxel_pts_counts, voxel_coors = [], []
            for i in range(batch_size):
                inds = torch.where(coors[:, 0] == i)
                voxel_coor, pts_voxel_map, voxel_pts_count = scatter_index(coors[inds][:, 1:].contiguous())
                pts_voxel_map[pts_voxel_map.ge(0)] += previous_voxels
                pts_voxel_maps[inds] = pts_voxel_map
                previous_voxels += voxel_coor.size(0)
                voxel_pts_counts.append(voxel_pts_count)
                voxel_coors.append(nn.functional.pad(voxel_coor, (1, 0), mode='constant', value=i))
            voxel_coors = torch.cat(voxel_coors, dim=0)
            voxel_pts_counts = torch.cat(voxel_pts_counts, dim=0)
        self.voxel_coors = voxel_coors
        self.pts_voxel_maps = pts_voxel_maps
        self.voxel_pts_counts = voxel_pts_counts

    @property
    def pts_coors(self):
        return self._pts_coors

    @property
    def batch_size(self):
        return self._batch_size

    def mapback(self, voxel_feats, default_feat=0):
        invalid_mask = self.pts_voxel_maps.lt(0)
        point_feats = voxel_feats[self.pts_voxel_maps.clamp(min=0).long()]
        point_feats[invalid_mask] = default_feat
        return point_feats

    def reduce(self, pts_feats, reduce_op):
        assert reduce_op in ['max', 'mean', 'sum'], \
            f'For the arg "reduce", only "max", "mean" and "sum" are supported but got {reduce_op}'
        if self._grouping is None and self.pts_voxel_maps.numel() > 0:
            self._grouping = group_points(self.pts_voxel_maps, self.voxel_pts_counts)
        voxel_feats = scatter_reduce(pts_feats.contiguous(), self.pts_voxel_maps, self.voxel_pts_counts, reduce_op,
                                     self._grouping)
        return voxel_feats, self.voxel_coors

    def reduce_mapback(self, pts_feats, reduce_op, default_feat=0):
        voxel_feats, _ = self.reduce(pts_feats, reduce_op)
        return self.mapback(voxel_feats, default_feat)
