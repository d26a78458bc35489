"""oracle/voxel_oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of the reference's dynamic point-to-voxel scatter ops:
  scatter_index  — /root/reference/mmdet3d_gaussian/ops/voxel/src/scatter_points_cuda.cu:221-251
  scatter_reduce — feats_reduce_kernel :80-104 (+ the mean division :212-213)
  backward       — :106-179 (sum/mean trace-back; max: the SMALLEST point index equal to the reduced value, atomicMin)
PARITY UNPINNED: the reference implements these ops in CUDA only ("do not support cpu yet", voxelization.h:46,59,78),
so nothing can be executed here to produce golden vectors; this file restates the kernels' documented semantics.
Sum/mean in the reference accumulate with float atomics in arbitrary order; here in ascending point order (fp64 is used
by the tests as the arbiter)."""
import numpy as np


def scatter_index(coors):
    coors = np.asarray(coors)
    if coors.shape[0] == 0:
        return coors.copy(), np.zeros(0, np.int32), np.zeros(0, np.int32)
    clean = coors.copy()
    clean[(coors < 0).any(-1)] = -1
    uniq, inv, cnt = np.unique(clean, axis=0, return_inverse=True, return_counts=True)
    inv = inv.reshape(-1)
    if uniq[0, 0] < 0:
        uniq, cnt, inv = uniq[1:], cnt[1:], inv - 1
    return uniq, inv.astype(np.int32), cnt.astype(np.int32)


def scatter_reduce(feats, pmap, count, reduce_type, dtype=np.float64):
    feats = np.asarray(feats, dtype)
    v, c = len(count), feats.shape[1]
    out = np.full((v, c), -np.inf if reduce_type == 'max' else 0.0, dtype)
    arg = np.full((v, c), -1, np.int64)
    for i in range(feats.shape[0]):
        m = pmap[i]
        if m < 0:
            continue
        if reduce_type == 'max':
            better = feats[i] > out[m]
            out[m][better] = feats[i][better]
            arg[m][better] = i
        else:
            out[m] += feats[i]
    if reduce_type == 'mean':
        out = out / np.asarray(count, dtype)[:, None]
    return out, arg


def scatter_backward(grad_vox, feats, pmap, count, reduce_type, dtype=np.float64):
    grad_vox = np.asarray(grad_vox, dtype)
    feats = np.asarray(feats, dtype)
    g = np.zeros_like(feats)
    if reduce_type == 'max':
        red, _ = scatter_reduce(feats, pmap, count, 'max', dtype)
        first = np.full(red.shape, feats.shape[0], np.int64)
        for i in range(feats.shape[0]):               # atomicMin over points equal to the reduced value
            m = pmap[i]
            if m >= 0:
                eq = feats[i] == red[m]
                first[m][eq] = np.minimum(first[m][eq], i)
        vi, ci = np.nonzero(first < feats.shape[0])
        g[first[vi, ci], ci] = grad_vox[vi, ci]
        return g
    valid = pmap >= 0
    g[valid] = grad_vox[pmap[valid]]
    if reduce_type == 'mean':
        g[valid] /= np.asarray(count, dtype)[pmap[valid]][:, None]
    return g
