"""Anchor grids of the anchor heads: mmdet3d's `Anchor3DRangeGenerator` / `AlignedAnchor3DRangeGenerator` for one feature level
with `reshape_out=False` — what `GDAnchor3DHead.loss` / `get_bboxes` obtain from `self.anchor_generator.grid_anchors(...)`
(/root/reference/mmdet3d_gaussian/models/dense_heads/gd_anchor3d_head.py:224-225; configs/_base_/models/hv_pointpillars_secfpn_kitti.py:40-50,
..._waymo.py:46-57).  Third party, absent: restated from the published text.  Static data, computed once per geometry with a handful of
torch ops on the device the caller names (no kernel of this package is involved); it exists so that `anchor_head_get_targets`,
`gd_anchor_head_loss` and `anchor_head_get_bboxes` can be fed without mmdet3d.
"""
import torch


def anchor3d_range_anchors(feature_size, ranges, sizes, rotations, device, aligned=False, align_corner=False, scale=1.0):
    """feature_size : (H, W) of the level's maps (one z layer), or (D, H, W);
    ranges       : per size class [x_min, y_min, z_min, x_max, y_max, z_max] (one range may serve all sizes);
    sizes        : per size class the three box sizes as the config lists them;  rotations: the yaw values;
    aligned      : AlignedAnchor3DRangeGenerator — centres in the middle of the cells (align_corner=False) instead of
                   `linspace(min, max, n)` over the range.
    Returns (D, H, W, S, R, 7) fp32 on `device`: [x, y, z, size (3), yaw] in the (h, w, size, rotation) order of the head's maps."""
    fs = list(feature_size)
    if len(fs) == 2:
        fs = [1] + fs
    if len(ranges) == 1 and len(sizes) > 1:
        ranges = list(ranges) * len(sizes)
    if len(ranges) != len(sizes):
        raise RuntimeError(f'anchor3d_range_anchors: {len(ranges)} ranges for {len(sizes)} sizes')
    dev = torch.device(device)
    rot = torch.tensor(rotations, dtype=torch.float32, device=dev)
    per = []
    for rng, size in zip(ranges, sizes):
        rng = torch.tensor(rng, dtype=torch.float32, device=dev)
        if aligned:
            axes = []
            for lo, hi, n in ((rng[2], rng[5], fs[0]), (rng[1], rng[4], fs[1]), (rng[0], rng[3], fs[2])):
                c = torch.linspace(float(lo), float(hi), n + 1, device=dev)
                if not align_corner:
                    c = c + (c[1] - c[0]) / 2
                axes.append(c[:n])
            z, y, x = axes
        else:
            z = torch.linspace(float(rng[2]), float(rng[5]), fs[0], device=dev)
            y = torch.linspace(float(rng[1]), float(rng[4]), fs[1], device=dev)
            x = torch.linspace(float(rng[0]), float(rng[3]), fs[2], device=dev)
        gx, gy, gz, gr = torch.meshgrid(x, y, z, rot, indexing='ij')                       # (X, Y, Z, R)
        sz = (torch.tensor(size, dtype=torch.float32, device=dev) * scale).expand(gx.shape + (3,))
        a = torch.cat([gx.unsqueeze(-1), gy.unsqueeze(-1), gz.unsqueeze(-1), sz, gr.unsqueeze(-1)], dim=-1)   # (X, Y, Z, R, 7)
        per.append(a.permute(2, 1, 0, 3, 4).unsqueeze(3))                                # (Z, Y, X, 1, R, 7)
    return torch.cat(per, dim=3).contiguous()
