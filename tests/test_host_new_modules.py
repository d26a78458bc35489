"""Host-side logic of the round-3 modules that needs no GPU: output-structure handling of GraphedStep, loud failures on CPU tensors
(the product has no CPU path), config parsing of the heat-map loss."""
import pytest
import torch

import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd import graphed, heat_loss


def test_graphed_step_flatten_rebuilds_nested_outputs():
    out = dict(a=torch.zeros(1), b=(torch.ones(1), [torch.full((2,), 2.0)]), c=[])
    flat, rebuild = graphed._flatten(out)
    assert len(flat) == 3
    back = rebuild([x + 1 for x in flat])
    assert sorted(back) == ['a', 'b', 'c'] and isinstance(back['b'], tuple) and isinstance(back['b'][1], list)
    assert float(back['a']) == 1.0 and float(back['b'][0]) == 2.0 and back['b'][1][0].tolist() == [3.0, 3.0] and back['c'] == []
    with pytest.raises(TypeError, match='expected tensors'):
        graphed._flatten((torch.zeros(1), 3.0))


def test_new_entry_points_refuse_cpu_tensors():
    with pytest.raises(RuntimeError, match='GPU tensors'):
        amd.GraphedStep(lambda x: x.sum(), (torch.zeros(3),))
    with pytest.raises(RuntimeError, match='no CPU path'):
        amd.center_head_heatmap_loss(dict(type='GaussianFocalLoss'), [torch.zeros(1, 1, 4, 4)], [torch.zeros(1, 1, 4, 4)])
    with pytest.raises(RuntimeError, match='no CPU path'):
        amd.center_head_get_targets([torch.zeros(2, 9)], [torch.zeros(2, dtype=torch.long)], [['a']],
                                    dict(grid_size=[8, 8, 1], point_cloud_range=[0, 0, 0, 8, 8, 1], voxel_size=[1, 1, 1],
                                         out_size_factor=1, gaussian_overlap=0.1, min_radius=2))
    coder = amd.CenterPointBBoxYawCoder([0, 0], 1, [1, 1])
    with pytest.raises(RuntimeError, match='no CPU path'):
        amd.center_head_get_bboxes([dict(heatmap=torch.zeros(1, 1, 4, 4), height=torch.zeros(1, 1, 4, 4), dim=torch.zeros(1, 3, 4, 4),
                                         yaw=torch.zeros(1, 1, 4, 4), dir=torch.zeros(1, 2, 4, 4))], coder,
                                   dict(nms_type='rotate', max_per_img=4, nms_thr=0.2, pre_max_size=10, post_max_size=5), [1])
    with pytest.raises(RuntimeError, match='label sets'):
        amd.center_head_get_targets([], [], [['a']], {})


def test_heatmap_loss_config_parsing():
    assert heat_loss._cfg(dict(type='GaussianFocalLoss')) == (2.0, 4.0, 1.0)
    assert heat_loss._cfg(dict(type='GaussianFocalLoss', alpha=1.5, gamma=3.0, loss_weight=0.5, reduction='mean')) == (1.5, 3.0, 0.5)

    class GaussianFocalLoss:
        alpha, gamma, loss_weight, reduction = 2.0, 4.0, 2.0, 'mean'
    assert heat_loss._cfg(GaussianFocalLoss()) == (2.0, 4.0, 2.0)
    with pytest.raises(RuntimeError, match='GaussianFocalLoss'):
        heat_loss._cfg(dict(type='FocalLoss'))
    with pytest.raises(RuntimeError, match="'mean'"):
        heat_loss._cfg(dict(type='GaussianFocalLoss', reduction='sum'))
