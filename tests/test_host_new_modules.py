"""Host-side logic of the round-3 modules that needs no GPU: output-structure handling of GraphedStep, loud failures on CPU tensors
(the product has no CPU path), config parsing of the heat-map loss."""
import pytest
import torch

import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd import anchor_cls, graphed, heat_loss


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
        amd.extras.center_head_heatmap_loss(dict(type='GaussianFocalLoss'), [torch.zeros(1, 1, 4, 4)], [torch.zeros(1, 1, 4, 4)])
    with pytest.raises(RuntimeError, match='no CPU path'):
        amd.extras.center_head_get_targets([torch.zeros(2, 9)], [torch.zeros(2, dtype=torch.long)], [['a']],
                                    dict(grid_size=[8, 8, 1], point_cloud_range=[0, 0, 0, 8, 8, 1], voxel_size=[1, 1, 1],
                                         out_size_factor=1, gaussian_overlap=0.1, min_radius=2))
    coder = amd.CenterPointBBoxYawCoder([0, 0], 1, [1, 1])
    with pytest.raises(RuntimeError, match='no CPU path'):
        amd.extras.center_head_get_bboxes([dict(heatmap=torch.zeros(1, 1, 4, 4), height=torch.zeros(1, 1, 4, 4), dim=torch.zeros(1, 3, 4, 4),
                                         yaw=torch.zeros(1, 1, 4, 4), dir=torch.zeros(1, 2, 4, 4))], coder,
                                   dict(nms_type='rotate', max_per_img=4, nms_thr=0.2, pre_max_size=10, post_max_size=5), [1])
    with pytest.raises(RuntimeError, match='label sets'):
        amd.extras.center_head_get_targets([], [], [['a']], {})


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


def test_anchor_cls_config_parsing_and_cpu_refusal():
    assert anchor_cls._focal_cfg(dict(type='FocalLoss', use_sigmoid=True)) == (2.0, 0.25, 1.0)
    assert anchor_cls._focal_cfg(dict(type='FocalLoss', use_sigmoid=True, gamma=1.5, alpha=0.5, loss_weight=3.0)) == (1.5, 0.5, 3.0)
    assert anchor_cls._ce_cfg(dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.2)) == 0.2

    class FocalLoss:
        use_sigmoid, gamma, alpha, loss_weight, reduction = True, 2.0, 0.25, 1.0, 'mean'
    assert anchor_cls._focal_cfg(FocalLoss()) == (2.0, 0.25, 1.0)
    with pytest.raises(RuntimeError, match='FocalLoss'):
        anchor_cls._focal_cfg(dict(type='FocalLoss', use_sigmoid=False))
    with pytest.raises(RuntimeError, match='CrossEntropyLoss'):
        anchor_cls._ce_cfg(dict(type='CrossEntropyLoss', use_sigmoid=True))
    with pytest.raises(RuntimeError, match="'mean'"):
        anchor_cls._focal_cfg(dict(type='FocalLoss', reduction='sum'))
    with pytest.raises(RuntimeError, match='activated'):
        anchor_cls._focal_cfg(dict(type='FocalLoss', activated=True))
    with pytest.raises(RuntimeError, match='no CPU path'):
        amd.extras.anchor_head_cls_dir_loss(dict(type='FocalLoss'), dict(type='CrossEntropyLoss'), torch.zeros(1, 2, 3, 3), torch.zeros(1, 4, 3, 3),
                                     torch.zeros(1, 18, dtype=torch.long), torch.ones(1, 18), torch.zeros(1, 18, dtype=torch.long), torch.ones(1, 18), 1, 1.0)


def test_anchor_cls_oracle_known_answers():
    """oracle/anchor_cls_torch.py against hand-computed values of mmdet's formulas: at logit 0 every p = 1/2, so a positive of
    class 0 among C = 2 costs alpha/4 ln 2 + (1 - alpha)/4 ln 2 and a background anchor 2 (1 - alpha)/4 ln 2; the direction term
    of equal logits is ln 2 per positive."""
    import math
    from oracle import anchor_cls_torch as ORA
    cls = torch.zeros(1, 2, 1, 2, dtype=torch.float64)            # A = 1, C = 2, two cells
    dirs = torch.zeros(1, 2, 1, 2, dtype=torch.float64)
    labels = torch.tensor([[0, 2]])
    lc, ld = ORA.cls_dir_losses(cls, dirs, labels, torch.ones(1, 2, dtype=torch.float64), torch.tensor([[1, 0]]), torch.tensor([[1.0, 0.0]], dtype=torch.float64),
                                2, 4.0, cls_weight=1.0, dir_weight=0.2)
    ln2 = math.log(2.0)
    assert abs(lc.item() - (0.25 / 4 + 0.75 / 4 + 2 * 0.75 / 4) * ln2 / 4.0) < 1e-12
    assert abs(ld.item() - 0.2 * ln2 / 4.0) < 1e-12
