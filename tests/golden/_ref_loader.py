"""Load the REAL reference loss module from /root/reference (this container only).

Used solely by tests/golden/make_golden_*.py to generate the committed golden
vectors.  Nothing here runs on the GPU box (the reference tree does not exist
there) and nothing from the reference is copied: the module is executed from
where it lies, read-only, with two stubbed third-party symbols that are absent
from this image:

  * ``mmdet.models.builder.LOSSES``        -> a no-op registry
  * ``mmdet.models.losses.utils.weighted_loss`` -> restatement of mmdet 2.x
    ``weighted_loss`` / ``weight_reduce_loss`` semantics:
    elementwise ``loss * weight``; ``avg_factor is None``: none|mean|sum;
    else mean -> sum/avg_factor, none -> as is, sum -> ValueError.

The stubbed reduction is the only non-reference arithmetic in the fixtures.
"""
import functools
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get('GD3D_REFERENCE_ROOT', '/root/reference')
REF_LOSS = os.path.join(REF_ROOT, 'mmdet3d_gaussian', 'models', 'losses',
                        'gaussian_distance_loss.py')


def reference_available():
    return os.path.isfile(REF_LOSS)


class _NoopRegistry:
    def register_module(self, *a, **k):
        return lambda cls: cls


def _weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        if reduction == 'none':
            return loss
        return loss.mean() if reduction == 'mean' else loss.sum()
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction == 'none':
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


def _weighted_loss(fn):
    @functools.wraps(fn)
    def wrapper(pred, target, weight=None, reduction='mean', avg_factor=None,
                **kwargs):
        return _weight_reduce_loss(fn(pred, target, **kwargs), weight,
                                   reduction, avg_factor)
    return wrapper


def load_reference_loss():
    """Returns the reference module object (GDLoss, preprocess, ...)."""
    sys.dont_write_bytecode = True  # the reference tree is read-only
    for name in ('mmdet', 'mmdet.models', 'mmdet.models.builder',
                 'mmdet.models.losses', 'mmdet.models.losses.utils'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['mmdet.models.builder'].LOSSES = _NoopRegistry()
    sys.modules['mmdet.models.losses.utils'].weighted_loss = _weighted_loss
    spec = importlib.util.spec_from_file_location('_ref_gd_loss', REF_LOSS)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
