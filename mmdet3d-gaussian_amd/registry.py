"""Minimal stand-in for the mmcv/mmdet registry surface the reference relies on.

The reference registers ``GDLoss`` with ``@LOSSES.register_module()`` and heads build it with
``build_loss(dict(type='GDLoss', ...))`` (/root/reference/mmdet3d_gaussian/models/losses/
gaussian_distance_loss.py:251; built at models/dense_heads/gd_anchor3d_head.py:60 and
gd_centerpoint_head.py:370).  mmcv/mmdet are not installed in this image, so this module offers
the same two verbs; when mmdet IS importable the class is registered there too (force=True), which
is what makes the config dicts of the reference build this implementation unchanged.
"""
import inspect


class Registry:
    def __init__(self, name):
        self.name = name
        self._module_dict = {}

    def __contains__(self, key):
        return key in self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            key = name or cls.__name__
            if key in self._module_dict and not force:
                raise KeyError(f'{key} is already registered in {self.name}')
            self._module_dict[key] = cls
            return cls
        if module is not None:
            return _register(module)
        return _register

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict) or 'type' not in cfg:
            raise TypeError('cfg must be a dict containing the key "type"')
        args = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        obj_type = args.pop('type')
        if isinstance(obj_type, str):
            cls = self.get(obj_type)
            if cls is None:
                raise KeyError(f'{obj_type} is not in the {self.name} registry')
        elif inspect.isclass(obj_type):
            cls = obj_type
        else:
            raise TypeError(f'type must be a str or a class, got {type(obj_type)}')
        return cls(**args)


LOSSES = Registry('loss')


def build_loss(cfg):
    return LOSSES.build(cfg)


def register_with_mmdet(cls):
    """Also register in mmdet's LOSSES when mmdet is present (drop-in for the reference's configs)."""
    try:
        from mmdet.models.builder import LOSSES as MMDET_LOSSES  # type: ignore
    except Exception:  # noqa: BLE001 - mmdet absent or incompatible: local registry only
        return False
    MMDET_LOSSES.register_module(name=cls.__name__, force=True, module=cls)
    return True
