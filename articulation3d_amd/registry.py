"""Name -> constructor registries with the detectron2 `Registry` surface the reference uses
(`@X_REGISTRY.register()`, `X_REGISTRY.get(name)`; e.g. pkg/modeling/roi_heads/plane_head.py:12,15,135-137)."""
from __future__ import annotations


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        assert name not in self._obj_map, f"An object named '{name}' was already registered in '{self._name}' registry!"
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:
            def deco(func_or_class):
                self._do_register(func_or_class.__name__, func_or_class)
                return func_or_class

            return deco
        self._do_register(obj.__name__, obj)
        return obj

    def get(self, name):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return ret

    def __contains__(self, name):
        return name in self._obj_map

    def __iter__(self):
        return iter(self._obj_map.items())


META_ARCH_REGISTRY = Registry("META_ARCH")
BACKBONE_REGISTRY = Registry("BACKBONE")
PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")
RPN_HEAD_REGISTRY = Registry("RPN_HEAD")
ANCHOR_GENERATOR_REGISTRY = Registry("ANCHOR_GENERATOR")
ROI_HEADS_REGISTRY = Registry("ROI_HEADS")
ROI_BOX_HEAD_REGISTRY = Registry("ROI_BOX_HEAD")
ROI_MASK_HEAD_REGISTRY = Registry("ROI_MASK_HEAD")
ROI_PLANE_HEAD_REGISTRY = Registry("ROI_PLANE_HEAD")
ROI_AXIS_HEAD_REGISTRY = Registry("ROI_AXIS_HEAD")
DEPTH_HEAD_REGISTRY = Registry("DEPTH_HEAD")
