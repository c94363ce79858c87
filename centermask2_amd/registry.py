"""Name -> object registry with the detectron2 `Registry` surface.

The reference registers its plugins into detectron2 registries
(`BACKBONE_REGISTRY` vovnet.py:492,527; `PROPOSAL_GENERATOR_REGISTRY` fcos.py:28;
`ROI_HEADS_REGISTRY` center_heads.py:295; local `Registry("ROI_MASK_HEAD")`
mask_head.py:17, `Registry("ROI_MASKIOU_HEAD")` maskiou_head.py:10) and its scripts
poke `_obj_map` directly (tester.py:157).  detectron2 is a third-party package whose
source is not in the reference tree; this is a from-scratch class with the same
call surface (`register` as decorator or call, `get`, `_obj_map`, `in`, iteration).
"""
from typing import Any, Dict, Iterator, Optional, Tuple


class Registry:
    def __init__(self, name: str) -> None:
        self._name = name
        self._obj_map: Dict[str, Any] = {}

    def _do_register(self, name: str, obj: Any) -> None:
        if name in self._obj_map:
            raise AssertionError(
                "An object named '{}' was already registered in '{}' registry!".format(name, self._name)
            )
        self._obj_map[name] = obj

    def register(self, obj: Any = None) -> Any:
        if obj is None:
            def deco(func_or_class: Any) -> Any:
                self._do_register(func_or_class.__name__, func_or_class)
                return func_or_class
            return deco
        self._do_register(obj.__name__, obj)
        return obj

    def get(self, name: str) -> Any:
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError("No object named '{}' found in '{}' registry!".format(name, self._name))
        return ret

    def __contains__(self, name: str) -> bool:
        return name in self._obj_map

    def __iter__(self) -> Iterator[Tuple[str, Any]]:
        return iter(self._obj_map.items())

    def __repr__(self) -> str:
        return "Registry of {}: {}".format(self._name, sorted(self._obj_map))


def _d2_registry(path: str, attr: str) -> Optional[Any]:
    """Return detectron2's own registry when a real install is importable, so the
    plugins land where a d2 `build_model` looks for them (one code path, chosen at import)."""
    try:
        import importlib
        return getattr(importlib.import_module(path), attr)
    except Exception:
        return None


BACKBONE_REGISTRY = _d2_registry("detectron2.modeling.backbone.build", "BACKBONE_REGISTRY") or Registry("BACKBONE")
PROPOSAL_GENERATOR_REGISTRY = (
    _d2_registry("detectron2.modeling.proposal_generator.build", "PROPOSAL_GENERATOR_REGISTRY")
    or Registry("PROPOSAL_GENERATOR")
)
ROI_HEADS_REGISTRY = _d2_registry("detectron2.modeling.roi_heads.roi_heads", "ROI_HEADS_REGISTRY") or Registry("ROI_HEADS")
META_ARCH_REGISTRY = _d2_registry("detectron2.modeling.meta_arch.build", "META_ARCH_REGISTRY") or Registry("META_ARCH")
# local registries in the reference (mask_head.py:17, maskiou_head.py:10)
ROI_MASK_HEAD_REGISTRY = Registry("ROI_MASK_HEAD")
ROI_MASKIOU_HEAD_REGISTRY = Registry("ROI_MASKIOU_HEAD")
