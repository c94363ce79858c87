"""`get_cfg()` returns a clone of the defaults (reference config/config.py:4-13)."""
import os

from .cfgnode import CfgNode


def get_cfg() -> CfgNode:
    from .defaults import _C

    return _C.clone()


def config_path(name: str) -> str:
    """Path of a yaml shipped with the package (`configs/centermask/...`)."""
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "centermask", name)


__all__ = ["CfgNode", "get_cfg", "config_path"]
