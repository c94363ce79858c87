"""A small yacs/detectron2-`CfgNode`-compatible config node.

yacs and detectron2 are absent from the reference tree and from this image; the
reference only relies on: attribute access, `clone`, `merge_from_file` with `_BASE_`
inheritance (zy_model_config.yaml:1), `merge_from_list` (tester.py:144-148),
`freeze`/`defrost` (deploy_utils.py:46-57).  Values in yaml written as python tuples
`(a, b)` (Base-CenterMask-VoVNet.yaml:26-34) are literal-eval'ed like yacs does.
"""
import ast
import copy
import os
from typing import Any, List

import yaml

BASE_KEY = "_BASE_"


class CfgNode(dict):
    IMMUTABLE = "__immutable__"

    def __init__(self, init_dict=None):
        super().__init__()
        self.__dict__[CfgNode.IMMUTABLE] = False
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name: str) -> Any:
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name: str, value: Any) -> None:
        if self.is_frozen():
            raise AttributeError("Attempted to set {} to {}, but CfgNode is immutable".format(name, value))
        self[name] = value

    def is_frozen(self) -> bool:
        return self.__dict__[CfgNode.IMMUTABLE]

    def _set_immutable(self, flag: bool) -> None:
        self.__dict__[CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_immutable(flag)

    def freeze(self) -> None:
        self._set_immutable(True)

    def defrost(self) -> None:
        self._set_immutable(False)

    def clone(self) -> "CfgNode":
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        new = CfgNode()
        for k, v in self.items():
            dict.__setitem__(new, k, copy.deepcopy(v, memo))
        new.__dict__[CfgNode.IMMUTABLE] = self.is_frozen()
        return new

    # -- merging --------------------------------------------------------------
    @staticmethod
    def _decode(v: Any) -> Any:
        if isinstance(v, dict):
            return CfgNode(v)
        if isinstance(v, str):
            try:
                v = ast.literal_eval(v)
            except (ValueError, SyntaxError):
                pass
        return v

    @staticmethod
    def _coerce(new: Any, old: Any, key: str) -> Any:
        if old is None or type(new) == type(old):
            return new
        if isinstance(old, tuple) and isinstance(new, list):
            return tuple(new)
        if isinstance(old, list) and isinstance(new, tuple):
            return list(new)
        if isinstance(old, float) and isinstance(new, int):
            return float(new)
        if isinstance(old, bool) or isinstance(new, bool) or type(new) != type(old):
            raise ValueError("Type mismatch ({} vs. {}) for config key: {}".format(type(old), type(new), key))
        return new

    def _merge(self, other: dict, root_keys: List[str], new_allowed: bool = False) -> None:
        for k, v in other.items():
            full = ".".join(root_keys + [k])
            v = self._decode(copy.deepcopy(v))
            if k not in self:
                if new_allowed:
                    dict.__setitem__(self, k, v)
                    continue
                raise KeyError("Non-existent config key: {}".format(full))
            if isinstance(self[k], CfgNode) and isinstance(v, dict):
                self[k]._merge(v, root_keys + [k], new_allowed)
            else:
                dict.__setitem__(self, k, self._coerce(v, self[k], full))

    @classmethod
    def load_yaml_with_base(cls, filename: str) -> dict:
        with open(filename, "r") as f:
            cfg = yaml.safe_load(f) or {}

        def merge_a_into_b(a: dict, b: dict) -> None:
            for k, v in a.items():
                if isinstance(v, dict) and k in b and isinstance(b[k], dict):
                    merge_a_into_b(v, b[k])
                else:
                    b[k] = v

        if BASE_KEY in cfg:
            base_file = cfg.pop(BASE_KEY)
            if base_file.startswith("~"):
                base_file = os.path.expanduser(base_file)
            if not os.path.isabs(base_file):
                base_file = os.path.join(os.path.dirname(filename), base_file)
            base_cfg = cls.load_yaml_with_base(base_file)
            merge_a_into_b(cfg, base_cfg)
            return base_cfg
        return cfg

    def merge_from_file(self, cfg_filename: str, allow_unsafe: bool = False) -> None:
        if self.is_frozen():
            raise AttributeError("CfgNode is immutable")
        self._merge(self.load_yaml_with_base(cfg_filename), [])

    def merge_from_other_cfg(self, other: "CfgNode") -> None:
        self._merge(other, [])

    def merge_from_list(self, cfg_list: List[Any]) -> None:
        if self.is_frozen():
            raise AttributeError("CfgNode is immutable")
        assert len(cfg_list) % 2 == 0, "Override list has odd length: {}".format(cfg_list)
        for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
            d = self
            keys = full_key.split(".")
            for sub in keys[:-1]:
                if sub not in d:
                    raise KeyError("Non-existent key: {}".format(full_key))
                d = d[sub]
            if keys[-1] not in d:
                raise KeyError("Non-existent key: {}".format(full_key))
            dict.__setitem__(d, keys[-1], self._coerce(self._decode(v), d[keys[-1]], full_key))

    def dump(self) -> str:
        def to_dict(n):
            return {k: to_dict(v) if isinstance(v, CfgNode) else (list(v) if isinstance(v, tuple) else v) for k, v in n.items()}
        return yaml.safe_dump(to_dict(self))
