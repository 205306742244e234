"""``MODELS`` registry used to publish the module under the reference's name.

The reference registers the neck with ``@MODELS.register_module()`` on the
embodiedscan child registry (embodiedscan/registry.py:11-13,
preshape_norm_reverse_drop.py:280) and builds it with
``MODELS.build(dict(type='ProxyTransformationNormReverse', ...))``
(sparse_featfusion_grounder_preshape.py:95, config line 41).

Resolution order, so the module drops into a real EmbodiedScan install and
still works in a bare image:
1. ``embodiedscan.registry.MODELS`` if embodiedscan imports;
2. a fresh ``mmengine.Registry('model')`` if only mmengine is present;
3. the minimal stand-alone registry below (same ``register_module`` /
   ``build`` / ``get`` surface for the calls this path needs).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional


class _MiniRegistry:
    """Smallest useful subset of ``mmengine.Registry``."""

    def __init__(self, name: str):
        self.name = name
        self._module_dict: Dict[str, type] = {}

    def register_module(self, name: Optional[str] = None, force: bool = False,
                        module: Optional[type] = None):
        def _register(cls):
            key = name or cls.__name__
            if key in self._module_dict and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self._module_dict[key] = cls
            return cls
        if module is not None:
            return _register(module)
        return _register

    def get(self, key: str):
        return self._module_dict.get(key)

    def build(self, cfg: dict):
        if not isinstance(cfg, dict) or "type" not in cfg:
            raise KeyError('cfg must be a dict containing the key "type"')
        args = dict(cfg)
        obj_type = args.pop("type")
        cls = self.get(obj_type) if isinstance(obj_type, str) else obj_type
        if cls is None:
            raise KeyError(f"{obj_type} is not in the {self.name} registry")
        return cls(**args)

    def __contains__(self, key: str) -> bool:
        return key in self._module_dict


def _resolve():
    try:  # a real EmbodiedScan checkout
        from embodiedscan.registry import MODELS as models  # type: ignore
        return models, "embodiedscan"
    except Exception:
        pass
    try:
        from mmengine import Registry  # type: ignore
        return Registry("model"), "mmengine"
    except Exception:
        return _MiniRegistry("model"), "standalone"


MODELS, REGISTRY_BACKEND = _resolve()
