"""MI355X-native implementation of ProxyTransformation's point-cloud preshaping path.

Only what the path needs lives here:

* ``module.ProxyTransformationNormReverse`` -- the reference's registry entry / nn.Module
  surface (embodiedscan/models/necks/preshape_norm_reverse_drop.py:280-469);
* ``csrc/`` + ``libproxyt_hip.so``           -- hand-written HIP kernels for gfx950 behind
  the C ABI of ``include/proxyt.h`` (bound with ctypes in ``_abi``);
* ``registry.MODELS``                        -- embodiedscan/mmengine registry or a stand-alone shim;
* ``shard``                                  -- scene sharding across the GPUs of one node;
* ``synth``                                  -- seeded synthetic scenes / closed-form weights.

Importing the package does not need a GPU and does not load the shared library;
``forward`` fails loudly when either is missing (there is no CPU fallback).
"""
from .registry import MODELS, REGISTRY_BACKEND
from .module import ProxyTransformationNormReverse

__all__ = ["MODELS", "REGISTRY_BACKEND", "ProxyTransformationNormReverse"]
__version__ = "0.1.0"
