"""dfepe — MI355X-native weighted-8-point hot path of deepFEPE (hand-written HIP behind a C ABI).

The directory name carries a hyphen, so the package is imported through importlib:

    import importlib; dfepe = importlib.import_module("pytorch-deepfepe_amd")

After that first import the alias ``dfepe`` (and ``dfepe.<submodule>``) is registered in sys.modules,
so ``import dfepe.compat`` style imports work too.
"""
import sys as _sys

from . import _lib, ops, synth, pipeline, estimator, compat, dist  # noqa: F401
from ._lib import DfepeError, LIB_PATH, EXPORTED_SYMBOLS  # noqa: F401

__version__ = "0.1.0"

_sys.modules.setdefault("dfepe", _sys.modules[__name__])
for _name in ("_lib", "ops", "synth", "pipeline", "estimator", "dist", "compat", "compat.DeepFNet", "compat.ErrorEstimators", "compat.utils_F",
              "compat.utils_geo", "compat.train_good_utils"):
    _sys.modules.setdefault("dfepe." + _name, _sys.modules[__name__ + "." + _name])
