"""dfepe — MI355X-native weighted-8-point hot path of deepFEPE (hand-written HIP behind a C ABI).

The directory name carries a hyphen, so the package is imported through importlib:

    import importlib; dfepe = importlib.import_module("pytorch-deepfepe_amd")

After that first import the alias ``dfepe`` (and ``dfepe.<submodule>``) is registered in sys.modules,
so ``import dfepe.compat`` style imports work too.
"""
import os as _os
import sys as _sys

import torch as _torch

# ROCm 7.2 hipGraph workaround, applied before the HIP runtime initialises (its flags are read at the first HIP call).  With the
# runtime's "graph packet capture" (AQL packets recorded at the first launch of a graph and re-submitted afterwards) a captured
# training step of the full model replays WRONG from its second launch on: some kernels enqueued by the autograd engine's thread --
# the reductions behind the estimator's parameter gradients -- leave their outputs unwritten (scripts/est_capture_debug.py: first
# replay bit-identical to eager, every later one wrong; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: all bit-identical).  Without it the
# runtime dispatches the graph's nodes the ordinary way, which for this package's graphs (11-80 small kernels) is also 3 % FASTER
# (bench.py: 0.1063 vs 0.1101 ms per step).  An explicit setting in the environment wins.
HIP_GRAPH_PACKET_CAPTURE_OFF = _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0"
if "DEBUG_CLR_GRAPH_PACKET_CAPTURE" not in _os.environ:
    _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
    HIP_GRAPH_PACKET_CAPTURE_OFF = not _torch.cuda.is_initialized()  # too late otherwise: the runtime has read its flags

from . import _lib, ops, synth, pipeline, estimator, compat, dist  # noqa: F401
from ._lib import DfepeError, LIB_PATH, EXPORTED_SYMBOLS  # noqa: F401

__version__ = "0.1.0"

_sys.modules.setdefault("dfepe", _sys.modules[__name__])
for _name in ("_lib", "ops", "synth", "pipeline", "estimator", "dist", "compat", "compat.DeepFNet", "compat.ErrorEstimators", "compat.utils_F",
              "compat.utils_geo", "compat.train_good_utils"):
    _sys.modules.setdefault("dfepe." + _name, _sys.modules[__name__ + "." + _name])
