"""Mirror of the reference's Python call surface for the hot path: same class / function names,
argument order, defaults, return arity, tensor layouts and dict keys, backed by libdfepe_hip.so.

    reference module                         this package
    deepFEPE.models.DeepFNet            ->   compat.DeepFNet        (NormalizeAndExpand_HW, Fit, DeepFNet)
    deepFEPE.models.ErrorEstimators     ->   compat.ErrorEstimators (stock PyTorch; not part of the hot path)
    deepFEPE.dsac_tools.utils_F         ->   compat.utils_F
    deepFEPE.dsac_tools.utils_geo       ->   compat.utils_geo
    deepFEPE.train_good_utils           ->   compat.train_good_utils (get_all_loss_DeepF, get_Rt_loss, get_matches_from_SP)
    deepFEPE.dsac_tools.utils_misc      ->   compat.utils_misc      (homogeneous / rigid-transform helpers, crop_or_pad_choice)
    deepFEPE.dsac_tools.dsac            ->   compat.dsac            (DSAC hypothesis loop, all hypotheses per launch)
    superpoint.models.model_wrap        ->   compat.model_wrap      (PointTracker.nn_match_two_way only)

    (no counterpart: the reference's agent is eager)  compat.CapturedStep  (its training step as one replayed hipGraph)

See INTEGRATION.md for how train_good.py is pointed at these.
"""
from . import DeepFNet, ErrorEstimators, captured, dsac, model_wrap, train_good_utils, utils_F, utils_geo, utils_misc  # noqa: F401
from .captured import CapturedStep  # noqa: F401
