"""houv_amd -- MI355X-native (gfx950) HOUV partial-to-partial registration hot path.

Host-side mirror of the reference's Python surface over the C ABI in include/houv_hip.h:

    houv_amd.metrics                   <- utils/metrics            (cd, chamfer_3D shim)
    houv_amd.model_utils_completion    <- registration/model_utils_completion.py (calc_cd*, loss_view)
    houv_amd.models.houv               <- registration/models/houv.py (HOUV, predict_model, solve_model, Predict_loss)
    houv_amd.train_utils               <- registration/train_utils.py (solve, getPredict_angle, metrics)
    houv_amd.model_utils               <- registration/model_utils.py (SVDHead, nearest_neighbor)
    houv_amd.compat.install()          makes the reference's own import names resolve to these modules

Every op runs in hand-written HIP kernels; there is no CPU fallback.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
