"""Drop-in aliasing: make the import names the reference's scripts use resolve to houv_amd.

The reference's drivers do (registration/train_HOUV.py:26-38, model_utils_completion.py:17-18)

    sys.path.append("../utils"); from metrics import cd
    from models.houv import HOUV, predict_model, solve_model
    from train_utils import solve, rotation_error, translation_error, rmse_loss, AverageValueMeter
    from model_utils import SVDHead
    from models.dcp import Model                      # registration/models/dcp.py (inference)
    from mm3d_pn2 import furthest_point_sample, gather_points

After ``houv_amd.compat.install()`` those statements import the MI355X implementations."""
import importlib
import sys

_ALIASES = {
    "metrics": "houv_amd.metrics",
    "models": "houv_amd.models",
    "models.houv": "houv_amd.models.houv",
    "models.dcp": "houv_amd.models.dcp",
    "mm3d_pn2": "houv_amd.mm3d_pn2",
    "train_utils": "houv_amd.train_utils",
    "model_utils": "houv_amd.model_utils",
    "model_utils_completion": "houv_amd.model_utils_completion",
}


def install(force=False):
    """Register the aliases.  Existing modules of the same name are left alone unless ``force``."""
    done = []
    for name, target in _ALIASES.items():
        if name in sys.modules and not force:
            continue
        sys.modules[name] = importlib.import_module(target)
        done.append(name)
    return done


def uninstall():
    for name, target in _ALIASES.items():
        m = sys.modules.get(name)
        if m is not None and getattr(m, "__name__", "") == target:
            del sys.modules[name]
