"""Mirror of the reference's ``utils/metrics`` package for the hot path: ``cd`` (utils/metrics/CD/__init__.py:1,
utils/metrics/__init__.py:1-5).  ``fscore``/``emd`` are completion-net metrics outside the path."""
from .chamfer import chamfer_3D, chamfer_3DDist, chamfer_3DFunction

cd = chamfer_3DDist

__all__ = ["cd", "chamfer_3D", "chamfer_3DDist", "chamfer_3DFunction"]
