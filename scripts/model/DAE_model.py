"""`model.DAE_model` -- drop-in module path of the reference (scripts/model/DAE_model.py); implementation in
gesture2vec_amd.model.DAE_model."""
import os as _os
import sys as _sys

_ROOT = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _ROOT not in _sys.path:
    _sys.path.insert(0, _ROOT)

from gesture2vec_amd.model.DAE_model import DAE_Network  # noqa: E402,F401
