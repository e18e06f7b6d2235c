"""`model.text2embedding_model` -- drop-in module path of the reference (scripts/model/text2embedding_model.py);
implementation in gesture2vec_amd.model.text2embedding_model."""
import os as _os
import sys as _sys

_ROOT = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _ROOT not in _sys.path:
    _sys.path.insert(0, _ROOT)

from gesture2vec_amd.model.text2embedding_model import (  # noqa: E402,F401
    Attn, AttnDecoderRNN_New, BahdanauAttnDecoderRNN, DecoderRNN_New, EncoderRNN, EncoderRNN_New, Generator,
    text2embedding_model, text2embedding_model_New)
