"""`train_eval.train_seq2seq` -- drop-in module path of the reference; implementation in gesture2vec_amd.train_eval."""
import os as _os
import sys as _sys

_ROOT = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _ROOT not in _sys.path:
    _sys.path.insert(0, _ROOT)

from gesture2vec_amd.train_eval.train_seq2seq import (  # noqa: E402,F401
    FusedClipAdam, GraphedText2EmbeddingStep, custom_loss, train_iter_Autoencoder_VQ_seq2seq,
    train_iter_Autoencoder_VQ_seq2seq_dp, train_iter_DAE, train_iter_text2embedding)
