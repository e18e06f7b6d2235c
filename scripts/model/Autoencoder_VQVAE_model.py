"""`model.Autoencoder_VQVAE_model` -- drop-in module path of the reference (scripts/model/Autoencoder_VQVAE_model.py).
The implementation lives in gesture2vec_amd.model.Autoencoder_VQVAE_model (MI355X HIP kernels behind include/g2v.h)."""
import os as _os
import sys as _sys

_ROOT = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _ROOT not in _sys.path:
    _sys.path.insert(0, _ROOT)

from gesture2vec_amd.model.Autoencoder_VQVAE_model import (  # noqa: E402,F401
    Attn, Autoencoder_VQVAE, BahdanauAttnDecoderRNN, EncoderRNN, Generator, VQ_Payam, VQ_Payam_EMA, VQ_Payam_GSSoft,
    VectorQuantizer, VectorQuantizerEMA)
