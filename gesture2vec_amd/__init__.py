"""gesture2vec_amd -- MI355X-native (gfx950) implementation of Gesture2Vec's training hot path:
chunk VQ-VAE (GRU pose encoder -> EMA vector quantiser -> autoregressive GRU pose decoder) behind the
reference's own Python operator surface.  Arithmetic lives in hand-written HIP kernels (csrc/) reached
through the C-ABI declared in include/g2v.h; PyTorch only provides device memory, streams and RCCL."""
__version__ = "0.1.0"
