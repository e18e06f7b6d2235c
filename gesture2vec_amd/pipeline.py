"""The callers either side of the chunk VQ-VAE hot path, batched on the device.

* `chunk_latents` / `chunks_to_codes`: the bulk latent-extraction + code-assignment path.  The reference does this one
  chunk at a time on the CPU: `data_preprocessor.py:366-457` runs `net.encoder` per chunk and stores the chunk's
  `[h_l0 ; h_l1]` row, `TrinityDataset_sentencelevel.__getitem__` (`lmdb_data_loader.py:1255-1281`) then feeds the
  `(S, L*H)` rows to `vq_layer(...)` and takes `argmax(encodings)`; `Clustering.generate_gestures_latent_dataset`
  (`Clustering.py:151-157`) does the same for whole recordings.  Here all N chunks go through one encoder pass and one
  `g2v_vq_assign_fwd` launch (the shape at which that kernel reaches its roofline).
* `stacked_autoencode`: config 3, raw poses -> frozen `DAE_Network.encoder` (`lmdb_data_loader.py:649-653`) ->
  chunk VQ-VAE -> `DAE.decoder` (`inference_Autoencoder.py:239`).

Forward-only, eval-mode helpers; every arithmetic step is a HIP kernel behind include/g2v.h."""
from __future__ import annotations

from typing import Tuple

import torch


def _need_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what} runs on the MI355X kernels only (no CPU fallback)")


@torch.no_grad()
def chunk_latents(net, chunks: torch.Tensor) -> torch.Tensor:
    """(N,T,D) pose chunks -> (N, L*H) latent rows, row n = [h_l0_fwd(n) ; h_l0_bwd(n)] = encoder_hidden[:L] of chunk n
    laid out per chunk (the reference runs the encoder with batch 1, where `view(-1, E)` is exactly this concat)."""
    _need_cuda(chunks, "chunk_latents")
    L = net.encoder.n_layers
    _, hidden = net.encoder(chunks.transpose(0, 1).contiguous(), None)        # (2L,N,H)
    return hidden[:L].transpose(0, 1).reshape(chunks.shape[0], -1).contiguous()  # layout only


@torch.no_grad()
def chunks_to_codes(net, chunks: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(N,T,D) pose chunks -> (latents (N,L*H), code ids (N,) int64) with the net's EMA codebook."""
    lat = chunk_latents(net, chunks)
    return lat, net.vq_layer.assign(lat)


@torch.no_grad()
def stacked_autoencode(dae, net, poses: torch.Tensor):
    """Config 3: raw (B,T,D_raw) poses -> DAE encoder (per frame) -> chunk VQ-VAE -> DAE decoder -> (B,T,D_raw).
    Returns (reconstruction, vq-vae output in the DAE latent space, perplexity).  `net` decodes in whatever mode it is in
    (its inline Dropout(0.95) is always active, as in the reference)."""
    _need_cuda(poses, "stacked_autoencode")
    B, T, D = poses.shape
    lat = dae.encode(poses.reshape(B * T, D).contiguous()).view(B, T, -1) if dae.encoder is not None else poses
    out, _, _, perplexity = net(lat, lat)
    rec = dae.decode(out.reshape(B * T, -1).contiguous()).view(B, T, D) if dae.decoder is not None else out
    return rec, out, perplexity
