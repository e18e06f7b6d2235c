"""Part a: frame-level denoising autoencoder -- mirror of `scripts/model/DAE_model.py::DAE_Network` (:22-114) on the
MI355X dense-layer kernels.  Dropout(0.2) -> Linear(motion_dim, latent) + ReLU -> Linear(latent, motion_dim).
state_dict keys `encoder.0.*`, `decoder.0.*` as in the reference.  Sentinels: latent_dim == -1 is the identity
(:52-55), latent_dim == -2 is a linear 200-d bottleneck with 30 % dropout (:58-66)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import functional as Fn
from .. import ops


class DAE_Network(nn.Module):
    def __init__(self, motion_dim: int, latent_dim: int):
        super().__init__()
        self.dropout = nn.Dropout(0.2)             # container for p only; masks come from the Philox kernel
        self._relu = True
        self._explicit_keep = None
        self._rng_counter = None
        self.rng_seed = 0
        if latent_dim == -1:
            self.encoder = None
            self.decoder = None
            return
        if latent_dim == -2:
            self.encoder = nn.Sequential(nn.Linear(motion_dim, 200))
            self.decoder = nn.Sequential(nn.Linear(200, motion_dim))
            self.dropout = nn.Dropout(0.3)
            self._relu = False
            return
        self.encoder = nn.Sequential(nn.Linear(motion_dim, latent_dim), nn.ReLU())
        self.decoder = nn.Sequential(nn.Linear(latent_dim, motion_dim))

    def set_dropout_mask(self, keep):
        """Explicit uint8 keep mask for the next training forward (parity tests)."""
        self._explicit_keep = keep

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """`rep_model.encoder(x)` as the datasets call it (lmdb_data_loader.py:649-653): Linear(+ReLU), no dropout."""
        e = self.encoder[0]
        return Fn.linear(x, e.weight, e.bias, act=1 if self._relu else 0)

    def decode(self, lat: torch.Tensor) -> torch.Tensor:
        d = self.decoder[0]
        return Fn.linear(lat, d.weight, d.bias)

    def forward(self, x: torch.Tensor, get_latent: bool = False):
        if self.encoder is None:
            return (x, x) if get_latent else x
        inp = torch.squeeze(x)
        if not inp.is_cuda:
            raise RuntimeError("DAE_Network runs on the MI355X kernels only (no CPU fallback)")
        keep, scale = None, 1.0
        if self.training:
            p = self.dropout.p
            if self._explicit_keep is not None:
                keep = self._explicit_keep
            else:
                if self._rng_counter is None or self._rng_counter.device != inp.device:
                    self._rng_counter = torch.zeros(1, dtype=torch.int64, device=inp.device)
                keep = ops.keep_mask(torch.empty(inp.shape, dtype=torch.uint8, device=inp.device), 1.0 - p, self.rng_seed,
                                     self._rng_counter)
            scale = 1.0 / (1.0 - p)
        e = self.encoder[0]
        lat = Fn.linear(inp, e.weight, e.bias, keep=keep, scale=scale, act=1 if self._relu else 0)
        out = self.decode(lat)
        out = torch.unsqueeze(out, 2)
        return (out, lat.detach().clone()) if get_latent else out
