"""Reader of tests/golden/vqvae_h200.npz (tests/golden/make_fixtures_h200.py: the reference at its SHIPPED width, H = 200, E = 400).
The fixture carries no full weight tensors: the initial state is oracle.g2v_oracle.init_vqvae_state(..., seed) -- checked here
against the fixture's per-tensor sha256 -- and big tensors are compared through their float64 norm + a strided sample."""
import hashlib
import os

import numpy as np
import torch

from oracle import g2v_oracle as O


def sample_index(numel: int) -> np.ndarray:
    stride = max(1, numel // 512)
    return np.arange(0, numel, stride)[:512]


def load(golden_dir):
    fx = np.load(os.path.join(golden_dir, "vqvae_h200.npz"))
    B, T, D, H, L, K, n_steps = [int(v) for v in fx["cfg"]]
    sd = O.init_vqvae_state(D, H, L, K, seed=int(fx["seed"]))
    for k, v in sd.items():
        want = str(fx["w0_sha256/" + k])
        got = hashlib.sha256(v.contiguous().numpy().tobytes()).hexdigest()
        assert got == want, f"regenerated initial state differs from the one the reference ran on: {k}"
    return fx, sd


def masks(fx, step, B, T, D, H):
    return {"dec": O.unpack_mask(fx[f"s{step}/mask_dec"], (T - 1, B, D)),
            "in": torch.from_numpy(fx[f"s{step}/mask_in"].copy()),
            "enc_l0": O.unpack_mask(fx[f"s{step}/mask_enc_l0"], (T, B, 2 * H)),
            "dec_l0": torch.from_numpy(fx[f"s{step}/mask_dec_l0"].copy())}


def check_sampled(got, norm_ref, sample_ref, rtol, atol, what):
    """`got` (any tensor) against the fixture's (float64 L2 norm, strided sample) of the reference's tensor"""
    g = torch.as_tensor(got).detach().cpu().double().reshape(-1)
    n = float(g.norm())
    assert abs(n - float(norm_ref)) <= rtol * float(norm_ref) + atol, (what, "norm", n, float(norm_ref))
    s = g.numpy()[sample_index(g.numel())]
    scale = max(float(np.abs(sample_ref).max()), 1e-30)
    err = float(np.abs(s - sample_ref.astype(np.float64)).max())
    assert err <= rtol * scale + atol, (what, "sample", err, scale)
