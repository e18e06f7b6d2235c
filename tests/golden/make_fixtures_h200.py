#!/usr/bin/env python3
"""Golden vectors of the reference at the width it SHIPS (hidden_size = 200, E = 400: config/VQ-VAE.yml:21-22) -- SURVEY.md 8(c)'s
"cut-down native" fixture (B = 8, T = 20, D = 40, H = 200, L = 2, K = 512, dropout_prob = 0.2), two iterations of the
reference's own train_iter_Autoencoder_VQ_seq2seq (train_eval/train_seq2seq.py:664-758) on Autoencoder_VQVAE
(model/Autoencoder_VQVAE_model.py:686) with the EMA quantiser put back (as make_fixtures.py does).

Runs only in the build container (imports /root/reference); writes tests/golden/vqvae_h200.npz.  The model has 2.3 M
parameters, so unlike the small fixtures this one does NOT carry full weight tensors:
  * the INITIAL state is oracle/g2v_oracle.py's init_vqvae_state(D, H, L, K, seed=SEED) loaded into the reference model with
    load_state_dict(strict=True); the fixture stores a sha256 per tensor so that a test which regenerates it knows it has the
    same bits;
  * gradients (step 1) and post-training weights are stored as their float64 L2 norm + a strided sample of <= 512 elements
    (`sample_index` below); the EMA codebook state as the rows the batch touched + 32 fixed rows;
  * everything small is stored whole: inputs, every dropout mask, encoder_hidden, flat_input, code indices + top-2 gaps,
    quantized, outputs, losses, perplexity, _ema_cluster_size.

usage:  python tests/golden/make_fixtures_h200.py
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_fixtures as MF  # noqa: E402
from oracle import g2v_oracle as O  # noqa: E402

SEED = 11
B, T, D, H, L, K, P = 8, 20, 40, 200, 2, 512, 0.2
N_STEPS = 2


def sample_index(numel: int) -> np.ndarray:
    """the strided sample of a flattened tensor that fixture and tests agree on"""
    stride = max(1, numel // 512)
    return np.arange(0, numel, stride)[:512]


def digest(t: torch.Tensor) -> str:
    return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()


def codebook_rows(idx_all: np.ndarray) -> np.ndarray:
    return np.unique(np.concatenate([np.asarray(idx_all).reshape(-1), np.arange(0, K, K // 32)])).astype(np.int64)


def main():
    vq, _dae, ts = MF._import_reference()
    torch.set_num_threads(1)
    args = MF.make_args(rep_learning_dim=D, hidden_size=H, dropout_prob=P, autoencoder_vq_components=K, n_poses=T)
    net = MF.build_vqvae(vq, args, SEED)
    sd0 = O.init_vqvae_state(D, H, L, K, seed=SEED)
    net.load_state_dict(sd0, strict=True)
    net.train(True)
    x = torch.randn(B, T, D, generator=torch.Generator().manual_seed(1234))
    optim = torch.optim.Adam(net.parameters(), lr=args.learning_rate, betas=(0.5, 0.999))
    fx = {"x": x.numpy().copy(), "seed": np.int64(SEED),
          "cfg": np.array([B, T, D, H, L, K, N_STEPS], dtype=np.int64),
          "cfg_f": np.array([float(args.autoencoder_vq_commitment_cost), P, args.learning_rate, args.loss_l1_weight,
                             args.loss_cont_weight, args.loss_var_weight], dtype=np.float64)}
    for k_, v_ in sd0.items():
        fx["w0_sha256/" + k_] = np.array(digest(v_))
    idx_all = []
    for step in range(1, N_STEPS + 1):
        step_seed = 9000 + 17 * step + SEED
        hooks, probe, cap = {}, {}, {}

        def enc_hook(mod, inp, out):
            hooks["encoder_hidden"] = out[1].detach().numpy().copy()

        def vq_hook(mod, inp, out):
            hooks["loss_vq"] = float(out[0])
            hooks["quantized"] = out[1].detach().numpy().copy()
            hooks["encodings_idx"] = out[3].argmax(1).numpy().copy()

        def vq_pre_hook(mod, inp):
            f, d, i, gp = MF.vq_probe(mod, inp[0].detach())
            probe.update(flat=f, dist=d, idx=i, gap=gp)

        h1 = net.encoder.register_forward_hook(enc_hook)
        h2 = net.vq_layer.register_forward_hook(vq_hook)
        h3 = net.vq_layer.register_forward_pre_hook(vq_pre_hook)
        orig = ts.custom_loss

        def spy(output, target, a):
            cap["outputs"] = output.detach().numpy().copy()
            v = orig(output, target, a)
            cap["custom_loss"] = float(v)
            return v

        ts.custom_loss = spy
        torch.manual_seed(step_seed)
        with MF.MaskRecorder() as rec:
            loss, perp = ts.train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
        ts.custom_loss = orig
        h1.remove(); h2.remove(); h3.remove()
        # masks: same bookkeeping as make_fixtures.gen_vqvae_train (F.dropout draws recorded, nn.GRU's re-derived by RNG replay)
        masks = rec.masks
        fx[f"s{step}/mask_in"] = masks[0]
        dec = np.stack([m.reshape(B, D) for m in masks[1:]])
        assert dec.shape[0] == T - 1
        fx[f"s{step}/mask_dec"] = np.packbits(dec, axis=None)
        plan = [(("T01", (T, B, D)), P), ((T, B, 2 * H), P)]
        for _ in range(T - 1):
            plan += [((1, B, D), 0.95), ((1, B, H), P)]
        rp = MF.replay_gru_masks(step_seed, plan)
        assert np.array_equal(rp[0], masks[0]), "RNG replay misaligned (input dropout)"
        for t in range(T - 1):
            assert np.array_equal(rp[2 + 2 * t].reshape(B, D), dec[t]), "RNG replay misaligned (dec)"
        fx[f"s{step}/mask_enc_l0"] = np.packbits(rp[1], axis=None)          # (T,B,2H) bits
        fx[f"s{step}/mask_dec_l0"] = np.stack([rp[3 + 2 * t].reshape(B, H) for t in range(T - 1)])

        assert np.array_equal(probe["idx"], hooks["encodings_idx"])
        idx_all.append(probe["idx"])
        fx[f"s{step}/loss"] = np.float64(loss["loss"])
        fx[f"s{step}/perplexity"] = np.float64(float(perp))
        fx[f"s{step}/loss_vq"] = np.float64(hooks["loss_vq"])
        fx[f"s{step}/custom_loss"] = np.float64(cap["custom_loss"])
        fx[f"s{step}/idx"] = probe["idx"].astype(np.int64)
        fx[f"s{step}/gap"] = probe["gap"]
        fx[f"s{step}/dist_min"] = probe["dist"].min(1)
        fx[f"s{step}/encoder_hidden"] = hooks["encoder_hidden"]
        fx[f"s{step}/flat_input"] = probe["flat"]
        fx[f"s{step}/quantized"] = hooks["quantized"]
        fx[f"s{step}/outputs"] = cap["outputs"]
        fx[f"s{step}/ema_cluster_size"] = net.vq_layer._ema_cluster_size.detach().numpy().copy()
        rows = codebook_rows(probe["idx"])
        fx[f"s{step}/rows"] = rows
        fx[f"s{step}/ema_w_rows"] = net.vq_layer._ema_w.detach().numpy()[rows].copy()
        fx[f"s{step}/codebook_after_rows"] = net.vq_layer._embedding.weight.detach().numpy()[rows].copy()
        fx[f"s{step}/codebook_after_norm"] = np.float64(net.vq_layer._embedding.weight.detach().double().norm())
        if step == 1:
            for n_, p_ in net.named_parameters():
                if p_.grad is not None:
                    g = p_.grad.detach()
                    fx[f"s1/grad_norm/{n_}"] = np.float64(g.double().norm())
                    fx[f"s1/grad_sample/{n_}"] = g.reshape(-1).numpy()[sample_index(g.numel())].copy()
                else:
                    fx[f"s1/gradnone/{n_}"] = np.zeros(0, dtype=np.float32)
    for k_, v_ in net.state_dict().items():
        v_ = v_.detach()
        if v_.dtype.is_floating_point:
            fx[f"wN_norm/{k_}"] = np.float64(v_.double().norm())
            fx[f"wN_sample/{k_}"] = v_.reshape(-1).numpy()[sample_index(v_.numel())].copy()
        else:
            fx[f"wN_int/{k_}"] = v_.numpy().copy()
    out = os.path.join(HERE, "vqvae_h200.npz")
    np.savez_compressed(out, **fx)
    print("[vqvae_h200] losses", [float(fx[f"s{s}/loss"]) for s in (1, 2)], "perp", [float(fx[f"s{s}/perplexity"]) for s in (1, 2)],
          "min gap", [float(fx[f"s{s}/gap"].min()) for s in (1, 2)], "bytes", os.path.getsize(out))


if __name__ == "__main__":
    main()
