#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference).  The reference's source
never travels: only the .npz data files written here are committed.  Every
fixture records the inputs, the weights, every dropout mask that was drawn, and
the outputs of the reference's own code (reference on this container's torch).

Reference entry points exercised (paths relative to /root/reference/scripts):
  model/Autoencoder_VQVAE_model.py : Autoencoder_VQVAE (:686), VQ_Payam_EMA (:1182),
                                     VQ_Payam (:1088), VectorQuantizerEMA (:1713)
  train_eval/train_seq2seq.py      : custom_loss (:40), train_iter_Autoencoder_VQ_seq2seq (:664),
                                     train_iter_DAE (:161), train_iter_text2embedding (:462)
  model/DAE_model.py               : DAE_Network (:22)
  model/text2embedding_model.py    : text2embedding_model (:488)

Oracle adjustments made from OUTSIDE the reference, no file edits:
  * `configargparse` / `fasttext` are absent -> stub modules.
  * Autoencoder_VQVAE.__init__ overwrites the EMA quantizer with GSSoft (:816-820);
    we put `VQ_Payam_EMA(K, H*L, beta, 0.85)` back (same ctor call as :801-807).
  * torch.nn.functional.dropout is replaced by an equivalent that RECORDS the
    keep-mask it draws (noise = bernoulli(1-p); out = x * noise / (1-p), the
    same formula ATen uses), so nn.Dropout masks are replayable.
  * nn.GRU's inter-layer dropout is drawn inside ATen and cannot be recorded;
    fixtures with dropout_prob > 0 re-derive those masks by replaying the global
    CPU RNG stream with identical draw shapes and order (checked: the recorded
    nn.Dropout masks must coincide in the replay).

usage:  python tests/golden/make_fixtures.py            (writes tests/golden/*.npz)
"""
from __future__ import annotations

import argparse
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/scripts"
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    sys.dont_write_bytecode = True
    m = types.ModuleType("configargparse")
    m.argparse = argparse
    sys.modules["configargparse"] = m
    sys.modules["fasttext"] = types.ModuleType("fasttext")
    sys.path.insert(0, REF)
    import model.Autoencoder_VQVAE_model as vq  # noqa
    import model.DAE_model as dae  # noqa

    spec = importlib.util.spec_from_file_location("ref_train_seq2seq", REF + "/train_eval/train_seq2seq.py")
    ts = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ts)
    return vq, dae, ts


class MaskRecorder:
    """Replaces torch.nn.functional.dropout; records keep-masks (uint8) in call order."""

    def __init__(self):
        self.masks = []
        self.ps = []
        self._orig = None

    def __enter__(self):
        import torch.nn.functional as F

        self._orig = F.dropout

        def dropout(input, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return input
            noise = torch.empty_like(input).bernoulli_(1 - p)
            self.masks.append(noise.to(torch.uint8).numpy().copy())
            self.ps.append(p)
            return input * noise / (1 - p)

        F.dropout = dropout
        return self

    def __exit__(self, *a):
        import torch.nn.functional as F

        F.dropout = self._orig


def make_args(**kw):
    d = dict(
        rep_learning_dim=135, hidden_size=64, n_layers=2, dropout_prob=0.0,
        autoencoder_vae="False", autoencoder_vq="True", autoencoder_vq_components=64,
        autoencoder_vq_commitment_cost=0.25, n_pre_poses=1, autoencoder_conditioned="True",
        autoencoder_att="False", autoencoder_fixed_weight="False", n_poses=34,
        loss_l1_weight=5.0, loss_cont_weight=0.1, loss_var_weight=0.5,
        learning_rate=5e-4, epochs=10, text2_embedding_discrete="True",
        autoencoder_freeze_encoder="False",
    )
    d.update(kw)
    return argparse.Namespace(**d)


def build_vqvae(vq, args, seed):
    torch.manual_seed(seed)
    net = vq.Autoencoder_VQVAE(args, args.rep_learning_dim, args.n_poses)
    # undo the GSSoft override (:816-820): same ctor call as :801-807
    net.vq_layer = vq.VQ_Payam_EMA(
        int(args.autoencoder_vq_components), args.hidden_size * args.n_layers,
        float(args.autoencoder_vq_commitment_cost), 0.85)
    return net


def sd_np(net, prefix="w/"):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}


def vq_probe(vq_layer, hidden):
    """Recompute the quantizer's intermediates exactly as :1229-1244 does (no state change)."""
    with torch.no_grad():
        flat = vq_layer.pre_linear(hidden.reshape(-1, vq_layer._embedding_dim))
        w = vq_layer._embedding.weight
        d = (flat ** 2).sum(1, keepdim=True) + (w ** 2).sum(1) - 2 * flat @ w.t()
        idx = d.argmin(1)
        top2 = torch.topk(d, 2, dim=1, largest=False).values
        gap = top2[:, 1] - top2[:, 0]
    return flat.numpy().copy(), d.numpy().copy(), idx.numpy().copy(), gap.numpy().copy()


def replay_gru_masks(seed, plan):
    """Replay the global CPU RNG with the same draw order/shapes to recover every mask,
    including the ones ATen draws inside nn.GRU.  plan = [(shape, p), ...] in draw order."""
    torch.manual_seed(seed)
    out = []
    for shape, p in plan:
        if isinstance(shape, tuple) and shape and shape[0] == "T01":
            # in_poses.transpose(0,1) is a non-contiguous view and ATen draws into empty_like(view)
            _, (T, B, D) = shape
            e = torch.empty(B, T, D).transpose(0, 1)
        else:
            e = torch.empty(shape)
        out.append(e.bernoulli_(1 - p).to(torch.uint8).numpy().copy())
    return out


def gen_vqvae_train(vq, ts, name, args, B, seed, n_steps, store_grads=True):
    """Runs the reference's train_iter_Autoencoder_VQ_seq2seq n_steps times on one batch."""
    T, D, H, L = args.n_poses, args.rep_learning_dim, args.hidden_size, args.n_layers
    p = args.dropout_prob
    net = build_vqvae(vq, args, seed)
    net.train(True)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, T, D, generator=g)
    optim = torch.optim.Adam(net.parameters(), lr=args.learning_rate, betas=(0.5, 0.999))
    fx = dict(sd_np(net, "w0/"))
    fx["x"] = x.numpy().copy()
    fx["cfg"] = np.array([B, T, D, H, L, int(args.autoencoder_vq_components), n_steps], dtype=np.int64)
    fx["cfg_f"] = np.array([float(args.autoencoder_vq_commitment_cost), p, args.learning_rate,
                            args.loss_l1_weight, args.loss_cont_weight, args.loss_var_weight], dtype=np.float64)

    for step in range(1, n_steps + 1):
        # probe pre-update quantizer intermediates with a mask-replayed, side-effect-free forward
        step_seed = 9000 + 17 * step + seed
        # --- the reference train step itself ---
        hooks = {}

        def enc_hook(mod, inp, out):
            hooks["encoder_hidden"] = out[1].detach().numpy().copy()

        def vq_hook(mod, inp, out):
            hooks["vq_in"] = inp[0].detach().clone()
            hooks["loss_vq"] = float(out[0])
            hooks["quantized"] = out[1].detach().numpy().copy()
            hooks["encodings_idx"] = out[3].argmax(1).numpy().copy()

        # pre-hook to probe the PRE-update codebook with the real vq input
        probe = {}

        def vq_pre_hook(mod, inp):
            f, d, i, gp = vq_probe(mod, inp[0].detach())
            probe.update(flat=f, dist=d, idx=i, gap=gp,
                         codebook=mod._embedding.weight.detach().numpy().copy())

        h1 = net.encoder.register_forward_hook(enc_hook)
        h2 = net.vq_layer.register_forward_hook(vq_hook)
        h3 = net.vq_layer.register_forward_pre_hook(vq_pre_hook)
        cap = {}
        orig_custom_loss = ts.custom_loss

        def custom_loss_spy(output, target, a):
            cap["outputs"] = output.detach().numpy().copy()
            val = orig_custom_loss(output, target, a)
            cap["custom_loss"] = float(val)
            return val

        ts.custom_loss = custom_loss_spy
        torch.manual_seed(step_seed)
        with MaskRecorder() as rec:
            loss, perp = ts.train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
        ts.custom_loss = orig_custom_loss
        h1.remove(); h2.remove(); h3.remove()

        # mask bookkeeping. F.dropout call order in forward (:956-1054):
        #   [self.do on in_poses (T,B,D) if p>0] then per t: Dropout(0.95) on (1,B,D)
        # ATen-internal draws: encoder GRU inter-layer (T,B,2H) if p>0 (after self.do),
        #   decoder GRU inter-layer (1,B,H) per step if p>0 (after that step's 0.95 draw).
        masks = rec.masks
        k = 0
        if p > 0:
            fx[f"s{step}/mask_in"] = masks[0]  # (T,B,D)
            k = 1
        dec = np.stack([m.reshape(B, D) for m in masks[k:]])  # (T-1,B,D)
        assert dec.shape[0] == T - 1
        fx[f"s{step}/mask_dec"] = np.packbits(dec, axis=None)
        if p > 0:
            plan = [(("T01", (T, B, D)), p), ((T, B, 2 * H), p)]
            for _ in range(T - 1):
                plan += [((1, B, D), 0.95), ((1, B, H), p)]
            rp = replay_gru_masks(step_seed, plan)
            assert np.array_equal(rp[0], masks[0]), "RNG replay misaligned (input dropout)"
            for t in range(T - 1):
                assert np.array_equal(rp[2 + 2 * t].reshape(B, D), dec[t]), "RNG replay misaligned (dec)"
            fx[f"s{step}/mask_enc_l0"] = rp[1]  # (T,B,2H)
            fx[f"s{step}/mask_dec_l0"] = np.stack([rp[3 + 2 * t].reshape(B, H) for t in range(T - 1)])

        fx[f"s{step}/loss"] = np.float64(loss["loss"])
        fx[f"s{step}/perplexity"] = np.float64(float(perp))
        fx[f"s{step}/loss_vq"] = np.float64(hooks["loss_vq"])
        fx[f"s{step}/custom_loss"] = np.float64(cap["custom_loss"])
        fx[f"s{step}/idx"] = probe["idx"].astype(np.int64)
        assert np.array_equal(probe["idx"], hooks["encodings_idx"])
        fx[f"s{step}/gap"] = probe["gap"]
        fx[f"s{step}/ema_cluster_size"] = net.vq_layer._ema_cluster_size.detach().numpy().copy()
        fx[f"s{step}/ema_w"] = net.vq_layer._ema_w.detach().numpy().copy()
        fx[f"s{step}/codebook_after"] = net.vq_layer._embedding.weight.detach().numpy().copy()
        if step == 1 or step == n_steps:
            fx[f"s{step}/encoder_hidden"] = hooks["encoder_hidden"]
            fx[f"s{step}/flat_input"] = probe["flat"]
            fx[f"s{step}/quantized"] = hooks["quantized"]
            fx[f"s{step}/outputs"] = cap["outputs"]
            fx[f"s{step}/dist_min"] = probe["dist"].min(1)
        if store_grads and step == 1:
            for n_, p_ in net.named_parameters():
                if p_.grad is not None:
                    fx[f"s{step}/grad/{n_}"] = p_.grad.detach().numpy().copy()
                else:
                    fx[f"s{step}/gradnone/{n_}"] = np.zeros(0, dtype=np.float32)
    fx.update(sd_np(net, "wN/"))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)

    # eval-mode forward from the final state with recorded masks (a12 / §3.5)
    net.train(False)
    torch.manual_seed(4242)
    with torch.no_grad(), MaskRecorder() as rec:
        outs, first_hidden, loss_vq, perp = net(x, x)
    ev = {
        "mask_dec": np.packbits(np.stack([m.reshape(B, D) for m in rec.masks]), axis=None),
        "outputs": outs.numpy().copy(), "first_hidden": first_hidden.numpy().copy(),
        "loss_vq": np.float64(float(loss_vq)), "perplexity": np.float64(float(perp)),
    }
    assert len(rec.masks) == T - 1
    np.savez_compressed(os.path.join(OUT, name + "_eval.npz"), **ev)
    print(f"[{name}] steps={n_steps} losses=", [float(fx[f's{s}/loss']) for s in range(1, n_steps + 1)],
          "perp=", [float(fx[f's{s}/perplexity']) for s in range(1, n_steps + 1)])


def gen_vq_layers(vq):
    """Quantizer operators alone (a5, a7, a8) at N=256, E=128, K=512, train mode, 2 calls."""
    torch.manual_seed(7)
    K, H, L, B = 512, 64, 2, 256
    E = H * L
    fx = {}
    z1 = torch.randn(L, B, H)
    z2 = torch.randn(L, B, H) * 0.5
    fx["z1"], fx["z2"] = z1.numpy().copy(), z2.numpy().copy()
    # --- VQ_Payam_EMA (:1182)
    q = vq.VQ_Payam_EMA(K, E, 0.25, 0.85)
    q.train(True)
    fx.update(sd_np(q, "ema/w0/"))
    for i, z in enumerate([z1, z2], 1):
        zz = z.clone().requires_grad_(True)
        flat, d, idx, gap = vq_probe(q, zz.detach())
        loss, quant, perp, enc = q(zz)
        gq = torch.randn(quant.shape, generator=torch.Generator().manual_seed(100 + i))
        (quant * gq).sum().backward(retain_graph=True)
        gz_from_q = zz.grad.clone(); zz.grad = None
        loss.backward()
        gz_from_loss = zz.grad.clone()
        fx[f"ema/c{i}/flat"], fx[f"ema/c{i}/idx"], fx[f"ema/c{i}/gap"] = flat, idx.astype(np.int64), gap
        fx[f"ema/c{i}/dist_min"] = d.min(1)
        fx[f"ema/c{i}/loss"] = np.float64(float(loss)); fx[f"ema/c{i}/perplexity"] = np.float64(float(perp))
        fx[f"ema/c{i}/quantized"] = quant.detach().numpy().copy()
        fx[f"ema/c{i}/gq"] = gq.numpy().copy()
        fx[f"ema/c{i}/gz_from_q"] = gz_from_q.numpy().copy()
        fx[f"ema/c{i}/gz_from_loss"] = gz_from_loss.numpy().copy()
        assert np.array_equal(enc.argmax(1).numpy(), idx)
        fx.update(sd_np(q, f"ema/w{i}/"))
    # eval-mode call (no EMA update)
    q.train(False)
    with torch.no_grad():
        loss, quant, perp, enc = q(z1)
    fx["ema/eval/idx"] = enc.argmax(1).numpy().astype(np.int64)
    fx["ema/eval/loss"] = np.float64(float(loss)); fx["ema/eval/perplexity"] = np.float64(float(perp))
    fx["ema/eval/quantized"] = quant.numpy().copy()

    # --- VQ_Payam (:1088) non-EMA: codebook learns by gradient
    torch.manual_seed(8)
    q2 = vq.VQ_Payam(K, E, 0.25)
    q2.train(True)
    fx.update(sd_np(q2, "plain/w0/"))
    zz = z1.clone().requires_grad_(True)
    loss, quant, perp, enc = q2(zz)
    gq = torch.randn(quant.shape, generator=torch.Generator().manual_seed(300))
    ((quant * gq).sum() + loss).backward()
    fx["plain/idx"] = enc.argmax(1).numpy().astype(np.int64)
    fx["plain/loss"] = np.float64(float(loss)); fx["plain/perplexity"] = np.float64(float(perp))
    fx["plain/quantized"] = quant.detach().numpy().copy()
    fx["plain/gq"] = gq.numpy().copy()
    fx["plain/gz"] = zz.grad.numpy().copy()
    fx["plain/g_embedding"] = q2._embedding.weight.grad.numpy().copy()

    # --- VectorQuantizerEMA (:1713) wrapper semantics (hstack in, reshape out, pre_lin in graph)
    torch.manual_seed(9)
    q3 = vq.VectorQuantizerEMA(K, E, 0.25, 0.85)
    q3.train(True)
    fx.update(sd_np(q3, "vqema/w0/"))
    zz = z1.clone().requires_grad_(True)
    out = q3(zz)
    loss, quant, perp, enc = out
    gq = torch.randn(quant.shape, generator=torch.Generator().manual_seed(301))
    ((quant * gq).sum() + loss).backward()
    fx["vqema/idx"] = enc.argmax(1).numpy().astype(np.int64)
    fx["vqema/loss"] = np.float64(float(loss)); fx["vqema/perplexity"] = np.float64(float(perp))
    fx["vqema/quantized"] = quant.detach().numpy().copy()
    fx["vqema/gq"] = gq.numpy().copy()
    fx["vqema/gz"] = zz.grad.numpy().copy()
    for n_, p_ in q3.named_parameters():
        if p_.grad is not None:
            fx[f"vqema/grad/{n_}"] = p_.grad.numpy().copy()
    fx.update(sd_np(q3, "vqema/w1/"))
    np.savez_compressed(os.path.join(OUT, "vq_layers.npz"), **fx)
    print("[vq_layers] ema perp:", fx["ema/c1/perplexity"], fx["ema/c2/perplexity"],
          "min gap:", fx["ema/c1/gap"].min(), fx["ema/c2/gap"].min())


def gen_vq_gssoft(vq):
    """VQ_Payam_GSSoft (:1304-1438), the quantiser the reference's Autoencoder_VQVAE actually ships with (:816-820):
    soft assignment probabilities, q = probs @ W, both latent losses; forward + every gradient, N=256, E=128, K=512."""
    torch.manual_seed(21)
    K, H, L, B = 512, 64, 2, 256
    E = H * L
    fx = {}
    q = vq.VQ_Payam_GSSoft(K, E, 0.25)
    q.train(True)
    fx.update(sd_np(q, "w0/"))
    for i, scale in enumerate((1.0, 0.3), 1):
        z = (torch.randn(L, B, H, generator=torch.Generator().manual_seed(40 + i)) * scale).requires_grad_(True)
        for p_ in q.parameters():
            p_.grad = None
        loss, quant, perp, probs = q(z)
        gq = torch.randn(quant.shape, generator=torch.Generator().manual_seed(400 + i))
        ((quant * gq).sum() + 3.0 * loss).backward()
        fx[f"c{i}/z"], fx[f"c{i}/gq"] = z.detach().numpy().copy(), gq.numpy().copy()
        fx[f"c{i}/loss"] = np.float64(float(loss)); fx[f"c{i}/perplexity"] = np.float64(float(perp))
        fx[f"c{i}/quantized"] = quant.detach().numpy().copy()
        fx[f"c{i}/probs"] = probs.detach().numpy().copy()
        fx[f"c{i}/gz"] = z.grad.numpy().copy()
        for n_, p_ in q.named_parameters():
            if p_.grad is not None:
                fx[f"c{i}/grad/{n_}"] = p_.grad.numpy().copy()
            else:
                fx[f"c{i}/gradnone/{n_}"] = np.zeros(0, dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "vq_gssoft.npz"), **fx)
    print("[vq_gssoft] loss", fx["c1/loss"], fx["c2/loss"], "perp", fx["c1/perplexity"], fx["c2/perplexity"],
          "none:", [k for k in fx if "gradnone" in k])


def gen_custom_loss(ts):
    g = torch.Generator().manual_seed(5)
    fx = {}
    for tag, (B, T, D) in {"a": (8, 20, 40), "b": (6, 34, 135)}.items():
        out = torch.randn(B, T, D, generator=g, requires_grad=True)
        tgt = torch.randn(B, T, D, generator=g)
        a = make_args()
        v = ts.custom_loss(out, tgt, a)
        v.backward()
        fx[f"{tag}/output"], fx[f"{tag}/target"] = out.detach().numpy().copy(), tgt.numpy().copy()
        fx[f"{tag}/loss"] = np.float64(float(v))
        fx[f"{tag}/grad"] = out.grad.numpy().copy()
    fx["weights"] = np.array([5.0, 0.1, 0.5])
    np.savez_compressed(os.path.join(OUT, "custom_loss.npz"), **fx)
    print("[custom_loss]", fx["a/loss"], fx["b/loss"])


def gen_dae(dae, ts):
    """DAE_Network (:22) forward (train w/ recorded mask, eval) + 2 train_iter_DAE steps (a17)."""
    torch.manual_seed(1235)
    net = dae.DAE_Network(135, 40)
    fx = dict(sd_np(net, "w0/"))
    g = torch.Generator().manual_seed(11)
    x = torch.randn(64, 135, 1, generator=g)
    fx["x"] = x.numpy().copy()
    args = make_args(autoencoder_vq="False", autoencoder_vae="False")
    optim = torch.optim.Adam(net.parameters(), lr=1e-3, betas=(0.5, 0.999))
    net.train(True)
    for step in (1, 2):
        torch.manual_seed(50 + step)
        with MaskRecorder() as rec:
            loss = ts.train_iter_DAE(args, 1, x, x, net, optim)
        fx[f"s{step}/mask"] = rec.masks[0]
        fx[f"s{step}/loss"] = np.float64(loss["loss"])
        if step == 1:
            for n_, p_ in net.named_parameters():
                fx[f"s1/grad/{n_}"] = p_.grad.numpy().copy()
    fx.update(sd_np(net, "wN/"))
    net.train(False)
    with torch.no_grad():
        out, lat = net(x, get_latent=True)
        enc_only = net.encoder(x.squeeze())  # the dataset's stacking call (lmdb_data_loader.py:649-653)
    fx["eval/out"], fx["eval/latent"], fx["eval/enc_only"] = out.numpy().copy(), lat.numpy().copy(), enc_only.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "dae.npz"), **fx)
    print("[dae] losses", fx["s1/loss"], fx["s2/loss"])


def main():
    vq, dae, ts = _import_reference()
    torch.set_num_threads(1)  # deterministic summation order for the golden numbers
    if len(sys.argv) > 1 and sys.argv[1] == "gssoft":      # regenerate only vq_gssoft.npz
        gen_vq_gssoft(vq)
        return
    # 1. tiny (BASELINE configs[0]): B=32,T=34,D=135,H=64,L=2,K=64, p=0
    gen_vqvae_train(vq, ts, "vqvae_tiny", make_args(), B=32, seed=1, n_steps=3)
    # 2. native-lite with every dropout active: B=8,T=20,D=40,H=50,L=2,K=512, p=0.2
    gen_vqvae_train(vq, ts, "vqvae_lite_dropout",
                    make_args(rep_learning_dim=40, hidden_size=50, dropout_prob=0.2,
                              autoencoder_vq_components=512, n_poses=20),
                    B=8, seed=2, n_steps=2)
    gen_vq_layers(vq)
    gen_vq_gssoft(vq)
    gen_custom_loss(ts)
    gen_dae(dae, ts)


if __name__ == "__main__":
    main()
