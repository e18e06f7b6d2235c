"""GPU parity of the whole chunk VQ-VAE path (module surface + fused train step) against
(a) the golden vectors captured from the reference import and (b) the CPU oracle on seeded inputs.
Run on the MI355X box:  python -m pytest tests -m gpu -q"""
import argparse
import os

import numpy as np
import pytest
import torch

from oracle import g2v_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_args(**kw):
    d = dict(rep_learning_dim=135, hidden_size=64, n_layers=2, dropout_prob=0.0, autoencoder_vae="False",
             autoencoder_vq="True", autoencoder_vq_components=64, autoencoder_vq_commitment_cost=0.25, n_pre_poses=1,
             autoencoder_conditioned="True", autoencoder_att="False", autoencoder_fixed_weight="False", n_poses=34,
             loss_l1_weight=5.0, loss_cont_weight=0.1, loss_var_weight=0.5, learning_rate=5e-4, epochs=10)
    d.update(kw)
    return argparse.Namespace(**d)


def fixture_args(fx):
    B, T, D, H, L, K, n_steps = [int(v) for v in fx["cfg"]]
    beta, p, lr, w1, w2, w3 = [float(v) for v in fx["cfg_f"]]
    return (B, T, D, H, L, K, n_steps), make_args(rep_learning_dim=D, hidden_size=H, n_layers=L, dropout_prob=p,
                                                  autoencoder_vq_components=K, autoencoder_vq_commitment_cost=beta,
                                                  n_poses=T, learning_rate=lr, loss_l1_weight=w1, loss_cont_weight=w2,
                                                  loss_var_weight=w3)


def state_from(fx, prefix):
    return {k[len(prefix):]: torch.from_numpy(fx[k].copy()) for k in fx.files if k.startswith(prefix)}


def relerr(got, ref):
    got = got.detach().cpu().double().reshape(-1)
    ref = torch.as_tensor(ref).double().reshape(-1)
    return float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-12)


@pytest.mark.parametrize("name", ["vqvae_tiny", "vqvae_lite_dropout"])
def test_train_steps_match_reference_golden(golden_dir, name):
    """Drop-in surface: Autoencoder_VQVAE + train_iter_Autoencoder_VQ_seq2seq reproduce the REFERENCE's numbers
    (loss, perplexity, code indices, EMA codebook, outputs, gradients, post-Adam weights) with replayed masks."""
    from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
    from gesture2vec_amd.train_eval.train_seq2seq import FusedClipAdam, train_iter_Autoencoder_VQ_seq2seq

    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    (B, T, D, H, L, K, n_steps), args = fixture_args(fx)
    p = args.dropout_prob
    net = Autoencoder_VQVAE(args, D, T)
    net.load_state_dict(state_from(fx, "w0/"), strict=True)
    net = net.to(DEV)
    net.train(True)
    optim = FusedClipAdam(net, lr=args.learning_rate, betas=(0.5, 0.999))
    x = torch.from_numpy(fx["x"].copy()).to(DEV)
    for step in range(1, n_steps + 1):
        keep95 = O.unpack_mask(fx[f"s{step}/mask_dec"], (T - 1, B, D)).to(DEV)
        if p > 0:
            net.set_dropout_masks(keep95, torch.from_numpy(fx[f"s{step}/mask_in"].copy()).to(DEV),
                                  torch.from_numpy(fx[f"s{step}/mask_dec_l0"].copy()).to(DEV))
        else:
            net.set_dropout_masks(keep95)
        loss, perp = train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
        eng = net.engine()
        b = eng.buffers(B)
        # code indices: bit-exact wherever the reference's own top-2 gap is above fp32 rounding noise
        gap = fx[f"s{step}/gap"]
        safe = gap > 1e-4
        assert np.array_equal(b["idx"].cpu().numpy()[safe], fx[f"s{step}/idx"][safe]), "code indices differ"
        assert abs(loss["loss"] - float(fx[f"s{step}/loss"])) <= 1e-5 * abs(float(fx[f"s{step}/loss"]))
        assert abs(float(perp) - float(fx[f"s{step}/perplexity"])) <= 1e-4 * float(fx[f"s{step}/perplexity"])
        assert relerr(net.vq_layer._ema_cluster_size, fx[f"s{step}/ema_cluster_size"]) < 1e-5
        assert relerr(net.vq_layer._ema_w, fx[f"s{step}/ema_w"]) < 1e-5
        assert relerr(net.vq_layer._embedding.weight, fx[f"s{step}/codebook_after"]) < 1e-4
        if step in (1, n_steps):
            assert relerr(b["enc_hidden"], fx[f"s{step}/encoder_hidden"][:2]) < 1e-4
            assert relerr(b["quant"], fx[f"s{step}/quantized"]) < 1e-4
            assert relerr(b["y"].transpose(0, 1), fx[f"s{step}/outputs"]) < 1e-4, "reconstructed poses"
        if step == 1:
            for k in fx.files:
                if k.startswith("s1/grad/"):
                    n = k[len("s1/grad/"):]
                    ref = fx[k]
                    prm = net.get_parameter(n)
                    if n in eng.offsets:
                        g = eng.view(n, True)
                        scale = np.abs(ref).max()
                        if n == "decoder.decoder.pre_linear.0.bias":
                            # mathematically zero (feeds BatchNorm): both sides are rounding noise
                            assert float(g.abs().max()) < 1e-6 and scale < 1e-6
                        else:
                            assert relerr(g, ref) < 5e-4, (n, relerr(g, ref))
                    else:
                        raise AssertionError(f"{n} has a gradient in the reference but is not trainable here")
    fin = state_from(fx, "wN/")
    for n, ref in fin.items():
        got = net.state_dict()[n]
        if n in ("decoder.decoder.pre_linear.0.bias", "decoder.decoder.pre_linear.1.running_mean"):
            # Adam turns the (mathematically zero) pre-BN bias gradient's rounding noise into +-lr steps
            assert float((got.cpu() - ref).abs().max()) <= 1.01 * n_steps * args.learning_rate, n
        elif ref.dtype.is_floating_point:
            err = float((got.cpu().double() - ref.double()).abs().max())
            assert err <= 1e-4 * float(ref.abs().max()) + 2e-6, (n, err)
        else:
            assert torch.equal(got.cpu(), ref), n


@pytest.mark.parametrize("name", ["vqvae_tiny", "vqvae_lite_dropout"])
def test_eval_forward_matches_reference_golden(golden_dir, name):
    from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    ev = np.load(os.path.join(golden_dir, name + "_eval.npz"))
    (B, T, D, H, L, K, n_steps), args = fixture_args(fx)
    net = Autoencoder_VQVAE(args, D, T)
    net.load_state_dict(state_from(fx, "wN/"), strict=True)
    net = net.to(DEV)
    net.train(False)
    x = torch.from_numpy(fx["x"].copy()).to(DEV)
    net.set_dropout_masks(O.unpack_mask(ev["mask_dec"], (T - 1, B, D)).to(DEV))
    with torch.no_grad():
        outputs, first_hidden, loss_vq, perp = net(x, x)
    assert outputs.shape == (B, T, D) and first_hidden.shape == (L, B, H)
    assert relerr(outputs, ev["outputs"]) < 1e-4
    assert relerr(first_hidden, ev["first_hidden"]) < 1e-4
    assert abs(float(loss_vq) - float(ev["loss_vq"])) <= 1e-4 * float(ev["loss_vq"])
    assert abs(float(perp) - float(ev["perplexity"])) <= 1e-4 * float(ev["perplexity"])
    # eval mode must not move any state
    after = net.state_dict()
    for n, ref in state_from(fx, "wN/").items():
        assert torch.equal(after[n].cpu(), ref), n


def _engine_from_state(sd, D, H, K, T, p, beta=0.25):
    from gesture2vec_amd.engine import VQVAEEngine
    eng = VQVAEEngine(D, H, 2, K, T, beta=beta, dropout_prob=p, device=DEV)
    for name, _ in eng.layout:
        eng.view(name).copy_(sd[name])
    eng.vq_pre_w.copy_(sd["vq_layer.pre_linear.weight"]); eng.vq_pre_b.copy_(sd["vq_layer.pre_linear.bias"])
    eng.codebook.copy_(sd["vq_layer._embedding.weight"]); eng.ema_w.copy_(sd["vq_layer._ema_w"])
    eng.ema_cs.copy_(sd["vq_layer._ema_cluster_size"])
    return eng


@pytest.mark.parametrize("B,T,D,H,K,p", [(256, 34, 135, 64, 512, 0.0), (48, 20, 40, 200, 512, 0.2), (37, 10, 45, 200, 400, 0.0),
                                         (1040, 4, 40, 48, 64, 0.1)])     # last: generic dims ABOVE the small-batch split thresholds
def test_fused_train_step_vs_oracle(B, T, D, H, K, p):
    """Engine.train_step (the path bench.py times) against the CPU oracle on seeded inputs, two steps."""
    sd = O.init_vqvae_state(D, H, 2, K, seed=3)
    # spread the encoder states so that the quantiser sees a non-degenerate assignment problem
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, T, D, generator=g)
    cfg = dict(n_layers=2, dropout_prob=p, commitment_cost=0.25, n_pre_poses=1, conditioned=True, w_l1=5.0, w_cont=0.1,
               w_var=0.5, lr=5e-4)
    eng = _engine_from_state(sd, D, H, K, T, p)
    xd = x.to(DEV)
    adam = {}
    for step in range(2):
        masks = {"dec": (torch.rand(T - 1, B, D, generator=g) < 0.05).to(torch.uint8)}
        if p > 0:
            masks["in"] = (torch.rand(T, B, D, generator=g) < 1 - p).to(torch.uint8)
            masks["enc_l0"] = torch.ones(T, B, 2 * H, dtype=torch.uint8)   # layer 1 is dead compute: any mask works
            masks["dec_l0"] = (torch.rand(T - 1, B, H, generator=g) < 1 - p).to(torch.uint8)
        r = O.vqvae_train_step(sd, adam, x, masks, cfg)
        eng.set_masks(B, masks["dec"].to(DEV), masks["in"].to(DEV) if p > 0 else None,
                      masks["dec_l0"].to(DEV) if p > 0 else None)
        eng.train_step(xd, xd, lr=5e-4, w_l1=5.0, w_cont=0.1, w_var=0.5, draw_masks=False)
        b = eng.buffers(B)
        d = r["dist"]
        top2 = torch.topk(d, 2, dim=1, largest=False).values
        safe = ((top2[:, 1] - top2[:, 0]) > 1e-4 * top2[:, 0].abs().clamp(min=1)).numpy()
        assert safe.mean() > 0.95
        assert np.array_equal(b["idx"].cpu().numpy()[safe], r["idx"].numpy()[safe]), "argmin code indices"
        assert relerr(b["y"].transpose(0, 1), r["outputs"]) < 1e-4, "reconstructed poses"
        total = eng.loss_terms[0].item() + eng.vq_scalars[0].item() / 400
        assert abs(total - float(r["loss"])) <= 1e-5 * abs(float(r["loss"]))
        assert abs(eng.vq_scalars[1].item() - float(r["perplexity"])) <= 1e-4 * float(r["perplexity"])
        assert abs(eng.gnorm.item() - float(r["grad_norm"])) <= 2e-4 * float(r["grad_norm"])
        for name, _ in eng.layout:
            if name == "decoder.decoder.pre_linear.0.bias":
                continue
            ref = r["grads"][name]
            if float(ref.abs().max()) == 0.0:
                assert float(eng.view(name, True).abs().max()) == 0.0, name     # encoder layer 1: exactly zero
            else:
                assert relerr(eng.view(name, True), ref) < 5e-4, (name, relerr(eng.view(name, True), ref))
    for name, _ in eng.layout:
        if name == "decoder.decoder.pre_linear.0.bias":
            continue
        # Adam normalises each element's gradient by its own magnitude, so elements whose gradient sits at the
        # rounding-noise floor can move differently by a fraction of the total travel (2 steps x lr): allow 2 % of it.
        err = float((eng.view(name).cpu().double() - sd[name].double()).abs().max())
        assert err <= 1e-4 * float(sd[name].abs().max()) + 0.02 * 2 * 5e-4, (name, err)
    assert relerr(eng.codebook, sd["vq_layer._embedding.weight"]) < 1e-4


def test_full_size_properties():
    """BASELINE configs[1] size (B=4096,T=34,D=135,H=64,K=512): size-independent properties."""
    B, T, D, H, K = 4096, 34, 135, 64, 512
    sd = O.init_vqvae_state(D, H, 2, K, seed=1)
    x = torch.randn(B, T, D, generator=torch.Generator().manual_seed(1234)).to(DEV)
    outs = []
    for rep in range(2):
        eng = _engine_from_state(sd, D, H, K, T, 0.0)
        eng.rng_counter.zero_()
        for _ in range(2):
            eng.train_step(x, x, lr=5e-4, w_l1=5.0, w_cont=0.1, w_var=0.5)
        b = eng.buffers(B)
        outs.append((b["idx"].clone(), b["y"].clone(), eng.flat.clone(), eng.codebook.clone(), eng.vq_stats.clone(),
                     eng.vq_scalars.clone()))
    # bitwise reproducible run-to-run (no float atomics anywhere on the path)
    for a, c in zip(outs[0], outs[1]):
        assert torch.equal(a, c)
    idx, y, flat, cb, stats, sc = outs[0]
    assert int(idx.min()) >= 0 and int(idx.max()) < K
    assert torch.isfinite(y).all() and torch.isfinite(flat).all() and torch.isfinite(cb).all()
    cnt = torch.bincount(idx, minlength=K).float()
    assert torch.equal(stats[:K], cnt) and float(cnt.sum()) == B          # checksum of the assignment histogram
    p = cnt / B
    perp = torch.exp(-(p * torch.log(p + 1e-10)).sum())
    assert abs(sc[1].item() - perp.item()) <= 1e-4 * perp.item()
    # y[0] is the target's first frame, verbatim
    assert torch.equal(y[0], x[:, 0, :])
