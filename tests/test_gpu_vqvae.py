"""GPU parity of the whole chunk VQ-VAE path (module surface + fused train step) against
(a) the golden vectors captured from the reference import and (b) the CPU oracle on seeded inputs.
Run on the MI355X box:  python -m pytest tests -m gpu -q"""
import argparse
import os

import numpy as np
import pytest
import torch

from oracle import g2v_oracle as O
from _f64 import as64, default64

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


def check_code_indices(got_idx, ref_idx, dist, gap_rel_floor=None, gap=None):
    """The north star's "bit-exact argmin" as every test here applies it: rows whose reference top-2 distance gap is above
    fp32 rounding noise (1e-4 on distances of order 100) must carry EXACTLY the reference's code; the remaining (in-band)
    rows must still pick a code whose reference distance is within 1e-3 of that row's minimum.  `dist` (N,K) are the
    reference-side distances (fixture / oracle)."""
    got = np.asarray(got_idx).reshape(-1)
    ref = np.asarray(ref_idx).reshape(-1)
    d = torch.as_tensor(dist)
    if gap is None:
        top2 = torch.topk(d, 2, dim=1, largest=False).values
        gap = (top2[:, 1] - top2[:, 0]).numpy()
        floor = 1e-4 * np.maximum(1.0, np.abs(top2[:, 0].numpy())) if gap_rel_floor else 1e-4
    else:
        floor = 1e-4
    safe = np.asarray(gap) > floor
    assert np.array_equal(got[safe], ref[safe]), "argmin code indices differ on rows outside the rounding band"
    assert got.min() >= 0 and got.max() < d.shape[1]
    chosen = d[torch.arange(d.shape[0]), torch.from_numpy(got.astype(np.int64))]
    assert float((chosen - d.min(1).values).max()) <= 1e-3, "an in-band row picked a code that is not within 1e-3 of the minimum"
    return safe


WORST = []      # (shape, step, tensor, L2, max) of the large-batch gradient comparisons: printed by the last test of the module


def relerr_l2(got, ref):
    got = got.detach().cpu().double().reshape(-1)
    ref = torch.as_tensor(ref).double().reshape(-1)
    return float((got - ref).norm()) / max(float(ref.norm()), 1e-30)


def sync_engine_from_oracle(eng, sd, adam, step_count):
    """Put the engine into EXACTLY the oracle's state (weights, Adam moments / step count, EMA codebook state, BatchNorm running
    statistics).  Multi-step comparisons re-synchronise between steps: Adam moves an element whose gradient sits at the
    rounding-noise floor by ~lr in either direction, and one step later those 1e-3-relative weight differences are 1e-3-relative
    gradient differences that say nothing about the kernels."""
    for name, _ in eng.layout:
        eng.view(name).copy_(sd[name])
        off, n, shp = eng.offsets[name]
        st = adam.get(name)
        if st is not None:
            eng.m[off:off + n].copy_(st["m"].reshape(-1))
            eng.v[off:off + n].copy_(st["v"].reshape(-1))
    eng.step_counter.fill_(step_count)
    eng.codebook.copy_(sd["vq_layer._embedding.weight"]); eng.ema_w.copy_(sd["vq_layer._ema_w"])
    eng.ema_cs.copy_(sd["vq_layer._ema_cluster_size"])
    eng.bn_rm.copy_(sd["decoder.decoder.pre_linear.1.running_mean"]); eng.bn_rv.copy_(sd["decoder.decoder.pre_linear.1.running_var"])


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
    codebook_before = torch.from_numpy(fx["w0/vq_layer._embedding.weight"].copy())
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
        # code indices: bit-exact wherever the reference's own top-2 gap is above fp32 rounding noise, within 1e-3 of the
        # minimum elsewhere (distances rebuilt from the reference's own flat_input and its pre-update codebook)
        if f"s{step}/flat_input" in fx.files:
            d_ref = O.vq_distances(torch.from_numpy(fx[f"s{step}/flat_input"].copy()), codebook_before)
            check_code_indices(b["idx"].cpu().numpy(), fx[f"s{step}/idx"], d_ref, gap=fx[f"s{step}/gap"])
        else:
            safe = fx[f"s{step}/gap"] > 1e-4
            assert np.array_equal(b["idx"].cpu().numpy()[safe], fx[f"s{step}/idx"][safe]), "code indices differ"
        codebook_before = torch.from_numpy(fx[f"s{step}/codebook_after"].copy())
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


def test_train_steps_match_reference_golden_h200(golden_dir):
    """The module surface at the width the reference SHIPS (hidden_size = 200, E = 400, K = 512, dropout 0.2: config/VQ-VAE.yml)
    against two iterations of the REFERENCE's own train_iter_Autoencoder_VQ_seq2seq (tests/golden/make_fixtures_h200.py; SURVEY.md
    8(c)'s cut-down native fixture): the H = 200 kernels (generic GRU / decoder step kernels at this batch, the E = 400 quantiser)
    against the reference itself, not only against the oracle."""
    import _h200
    from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
    from gesture2vec_amd.train_eval.train_seq2seq import FusedClipAdam, train_iter_Autoencoder_VQ_seq2seq
    fx, sd0 = _h200.load(golden_dir)
    (B, T, D, H, L, K, n_steps), args = fixture_args(fx)
    net = Autoencoder_VQVAE(args, D, T)
    net.load_state_dict(sd0, strict=True)
    net = net.to(DEV)
    net.train(True)
    optim = FusedClipAdam(net, lr=args.learning_rate, betas=(0.5, 0.999))
    x = torch.from_numpy(fx["x"].copy()).to(DEV)
    codebook_before = sd0["vq_layer._embedding.weight"].clone()
    for step in range(1, n_steps + 1):
        m = _h200.masks(fx, step, B, T, D, H)
        net.set_dropout_masks(m["dec"].to(DEV), m["in"].to(DEV), m["dec_l0"].to(DEV))
        loss, perp = train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
        eng = net.engine()
        b = eng.buffers(B)
        d_ref = O.vq_distances(torch.from_numpy(fx[f"s{step}/flat_input"].copy()), codebook_before)
        check_code_indices(b["idx"].cpu().numpy(), fx[f"s{step}/idx"], d_ref, gap=fx[f"s{step}/gap"])
        assert abs(loss["loss"] - float(fx[f"s{step}/loss"])) <= 1e-5 * abs(float(fx[f"s{step}/loss"]))
        assert abs(float(perp) - float(fx[f"s{step}/perplexity"])) <= 1e-4 * float(fx[f"s{step}/perplexity"])
        assert relerr(net.vq_layer._ema_cluster_size, fx[f"s{step}/ema_cluster_size"]) < 1e-5
        rows = torch.from_numpy(fx[f"s{step}/rows"])
        assert relerr(net.vq_layer._ema_w.detach().cpu()[rows], fx[f"s{step}/ema_w_rows"]) < 1e-5
        assert relerr(net.vq_layer._embedding.weight.detach().cpu()[rows], fx[f"s{step}/codebook_after_rows"]) < 1e-4
        cn = float(net.vq_layer._embedding.weight.detach().double().norm())
        assert abs(cn - float(fx[f"s{step}/codebook_after_norm"])) <= 1e-5 * cn
        codebook_before = net.vq_layer._embedding.weight.detach().cpu().clone()
        assert relerr(b["enc_hidden"], fx[f"s{step}/encoder_hidden"][:2]) < 1e-4
        assert relerr(b["quant"], fx[f"s{step}/quantized"]) < 1e-4
        assert relerr(b["y"].transpose(0, 1), fx[f"s{step}/outputs"]) < 1e-4, "reconstructed poses"
        if step == 1:
            n_checked = 0
            for k in fx.files:
                if not k.startswith("s1/grad_norm/"):
                    continue
                n = k[len("s1/grad_norm/"):]
                assert n in eng.offsets, f"{n} has a gradient in the reference but is not trainable here"
                g = eng.view(n, True)
                if n == "decoder.decoder.pre_linear.0.bias":
                    assert float(g.abs().max()) < 1e-6 and float(fx[k]) < 1e-5       # mathematically zero (feeds BatchNorm)
                elif float(fx[k]) == 0.0:
                    assert float(g.abs().max()) == 0.0, n                               # encoder layer 1: dead compute, exactly zero
                else:
                    _h200.check_sampled(g, fx[k], fx["s1/grad_sample/" + n], 5e-4, 1e-9, n)
                n_checked += 1
            assert n_checked >= 20
    after = net.state_dict()
    for k in fx.files:
        if k.startswith("wN_norm/"):
            n = k[len("wN_norm/"):]
            if n in ("decoder.decoder.pre_linear.0.bias", "decoder.decoder.pre_linear.1.running_mean"):
                continue            # Adam turns the pre-BatchNorm bias gradient's rounding noise into +-lr steps
            _h200.check_sampled(after[n], fx[k], fx["wN_sample/" + n], 1e-4, 1e-4 if n.startswith("vq_layer._e") else 2e-6, n)
        if k.startswith("wN_int/"):
            assert np.array_equal(after[k[len("wN_int/"):]].cpu().numpy(), fx[k]), k


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


@pytest.mark.parametrize("B,T,D,H,K,p", [(4096, 34, 135, 64, 512, 0.0), (1024, 34, 135, 64, 512, 0.2), (4096, 10, 45, 200, 400, 0.0)])
def test_deferred_slab_reductions_leave_the_step_bitwise_unchanged(B, T, D, H, K, p):
    """Round 6: the backward tail's slab reductions run as one launch per branch (VQVAEEngine.defer_reduce, include/g2v.h:
    g2v_linear_bwd_weight_deferred / _reduce) instead of one behind every product: same products, same slabs, same summation order
    -- every gradient, the loss and the post-step weights are BITWISE those of the immediate calls."""
    sd = O.init_vqvae_state(D, H, 2, K, seed=3)
    x = torch.randn(B, T, D, generator=torch.Generator().manual_seed(5)).to(DEV)
    g = torch.Generator().manual_seed(6)
    keep95 = (torch.rand(T - 1, B, D, generator=g) < 0.05).to(torch.uint8).to(DEV)
    m_in = (torch.rand(T, B, D, generator=g) < 1 - p).to(torch.uint8).to(DEV) if p > 0 else None
    m_l0 = (torch.rand(T - 1, B, H, generator=g) < 1 - p).to(torch.uint8).to(DEV) if p > 0 else None
    res = []
    for defer in (True, False):
        eng = _engine_from_state(sd, D, H, K, T, p)
        eng.defer_reduce = defer
        for _ in range(2):
            eng.set_masks(B, keep95, m_in, m_l0)
            eng.train_step(x, x, lr=5e-4, w_l1=5.0, w_cont=0.1, w_var=0.5, draw_masks=False)
        torch.cuda.synchronize()
        res.append((eng.gflat.clone(), eng.flat.clone(), eng.loss_terms.clone(), eng.gnorm.clone()))
    for a, b_ in zip(res[0], res[1]):
        assert torch.equal(a, b_)
    assert float(res[0][0].abs().max()) > 0


@pytest.mark.parametrize("B,T,D,H,K,p", [(256, 34, 135, 64, 512, 0.0), (48, 20, 40, 200, 512, 0.2), (37, 10, 45, 200, 400, 0.0),
                                         (128, 20, 40, 200, 512, 0.2),   # config/VQ-VAE.yml AS SHIPPED (the cluster kernels at their own batch size:
                                                                         # the shape bench.py's `shipped_config` times)
                                         (128, 10, 45, 200, 400, 0.0),   # config/VQ-VAE_GENEA.yml dims at the reference's batch size
                                         (1040, 4, 40, 48, 64, 0.1),     # generic dims ABOVE the small-batch split thresholds
                                         (4096, 34, 135, 64, 512, 0.0),  # BASELINE configs[1] = the shape bench.py times
                                         (4096, 34, 135, 64, 512, 0.2),  # ... and with the yml's dropout_prob
                                         (4096, 20, 40, 200, 512, 0.2),  # config/VQ-VAE.yml dims at the bench batch (8-wave generic step kernels)
                                         (4096, 10, 45, 200, 400, 0.0),  # BASELINE configs[4] (config/VQ-VAE_GENEA.yml shape)
                                         (8192, 8, 135, 64, 512, 0.0),   # two row tiles per workgroup in the persistent rollouts
                                         (4100, 6, 135, 64, 512, 0.0)])  # ... and a ragged last tile
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
    big = B >= 4096          # large batch: float64 oracle (tests/_f64.py), same formulas
    if big:
        sd, x = as64(sd), x.double()
    for step in range(2):
        masks = {"dec": (torch.rand(T - 1, B, D, generator=g) < 0.05).to(torch.uint8)}
        if p > 0:
            masks["in"] = (torch.rand(T, B, D, generator=g) < 1 - p).to(torch.uint8)
            masks["enc_l0"] = torch.ones(T, B, 2 * H, dtype=torch.uint8)   # layer 1 is dead compute: any mask works
            masks["dec_l0"] = (torch.rand(T - 1, B, H, generator=g) < 1 - p).to(torch.uint8)
        eng.set_masks(B, masks["dec"].to(DEV), masks["in"].to(DEV) if p > 0 else None,
                      masks["dec_l0"].to(DEV) if p > 0 else None)
        eng.train_step(xd, xd, lr=5e-4, w_l1=5.0, w_cont=0.1, w_var=0.5, draw_masks=False)
        b = eng.buffers(B)
        if big:
            # Large batch: the float64 oracle is told the DISCRETE decisions the kernels took -- code indices, the decoder's ReLU
            # pattern, the signs of custom_loss's |.| terms (oracle/g2v_oracle.py: cfg["forced"]).  A decision whose argument is
            # inside fp32 rounding of its threshold may legitimately fall the other way in float64, and ONE flipped ReLU / sign
            # moves single gradient elements by 1e-3..1e-2 of a tensor's maximum: until round 4 this test carried tolerances wide
            # enough for such flips (L2 1e-3, max 5e-3), i.e. wide enough for a real kernel error of that size.  With the
            # decisions pinned both sides evaluate the same smooth function; what is left is fp32 rounding.  The decisions
            # themselves are checked separately: the indices against the oracle's own argmin (check_code_indices below), the
            # signs against the oracle's own values wherever those are outside a rounding band; a wrong ReLU pattern would show
            # in the reconstructed poses (1e-4 below), which the pinned oracle computes with that pattern.
            y_e = b["y"].transpose(0, 1)
            forced = {"idx": b["idx"].cpu(), "relu": (b["a"] > 0).cpu().double(),
                      "sign_l1": torch.sign(y_e - xd).cpu().double(),
                      "sign_cont": torch.sign(y_e[:, 1:] - y_e[:, :-1]).cpu().double()}
            with default64():
                r = O.vqvae_train_step(sd, adam, x, masks, dict(cfg, forced=forced))
            # the pinned decisions are the oracle's own wherever the oracle's argument is clear of the threshold
            ro = r["outputs"]
            clear = (ro - x).abs() > 1e-5 * (1 + ro.abs())
            assert torch.equal(torch.sign(ro - x)[clear], forced["sign_l1"][clear]), "sign(y - target) outside the rounding band"
            dc = ro[:, 1:] - ro[:, :-1]
            clear = dc.abs() > 1e-5 * (1 + ro[:, 1:].abs())
            assert torch.equal(torch.sign(dc)[clear], forced["sign_cont"][clear]), "sign(y_t - y_{t-1}) outside the rounding band"
        else:
            r = O.vqvae_train_step(sd, adam, x, masks, cfg)
        ref_idx = torch.argmin(torch.as_tensor(r["dist"]), dim=1).numpy()       # (the oracle's OWN argmin, also when idx was pinned)
        safe = check_code_indices(b["idx"].cpu().numpy(), ref_idx, r["dist"], gap_rel_floor=True)
        # rows whose top-2 gap is inside the rounding band (1e-4 relative: 0.01 on distances of order 100): measured <= 3 rows at
        # B <= 1040 and 0.2 .. 0.6 % of the rows at B >= 4096 on these seeds (round 5; the bound used to be 5 %)
        assert (~safe).sum() <= max(3, B // 100), int((~safe).sum())
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
            elif big:
                # (decisions pinned, see above: fp32 rounding of sums over 1e5..1e6 rows is what is left)
                l2, mx = relerr_l2(eng.view(name, True), ref), relerr(eng.view(name, True), ref)
                WORST.append((B, T, H, K, p, step, name, l2, mx))
                # measured on all eight large shapes, both steps: <= 4.0e-6 / 4.6e-6 (profiles/r05_i_large_batch_grad_errors.txt)
                assert l2 < 2e-5, (name, l2)
                assert mx < 5e-5, (name, mx)
            else:
                assert relerr(eng.view(name, True), ref) < 5e-4, (name, relerr(eng.view(name, True), ref))
        # post-step weights: Adam normalises each element's gradient by its own magnitude, so elements whose gradient sits at
        # the rounding-noise floor can move differently by a fraction of the step (lr): allow 2 % of it (10 % at B = 4096,
        # where batch-mean gradients are ~sqrt(16) smaller against the same summation noise; gradients are held above)
        for name, _ in eng.layout:
            if name == "decoder.decoder.pre_linear.0.bias":
                continue
            err = float((eng.view(name).cpu().double() - sd[name].double()).abs().max())
            if big:
                # the first Adam steps are ~lr * sign(g): an element whose gradient is inside the rounding band (see the
                # gradient check above) may take the opposite sign, i.e. end up to 2 lr away -- but only a handful may
                n_off = int(((eng.view(name).cpu().double() - sd[name].double()).abs() > 0.1 * 5e-4).sum())
                assert err <= 2.1 * 5e-4 and n_off <= max(2, eng.view(name).numel() // 500), (name, err, n_off)
            else:
                assert err <= 1e-4 * float(sd[name].abs().max()) + 0.02 * 5e-4, (name, err)
        assert relerr(eng.codebook, sd["vq_layer._embedding.weight"]) < 1e-4
        sync_engine_from_oracle(eng, sd, adam, step + 1)          # the next step starts from identical states


def test_report_large_batch_gradient_errors():
    """not a check: prints the worst pinned-decision gradient errors of the large-batch cases above (pytest -s / -rP)"""
    for rec in sorted(WORST, key=lambda r: -r[7])[:8]:
        print("worst L2", rec)
    for rec in sorted(WORST, key=lambda r: -r[8])[:8]:
        print("worst max", rec)


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


def test_fixed_weight_freezes_decoder_gru():
    """autoencoder_fixed_weight == "True" (reference :483-486): the decoder GRU has requires_grad False, so the reference's
    clip_grad_norm_ and Adam skip it.  Here its gradient slots are zeroed before the fused clip+Adam: the tensors must not
    move by a single bit over several steps, and the clip norm must be the norm of the OTHER gradients only."""
    from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
    from gesture2vec_amd.train_eval.train_seq2seq import FusedClipAdam, train_iter_Autoencoder_VQ_seq2seq
    B, T, D = 64, 34, 135
    x = torch.randn(B, T, D, generator=torch.Generator().manual_seed(3)).to(DEV)
    norms = {}
    for fixed in ("False", "True"):
        torch.manual_seed(0)
        args = make_args(autoencoder_fixed_weight=fixed, autoencoder_vq_components=512)
        net = Autoencoder_VQVAE(args, D, T).to(DEV)
        net.train(True)
        net.rng_seed = 5
        optim = FusedClipAdam(net, lr=5e-4)
        eng = net.engine()
        before = {n: p.detach().clone() for n, p in net.named_parameters()}
        for step in range(3):
            train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
            if step == 0:
                norms[fixed] = eng.gnorm.item()
                if fixed == "False":
                    norms["gru_part"] = float(sum(float((eng.view(n, True).double() ** 2).sum())
                                                  for n, _ in eng.layout if n.startswith("decoder.decoder.gru.")) ** 0.5)
        moved = {n: not torch.equal(p.detach(), before[n]) for n, p in net.named_parameters()}
        for n, m in moved.items():
            if n.startswith("decoder.decoder.gru."):
                assert m == (fixed == "False"), (fixed, n, m)
        assert moved["decoder.decoder.out_layer.weight"] and moved["encoder.in_layer.weight"]
    expect = (norms["False"] ** 2 - norms["gru_part"] ** 2) ** 0.5
    assert abs(norms["True"] - expect) <= 1e-4 * expect, (norms, expect)
