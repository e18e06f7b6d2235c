"""GPU parity of the thin models against the reference's golden vectors: DAE_Network + train_iter_DAE (a17),
VQ_Payam_EMA standalone, VQ_Payam (non-EMA) and VectorQuantizerEMA (a5, a7, a8)."""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def relerr(got, ref):
    got = got.detach().cpu().double().reshape(-1)
    ref = torch.as_tensor(ref).detach().cpu().double().reshape(-1)
    return float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-12)


def state_from(fx, prefix):
    return {k[len(prefix):]: torch.from_numpy(fx[k].copy()) for k in fx.files if k.startswith(prefix)}


def test_dae_matches_reference_golden(golden_dir):
    from gesture2vec_amd.flat import FlatClipAdam
    from gesture2vec_amd.model.DAE_model import DAE_Network
    from gesture2vec_amd.train_eval.train_seq2seq import train_iter_DAE
    fx = np.load(os.path.join(golden_dir, "dae.npz"))
    net = DAE_Network(135, 40)
    net.load_state_dict(state_from(fx, "w0/"), strict=True)
    net = net.to(DEV)
    net.train(True)
    optim = FlatClipAdam(net.parameters(), lr=1e-3, betas=(0.5, 0.999))
    args = argparse.Namespace(autoencoder_vq="False", autoencoder_vae="False")
    x = torch.from_numpy(fx["x"].copy()).to(DEV)
    for step in (1, 2):
        net.set_dropout_mask(torch.from_numpy(fx[f"s{step}/mask"].copy()).to(DEV))
        loss = train_iter_DAE(args, 1, x, x, net, optim)
        assert abs(loss["loss"] - float(fx[f"s{step}/loss"])) <= 2e-6 * float(fx[f"s{step}/loss"])
        if step == 1:
            for n, p in net.named_parameters():
                assert relerr(p.grad, fx[f"s1/grad/{n}"]) < 1e-4, n
    for n, v in net.state_dict().items():
        assert relerr(v, fx["wN/" + n]) < 1e-5, n
    net.train(False)
    with torch.no_grad():
        out, lat = net(x, get_latent=True)
        enc_only = net.encode(x.squeeze())
    assert relerr(out, fx["eval/out"]) < 1e-5 and relerr(lat, fx["eval/latent"]) < 1e-5
    assert relerr(enc_only, fx["eval/enc_only"]) < 1e-5
    assert DAE_Network(135, -1)(x) is x            # ablation sentinel: identity (:52-55)


def test_vq_payam_ema_module_matches_reference_golden(golden_dir):
    from gesture2vec_amd.model.Autoencoder_VQVAE_model import VQ_Payam_EMA
    fx = np.load(os.path.join(golden_dir, "vq_layers.npz"))
    q = VQ_Payam_EMA(512, 128, 0.25, 0.85)
    q.load_state_dict(state_from(fx, "ema/w0/"), strict=True)
    q = q.to(DEV)
    q.train(True)
    for i, zk in ((1, "z1"), (2, "z2")):
        z = torch.from_numpy(fx[zk].copy()).to(DEV).requires_grad_(True)
        loss, quant, perp, enc = q(z)
        safe = fx[f"ema/c{i}/gap"] > 1e-4
        assert np.array_equal(enc.argmax(1).cpu().numpy()[safe], fx[f"ema/c{i}/idx"][safe])
        assert enc.shape == (256, 512) and float(enc.sum()) == 256
        assert abs(float(loss) - float(fx[f"ema/c{i}/loss"])) <= 1e-5 * float(fx[f"ema/c{i}/loss"])
        assert abs(float(perp) - float(fx[f"ema/c{i}/perplexity"])) <= 1e-4 * float(fx[f"ema/c{i}/perplexity"])
        assert relerr(quant, fx[f"ema/c{i}/quantized"]) < 1e-5
        gq = torch.from_numpy(fx[f"ema/c{i}/gq"].copy()).to(DEV)
        (g1,) = torch.autograd.grad((quant * gq).sum(), z, retain_graph=True)
        (g2,) = torch.autograd.grad(loss, z)
        assert relerr(g1, fx[f"ema/c{i}/gz_from_q"]) < 1e-6
        assert relerr(g2, fx[f"ema/c{i}/gz_from_loss"]) < 1e-5
        sd = q.state_dict()
        for k in ("_ema_cluster_size", "_ema_w", "_embedding.weight"):
            assert relerr(sd[k], fx[f"ema/w{i}/{k}"]) < 2e-5, k
    q.train(False)
    with torch.no_grad():
        loss, quant, perp, enc = q(torch.from_numpy(fx["z1"].copy()).to(DEV))
    assert np.array_equal(enc.argmax(1).cpu().numpy(), fx["ema/eval/idx"])
    assert relerr(quant, fx["ema/eval/quantized"]) < 1e-5
    assert np.array_equal(q.assign(torch.from_numpy(fx["z1"].copy()).to(DEV)).cpu().numpy(), fx["ema/eval/idx"])


def test_vq_payam_plain_matches_reference_golden(golden_dir):
    from gesture2vec_amd.model.Autoencoder_VQVAE_model import VQ_Payam
    fx = np.load(os.path.join(golden_dir, "vq_layers.npz"))
    q = VQ_Payam(512, 128, 0.25)
    q.load_state_dict(state_from(fx, "plain/w0/"), strict=True)
    q = q.to(DEV)
    z = torch.from_numpy(fx["z1"].copy()).to(DEV).requires_grad_(True)
    loss, quant, perp, enc = q(z)
    assert np.array_equal(enc.argmax(1).cpu().numpy(), fx["plain/idx"])
    assert abs(float(loss) - float(fx["plain/loss"])) <= 1e-5 * float(fx["plain/loss"])
    assert abs(float(perp) - float(fx["plain/perplexity"])) <= 1e-4 * float(fx["plain/perplexity"])
    gq = torch.from_numpy(fx["plain/gq"].copy()).to(DEV)
    ((quant * gq).sum() + loss).backward()
    assert relerr(z.grad, fx["plain/gz"]) < 1e-5
    assert relerr(q._embedding.weight.grad, fx["plain/g_embedding"]) < 1e-4


def test_vector_quantizer_ema_matches_reference_golden(golden_dir):
    from gesture2vec_amd.model.Autoencoder_VQVAE_model import VectorQuantizerEMA
    fx = np.load(os.path.join(golden_dir, "vq_layers.npz"))
    q = VectorQuantizerEMA(512, 128, 0.25, 0.85)
    q.load_state_dict(state_from(fx, "vqema/w0/"), strict=True)
    q = q.to(DEV)
    q.train(True)
    z = torch.from_numpy(fx["z1"].copy()).to(DEV).requires_grad_(True)
    loss, quant, perp, enc = q(z)
    assert quant.shape == (2, 256, 64)
    assert np.array_equal(enc.argmax(1).cpu().numpy(), fx["vqema/idx"])
    assert abs(float(loss) - float(fx["vqema/loss"])) <= 1e-5 * float(fx["vqema/loss"])
    assert relerr(quant, fx["vqema/quantized"]) < 1e-5
    gq = torch.from_numpy(fx["vqema/gq"].copy()).to(DEV)
    ((quant * gq).sum() + loss).backward()
    assert relerr(z.grad, fx["vqema/gz"]) < 1e-4
    assert relerr(q.pre_lin.weight.grad, fx["vqema/grad/pre_lin.weight"]) < 1e-4
    assert relerr(q.pre_lin.bias.grad, fx["vqema/grad/pre_lin.bias"]) < 1e-4
    sd = q.state_dict()
    for k in ("_ema_cluster_size", "_ema_w", "_embedding.weight"):
        assert relerr(sd[k], fx[f"vqema/w1/{k}"]) < 2e-5, k


def _vq_args(D, H, L, K):
    import argparse
    return argparse.Namespace(rep_learning_dim=D, hidden_size=H, n_layers=L, dropout_prob=0.0, autoencoder_vq="True",
                              autoencoder_vae="False", autoencoder_vq_components=K, autoencoder_vq_commitment_cost=0.25,
                              autoencoder_conditioned="True", autoencoder_att="False", autoencoder_fixed_weight="False",
                              n_pre_poses=1, n_poses=34)


def test_bulk_code_assignment_matches_oracle():
    """§8f-2: N chunks -> latents -> codes in one encoder pass + one assign launch; oracle: the reference's per-chunk
    route (encoder with batch 1, [l0;l1] row, argmin of distances on pre_linear(row))."""
    from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
    from gesture2vec_amd.pipeline import chunks_to_codes
    from oracle import g2v_oracle as O
    torch.manual_seed(11)
    D, H, L, K, T, N = 40, 64, 2, 128, 20, 67
    net = Autoencoder_VQVAE(_vq_args(D, H, L, K), D, T).to(DEV)
    net.train(False)
    chunks = torch.randn(N, T, D)
    lat, codes = chunks_to_codes(net, chunks.to(DEV))
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    _, hidden = O.encoder_forward(chunks.transpose(0, 1), sd, L, 0.0, None)
    lat_ref = hidden[:L].transpose(0, 1).reshape(N, L * H)
    assert relerr(lat, lat_ref) < 2e-5
    flat = O.linear(lat_ref, sd["vq_layer.pre_linear.weight"], sd["vq_layer.pre_linear.bias"])
    d = O.vq_distances(flat, sd["vq_layer._embedding.weight"])
    top2 = d.topk(2, dim=1, largest=False)
    clear = (top2.values[:, 1] - top2.values[:, 0]) > 1e-3 * top2.values[:, 1].abs()
    assert clear.sum() > N // 2
    assert torch.equal(codes.cpu()[clear], top2.indices[:, 0][clear])     # bit-exact wherever the decision is not a near-tie


def test_stacked_dae_vqvae_dae_matches_oracle():
    """Config 3: raw (B,T,135) -> DAE encoder -> VQ-VAE(D=40) -> DAE decoder, eval mode with replayed Dropout(0.95) masks."""
    from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
    from gesture2vec_amd.model.DAE_model import DAE_Network
    from gesture2vec_amd.pipeline import stacked_autoencode
    from oracle import g2v_oracle as O
    torch.manual_seed(12)
    B, T, DR, D, H, L, K = 24, 34, 135, 40, 64, 2, 64
    dae = DAE_Network(DR, D).to(DEV)
    dae.train(False)
    net = Autoencoder_VQVAE(_vq_args(D, H, L, K), D, T).to(DEV)
    net.train(False)
    x = torch.randn(B, T, DR)
    keep95 = (torch.rand(T - 1, B, D) < 0.05)                 # Dropout(0.95) keeps 5 %
    net.set_dropout_masks(keep95=keep95.to(DEV).to(torch.uint8))
    rec, lat_out, _ = stacked_autoencode(dae, net, x.to(DEV))
    dsd = {k: v.detach().cpu() for k, v in dae.state_dict().items()}
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    lat = torch.relu(O.linear(x.reshape(B * T, DR), dsd["encoder.0.weight"], dsd["encoder.0.bias"])).view(B, T, D)
    cfg = dict(n_layers=L, dropout_prob=0.0, n_pre_poses=1, commitment_cost=0.25, decay=0.85, epsilon=1e-5)
    fw = O.vqvae_forward(sd, lat, lat, cfg, False, {"dec": keep95})
    rec_ref = O.linear(fw["outputs"].reshape(B * T, D), dsd["decoder.0.weight"], dsd["decoder.0.bias"]).view(B, T, DR)
    assert relerr(lat_out, fw["outputs"]) < 1e-4
    assert relerr(rec, rec_ref) < 1e-4


def test_vq_gssoft_matches_reference_golden(golden_dir):
    """a8: VQ_Payam_GSSoft (:1304-1438) forward values + every gradient against the reference fixture."""
    from gesture2vec_amd.model.Autoencoder_VQVAE_model import VQ_Payam_GSSoft
    fx = np.load(os.path.join(golden_dir, "vq_gssoft.npz"))
    q = VQ_Payam_GSSoft(512, 128, 0.25)
    q.load_state_dict(state_from(fx, "w0/"), strict=True)
    q = q.to(DEV)
    q.train(True)
    for i in (1, 2):
        for p_ in q.parameters():
            p_.grad = None
        z = torch.from_numpy(fx[f"c{i}/z"].copy()).to(DEV).requires_grad_(True)
        loss, quant, perp, probs = q(z)
        gq = torch.from_numpy(fx[f"c{i}/gq"].copy()).to(DEV)
        ((quant * gq).sum() + 3.0 * loss).backward()
        assert abs(float(loss.detach()) - float(fx[f"c{i}/loss"])) <= 1e-5 * float(fx[f"c{i}/loss"])
        assert abs(float(perp) - float(fx[f"c{i}/perplexity"])) <= 1e-5 * float(fx[f"c{i}/perplexity"])
        assert relerr(probs, fx[f"c{i}/probs"]) < 5e-5
        assert relerr(quant, fx[f"c{i}/quantized"]) < 1e-5
        assert relerr(z.grad, fx[f"c{i}/gz"]) < 1e-4
        for k in fx.files:
            if k.startswith(f"c{i}/grad/"):
                n = k[len(f"c{i}/grad/"):]
                g = dict(q.named_parameters())[n].grad
                assert g is not None and relerr(g, fx[k]) < 3e-4, (n, relerr(g, fx[k]))
            if k.startswith(f"c{i}/gradnone/"):
                assert dict(q.named_parameters())[k[len(f"c{i}/gradnone/"):]].grad is None


@pytest.mark.parametrize("N,K,scale", [(4096, 512, 0.3), (100, 512, 1.0), (37, 1024, 0.1), (16, 128, 0.5)])
def test_fused_soft_quantiser_kernels_equal_the_separate_kernels(N, K, scale):
    """csrc/vq_soft.hip (one launch forward, one backward) against the sequence of separate launches it replaces (dense products,
    vq_soft_fwd / _bwd, rowscale_combine, mse, ste, vq_bwd): same element-wise expressions, K- and E-long sums in another order.
    N % 16 != 0 exercises the ragged row tile."""
    from gesture2vec_amd import ops
    E, beta, gs = 128, 0.25, 0.7
    assert ops.vq_soft_fused_ok(N, E, K) and not ops.vq_soft_fused_ok(N, 400, K) and not ops.vq_soft_fused_ok(N, E, 400)
    g = torch.Generator().manual_seed(N + K)
    r = lambda *s, m=1.0: (torch.randn(*s, generator=g) * m).to(DEV)
    x = r(N, E, m=scale)
    Wm, bm = r(E, E, m=E ** -0.5), r(E, m=0.1)
    Wl, bl = r(K, E, m=0.3 * E ** -0.5), r(K, m=0.1)
    W = r(K, E, m=scale)
    dh, gl = r(N, E, m=1e-3), torch.full((1,), 0.4, device=DEV)
    # ---- the separate kernels ---------------------------------------------------------------------------------------------------
    flat = ops.linear_fwd(x, Wm, bm)
    logvar = ops.linear_fwd(flat, Wl, bl)
    dots = ops.linear_fwd(flat, W, None)
    probs, dist, perp = ops.vq_soft_fwd(flat, dots, logvar, ops.vq_code_sqnorm(W))
    q = ops.linear_bwd_data(probs, W)
    mse, dq = ops.mse_fwd_bwd(q, x, True, gs)
    quant = ops.ste(x, q)
    gz = ops.vq_bwd(dh, gl, x, q, None, beta)
    dprobs = ops.linear_fwd(dq, W, None)
    dd, dlv, rowsum = ops.vq_soft_bwd(probs, dprobs, dist, logvar)
    dflat = ops.rowscale_combine(flat, rowsum, ops.linear_bwd_data(dd, W))
    ops.linear_bwd_data(dlv, Wl, out=dflat, accumulate=True)
    ops.linear_bwd_data(dflat, Wm, out=gz, accumulate=True)
    # ---- fused ------------------------------------------------------------------------------------------------------------------
    o = ops.vq_soft_fused_fwd(x, Wm, bm, Wl, bl, W, beta, gs)
    o2 = ops.vq_soft_fused_fwd(x, Wm, bm, Wl, bl, W, beta, gs)
    gz_f, dd_f, dlv_f, dflat_f = ops.vq_soft_fused_bwd(dh, gl, x, o, Wm, Wl, W, beta)
    gz_f2 = ops.vq_soft_fused_bwd(dh, gl, x, o2, Wm, Wl, W, beta)[0]
    for k in ("flat", "logvar", "dist", "probs", "q", "dq", "quant"):
        assert torch.equal(o[k], o2[k]), f"{k}: not reproducible"
    assert torch.equal(gz_f, gz_f2)
    for name, got, ref, tol in (("flat", o["flat"], flat, 1e-5), ("logvar", o["logvar"], logvar, 1e-5), ("dist", o["dist"], dist, 2e-5),
                                ("probs", o["probs"], probs, 5e-5), ("q", o["q"], q, 2e-5), ("dq", o["dq"], dq, 1e-4),
                                ("quant", o["quant"], quant, 2e-5), ("dd", dd_f, dd, 3e-4), ("dlogvar", dlv_f, dlv, 3e-4),
                                ("dflat", dflat_f, dflat, 3e-4), ("gz", gz_f, gz, 3e-4)):
        assert relerr(got, ref) < tol, (name, relerr(got, ref))
    assert abs(float(o["mse"]) - float(mse)) <= 1e-5 * float(mse)
    assert abs(float(o["loss_vq"]) - (1 + beta) * float(mse)) <= 1e-5 * float(mse) * (1 + beta)
    assert abs(float(o["perplexity"]) - float(perp)) <= 1e-5 * float(perp)


def test_bulk_assign_route_of_the_quantiser_module():
    """VQ_Payam_EMA.assign switches to g2v_vq_assign_bulk (bf16 split screening + exact re-check) from 2^17 rows: same indices
    as the fp32 kernel on the projected rows."""
    from gesture2vec_amd import ops
    from gesture2vec_amd.model.Autoencoder_VQVAE_model import VQ_Payam_EMA
    torch.manual_seed(3)
    q = VQ_Payam_EMA(512, 128, 0.25, 0.85).to(DEV)
    z = torch.randn(131072 + 5, 128, device=DEV) * 0.3
    idx = q.assign(z)
    flat = ops.linear_fwd(z, q.pre_linear.weight.data, q.pre_linear.bias.data)
    W = q._embedding.weight.data
    ref = ops.vq_assign(flat, None, W, ops.vq_code_sqnorm(W), want_quantized=False)[0]
    assert idx.dtype == torch.int64 and torch.equal(idx, ref)


def test_bulk_assign_route_of_the_quantiser_module_at_the_shipped_width():
    """E = 400 (hidden_size 200 x 2 layers: every YAML the reference ships, i.e. every real checkpoint): VQ_Payam_EMA.assign from 2^17
    rows against the ORACLE's distances (reference :1230-1259: pre_linear, ||x||^2 + ||w||^2 - 2 x w^T, argmin) -- exact indices on
    every row whose top-2 gap is above fp32 rounding noise, within 1e-3 of the minimum elsewhere -- and against the fp32 kernel on
    the projected rows (every row)."""
    from gesture2vec_amd import ops
    from gesture2vec_amd.model.Autoencoder_VQVAE_model import VQ_Payam_EMA
    from oracle import g2v_oracle as O
    torch.manual_seed(5)
    K, E, N = 512, 400, 131072 + 37
    q = VQ_Payam_EMA(K, E, 0.25, 0.85).to(DEV)
    z = torch.randn(N, E, device=DEV) * 0.3
    idx = q.assign(z)
    assert idx.dtype == torch.int64 and idx.shape == (N,)
    flat = ops.linear_fwd(z, q.pre_linear.weight.data, q.pre_linear.bias.data)
    W = q._embedding.weight.data
    ref_k = ops.vq_assign(flat, None, W, ops.vq_code_sqnorm(W), want_quantized=False)[0]
    assert torch.equal(idx, ref_k), "bulk route differs from the fp32 kernel"
    # the oracle on a slice that a CPU finishes in seconds: 8192 rows spread over the batch (incl. the ragged tail)
    rows = torch.cat([torch.arange(0, N, N // 8000)[:8000], torch.arange(N - 192, N)])
    flat_o = O.linear(z[rows].cpu(), q.pre_linear.weight.data.cpu(), q.pre_linear.bias.data.cpu())
    d = O.vq_distances(flat_o, W.cpu())
    top2 = torch.topk(d, 2, dim=1, largest=False).values
    safe = (top2[:, 1] - top2[:, 0]) > 1e-4 * torch.clamp(top2[:, 0].abs(), min=1.0)
    got = idx[rows.to(DEV)].cpu()
    assert torch.equal(got[safe], d.argmin(1)[safe]), "bulk route differs from the oracle outside the rounding band"
    assert float((d[torch.arange(len(rows)), got] - top2[:, 0]).max()) <= 1e-3
    assert int((~safe).sum()) <= len(rows) // 100
