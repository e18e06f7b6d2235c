"""GPU parity of Part d (text -> gesture-code seq2seq) against the reference's golden vectors (2 train steps + eval)."""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def relerr(got, ref):
    got = got.detach().cpu().double().reshape(-1)
    ref = torch.as_tensor(ref).detach().double().reshape(-1)
    return float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-12)


@pytest.mark.parametrize("kernels", ["per_operator", "fused_step", "cluster_forward"])
@pytest.mark.parametrize("name,att", [("t2e_noatt", "False"), ("t2e_att", "True")])
def test_text2embedding_matches_reference_golden(golden_dir, name, att, kernels, monkeypatch, request):
    """kernels = "fused_step": the decode steps run as g2v_attn_code_rollout_fwd / _bwd (csrc/t2e_rollout.hip; selected from 1024
    rows per batch in production, forced here) -- the fused per-step kernels against the REFERENCE's own numbers."""
    from gesture2vec_amd import rollout_t2e
    from gesture2vec_amd.flat import FlatClipAdam
    monkeypatch.setattr(rollout_t2e, "FUSED_MIN_ROWS", 1 if kernels == "fused_step" else 1 << 30)
    # "cluster_forward" (round 5, no attention only): the forward rollout as one persistent cluster launch + the per-operator backward
    monkeypatch.setattr(rollout_t2e, "CLUSTER_FORWARD", kernels == "cluster_forward")
    if kernels == "cluster_forward" and att == "True":
        pytest.skip("the cluster forward serves the attention-free decoder")
    from gesture2vec_amd import _lib as _l
    prev_persist = _l.load().g2v_dec_rollout_set_persistent(0 if kernels == "fused_step" else 1)     # (fused_step: the step kernels, not the
    request.addfinalizer(lambda: _l.load().g2v_dec_rollout_set_persistent(prev_persist))              #  cluster launch behind the same entry)
    calls0, ccalls0 = rollout_t2e.FUSED_CALLS, rollout_t2e.CLUSTER_CALLS
    from gesture2vec_amd.model.text2embedding_model import text2embedding_model
    from gesture2vec_amd.train_eval.train_seq2seq import train_iter_text2embedding
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    B, Tw, S, H, L, K, NW, EMB = [int(v) for v in fx["cfg"]]
    p, lr = [float(v) for v in fx["cfg_f"]]
    args = argparse.Namespace(hidden_size=H, n_layers=L, dropout_prob=p, autoencoder_vq_components=K, autoencoder_att=att,
                              n_pre_poses=1, n_poses=20, sentence_frame_length=120, text2_embedding_discrete="True",
                              autoencoder_conditioned="True", autoencoder_fixed_weight="False")
    net = text2embedding_model(args, 135, 20, NW, EMB, np.zeros((NW, EMB), dtype=np.float32), None)
    sd0 = {k[3:]: torch.from_numpy(fx[k].copy()) for k in fx.files if k.startswith("w0/")}
    net.load_state_dict(sd0, strict=True)
    net = net.to(DEV)
    net.train(True)
    optim = FlatClipAdam(net.parameters(), lr=lr, betas=(0.5, 0.999))
    ids = torch.from_numpy(fx["ids"].copy()).to(DEV)
    lengths = torch.from_numpy(fx["lengths"].copy())
    codes = torch.from_numpy(fx["codes"].copy()).to(DEV)
    for step in (1, 2):
        net.set_dropout_masks(torch.from_numpy(fx[f"s{step}/mask_emb"].copy()).to(DEV),
                              torch.from_numpy(fx[f"s{step}/mask_dec_l0"].copy()).to(DEV),
                              torch.from_numpy(fx[f"s{step}/mask_enc_l0"].copy()).to(DEV).to(torch.uint8))
        outputs_ref = fx[f"s{step}/outputs"]
        loss = train_iter_text2embedding(args, 1, ids, lengths, None, None, codes, None, net, optim)
        assert abs(loss["loss"] - float(fx[f"s{step}/loss"])) <= 1e-5 * float(fx[f"s{step}/loss"]), (loss, float(fx[f"s{step}/loss"]))
        if step == 1:
            for n, prm in net.named_parameters():
                ref = fx["s1/grad/" + n]
                if n == "decoder.decoder.pre_linear.0.bias":
                    continue                       # feeds BatchNorm: mathematically zero
                if np.abs(ref).max() == 0:
                    assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, n     # encoder layer 1: dead compute
                else:
                    assert prm.grad is not None and relerr(prm.grad, ref) < 5e-4, (n, relerr(prm.grad, ref))
    assert rollout_t2e.FUSED_CALLS - calls0 == (2 if kernels == "fused_step" else 0)
    assert rollout_t2e.CLUSTER_CALLS - ccalls0 == (2 if kernels == "cluster_forward" else 0)
    for k in fx.files:
        if k.startswith("wN/"):
            n = k[3:]
            ref, got = torch.from_numpy(fx[k].copy()), net.state_dict()[n].cpu()
            if n in ("decoder.decoder.pre_linear.0.bias", "decoder.decoder.pre_linear.1.running_mean"):
                assert float((got - ref).abs().max()) <= 1.01 * 2 * lr, n
            elif ref.dtype.is_floating_point:
                err = float((got.double() - ref.double()).abs().max())
                assert err <= 1e-4 * float(ref.abs().max()) + 0.02 * 2 * lr, (n, err)
            else:
                assert torch.equal(got, ref), n
    net.train(False)
    with torch.no_grad():
        out, attn_list = net(ids, lengths, None, codes, None, None)
    assert out.shape == (B, S, K)
    if att == "True":
        assert len(attn_list) == S - 1 and attn_list[0].shape == (B, 1, Tw)
        assert float((attn_list[0].sum(2) - 1).abs().max()) < 1e-5
    else:
        assert attn_list == []
    assert relerr(out, fx["eval/outputs"]) < 1e-4
    # greedy codes of the eval rollout are exactly the reference's
    assert np.array_equal(out[:, 1:].argmax(2).cpu().numpy(), fx["eval/outputs"][:, 1:].argmax(2))
    # the inference branch (:685-692): an extra first step fed with vid_indices fills outputs[:, 0]
    vid = torch.from_numpy(fx["eval_vid/vid_indices"].copy()).to(DEV)
    with torch.no_grad():
        out_v, _ = net(ids, lengths, None, codes, None, vid)
    assert relerr(out_v, fx["eval_vid/outputs"]) < 1e-4
    assert np.array_equal(out_v.argmax(2).cpu().numpy(), fx["eval_vid/outputs"].argmax(2))
    net.train(True)
    with pytest.raises(NotImplementedError):
        net(ids, lengths, None, codes, None, vid)
    net.train(False)


def test_attention_op_matches_torch_reference():
    """g2v_attn_fwd / g2v_attn_bwd vs the plain torch fp32 formula (reference Attn :160-198 + context :353-359), at a
    ragged shape (H not a multiple of 64, B not a multiple of 4)."""
    from gesture2vec_amd import functional as Fn
    torch.manual_seed(5)
    T, B, H = 11, 37, 40
    hp = torch.randn(B, H, device=DEV, requires_grad=True)
    ep = torch.randn(T, B, H, device=DEV, requires_grad=True)
    enc = torch.randn(T, B, H, device=DEV, requires_grad=True)
    v = torch.randn(H, device=DEV, requires_grad=True)
    ctx, w = Fn.AttnFn.apply(hp, ep, enc, v)
    g = torch.randn(B, H, device=DEV)
    (ctx * g).sum().backward()
    got = [t.grad.clone() for t in (hp, ep, enc, v)]
    ref_in = [t.detach().clone().cpu().double().requires_grad_(True) for t in (hp, ep, enc, v)]
    hp2, ep2, enc2, v2 = ref_in
    score = (torch.tanh(hp2.unsqueeze(0) + ep2) * v2).sum(2)            # (T,B)
    w2 = torch.softmax(score.t(), 1)
    ctx2 = torch.einsum("bt,tbh->bh", w2, enc2)
    (ctx2 * g.cpu().double()).sum().backward()
    assert relerr(w, w2) < 1e-5 and relerr(ctx, ctx2) < 1e-5
    for a, b2, n in zip(got, ref_in, ("hp", "ep", "enc", "v")):
        assert relerr(a, b2.grad) < 2e-5, (n, relerr(a, b2.grad))


@pytest.mark.parametrize("T,B,H,K", [(20, 128, 200, 512), (11, 37, 40, 70), (3, 5, 16, 16)])
def test_fused_step_head_equals_its_three_operators(T, B, H, K):
    """g2v_attn_step_fwd (round 6: argmax of the previous logits + code embedding with its mask + attention in one launch) against
    g2v_argmax_rows, g2v_embedding_fwd and g2v_attn_fwd: ids equal (ties included), both halves of the decoder input row and the
    attention weights bitwise; with given ids (teacher-forced step) likewise.  g2v_linear_fwd_dual against two g2v_linear_fwd calls
    (bitwise), g2v_slab_sum against a float64 sum, g2v_attn_bwd's unsummed d_v partials against its own reduction."""
    from gesture2vec_amd import ops
    g = torch.Generator().manual_seed(11)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)
    logits = r(B, K)
    logits[1, 3] = logits[1, 7] = 50.0                                     # a tie: the lowest index wins
    table, hp, ep, enc, v = r(K, H), r(B, H), r(T, B, H), r(T, B, H), r(H)
    keep = (torch.rand(B, H, generator=g) < 0.5).to(torch.uint8).to(DEV)
    ids_ref = ops.argmax_rows(logits)
    assert int(ids_ref[1]) == 3
    ec_ref = torch.full((B, 2 * H), 7.0, device=DEV)
    ops.embedding_fwd(table, ids_ref, keep, 2.0, out=ec_ref, ldo=2 * H)
    w_ref, _ = ops.attn_fwd(hp, ep, enc, v, ctx_out=ec_ref[:, H:], ldctx=2 * H)
    for given in (False, True):
        ids = ids_ref.clone() if given else torch.full((B,), -5, dtype=torch.int64, device=DEV)
        ec, w = torch.full((B, 2 * H), 9.0, device=DEV), torch.empty((B, T), device=DEV)
        ops.attn_step_fwd(None if given else logits, ids, table, keep, 2.0, ec, hp, ep, enc, v, w)
        assert torch.equal(ids, ids_ref)
        assert torch.equal(ec, ec_ref), given
        assert torch.equal(w, w_ref)
    # the two products of the top state in one launch
    x, w_a, b_a, w_b, b_b = r(B, H), r(K, H), r(K), r(H, H), r(H)
    ya, yb = torch.empty((B, K), device=DEV), torch.empty((B, H), device=DEV)
    ops.linear_fwd_dual(x, w_a, b_a, ya, w_b, b_b, yb)
    assert torch.equal(ya, ops.linear_fwd(x, w_a, b_a)) and torch.equal(yb, ops.linear_fwd(x, w_b, b_b))
    # unsummed d_v partials + one slab sum == the per-call reduction (to summation order)
    d_ctx = r(B, H)
    ref = ops.attn_bwd(d_ctx, hp, ep, enc, v, w_ref)
    slabs = torch.zeros((2, B, H), device=DEV)
    out = (torch.empty((B, H), device=DEV), torch.empty((T, B, H), device=DEV), torch.empty((T, B, H), device=DEV), torch.full((H,), 3.0, device=DEV))
    ops.attn_bwd(d_ctx, hp, ep, enc, v, w_ref, out=out, dv_slab=slabs[1])
    for a_, b_ in zip(out[:3], ref[:3]):
        assert torch.equal(a_, b_)
    assert float((out[3] - 3.0).abs().max()) == 0.0                      # d_v untouched
    dv = ops.slab_sum(slabs.view(2 * B, H), torch.empty((H,), device=DEV))
    assert relerr(dv.cpu(), ref[3].cpu().double()) < 1e-5
    assert relerr(dv.cpu(), slabs.double().sum((0, 1)).cpu()) < 1e-6


def test_graphed_text2embedding_step_trains():
    """GraphedText2EmbeddingStep: the whole Part-d train iteration replayed from one hipGraph lowers the loss like the
    eager iteration does (same kernels; dropout masks are drawn on the device at every replay)."""
    from gesture2vec_amd.flat import FlatClipAdam
    from gesture2vec_amd.model.text2embedding_model import text2embedding_model
    from gesture2vec_amd.train_eval.train_seq2seq import GraphedText2EmbeddingStep
    torch.manual_seed(0)
    B, Tw, S, H, L, K, NW, EMB = 32, 10, 6, 32, 2, 64, 100, 300
    args = argparse.Namespace(hidden_size=H, n_layers=L, dropout_prob=0.1, autoencoder_vq_components=K, autoencoder_att="True",
                              n_pre_poses=1, n_poses=20, sentence_frame_length=120, text2_embedding_discrete="True")
    net = text2embedding_model(args, 135, 20, NW, EMB, np.random.RandomState(0).randn(NW, EMB).astype(np.float32), None).to(DEV)
    net.train(True)
    optim = FlatClipAdam(net.parameters(), lr=2e-3, betas=(0.5, 0.999))
    lengths = torch.randint(3, Tw + 1, (B,)).sort(descending=True).values
    lengths[0] = Tw
    ids = torch.zeros(B, Tw, dtype=torch.int64)
    for b in range(B):
        ids[b, : lengths[b]] = torch.randint(4, NW, (int(lengths[b]),))
    codes = torch.randint(0, K, (B, S))
    step = GraphedText2EmbeddingStep(args, net, optim, ids.to(DEV), lengths, codes.to(DEV))
    first = float(step.replay().detach())
    for _ in range(60):
        step.replay()
    last = float(step.loss.detach())
    assert np.isfinite(first) and np.isfinite(last) and last < 0.8 * first, (first, last)


def test_new_tutorial_classes_match_reference_golden(golden_dir):
    """a15': text2embedding_model_New / EncoderRNN_New / DecoderRNN_New against the reference fixture, both branches of the
    teacher-forcing coin flip (python `random` seeded as in the fixture)."""
    import random
    from gesture2vec_amd.model.text2embedding_model import text2embedding_model_New
    fx = np.load(os.path.join(golden_dir, "t2e_new.npz"))
    H, K, B, Tw, S, NW = [int(v) for v in fx["cfg"]]
    emb = np.random.RandomState(0).randn(NW, 300).astype(np.float32)
    args = argparse.Namespace(hidden_size=H, autoencoder_vq_components=K)
    net = text2embedding_model_New(args, 135, 20, NW, 300, emb, None)
    sd = {k[3:]: torch.from_numpy(fx[k].copy()) for k in fx.files if k.startswith("w0/")}
    sd["encoder.embedding.weight"] = torch.from_numpy(emb)
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV)
    wts = torch.randn(S, B, K + 2, generator=torch.Generator().manual_seed(6)).to(DEV)
    ids, codes = torch.from_numpy(fx["ids"].copy()).to(DEV), torch.from_numpy(fx["codes"].copy()).to(DEV)
    used = fx["used_rows"]
    for tag in ("tf", "free"):
        for p_ in net.parameters():
            p_.grad = None
        random.seed(int(fx[f"{tag}/seed"]))
        out = net(ids, None, codes, None)
        (out * wts).sum().backward()
        assert relerr(out[:, ::2], fx[f"{tag}/out_even_rows"]) < 2e-5
        params = dict(net.named_parameters())
        for k in fx.files:
            if k.startswith(f"{tag}/grad/"):
                n = k[len(f"{tag}/grad/"):]
                g = params[n.split("@")[0]].grad
                assert g is not None, n
                if n.endswith("@used"):
                    rest = g.clone(); rest[torch.from_numpy(used).to(DEV)] = 0
                    assert float(rest.abs().max()) == 0.0
                    g = g[torch.from_numpy(used).to(DEV)]
                assert relerr(g, fx[k]) < 3e-4, (tag, n, relerr(g, fx[k]))


@pytest.mark.parametrize("att,p,B", [("False", 0.2, 24), ("True", 0.2, 24), ("False", 0.0, 130), ("True", 0.0, 7)])
def test_fused_decoder_rollout_matches_per_operator_path(att, p, B):
    """gesture2vec_amd/rollout_t2e.py (the S-1 decode steps as ONE autograd node, weight gradients batched over the steps)
    against the step-at-a-time chain of per-operator nodes: forward equal to rounding, gradients equal up to the summation
    order over the steps."""
    from gesture2vec_amd.model.text2embedding_model import text2embedding_model
    H, L, K, NW, EMB, Tw, S = 48, 2, 40, 50, 30, 9, 6
    args = argparse.Namespace(hidden_size=H, n_layers=L, dropout_prob=p, autoencoder_vq_components=K, autoencoder_att=att,
                              n_pre_poses=1, n_poses=20, sentence_frame_length=120, text2_embedding_discrete="True",
                              autoencoder_conditioned="True", autoencoder_fixed_weight="False")
    torch.manual_seed(5)
    g = torch.Generator().manual_seed(6)
    nets = []
    for fused in (True, False):
        torch.manual_seed(5)
        net = text2embedding_model(args, 135, 20, NW, EMB, np.random.RandomState(0).randn(NW, EMB).astype(np.float32), None).to(DEV)
        net.train(True)
        net.fused_rollout = fused
        nets.append(net)
    nets[1].load_state_dict(nets[0].state_dict())
    ids = torch.randint(1, NW, (B, Tw), generator=g).to(DEV)
    lengths = torch.sort(torch.randint(3, Tw + 1, (B,), generator=g), descending=True).values
    lengths[0] = Tw
    codes = torch.randint(0, K, (B, S), generator=g).to(DEV)
    masks = ((torch.rand(S - 1, B, H, generator=g) < 0.5).to(torch.uint8).to(DEV),
             (torch.rand(S - 1, B, H, generator=g) < 1 - p).to(torch.uint8).to(DEV) if p > 0 else None,
             (torch.rand(Tw, B, 2 * H, generator=g) < 1 - p).to(torch.uint8).to(DEV) if p > 0 else None)
    w = torch.randn(B, S, K, generator=g).to(DEV)
    outs = []
    for net in nets:
        net.set_dropout_masks(*masks)
        out, attn = net(ids, lengths, None, codes, None, None)
        (out * w).sum().backward()
        outs.append((out.detach(), attn))
    # same operators in the same order, except the one-launch GRU cell of the fused node (h_new within an ulp of the
    # two-launch route): equal to rounding, and the greedy codes fed back are the same
    assert relerr(outs[0][0], outs[1][0].cpu()) < 2e-6
    assert torch.equal(outs[0][0][:, 1:].argmax(2), outs[1][0][:, 1:].argmax(2))
    if att == "True":
        assert len(outs[0][1]) == S - 1
        for a, b in zip(outs[0][1], outs[1][1]):
            assert a.shape == (B, 1, Tw) and relerr(a, b.cpu()) < 2e-6
    bn0, bn1 = (n.decoder.decoder.pre_linear[1] for n in nets)
    assert relerr(bn0.running_mean, bn1.running_mean.cpu()) < 2e-6 and relerr(bn0.running_var, bn1.running_var.cpu()) < 2e-6
    assert int(bn0.num_batches_tracked) == int(bn1.num_batches_tracked) == S - 1
    checked = 0
    for (n, pa), (_, pb) in zip(nets[0].named_parameters(), nets[1].named_parameters()):
        if pb.grad is None:
            assert pa.grad is None or float(pa.grad.abs().max()) == 0.0, n
            continue
        assert pa.grad is not None, n
        if n == "decoder.decoder.pre_linear.0.bias":          # feeds BatchNorm: rounding noise around zero on both sides
            continue
        assert relerr(pa.grad, pb.grad.cpu()) < 2e-5, (n, relerr(pa.grad, pb.grad.cpu()))
        checked += 1
    assert checked >= 20


@pytest.mark.parametrize("att,p,B,n_pre,H,K", [("False", 0.2, 24, 1, 48, 40), ("True", 0.2, 24, 1, 48, 40), ("False", 0.0, 130, 3, 48, 40),
                                                ("True", 0.0, 7, 1, 48, 40), ("False", 0.2, 37, 1, 200, 512), ("True", 0.2, 52, 2, 200, 512),
                                                ("True", 0.0, 260, 1, 64, 128), ("False", 0.1, 1040, 1, 200, 512)])
def test_fused_step_kernels_match_per_operator_path(att, p, B, n_pre, H, K, monkeypatch):
    """g2v_attn_code_rollout_fwd / _bwd (one kernel per decode step; the whole BPTT in one launch without attention) against the
    step-at-a-time chain of per-operator autograd nodes on the same weights, masks and codes: ragged row tiles (B % 16 != 0),
    H with and without MFMA padding (200 -> 208), teacher-forced prefixes, inter-layer dropout, both attention settings.  The
    BatchNorm statistics come from per-workgroup partial sums (E[x^2] - mean^2) instead of the two-pass form and the attention
    sums run in another order: equal to a few 1e-6, the greedy codes fed back are the same."""
    from gesture2vec_amd import rollout_t2e
    from gesture2vec_amd.model.text2embedding_model import text2embedding_model
    L, NW, EMB, Tw, S = 2, 50, 30, 9, 6
    args = argparse.Namespace(hidden_size=H, n_layers=L, dropout_prob=p, autoencoder_vq_components=K, autoencoder_att=att,
                              n_pre_poses=n_pre, n_poses=20, sentence_frame_length=120, text2_embedding_discrete="True",
                              autoencoder_conditioned="True", autoencoder_fixed_weight="False")
    g = torch.Generator().manual_seed(6)
    nets = []
    for fused in (True, False):
        torch.manual_seed(5)
        net = text2embedding_model(args, 135, 20, NW, EMB, np.random.RandomState(0).randn(NW, EMB).astype(np.float32), None).to(DEV)
        net.train(True)
        net.fused_rollout = fused
        nets.append(net)
    nets[1].load_state_dict(nets[0].state_dict())
    ids = torch.randint(1, NW, (B, Tw), generator=g).to(DEV)
    lengths = torch.sort(torch.randint(3, Tw + 1, (B,), generator=g), descending=True).values
    lengths[0] = Tw
    codes = torch.randint(0, K, (B, S), generator=g).to(DEV)
    masks = ((torch.rand(S - 1, B, H, generator=g) < 0.5).to(torch.uint8).to(DEV),
             (torch.rand(S - 1, B, H, generator=g) < 1 - p).to(torch.uint8).to(DEV) if p > 0 else None,
             (torch.rand(Tw, B, 2 * H, generator=g) < 1 - p).to(torch.uint8).to(DEV) if p > 0 else None)
    w = torch.randn(B, S, K, generator=g).to(DEV)
    outs = []
    calls0 = rollout_t2e.FUSED_CALLS
    monkeypatch.setattr(rollout_t2e, "CLUSTER_FORWARD", False)      # (the per-operator leg really is per-operator)
    from gesture2vec_amd import _lib
    lib = _lib.load()
    prev = lib.g2v_dec_rollout_set_persistent(0)                    # (the fused leg really runs the step kernels: no cluster launch behind the entry)
    try:
        for net, min_rows in zip(nets, (1, 1 << 30)):
            monkeypatch.setattr(rollout_t2e, "FUSED_MIN_ROWS", min_rows)
            net.set_dropout_masks(*masks)
            out, attn = net(ids, lengths, None, codes, None, None)
            (out * w).sum().backward()
            outs.append((out.detach(), attn))
    finally:
        lib.g2v_dec_rollout_set_persistent(prev)
    assert rollout_t2e.FUSED_CALLS - calls0 == 1, "the fused step kernels did not serve this shape (g2v_attn_code_rollout_ok)"
    assert relerr(outs[0][0], outs[1][0].cpu()) < 3e-5
    same = (outs[0][0][:, 1:].argmax(2) == outs[1][0][:, 1:].argmax(2)).all(1)
    assert float(same.float().mean()) > 0.995           # (a greedy decision inside fp32 rounding of a tie changes that row's later steps)
    if att == "True":
        assert len(outs[0][1]) == S - 1
        for a, b in zip(outs[0][1], outs[1][1]):
            assert a.shape == (B, 1, Tw) and relerr(a, b.cpu()) < 3e-5
    bn0, bn1 = (n.decoder.decoder.pre_linear[1] for n in nets)
    assert relerr(bn0.running_mean, bn1.running_mean.cpu()) < 1e-5 and relerr(bn0.running_var, bn1.running_var.cpu()) < 1e-5
    assert int(bn0.num_batches_tracked) == int(bn1.num_batches_tracked) == S - 1
    checked = 0
    for (n, pa), (_, pb) in zip(nets[0].named_parameters(), nets[1].named_parameters()):
        if pb.grad is None:
            assert pa.grad is None or float(pa.grad.abs().max()) == 0.0, n
            continue
        assert pa.grad is not None, n
        if n == "decoder.decoder.pre_linear.0.bias":          # feeds BatchNorm: rounding noise around zero on both sides
            continue
        tol = 2e-4 if bool(same.all()) else 5e-2
        assert relerr(pa.grad, pb.grad.cpu()) < tol, (n, relerr(pa.grad, pb.grad.cpu()))
        checked += 1
    assert checked >= 20


@pytest.mark.parametrize("bptt", [True, False])
@pytest.mark.parametrize("p,B,n_pre,H,K", [(0.2, 128, 1, 200, 512), (0.0, 128, 1, 200, 512), (0.2, 37, 3, 200, 512), (0.1, 300, 1, 52, 64),
                                           (0.2, 16, 1, 48, 40), (0.0, 130, 2, 64, 176)])
def test_cluster_forward_matches_per_operator_path(p, B, n_pre, H, K, bptt, monkeypatch):
    """Small batch, no attention (round 5): the greedy forward rollout as ONE persistent cluster launch (csrc/t2e_rollout.hip:
    code_cluster_fwd_kernel, behind g2v_attn_code_rollout_fwd) with the PER-OPERATOR backward on the arrays it saved, against
    the per-operator forward + backward (bptt = True: the GRU cells' BPTT as one persistent cluster launch too, g2v_code_cluster_bptt):
    the reference's B = 128 / H = 200 / K = 512, ragged row groups, teacher-forced prefixes,
    H % 16 != 0, 19 row groups, K tiles unevenly spread over the workgroups.  Equal to a few 1e-6 (BatchNorm sums from per-workgroup
    partial sums, k-split products), the greedy codes fed back are the same."""
    from gesture2vec_amd import rollout_t2e
    from gesture2vec_amd.model.text2embedding_model import text2embedding_model
    L, NW, EMB, Tw, S = 2, 50, 30, 9, 6
    args = argparse.Namespace(hidden_size=H, n_layers=L, dropout_prob=p, autoencoder_vq_components=K, autoencoder_att="False",
                              n_pre_poses=n_pre, n_poses=20, sentence_frame_length=120, text2_embedding_discrete="True",
                              autoencoder_conditioned="True", autoencoder_fixed_weight="False")
    g = torch.Generator().manual_seed(16)
    nets = []
    for _ in range(2):
        torch.manual_seed(5)
        net = text2embedding_model(args, 135, 20, NW, EMB, np.random.RandomState(0).randn(NW, EMB).astype(np.float32), None).to(DEV)
        net.train(True)
        nets.append(net)
    nets[1].load_state_dict(nets[0].state_dict())
    ids = torch.randint(1, NW, (B, Tw), generator=g).to(DEV)
    lengths = torch.sort(torch.randint(3, Tw + 1, (B,), generator=g), descending=True).values
    lengths[0] = Tw
    codes = torch.randint(0, K, (B, S), generator=g).to(DEV)
    masks = ((torch.rand(S - 1, B, H, generator=g) < 0.5).to(torch.uint8).to(DEV),
             (torch.rand(S - 1, B, H, generator=g) < 1 - p).to(torch.uint8).to(DEV) if p > 0 else None, None)
    w = torch.randn(B, S, K, generator=g).to(DEV)
    outs = []
    calls0, bcalls0 = rollout_t2e.CLUSTER_CALLS, rollout_t2e.CLUSTER_BPTT_CALLS
    monkeypatch.setattr(rollout_t2e, "CLUSTER_BACKWARD", bptt)
    for net, cluster in zip(nets, (True, False)):
        monkeypatch.setattr(rollout_t2e, "CLUSTER_FORWARD", cluster)
        net.set_dropout_masks(*masks)
        out, _ = net(ids, lengths, None, codes, None, None)
        (out * w).sum().backward()
        outs.append(out.detach())
    assert rollout_t2e.CLUSTER_CALLS - calls0 == 1, "the cluster kernel did not serve this shape (g2v_attn_code_rollout_cluster_ok)"
    assert rollout_t2e.CLUSTER_BPTT_CALLS - bcalls0 == (1 if bptt else 0)
    from gesture2vec_amd import _lib
    assert _lib.load().g2v_dec_rollout_persist_fault(0) == 0
    assert relerr(outs[0], outs[1].cpu()) < 3e-5
    same = (outs[0][:, 1:].argmax(2) == outs[1][:, 1:].argmax(2)).all(1)
    assert float(same.float().mean()) > 0.995           # (a greedy decision inside fp32 rounding of a tie changes that row's later steps)
    bn0, bn1 = (n.decoder.decoder.pre_linear[1] for n in nets)
    assert relerr(bn0.running_mean, bn1.running_mean.cpu()) < 1e-5 and relerr(bn0.running_var, bn1.running_var.cpu()) < 1e-5
    checked = 0
    for (n, pa), (_, pb) in zip(nets[0].named_parameters(), nets[1].named_parameters()):
        if pb.grad is None:
            assert pa.grad is None or float(pa.grad.abs().max()) == 0.0, n
            continue
        assert pa.grad is not None, n
        if n == "decoder.decoder.pre_linear.0.bias":          # feeds BatchNorm: rounding noise around zero on both sides
            continue
        tol = 2e-4 if bool(same.all()) else 5e-2
        assert relerr(pa.grad, pb.grad.cpu()) < tol, (n, relerr(pa.grad, pb.grad.cpu()))
        checked += 1
    assert checked >= 20


@pytest.mark.parametrize("att,p,B,H", [("False", 0.2, 24, 48), ("True", 0.2, 130, 48), ("False", 0.0, 2100, 200), ("True", 0.1, 3100, 200),
                                       ("False", 0.0, 7, 200)])
def test_packed_encoder_inputs_equal_the_padded_grid(att, p, B, H):
    """EncoderRNN with host lengths (round 5): layer 0's embeddings / input projections / their gradients on the sum(lengths)
    positions inside the sentences (g2v_gru_dir.gi_row_off: the recurrent kernels address the packed arrays through per-step
    row offsets -- per-step kernels at small batch, the sequence kernels above) against the padded (Tw,B) grid: outputs and
    hidden states bitwise (a position's arithmetic does not change), gradients to summation order (the padded grid adds exact
    zeros in other places of the row-split sums)."""
    from gesture2vec_amd.model.text2embedding_model import text2embedding_model
    L, K, NW, EMB, Tw, S = 2, 64, 50, 30, 13, 6
    args = argparse.Namespace(hidden_size=H, n_layers=L, dropout_prob=p, autoencoder_vq_components=K, autoencoder_att=att,
                              n_pre_poses=1, n_poses=20, sentence_frame_length=120, text2_embedding_discrete="True",
                              autoencoder_conditioned="True", autoencoder_fixed_weight="False")
    g = torch.Generator().manual_seed(8)
    nets = []
    for packed in (True, False):
        torch.manual_seed(5)
        net = text2embedding_model(args, 135, 20, NW, EMB, np.random.RandomState(0).randn(NW, EMB).astype(np.float32), None).to(DEV)
        net.train(True)
        net.encoder.packed_inputs = packed
        nets.append(net)
    nets[1].load_state_dict(nets[0].state_dict())
    lengths = torch.sort(torch.randint(2, Tw + 1, (B,), generator=g), descending=True).values
    lengths[0] = Tw
    ids = torch.zeros(B, Tw, dtype=torch.int64)
    for b in range(B):
        ids[b, : lengths[b]] = torch.randint(1, NW, (int(lengths[b]),), generator=g)
    ids = ids.to(DEV)
    codes = torch.randint(0, K, (B, S), generator=g).to(DEV)
    masks = ((torch.rand(S - 1, B, H, generator=g) < 0.5).to(torch.uint8).to(DEV),
             (torch.rand(S - 1, B, H, generator=g) < 1 - p).to(torch.uint8).to(DEV) if p > 0 else None,
             (torch.rand(Tw, B, 2 * H, generator=g) < 1 - p).to(torch.uint8).to(DEV) if p > 0 else None)
    w = torch.randn(B, S, K, generator=g).to(DEV)
    outs = []
    for net in nets:
        net.set_dropout_masks(*masks)
        out, _ = net(ids, lengths, None, codes, None, None)
        (out * w).sum().backward()
        outs.append(out.detach())
    assert nets[0].encoder._len_cache["packed"] is not None and nets[1].encoder._len_cache.get("packed") is None
    assert torch.equal(outs[0], outs[1])
    checked = 0
    for (n, pa), (_, pb) in zip(nets[0].named_parameters(), nets[1].named_parameters()):
        if pb.grad is None:
            assert pa.grad is None or float(pa.grad.abs().max()) == 0.0, n
            continue
        assert pa.grad is not None, n
        if n.startswith("decoder."):
            assert torch.equal(pa.grad, pb.grad), n         # behind the encoder everything is the same launch sequence
        else:
            assert relerr(pa.grad, pb.grad.cpu()) < 3e-6, (n, relerr(pa.grad, pb.grad.cpu()))
        checked += 1
    assert checked >= 20


def test_train_iter_repeats_an_iteration_whose_persistent_kernels_faulted():
    """The encoder's small-batch GRU kernels are persistent (g2v_gru_seq_set_cluster) and latch the fault word of the persistent
    kernels when a bounded wait runs out.  A latched iteration must change NOTHING (clip + Adam and BatchNorm's running statistics
    read the latch on the device) and train_iter_text2embedding repeats it once on the per-step kernels: the model after a
    faulted-and-repeated iteration equals the model after a clean one."""
    from gesture2vec_amd import _lib
    from gesture2vec_amd.flat import FlatClipAdam
    from gesture2vec_amd.model.text2embedding_model import text2embedding_model
    from gesture2vec_amd.train_eval.train_seq2seq import train_iter_text2embedding
    lib = _lib.load()
    B, H, L, K, NW, EMB, Tw, S, p = 32, 48, 2, 64, 50, 30, 9, 6, 0.1
    args = argparse.Namespace(hidden_size=H, n_layers=L, dropout_prob=p, autoencoder_vq_components=K, autoencoder_att="False",
                              n_pre_poses=1, n_poses=20, sentence_frame_length=120, text2_embedding_discrete="True",
                              autoencoder_conditioned="True", autoencoder_fixed_weight="False")
    g = torch.Generator().manual_seed(3)
    lengths = torch.sort(torch.randint(2, Tw + 1, (B,), generator=g), descending=True).values
    lengths[0] = Tw
    ids = torch.randint(1, NW, (B, Tw), generator=g).to(DEV)
    codes = torch.randint(0, K, (B, S), generator=g).to(DEV)
    masks = ((torch.rand(S - 1, B, H, generator=g) < 0.5).to(torch.uint8).to(DEV),
             (torch.rand(S - 1, B, H, generator=g) < 1 - p).to(torch.uint8).to(DEV), None)
    states = []
    prev = lib.g2v_gru_seq_set_cluster(1)
    try:
        for fault in (False, True):
            torch.manual_seed(9)
            net = text2embedding_model(args, 135, 20, NW, EMB, np.random.RandomState(2).randn(NW, EMB).astype(np.float32), None).to(DEV)
            net.train(True)
            optim = FlatClipAdam(net.parameters(), lr=1e-3, betas=(0.5, 0.999))
            net.set_dropout_masks(*masks)
            assert lib.g2v_dec_rollout_persist_fault(1) == 0
            lib.g2v_gru_seq_set_cluster(1 if fault else 0)      # (the clean run on the kernels the repeated iteration ends up on)
            lib.g2v_dec_rollout_set_persistent(1 if fault else 0)
            if fault:
                lib.g2v_dec_rollout_persist_fault(-1)          # as a bounded wait running out would
                orig = net.forward                               # (the repeated iteration needs the same explicit masks again)

                def fwd(*a, **k):
                    net.set_dropout_masks(*masks)
                    return orig(*a, **k)
                net.forward = fwd
                with pytest.warns(RuntimeWarning, match="repeated on the per-step kernels"):
                    r = train_iter_text2embedding(args, 1, ids, lengths, None, None, codes, None, net, optim)
                assert lib.g2v_gru_seq_set_cluster(1) == 0      # the per-step kernels were selected
            else:
                r = train_iter_text2embedding(args, 1, ids, lengths, None, None, codes, None, net, optim)
            assert lib.g2v_dec_rollout_persist_fault(0) == 0
            states.append((r["loss"], {k: v.detach().clone() for k, v in net.state_dict().items()}))
    finally:
        lib.g2v_dec_rollout_persist_fault(1)
        lib.g2v_gru_seq_set_cluster(prev)
        lib.g2v_dec_rollout_set_persistent(1)
    assert states[0][0] == states[1][0]
    for k, v in states[0][1].items():      # the same kernels on the same inputs: the discarded attempt left no trace at all
        assert torch.equal(v, states[1][1][k]), k


@pytest.mark.parametrize("att,B,force_fused", [("False", 40, False), ("True", 24, True), ("False", 1040, True)])
def test_loss_on_the_step_major_outputs_equals_the_reference_form(att, B, force_fused, monkeypatch):
    """train_iter_text2embedding takes CrossEntropyLoss on the step-major (S,B,K) array behind the model's outputs view
    (train_seq2seq._code_loss: no strided copy of the logits, no zero-filled slice gradient) -- against the reference's literal
    form outputs[:, 1:, :].reshape(-1, K) on the view: same loss (the mean runs over the same rows in another order), same
    gradients; slot 0 of the outputs is one_hot(codes[:, 0]) (reference :676-677)."""
    from gesture2vec_amd import rollout_t2e
    from gesture2vec_amd.functional import cross_entropy
    from gesture2vec_amd.model.text2embedding_model import text2embedding_model
    from gesture2vec_amd.train_eval.train_seq2seq import _code_loss
    monkeypatch.setattr(rollout_t2e, "FUSED_MIN_ROWS", 1 if force_fused else 1 << 30)
    H, L, K, NW, EMB, Tw, S, p = 48, 2, 64, 50, 30, 9, 6, 0.1
    args = argparse.Namespace(hidden_size=H, n_layers=L, dropout_prob=p, autoencoder_vq_components=K, autoencoder_att=att,
                              n_pre_poses=1, n_poses=20, sentence_frame_length=120, text2_embedding_discrete="True",
                              autoencoder_conditioned="True", autoencoder_fixed_weight="False")
    g = torch.Generator().manual_seed(21)
    torch.manual_seed(6)
    net = text2embedding_model(args, 135, 20, NW, EMB, np.random.RandomState(1).randn(NW, EMB).astype(np.float32), None).to(DEV)
    net.train(True)
    lengths = torch.sort(torch.randint(2, Tw + 1, (B,), generator=g), descending=True).values
    lengths[0] = Tw
    ids = torch.randint(1, NW, (B, Tw), generator=g).to(DEV)
    codes = torch.randint(0, K, (B, S), generator=g).to(DEV)
    masks = ((torch.rand(S - 1, B, H, generator=g) < 0.5).to(torch.uint8).to(DEV),
             (torch.rand(S - 1, B, H, generator=g) < 1 - p).to(torch.uint8).to(DEV),
             (torch.rand(Tw, B, 2 * H, generator=g) < 1 - p).to(torch.uint8).to(DEV))
    res = []
    for form in ("step_major", "reference"):
        net.zero_grad(set_to_none=True)
        net.set_dropout_masks(*masks)
        out, _ = net(ids, lengths, None, codes, None, None)
        assert out.shape == (B, S, K) and getattr(out, "_g2v_step_major").shape == (S, B, K)
        assert torch.equal(out[:, 0], torch.nn.functional.one_hot(codes[:, 0], K).float())
        if form == "step_major":
            loss = _code_loss(out, codes)
        else:
            loss = cross_entropy(out[:, 1:, :].reshape(-1, K), codes[:, 1:].reshape(-1).long())
        loss.backward()
        res.append((float(loss), {n: q.grad.clone() for n, q in net.named_parameters() if q.grad is not None}))
    assert abs(res[0][0] - res[1][0]) <= 2e-6 * abs(res[1][0]), (res[0][0], res[1][0])
    assert res[0][1].keys() == res[1][1].keys() and len(res[0][1]) >= 20
    for n, ga in res[0][1].items():
        if n == "decoder.decoder.pre_linear.0.bias":          # feeds BatchNorm: rounding noise around zero on both sides
            continue
        assert relerr(ga, res[1][1][n].cpu()) < 1e-5, (n, relerr(ga, res[1][1][n].cpu()))


@pytest.mark.parametrize("att,B", [("False", 4096), ("True", 1024),
                                   ("True", 128),      # config/seq2seqtxt.yml as shipped (autoencoder_att: True, B = 128): the cluster
                                                       # encoder + the small-batch decoder with attention (round-5 verdict, 5d)
                                   ("False", 128)])    # config/seq2seq.yml as shipped: cluster encoder, cluster decoder rollout + BPTT
def test_text2embedding_train_step_vs_oracle_at_large_batch(att, B):
    """Part d END TO END against the CPU oracle (oracle/g2v_oracle.py: t2e_train_step, pinned to the reference's golden vectors)
    at the batch sizes bench.py times -- until round 4 it was checked against the oracle only at the fixtures' B <= 24 and
    fused-vs-per-operator above that.  One train_iter_text2embedding on synthetic sentences (config/seq2seq.yml dims: H = 200,
    2 layers, K = 512, S = 6, dropout 0.2), explicit dropout masks on both sides: loss, greedy codes fed back (free-running from
    step 1), every gradient.  A greedy-code decision inside fp32 rounding of a tie may differ on a handful of rows (each changes
    that row's later steps): since round 5 the oracle is told the codes and the ReLU pattern the kernels used."""
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from oracle import g2v_oracle as O
    from gesture2vec_amd.flat import FlatClipAdam
    from gesture2vec_amd.model.text2embedding_model import text2embedding_model
    from gesture2vec_amd.train_eval.train_seq2seq import train_iter_text2embedding
    from train_text2embedding import SyntheticSentences
    H, L, K, NW, EMB, p, lr = 200, 2, 512, 500, 300, 0.2, 5e-4
    args = argparse.Namespace(hidden_size=H, n_layers=L, dropout_prob=p, autoencoder_vq_components=K, autoencoder_att=att,
                              n_pre_poses=1, n_poses=20, sentence_frame_length=120, text2_embedding_discrete="True", batch_size=B)
    torch.manual_seed(4)
    net = text2embedding_model(args, K, 20, NW, EMB, np.random.RandomState(1).randn(NW, EMB).astype(np.float32), None)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.to(DEV)
    net.train(True)
    optim = FlatClipAdam(net.parameters(), lr=lr, betas=(0.5, 0.999))
    data = list(SyntheticSentences(args, NW, 1, seed=2))[0]
    ids, lengths, codes = data[0], data[1], data[6]
    S, Tw = codes.shape[1], ids.shape[1]
    g = torch.Generator().manual_seed(7)
    masks = {"emb": (torch.rand(S - 1, B, H, generator=g) < 0.5).to(torch.uint8),
             "dec_l0": (torch.rand(S - 1, B, H, generator=g) < 1 - p).to(torch.uint8),
             "enc_l0": (torch.rand(Tw, B, 2 * H, generator=g) < 1 - p).to(torch.uint8)}
    cfg = dict(n_layers=L, dropout_prob=p, n_pre_poses=1, lr=lr, att=(att == "True"))
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    from gesture2vec_amd import rollout_t2e
    net.set_dropout_masks(masks["emb"].to(DEV), masks["dec_l0"].to(DEV), masks["enc_l0"].to(DEV))
    loss = train_iter_text2embedding(args, 1, ids.to(DEV), lengths, None, None, codes.to(DEV), None, net, optim)
    # Round 5: the oracle is told the DISCRETE decisions of the run -- the greedy code fed at every decode step and the decoder's
    # ReLU pattern (oracle/g2v_oracle.py: cfg["forced"]); a greedy decision inside fp32 rounding of a tie used to change a row's
    # later steps on a handful of rows, which is why this test carried L2 < 1e-3 / max < 2e-2.  The decisions themselves: the
    # greedy codes must be the argmax of the oracle's own logits wherever its top-2 gap is clear of rounding (below).
    sv = rollout_t2e.LAST_SAVED
    assert sv is not None and sv["ids"].shape == (S - 1, B), "the decoder rollout did not record its decisions"
    if B >= 1024:
        assert rollout_t2e.FUSED_CALLS > 0, "the fused step kernels did not serve this shape"
    forced = {"ids": sv["ids"].cpu(), "relu": (sv["a"] > 0).cpu().to(torch.float32)}
    r = O.t2e_train_step(sd, {}, ids, lengths.long(), codes.long(), masks, dict(cfg, forced=forced))
    lo = r["outputs"][:, 1:S - 1]                                  # logits of decode steps 0 .. S-3 decide ids[1 ..]
    top2 = torch.topk(lo, 2, dim=2).values
    clear = (top2[..., 0] - top2[..., 1]) > 1e-4 * (1 + top2[..., 0].abs())
    assert float(clear.float().mean()) > 0.99
    assert torch.equal(lo.argmax(2)[clear], forced["ids"][1:].t()[clear]), "a greedy code outside the rounding band differs"
    assert abs(loss["loss"] - float(r["loss"])) <= 2e-5 * float(r["loss"]), (loss, float(r["loss"]))
    worst = ("", 0.0)
    for n, prm in net.named_parameters():
        ref = r["grads"].get(n)
        if ref is None or n == "decoder.decoder.pre_linear.0.bias":
            continue
        if float(ref.abs().max()) == 0.0:
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, n      # encoder layer 1 without attention: dead compute
            continue
        got = prm.grad.detach().cpu().double().reshape(-1)
        rf = ref.double().reshape(-1)
        l2 = float((got - rf).norm() / rf.norm())
        worst = max(worst, (n, l2, relerr(prm.grad, ref)), key=lambda v: v[1])
        assert l2 < 1e-4, (n, l2)
        assert relerr(prm.grad, ref) < 2e-4, (n, relerr(prm.grad, ref))
    assert worst[1] > 0.0
    print("worst gradient error (pinned decisions)", worst)


@pytest.mark.parametrize("V,E,H,n", [(60, 300, 200, 5000), (3863, 300, 200, 9000), (17, 20, 12, 700)])
def test_projecting_the_embedding_table_equals_projecting_the_gathered_rows(V, E, H, n):
    """Fn.EmbedProjectPairFn (Part d's encoder at many rows per vocabulary entry: project the table, gather the projected rows;
    scatter-add the gate gradients by word, then three V-row products) against EmbeddingFn + two Fn.linear: forward BITWISE,
    every gradient to summation order, and against float64."""
    from gesture2vec_amd import functional as Fn
    g = torch.Generator().manual_seed(5)
    table0 = torch.randn(V, E, generator=g)
    ids = torch.randint(0, V, (n,), generator=g).to(DEV)
    ws = [(torch.randn(3 * H, E, generator=g) * 0.1, torch.randn(3 * H, generator=g) * 0.1) for _ in range(2)]
    cot = [torch.randn(n, 3 * H, generator=g).to(DEV) for _ in range(2)]

    def leaves():
        t = table0.clone().to(DEV).requires_grad_(True)
        p = [(w.clone().to(DEV).requires_grad_(True), b.clone().to(DEV).requires_grad_(True)) for w, b in ws]
        return t, p

    t1, p1 = leaves()
    gi = Fn.EmbedProjectPairFn.apply(t1, ids, p1[0][0], p1[0][1], p1[1][0], p1[1][1])
    (gi[0] * cot[0]).sum().backward(retain_graph=True)
    (gi[1] * cot[1]).sum().backward()
    t2, p2 = leaves()
    x = Fn.EmbeddingFn.apply(t2, ids, None, 1.0)
    ref = [Fn.linear(x, w, b) for w, b in p2]
    ((ref[0] * cot[0]).sum() + (ref[1] * cot[1]).sum()).backward()
    for k in range(2):
        assert torch.equal(gi[k], ref[k]), f"direction {k}: the projected-table rows differ from the projected rows"
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
    assert rel(t1.grad, t2.grad) < 5e-6, ("d table", rel(t1.grad, t2.grad))
    for k in range(2):
        assert rel(p1[k][0].grad, p2[k][0].grad) < 5e-6, (k, "dW", rel(p1[k][0].grad, p2[k][0].grad))
        assert rel(p1[k][1].grad, p2[k][1].grad) < 5e-6, (k, "db", rel(p1[k][1].grad, p2[k][1].grad))
    # float64: d table = sum_p onehot^T cot_p W_p
    oh = torch.zeros(n, V, dtype=torch.float64)
    oh[torch.arange(n), ids.cpu()] = 1.0
    d64 = sum(oh.t() @ cot[k].cpu().double() @ ws[k][0].double() for k in range(2))
    assert rel(t1.grad.cpu(), d64) < 5e-6, ("d table vs float64", rel(t1.grad.cpu(), d64))
    dw64 = (oh.t() @ cot[0].cpu().double()).t() @ table0.double()
    assert rel(p1[0][0].grad.cpu(), dw64) < 5e-6, ("dW vs float64", rel(p1[0][0].grad.cpu(), dw64))


@pytest.mark.parametrize("M,K,N", [(1536, 300, 600), (40, 20, 36), (5000, 400, 600)])
def test_linear_pair_function_equals_two_linear_functions(M, K, N):
    """Fn.LinearPairFn (both directions' input projections of a bidirectional layer as one autograd Function) against two
    Fn.linear: outputs bitwise, the input gradient (the two directions' sum) and the weight / bias gradients to summation order."""
    from gesture2vec_amd import functional as Fn
    g = torch.Generator().manual_seed(9)
    x0 = torch.randn(M, K, generator=g)
    ws = [(torch.randn(N, K, generator=g) * 0.1, torch.randn(N, generator=g) * 0.1) for _ in range(2)]
    cot = [torch.randn(M, N, generator=g).to(DEV) for _ in range(2)]

    def leaves():
        x = x0.clone().to(DEV).requires_grad_(True)
        p = [(w.clone().to(DEV).requires_grad_(True), b.clone().to(DEV).requires_grad_(True)) for w, b in ws]
        return x, p

    x1, p1 = leaves()
    ya, yb = Fn.LinearPairFn.apply(x1, p1[0][0], p1[0][1], p1[1][0], p1[1][1])
    ((ya * cot[0]).sum() + (yb * cot[1]).sum()).backward()
    x2, p2 = leaves()
    ref = [Fn.linear(x2, w, b) for w, b in p2]
    ((ref[0] * cot[0]).sum() + (ref[1] * cot[1]).sum()).backward()
    assert torch.equal(ya, ref[0]) and torch.equal(yb, ref[1]), "outputs differ"
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
    assert rel(x1.grad, x2.grad) < 5e-6, ("dx", rel(x1.grad, x2.grad))
    for k in range(2):
        assert rel(p1[k][0].grad, p2[k][0].grad) < 5e-6, (k, "dW", rel(p1[k][0].grad, p2[k][0].grad))
        assert rel(p1[k][1].grad, p2[k][1].grad) < 5e-6, (k, "db", rel(p1[k][1].grad, p2[k][1].grad))


def _small_t2e(att="False", B=32, H=48, K=64, NW=50, EMB=30, Tw=9, S=6, p=0.1, seed=9):
    from gesture2vec_amd.flat import FlatClipAdam
    from gesture2vec_amd.model.text2embedding_model import text2embedding_model
    args = argparse.Namespace(hidden_size=H, n_layers=2, dropout_prob=p, autoencoder_vq_components=K, autoencoder_att=att,
                              n_pre_poses=1, n_poses=20, sentence_frame_length=20 * S, text2_embedding_discrete="True",
                              autoencoder_conditioned="True", autoencoder_fixed_weight="False")
    g = torch.Generator().manual_seed(3)
    lengths = torch.sort(torch.randint(2, Tw + 1, (B,), generator=g), descending=True).values
    lengths[0] = Tw
    ids = torch.randint(1, NW, (B, Tw), generator=g).to(DEV)
    codes = torch.randint(0, K, (B, S), generator=g).to(DEV)
    masks = ((torch.rand(S - 1, B, H, generator=g) < 0.5).to(torch.uint8).to(DEV),
             (torch.rand(S - 1, B, H, generator=g) < 1 - p).to(torch.uint8).to(DEV), None)
    torch.manual_seed(seed)
    net = text2embedding_model(args, 135, 20, NW, EMB, np.random.RandomState(2).randn(NW, EMB).astype(np.float32), None).to(DEV)
    net.train(True)
    optim = FlatClipAdam(net.parameters(), lr=1e-3, betas=(0.5, 0.999))
    return args, net, optim, ids, lengths, codes, masks


def test_a_fault_latched_after_the_forward_leaves_batchnorm_running_statistics_alone(monkeypatch):
    """Round-5 advisor finding: Part d's decoder used to commit BatchNorm's running statistics INSIDE the forward rollout, once per
    decode step, gated only on the fault latch as it stood at that step.  A fault latched LATER in the iteration (a later step,
    the cells' BPTT, the encoder's backward) left those updates applied, train_iter_text2embedding repeated the iteration, and the
    same batch statistics were folded in twice.  Now the commit sits behind the backward (g2v_bn_running_update_invstd, latch-
    gated on the device): the model after [forward, fault, repeated iteration] equals the model after ONE clean iteration, the
    running statistics included."""
    from gesture2vec_amd import _lib
    from gesture2vec_amd.fault_policy import POLICY
    from gesture2vec_amd.train_eval import train_seq2seq as TS
    lib = _lib.load()
    states = []
    prev_c, prev_p = lib.g2v_gru_seq_set_cluster(1), lib.g2v_dec_rollout_set_persistent(1)
    saved_policy = (POLICY.off, POLICY.clean, POLICY.faults, POLICY.rearms)
    try:
        for fault in (False, True):
            args, net, optim, ids, lengths, codes, masks = _small_t2e()
            lib.g2v_dec_rollout_persist_fault(1)
            lib.g2v_gru_seq_set_cluster(1 if fault else 0)      # (the clean run on the kernels the repeated iteration ends up on)
            lib.g2v_dec_rollout_set_persistent(1 if fault else 0)
            orig_fwd = net.forward

            def fwd(*a, **k):
                net.set_dropout_masks(*masks)                    # (the repeated iteration needs the same explicit masks again)
                return orig_fwd(*a, **k)
            net.forward = fwd
            if fault:
                orig_loss, fired = TS._code_loss_backward, []

                def fault_then_loss_and_backward(outputs, codes_):
                    if not fired:                                # behind the FORWARD of the first attempt, in front of its backward
                        fired.append(1)
                        torch.cuda.synchronize()
                        lib.g2v_dec_rollout_persist_fault(-1)
                    return orig_loss(outputs, codes_)
                monkeypatch.setattr(TS, "_code_loss_backward", fault_then_loss_and_backward)
                with pytest.warns(RuntimeWarning, match="repeated on the per-step kernels"):
                    r = TS.train_iter_text2embedding(args, 1, ids, lengths, None, None, codes, None, net, optim)
                monkeypatch.setattr(TS, "_code_loss_backward", orig_loss)
                assert fired
            else:
                r = TS.train_iter_text2embedding(args, 1, ids, lengths, None, None, codes, None, net, optim)
            assert lib.g2v_dec_rollout_persist_fault(0) == 0
            states.append((r["loss"], {k: v.detach().clone() for k, v in net.state_dict().items()}))
    finally:
        lib.g2v_dec_rollout_persist_fault(1)
        lib.g2v_gru_seq_set_cluster(prev_c)
        lib.g2v_dec_rollout_set_persistent(prev_p)
        POLICY.off, POLICY.clean, POLICY.faults, POLICY.rearms = saved_policy
    bn0 = states[0][1]["decoder.decoder.pre_linear.1.running_mean"]
    assert float(bn0.abs().max()) > 0, "the clean iteration did not update the running statistics at all"
    assert states[0][0] == states[1][0]
    for k, v in states[0][1].items():
        assert torch.equal(v, states[1][1][k]), k


def test_deferred_batchnorm_commit_equals_the_in_rollout_updates():
    """The running statistics committed behind the backward from the saved (mean, 1/sqrt(var + eps)) equal the step-by-step updates
    the rollout makes when it is handed the running statistics (a plain forward in train mode): every decoder route."""
    from gesture2vec_amd import rollout_t2e
    for att, B, route in (("False", 32, "cluster"), ("True", 24, "per_operator"), ("False", 40, "per_operator")):
        got = []
        for defer in (False, True):
            args, net, optim, ids, lengths, codes, masks = _small_t2e(att=att, B=B)
            old = rollout_t2e.CLUSTER_FORWARD
            rollout_t2e.CLUSTER_FORWARD = route == "cluster"
            try:
                net.set_dropout_masks(*masks)
                if defer:
                    net.deferred_bn = []
                out, _ = net(ids, lengths, None, codes, None, None)
                bn = net.decoder.decoder.pre_linear[1]
                if defer:
                    assert float(bn.running_mean.abs().max()) == 0.0, "a deferred forward touched the running statistics"
                    net.commit_bn_running_stats()
                    assert net.deferred_bn is None
                got.append((bn.running_mean.detach().clone(), bn.running_var.detach().clone(), int(bn.num_batches_tracked)))
            finally:
                rollout_t2e.CLUSTER_FORWARD = old
        assert got[0][2] == got[1][2] == codes.shape[1] - 1
        assert float(got[0][0].abs().max()) > 0
        assert relerr(got[1][0].cpu(), got[0][0].cpu()) < 1e-6, (att, B, route)
        assert relerr(got[1][1].cpu(), got[0][1].cpu()) < 2e-6, (att, B, route)


def test_graphed_step_detects_a_fault_recaptures_and_the_policy_rearms():
    """GraphedText2EmbeddingStep (round-5 advisor finding): the captured step holds persistent cluster kernels whose fault latch is
    sticky and gates every commit -- ONE timeout used to make every later replay a silent no-op.  replay() now reads the latch
    every `check_every` replays: the unapplied replays are counted, the graph is captured again on the per-step kernels, the step
    repeated; and fault_policy.POLICY switches the fast path back on after `rearm_after` clean iterations (round-5 verdict,
    missing #6: one hiccup used to leave a run on the per-step kernels for the rest of the process)."""
    from gesture2vec_amd import _lib
    from gesture2vec_amd.fault_policy import POLICY
    from gesture2vec_amd.train_eval.train_seq2seq import GraphedText2EmbeddingStep
    lib = _lib.load()
    prev_c, prev_p = lib.g2v_gru_seq_set_cluster(1), lib.g2v_dec_rollout_set_persistent(1)
    saved = (POLICY.off, POLICY.clean, POLICY.faults, POLICY.rearms, POLICY.rearm_after, POLICY.max_rearms)
    try:
        lib.g2v_dec_rollout_persist_fault(1)
        POLICY.off, POLICY.clean, POLICY.rearms, POLICY.rearm_after, POLICY.max_rearms = False, 0, 0, 6, 1
        args, net, optim, ids, lengths, codes, masks = _small_t2e()
        step = GraphedText2EmbeddingStep(args, net, optim, ids, lengths, codes, check_every=4)
        for _ in range(4):
            step.replay()
        assert step.lost_replays == 0 and step.recaptures == 0
        w0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
        lib.g2v_dec_rollout_persist_fault(-1)                    # as a bounded wait running out would
        with pytest.warns(RuntimeWarning, match="were not applied"):
            for _ in range(4):
                step.replay()
        # the four replays behind the fault changed nothing; then ONE repeated step on the per-step kernels moved the weights
        assert step.lost_replays == 4 and step.recaptures == 1
        assert lib.g2v_dec_rollout_persist_fault(0) == 0
        assert lib.g2v_gru_seq_set_cluster(0) == 0 and POLICY.off      # the per-step kernels are selected
        moved = [k for k, v in net.state_dict().items() if v.dtype.is_floating_point and not torch.equal(v, w0[k])]
        assert moved, "the repeated step was not applied"
        # (3 warm-up steps of the first capture + 4 applied replays + the re-capture's eager step; the 4 unapplied ones do not count)
        assert int(net.decoder.decoder.pre_linear[1].num_batches_tracked) == (3 + 4 + 1) * (codes.shape[1] - 1)
        # six clean replays later the fast path is back, and the graph is captured again for it
        for _ in range(8):
            step.replay()
        assert not POLICY.off and POLICY.rearms == 1
        assert lib.g2v_gru_seq_set_cluster(1) == 1 and lib.g2v_dec_rollout_set_persistent(1) == 1
        assert step.recaptures == 2
        loss = step.read_loss()
        assert np.isfinite(loss)
        # a static-lengths graph refuses other lengths instead of replaying stale packing (advisor finding)
        st2 = GraphedText2EmbeddingStep(args, net, optim, ids, lengths, codes, static_lengths=True, check_every=0)
        other = lengths.clone(); other[-1] = max(1, int(other[-1]) - 1)
        with pytest.raises(ValueError, match="set_lengths"):
            st2.replay(other)
        st2.set_lengths(other)
        st2.replay(other)
    finally:
        lib.g2v_dec_rollout_persist_fault(1)
        lib.g2v_gru_seq_set_cluster(prev_c)
        lib.g2v_dec_rollout_set_persistent(prev_p)
        POLICY.off, POLICY.clean, POLICY.faults, POLICY.rearms, POLICY.rearm_after, POLICY.max_rearms = saved


@pytest.mark.parametrize("att,B,H,K", [("False", 2048, 200, 512), ("True", 2048, 200, 512), ("False", 128, 200, 512), ("True", 24, 48, 64)])
@pytest.mark.parametrize("graphed", [False, True])
def test_side_branches_of_the_backward_change_nothing(att, B, H, K, graphed, monkeypatch):
    """Round 6: inside train_iter_text2embedding / GraphedText2EmbeddingStep the encoder's weight gradients run on a side stream
    beside its input-gradient chain at large batches (ops.side_branches / ops.side_branch: forked in front of the chain, joined in
    front of the optimiser; below the measured row counts a branch stays inline).  Every kernel is deterministic, so three
    iterations with and without the branches must leave bitwise the same model, and the explicit-gradient path (backward()
    outside the scope) must never see a side stream."""
    from gesture2vec_amd import _lib, ops
    from gesture2vec_amd.train_eval import train_seq2seq as TS
    states = []
    mk = lambda: _small_t2e(att=att, B=B, H=H, K=K, NW=300, EMB=300 if H == 200 else 30, Tw=20 if B >= 2048 else 12)
    # (the W_hh-resident BPTT is taken only while no side branch is in flight -- with the branches off it would serve BOTH encoder
    #  layers, with them on only the first, and it equals the streaming kernel to summation order only: pinned off here so that
    #  both runs launch the same kernels and the comparison stays bit for bit)
    lib = _lib.load()
    prev_res = lib.g2v_ctx_set_option(None, _lib.OPT_GRU_RESIDENT_BWD, 0)
    monkeypatch.setattr(ops, "side_pending", lambda: False)
    try:
        _run_side_on_off(att, B, graphed, monkeypatch, ops, TS, mk, states)
    finally:
        lib.g2v_ctx_set_option(None, _lib.OPT_GRU_RESIDENT_BWD, prev_res)
    assert states[0][0] == states[1][0], (states[0][0], states[1][0])
    for k, v in states[0][1].items():
        assert torch.equal(v, states[1][1][k]), k
    # outside the scope a branch is inline: plain backward() + .grad needs no join
    monkeypatch.setattr(ops, "SIDE_BRANCHES", True)
    args, net, optim, ids, lengths, codes, masks = mk()
    net.set_dropout_masks(*masks)
    out, _ = net(ids, lengths, None, codes, None, None)
    TS._code_loss(out, codes).backward()
    assert not ops._side_state["pending"]


def _run_side_on_off(att, B, graphed, monkeypatch, ops, TS, mk, states):
    for side in (False, True):
        monkeypatch.setattr(ops, "SIDE_BRANCHES", side)
        args, net, optim, ids, lengths, codes, masks = mk()
        orig_fwd = net.forward
        taken = []

        def fwd(*a, _o=orig_fwd, _n=net, _m=masks, **k):
            _n.set_dropout_masks(*_m)
            return _o(*a, **k)
        net.forward = fwd
        orig_branch = ops.side_branch

        class Spy:
            def __init__(self, *a, **k):
                self.cm = orig_branch(*a, **k)

            def __enter__(self):
                v = self.cm.__enter__()
                taken.append(bool(v))
                return v

            def __exit__(self, *e):
                return self.cm.__exit__(*e)
        monkeypatch.setattr(ops, "side_branch", Spy)
        losses = []
        if graphed:
            g = TS.GraphedText2EmbeddingStep(args, net, optim, ids, lengths, codes, warmup=1, static_lengths=True, check_every=0)
            for _ in range(2):
                g.replay()
            losses.append(g.read_loss())
            del g
        else:
            for _ in range(3):
                losses.append(TS.train_iter_text2embedding(args, 1, ids, lengths, None, None, codes, None, net, optim)["loss"])
        monkeypatch.setattr(ops, "side_branch", orig_branch)
        torch.cuda.synchronize()
        assert not ops._side_state["pending"] and ops._side_state["active"] == 0
        assert taken, "no side_branch site was reached"
        assert any(taken) == (side and B >= 2048), (side, B, taken)       # forked exactly where the size rule says so
        states.append((losses, {k: v.detach().clone() for k, v in net.state_dict().items()}))


def test_many_multi_stream_graphs_in_one_process_replay():
    """ROCm 7.2: the replay of the fourth multi-stream graph captured in one process over the SAME side streams, with the third still
    alive, died inside hipGraphLaunch.  GraphedText2EmbeddingStep releases its previous graph and draws fresh side
    streams before every capture; this runs the sequences that crashed (four graphs over the bench's four configurations, six
    large-batch graphs in a row, every one captured while its predecessor is alive) in a CHILD process -- a regression would be a
    segmentation fault of the interpreter, not an exception."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "gpurun_tools", "r06_side_repro.py"), "seq4", "seq6big"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("rc 0") == 2, r.stdout[-2000:]


@pytest.mark.parametrize("att", ["False", "True"])
def test_encoder_gathers_the_projected_table_inside_the_recurrent_kernel(att, monkeypatch):
    """Round 6: at large batch the W_hh-resident GRU forward reads row gather[r] of the projected embedding TABLE for packed
    position r (g2v_gru_dir.gi_gather) instead of a materialised (positions x 3H) array, and the tables' gradients come back as
    the scatter-add of dgi by word.  Same arithmetic as the materialised route (Fn.EmbedProjectPairFn): loss bitwise, every
    gradient to summation order."""
    from gesture2vec_amd import ops
    from gesture2vec_amd.train_eval import train_seq2seq as TS
    res = []
    for gather in (True, False):
        if not gather:
            monkeypatch.setattr(ops, "gru_gather_ok", lambda *a, **k: False)
        args, net, optim, ids, lengths, codes, masks = _small_t2e(att=att, B=1280, H=200, K=512, NW=300, EMB=300, Tw=12)
        assert ids.numel() >= 2 * 300                         # many rows per word: the table is projected, not the rows
        calls = []
        orig = ops.gru_dirs_fwd

        def spy(dirs, *a, _o=orig, **k):
            calls.append(dirs[0].get("gi_gather") is not None)
            return _o(dirs, *a, **k)
        monkeypatch.setattr(ops, "gru_dirs_fwd", spy)
        net.set_dropout_masks(*masks)
        out, _ = net(ids, lengths, None, codes, None, None)
        loss = TS._code_loss(out, codes)
        loss.backward()
        monkeypatch.setattr(ops, "gru_dirs_fwd", orig)
        assert calls and calls[0] == gather, calls            # layer 0 gathered (or not) as asked
        res.append((float(loss.detach()), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}))
    assert res[0][0] == res[1][0]
    assert res[0][1].keys() == res[1][1].keys()
    for n, g in res[0][1].items():
        e = float((g.double() - res[1][1][n].double()).abs().max() / res[1][1][n].double().abs().max().clamp_min(1e-30))
        assert e < 2e-6, (n, e)
