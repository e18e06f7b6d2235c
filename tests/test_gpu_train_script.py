"""GPU: the drop-in entry point runs end to end (config -> model -> epochs -> checkpoint -> reload)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_autoencoder_vqvae_synthetic(tmp_path):
    out = os.path.join(tmp_path, "run")
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "train_autoencoder_VQVAE.py"),
           "--config", os.path.join(ROOT, "config", "VQ-VAE_synthetic.yml"), "--synthetic", "--synthetic_batches", "3",
           "--batch_size", "64", "--epochs", "2", "--model_save_path", out, "--name", "t"]
    r = subprocess.run(cmd, cwd=os.path.join(ROOT, "scripts"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    log = r.stderr + r.stdout
    assert "[VAL] loss:" in log and "EP 2 (  3) |" in log and "samples/s | loss:" in log
    ckpt = os.path.join(out, "t_checkpoint_002.bin")
    assert os.path.exists(ckpt) and os.path.exists(os.path.join(out, "conf"))
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import utils.train_utils as tu
    args, net, loss_fn, lang, pose_dim = tu.load_checkpoint_and_model(ckpt, "cuda:0", "autoencoder_vq")
    assert pose_dim == 135 and not net.training
    x = torch.randn(8, 34, 135, device="cuda:0")
    with torch.no_grad():
        out_poses, latent, loss_vq, perp = net(x, x)
    assert out_poses.shape == (8, 34, 135) and torch.isfinite(out_poses).all()
    assert float(loss_fn(out_poses, x)) > 0


def test_train_autoencoder_vqvae_shipped_quantizer(tmp_path):
    """--autoencoder_vq_quantizer gssoft: the model as the reference ships it trains through the same entry point, its
    checkpoint carries the soft quantiser's tensors and reloads as such."""
    out = os.path.join(tmp_path, "run")
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "train_autoencoder_VQVAE.py"),
           "--config", os.path.join(ROOT, "config", "VQ-VAE_synthetic.yml"), "--synthetic", "--synthetic_batches", "3",
           "--batch_size", "64", "--epochs", "2", "--model_save_path", out, "--name", "g", "--autoencoder_vq_quantizer", "gssoft"]
    r = subprocess.run(cmd, cwd=os.path.join(ROOT, "scripts"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    log = r.stderr + r.stdout
    assert "[VAL] loss:" in log and "EP 2 (  3) |" in log
    ckpt = os.path.join(out, "g_checkpoint_002.bin")
    raw = torch.load(ckpt, map_location="cpu", weights_only=False)
    assert "vq_layer.mean_layer.weight" in raw["gen_dict"] and "vq_layer._ema_w" not in raw["gen_dict"]
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import utils.train_utils as tu
    args, net, loss_fn, lang, pose_dim = tu.load_checkpoint_and_model(ckpt, "cuda:0", "autoencoder_vq")
    assert type(net.vq_layer).__name__ == "VQ_Payam_GSSoft"
    x = torch.randn(8, 34, 135, device="cuda:0")
    with torch.no_grad():
        out_poses, latent, loss_vq, perp = net(x, x)
    assert out_poses.shape == (8, 34, 135) and torch.isfinite(out_poses).all() and float(perp) > 1


def test_train_text2embedding_synthetic(tmp_path):
    out = os.path.join(tmp_path, "run_t2e")
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "train_text2embedding.py"),
           "--config", os.path.join(ROOT, "config", "seq2seq_synthetic.yml"), "--synthetic", "--synthetic_batches", "2",
           "--batch_size", "32", "--epochs", "10", "--hidden_size", "64", "--model_save_path", out, "--name", "t"]
    r = subprocess.run(cmd, cwd=os.path.join(ROOT, "scripts"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    log = r.stderr + r.stdout
    assert "[VAL] loss:" in log and "EP 10 (  2) |" in log
    ckpt = torch.load(os.path.join(out, "t_checkpoint_010.bin"), map_location="cpu", weights_only=False)
    assert set(ckpt) == {"args", "epoch", "lang_model", "pose_dim", "gen_dict"} and ckpt["pose_dim"] == 512
    assert "encoder.embedding.weight" in ckpt["gen_dict"] and "decoder.decoder.out.weight" in ckpt["gen_dict"]


def test_train_dae_synthetic_and_reload(tmp_path):
    """Part a trainer (train_DAE.py counterpart): epochs -> checkpoint with the reference's keys / file name -> reload
    through load_checkpoint_and_model(what="DAE"); the loss goes down on the synthetic frames."""
    out = os.path.join(tmp_path, "run_dae")
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "train_DAE.py"),
           "--config", os.path.join(ROOT, "config", "DAE_synthetic.yml"), "--synthetic", "--synthetic_batches", "10",
           "--batch_size", "256", "--epochs", "20", "--model_save_path", out, "--name", "t"]
    r = subprocess.run(cmd, cwd=os.path.join(ROOT, "scripts"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    log = r.stderr + r.stdout
    vals = [float(l.split("[VAL] loss:")[1].split("/")[0]) for l in log.splitlines() if "[VAL] loss:" in l]
    assert len(vals) == 20 and vals[-1] < 0.9 * vals[0], vals
    path = os.path.join(out, "t_H40_checkpoint_020.bin")
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ckpt) == {"args", "epoch", "lang_model", "pose_dim", "gen_dict"} and ckpt["pose_dim"] == 135
    assert set(ckpt["gen_dict"]) == {"encoder.0.weight", "encoder.0.bias", "decoder.0.weight", "decoder.0.bias"}
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import utils.train_utils as tu
    args, net, loss_fn, lang, pose_dim = tu.load_checkpoint_and_model(path, "cuda:0", "DAE")
    x = torch.randn(16, 135, 1, device="cuda:0")
    with torch.no_grad():
        y = net(x)
    assert y.shape == (16, 135, 1) and not net.training and torch.isfinite(y).all()


def test_text2embedding_checkpoint_reload(tmp_path):
    out = os.path.join(tmp_path, "run_t2e2")
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "train_text2embedding.py"),
           "--config", os.path.join(ROOT, "config", "seq2seq_synthetic.yml"), "--synthetic", "--synthetic_batches", "1",
           "--batch_size", "16", "--epochs", "10", "--hidden_size", "32", "--autoencoder_att", "True", "--model_save_path", out,
           "--name", "t"]
    r = subprocess.run(cmd, cwd=os.path.join(ROOT, "scripts"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import utils.train_utils as tu
    args, net, loss_fn, lang, pose_dim = tu.load_checkpoint_and_model(os.path.join(out, "t_checkpoint_010.bin"), "cuda:0",
                                                                       "text2embedding")
    assert args.autoencoder_att == "True" and not net.training
    assert any(k.startswith("decoder.decoder.attn.") for k in net.state_dict())


def test_train_autoencoder_vqvae_resume_is_bit_identical(tmp_path):
    """Checkpoint/resume: a run continued from the epoch-2 checkpoint writes the same epoch-4 checkpoint,
    bit for bit (weights, EMA codebook, BatchNorm statistics), as the uninterrupted run."""
    def run(out, extra):
        cmd = [sys.executable, os.path.join(ROOT, "scripts", "train_autoencoder_VQVAE.py"),
               "--config", os.path.join(ROOT, "config", "VQ-VAE_synthetic.yml"), "--synthetic", "--synthetic_batches", "3",
               "--batch_size", "64", "--epochs", "4", "--save_every", "2", "--model_save_path", out, "--name", "t"] + extra
        r = subprocess.run(cmd, cwd=os.path.join(ROOT, "scripts"), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
    a, b = os.path.join(tmp_path, "full"), os.path.join(tmp_path, "resumed")
    run(a, [])
    run(b, ["--resume", os.path.join(a, "t_checkpoint_002.bin")])
    ca = torch.load(os.path.join(a, "t_checkpoint_004.bin"), map_location="cpu", weights_only=False)
    cb = torch.load(os.path.join(b, "t_checkpoint_004.bin"), map_location="cpu", weights_only=False)
    assert set(ca["gen_dict"]) == set(cb["gen_dict"])
    for k in ca["gen_dict"]:
        assert torch.equal(ca["gen_dict"][k], cb["gen_dict"][k]), k
    assert torch.equal(ca["resume"]["optim"]["m"], cb["resume"]["optim"]["m"])
    assert ca["loss_list"] == cb["loss_list"]


def test_train_autoencoder_vqvae_distributed_launch(tmp_path):
    """BASELINE config 5 entry point: the trainer under `python -m torch.distributed.run` (one process per GPU; here one
    rank, all a single-GPU box offers) runs the data-parallel iteration -- RCCL all-reduce of [grads | EMA stats], EMA update
    from the reduced statistics -- at the GENEA shape and writes the usual checkpoint from rank 0."""
    out = os.path.join(tmp_path, "run_dp")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "scripts", "train_autoencoder_VQVAE.py"),
           "--config", os.path.join(ROOT, "config", "VQ-VAE_GENEA_synthetic.yml"), "--synthetic", "--synthetic_batches", "3",
           "--batch_size", "256", "--epochs", "2", "--model_save_path", out, "--name", "t"]
    r = subprocess.run(cmd, cwd=os.path.join(ROOT, "scripts"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    log = r.stderr + r.stdout
    assert "EP 2 (  3) |" in log and "[VAL] loss:" in log
    ck = torch.load(os.path.join(out, "t_checkpoint_002.bin"), map_location="cpu", weights_only=False)
    assert ck["pose_dim"] == 45 and ck["gen_dict"]["vq_layer._embedding.weight"].shape == (400, 400)
    assert all(torch.isfinite(v).all() for v in ck["gen_dict"].values() if v.dtype.is_floating_point)


def _iter_setup(B, T=8):
    import argparse
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
    D, H, K = 135, 64, 512
    args = argparse.Namespace(rep_learning_dim=D, hidden_size=H, n_layers=2, dropout_prob=0.0, autoencoder_vq="True",
                              autoencoder_vae="False", autoencoder_vq_components=K, autoencoder_vq_commitment_cost=0.25,
                              autoencoder_conditioned="True", autoencoder_att="False", autoencoder_fixed_weight="False",
                              n_pre_poses=1, n_poses=T, loss_l1_weight=5.0, loss_cont_weight=0.1, loss_var_weight=0.5,
                              learning_rate=5e-4)
    torch.manual_seed(3)
    net = Autoencoder_VQVAE(args, D, T).to("cuda:0")
    net.train(True)
    net.rng_seed = 11
    return args, net


def test_train_iter_replayed_from_a_graph_equals_eager_iterations(monkeypatch):
    """From 1024 rows per batch train_iter_Autoencoder_VQ_seq2seq replays its fused step from a hipGraph (first iteration eager,
    second captures); batches arrive as NEW tensors every iteration (a data loader), so the static input buffer is exercised.
    Weights, Adam moments, codebook and the returned losses must equal the eager iterations' bitwise."""
    import gesture2vec_amd.train_eval.train_seq2seq as ts
    B = 1024
    runs = {}
    for mode in ("graph", "eager"):
        monkeypatch.setattr(ts, "_GRAPH_REPLAY", mode == "graph")
        args, net = _iter_setup(B)
        optim = ts.FusedClipAdam(net, 5e-4, betas=(0.5, 0.999))
        g = torch.Generator(device="cuda:0").manual_seed(5)
        losses = []
        for _ in range(5):
            x = torch.randn(B, args.n_poses, 135, generator=g, device="cuda:0")
            loss, perp = ts.train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
            losses.append((loss["loss"], float(perp)))
        eng = net.engine()
        if mode == "graph":
            assert eng._iter_graph["graph"] not in (None, False), "the iteration was not replayed from a graph"
        runs[mode] = (losses, eng.flat.clone(), eng.m.clone(), eng.v.clone(), eng.codebook.clone(), eng.ema_cs.clone(),
                      eng.bn_rm.clone(), int(net.decoder.decoder.pre_linear[1].num_batches_tracked))
    a, b = runs["graph"], runs["eager"]
    assert a[0] == b[0], (a[0], b[0])
    for ta, tb in zip(a[1:7], b[1:7]):
        assert torch.equal(ta, tb)
    assert a[7] == b[7] == 5 * 7


def test_train_iter_inputs_at_a_recurring_address_are_read_in_place(monkeypatch):
    """Batches that arrive at an address seen before (a loader's resident device buffers refilled in place) get a graph of their own
    from the second sighting on, which reads that address directly instead of copying into the static input buffers; contents
    change every iteration, input and target are separate tensors.  Bitwise equal to the eager iterations."""
    import gesture2vec_amd.train_eval.train_seq2seq as ts
    B = 1024
    runs = {}
    for mode in ("graph", "eager"):
        monkeypatch.setattr(ts, "_GRAPH_REPLAY", mode == "graph")
        args, net = _iter_setup(B)
        optim = ts.FusedClipAdam(net, 5e-4, betas=(0.5, 0.999))
        g = torch.Generator(device="cuda:0").manual_seed(8)
        bufs = [(torch.empty(B, args.n_poses, 135, device="cuda:0"), torch.empty(B, args.n_poses, 135, device="cuda:0")) for _ in range(2)]
        losses = []
        for it in range(9):
            x, t = bufs[it % 2]
            x.copy_(torch.randn(B, args.n_poses, 135, generator=g, device="cuda:0"))
            t.copy_(x + 0.01 * torch.randn(B, args.n_poses, 135, generator=g, device="cuda:0"))
            loss, perp = ts.train_iter_Autoencoder_VQ_seq2seq(args, 1, x, t, net, optim)
            losses.append((loss["loss"], float(perp)))
        eng = net.engine()
        if mode == "graph":
            st = eng._iter_graph
            assert st["graph"] not in (None, False) and len(st.get("by_addr", {})) == 2, st.get("by_addr")
        runs[mode] = (losses, eng.flat.clone(), eng.m.clone(), eng.v.clone(), eng.codebook.clone(),
                      int(net.decoder.decoder.pre_linear[1].num_batches_tracked))
    a, b = runs["graph"], runs["eager"]
    assert a[0] == b[0], (a[0], b[0])
    for ta, tb in zip(a[1:5], b[1:5]):
        assert torch.equal(ta, tb)
    assert a[5] == b[5] == 9 * 7


def test_train_iter_keeps_one_graph_per_batch_shape(monkeypatch):
    """A data loader's short last batch alternates with the full ones (and the reference's own batch of 128 is replayed from a
    graph too): each shape keeps its own captured graph, neither evicts the other, and the result equals the eager iterations'."""
    import gesture2vec_amd.train_eval.train_seq2seq as ts
    sizes = [128, 128, 128, 48, 128, 128, 48, 48, 128]
    runs = {}
    for mode in ("graph", "eager"):
        monkeypatch.setattr(ts, "_GRAPH_REPLAY", mode == "graph")
        args, net = _iter_setup(128)
        optim = ts.FusedClipAdam(net, 5e-4, betas=(0.5, 0.999))
        g = torch.Generator(device="cuda:0").manual_seed(6)
        losses = []
        for B in sizes:
            x = torch.randn(B, args.n_poses, 135, generator=g, device="cuda:0")
            loss, perp = ts.train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
            losses.append((loss["loss"], float(perp)))
        eng = net.engine()
        if mode == "graph":
            slots = eng._iter_graphs
            assert len(slots) == 2 and all(st["graph"] not in (None, False) for st in slots.values()), slots
        runs[mode] = (losses, eng.flat.clone(), eng.m.clone(), eng.codebook.clone())
    a, b = runs["graph"], runs["eager"]
    assert a[0] == b[0], (a[0], b[0])
    for ta, tb in zip(a[1:], b[1:]):
        assert torch.equal(ta, tb)


def test_train_iter_adopts_a_plain_torch_adam():
    """The reference's harness builds torch.optim.Adam(net.parameters(), lr, betas=(0.5, 0.999)) (train_autoencoder_VQVAE.py:193-195):
    handing THAT to train_iter must train exactly like the FusedClipAdam it is adopted into, including an lr change between
    iterations; unsupported variants are refused."""
    import gesture2vec_amd.train_eval.train_seq2seq as ts
    res = []
    for kind in ("torch", "fused"):
        args, net = _iter_setup(64)
        optim = (torch.optim.Adam(net.parameters(), lr=5e-4, betas=(0.5, 0.999)) if kind == "torch"
                 else ts.FusedClipAdam(net, 5e-4, betas=(0.5, 0.999)))
        g = torch.Generator(device="cuda:0").manual_seed(6)
        for it in range(3):
            if it == 2:
                if kind == "torch":
                    optim.param_groups[0]["lr"] = 1e-4
                else:
                    optim.lr = 1e-4
            x = torch.randn(64, args.n_poses, 135, generator=g, device="cuda:0")
            loss, _ = ts.train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
        res.append((loss["loss"], net.engine().flat.clone()))
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])
    args, net = _iter_setup(64)
    x = torch.randn(64, args.n_poses, 135, device="cuda:0")
    with pytest.raises(TypeError):
        ts.train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=0.1))
    with pytest.raises(TypeError):
        ts.train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, torch.optim.SGD(net.parameters(), lr=1e-3))
