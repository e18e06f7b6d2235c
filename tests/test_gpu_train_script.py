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
