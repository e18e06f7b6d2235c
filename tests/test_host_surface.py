"""CPU-only tests of the host-side drop-in surface: config parsing, module construction / state_dict keys,
checkpoint contract.  No kernel is launched."""
import argparse
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sys.path.insert(0, ROOT)


def test_parse_synthetic_config():
    from config.parse_args import parse_args
    a = parse_args(["-c", os.path.join(ROOT, "config", "VQ-VAE_synthetic.yml"), "--synthetic", "--batch_size", "128"])
    assert a.hidden_size == 64 and a.n_poses == 34 and a.rep_learning_dim == 135 and a.batch_size == 128
    # boolean-like options stay strings, exactly what the model code compares against
    assert a.autoencoder_vq == "True" and a.autoencoder_att == "False" and a.Modality_Gesture == "True"
    assert a.autoencoder_vq_components == "512" and float(a.autoencoder_vq_commitment_cost) == 0.25
    assert a.learning_rate == 0.0005 and a.epochs == 2


@pytest.mark.skipif(not os.path.isdir("/root/reference/config"), reason="reference tree not present")
@pytest.mark.parametrize("name", ["VQ-VAE.yml", "VQ-VAE_GENEA.yml", "DAE.yml", "seq2seq.yml"])
def test_parse_reference_yaml_files(name):
    """The reference's own YAMLs parse (including the ones that omit Modality_* and break the reference's parser)."""
    from config.parse_args import parse_args
    a = parse_args(["-c", os.path.join("/root/reference/config", name)])
    assert isinstance(a.hidden_size, int) and a.n_layers == 2
    assert a.autoencoder_vq in ("True", "False")
    assert isinstance(a.train_data_path, list) and isinstance(a.data_mean, list)


def _args(**kw):
    d = dict(rep_learning_dim=40, hidden_size=200, n_layers=2, dropout_prob=0.2, autoencoder_vae="False",
             autoencoder_vq="True", autoencoder_vq_components=512, autoencoder_vq_commitment_cost=0.25, n_pre_poses=1,
             autoencoder_conditioned="True", autoencoder_att="False", autoencoder_fixed_weight="False", n_poses=20)
    d.update(kw)
    return argparse.Namespace(**d)


def test_state_dict_keys_match_reference_layout(golden_dir):
    """Same keys / shapes as the reference module (keys taken from the golden fixture captured from the reference)."""
    import numpy as np
    from model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
    fx = np.load(os.path.join(golden_dir, "vqvae_tiny.npz"))
    ref = {k[3:]: fx[k].shape for k in fx.files if k.startswith("w0/")}
    net = Autoencoder_VQVAE(_args(rep_learning_dim=135, hidden_size=64, dropout_prob=0.0, autoencoder_vq_components=64,
                                  n_poses=34), 135, 34)
    mine = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert set(mine) == set(ref)
    for k in ref:
        assert mine[k] == tuple(ref[k]), k
    n_params = sum(p.numel() for p in Autoencoder_VQVAE(_args(), 40, 20).parameters())
    assert n_params == 2_330_280          # parameter count with the EMA quantiser


def test_cpu_forward_fails_loudly():
    from model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
    net = Autoencoder_VQVAE(_args(), 40, 20)
    x = torch.zeros(2, 20, 40)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(x, x)


def test_unsupported_configs_are_refused():
    from model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
    with pytest.raises(NotImplementedError):
        Autoencoder_VQVAE(_args(autoencoder_vae="True"), 40, 20)


def test_attention_model_surface():
    """autoencoder_att == "True" builds the reference's parameter set (attn.attn.*, attn.v, pre_linear over D + H inputs,
    :466-480) and the module names SURVEY.md 8(b) lists are importable with a step-level forward."""
    import model.Autoencoder_VQVAE_model as M
    net = M.Autoencoder_VQVAE(_args(autoencoder_att="True"), 40, 20)
    sd = net.state_dict()
    H = net.hidden_size
    assert sd["decoder.decoder.attn.attn.weight"].shape == (H, 2 * H) and sd["decoder.decoder.attn.v"].shape == (H,)
    assert sd["decoder.decoder.pre_linear.0.weight"].shape == (H, 40 + H)
    for name in ("EncoderRNN", "Attn", "BahdanauAttnDecoderRNN", "Generator", "Autoencoder_VQVAE", "VQ_Payam", "VQ_Payam_EMA",
                 "VQ_Payam_GSSoft", "VectorQuantizer", "VectorQuantizerEMA"):
        assert hasattr(M, name), name
    for cls in (M.Attn, M.BahdanauAttnDecoderRNN, M.Generator):
        assert "forward" in cls.__dict__, cls.__name__


def test_checkpoint_roundtrip(tmp_path):
    import utils.train_utils as tu
    from model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
    args = _args(rep_learning_dim=12, hidden_size=16, autoencoder_vq_components=32, n_poses=6)
    net = Autoencoder_VQVAE(args, 12, 6)
    path = os.path.join(tmp_path, "m_checkpoint_001.bin")
    tu.save_checkpoint({"args": args, "epoch": 1, "lang_model": None, "pose_dim": 12, "gen_dict": net.state_dict()}, path)
    a2, net2, loss_fn, lang, pose_dim = tu.load_checkpoint_and_model(path, "cpu", "autoencoder_vq")
    assert pose_dim == 12 and lang is None and not net2.training
    for k, v in net.state_dict().items():
        assert torch.equal(v, net2.state_dict()[k]), k
    with pytest.raises(NotImplementedError):
        tu.load_checkpoint_and_model(path, "cpu", "c2g")


def test_vector_quantizer_passthrough_like_the_reference():
    """`VectorQuantizer.forward` returns on its first statement in the reference (:1617): identity + zero scalars."""
    from model.Autoencoder_VQVAE_model import VectorQuantizer
    q = VectorQuantizer(16, 8, 0.25)
    assert set(q.state_dict()) == {"pre_lin.weight", "pre_lin.bias", "_embedding.weight"}
    assert float(q._embedding.weight.abs().max()) <= 1 / 16
    x = torch.randn(2, 3, 4)
    loss, out, perp, enc = q(x)
    assert out is x and int(loss) == 0 and int(perp) == 0 and int(enc) == 0
