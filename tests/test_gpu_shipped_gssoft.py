"""Autoencoder_VQVAE exactly as the reference ships it (VQ_Payam_GSSoft override left in place, reference
model/Autoencoder_VQVAE_model.py:816-820) on the MI355X kernels: train iterations, eval forward and checkpoint interop
against golden vectors captured from the reference import (tests/golden/make_fixtures_shipped.py)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import g2v_oracle as O
from test_gpu_vqvae import fixture_args, relerr, state_from

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _shipped_args(fx):
    dims, args = fixture_args(fx)
    args.autoencoder_vq_quantizer = "gssoft"
    return dims, args


def test_shipped_model_train_steps_match_reference_golden(golden_dir):
    from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE, VQ_Payam_GSSoft
    from gesture2vec_amd.train_eval.train_seq2seq import FusedClipAdam, train_iter_Autoencoder_VQ_seq2seq
    fx = np.load(os.path.join(golden_dir, "vqvae_shipped.npz"))
    (B, T, D, H, L, K, n_steps), args = _shipped_args(fx)
    net = Autoencoder_VQVAE(args, D, T)
    assert isinstance(net.vq_layer, VQ_Payam_GSSoft)
    net.load_state_dict(state_from(fx, "w0/"), strict=True)          # identical key set, incl. the unused pre_linear
    net = net.to(DEV)
    net.train(True)
    optim = FusedClipAdam(net, lr=args.learning_rate, betas=(0.5, 0.999))
    x = torch.from_numpy(fx["x"].copy()).to(DEV)
    for step in range(1, n_steps + 1):
        net.set_dropout_masks(O.unpack_mask(fx[f"s{step}/mask_dec"], (T - 1, B, D)).to(DEV))
        loss, perp = train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
        eng = net.engine()
        b = eng.buffers(B)
        ref_loss = float(fx[f"s{step}/loss"])
        assert abs(loss["loss"] - ref_loss) <= 2e-5 * abs(ref_loss), (loss["loss"], ref_loss)
        assert abs(float(perp) - float(fx[f"s{step}/perplexity"])) <= 1e-4 * float(fx[f"s{step}/perplexity"])
        assert relerr(b["quant"], fx[f"s{step}/quantized"]) < 1e-4
        assert relerr(b["y"].transpose(0, 1), fx[f"s{step}/outputs"]) < 1e-4, "reconstructed poses"
        if step == 1:
            for k in fx.files:
                if k.startswith("s1/grad/"):
                    n = k[len("s1/grad/"):]
                    assert n in eng.offsets, f"{n} has a gradient in the reference but is not trainable here"
                    g, ref = eng.view(n, True), fx[k]
                    if n == "decoder.decoder.pre_linear.0.bias":      # mathematically zero (feeds BatchNorm)
                        assert float(g.abs().max()) < 1e-6 and np.abs(ref).max() < 1e-6
                    elif np.abs(ref).max() == 0:
                        assert float(g.abs().max()) == 0, n           # encoder GRU layer 1: exactly zero on both sides
                    else:
                        assert relerr(g, ref) < 5e-4, (n, relerr(g, ref))
                elif k.startswith("s1/gradnone/"):
                    assert k[len("s1/gradnone/"):] not in eng.offsets
    for n, ref in state_from(fx, "wN/").items():
        got = net.state_dict()[n]
        if n in ("decoder.decoder.pre_linear.0.bias", "decoder.decoder.pre_linear.1.running_mean"):
            assert float((got.cpu() - ref).abs().max()) <= 1.01 * n_steps * args.learning_rate, n
        elif ref.dtype.is_floating_point:
            err = float((got.cpu().double() - ref.double()).abs().max())
            assert err <= 1e-4 * float(ref.abs().max()) + 0.02 * args.learning_rate, (n, err)
        else:
            assert torch.equal(got.cpu(), ref), n


def _eval_check(net, fx, B, T, D):
    x = torch.from_numpy(fx["x"].copy()).to(DEV)
    net.train(False)
    net.set_dropout_masks(O.unpack_mask(fx["eval/mask_dec"], (T - 1, B, D)).to(DEV))
    with torch.no_grad():
        outs, first_hidden, loss_vq, perp = net(x, x)
    assert relerr(outs, fx["eval/outputs"]) < 1e-4
    assert relerr(first_hidden, fx["eval/first_hidden"]) < 1e-4
    assert abs(float(loss_vq) - float(fx["eval/loss_vq"])) <= 1e-4 * abs(float(fx["eval/loss_vq"]))
    assert abs(float(perp) - float(fx["eval/perplexity"])) <= 1e-4 * float(fx["eval/perplexity"])


def test_shipped_model_eval_forward_matches_reference_golden(golden_dir):
    from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
    fx = np.load(os.path.join(golden_dir, "vqvae_shipped.npz"))
    (B, T, D, H, L, K, _), args = _shipped_args(fx)
    net = Autoencoder_VQVAE(args, D, T)
    net.load_state_dict(state_from(fx, "wN/"), strict=True)
    _eval_check(net.to(DEV), fx, B, T, D)


def test_reference_checkpoint_file_loads_and_reproduces_eval(golden_dir):
    """A checkpoint written by the reference's own save path (pickled args Namespace + model.vocab.Vocab + gen_dict of the
    as-shipped GSSoft model) goes through load_checkpoint_and_model unchanged and reproduces the reference's eval forward."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from utils.train_utils import load_checkpoint_and_model
    fx = np.load(os.path.join(golden_dir, "vqvae_shipped.npz"))
    B, T, D = [int(v) for v in fx["cfg"][:3]]
    args, net, loss_fn, lang_model, pose_dim = load_checkpoint_and_model(os.path.join(golden_dir, "vqvae_shipped_ckpt.bin"),
                                                                        DEV, "autoencoder_vq")
    assert args.autoencoder_vq_quantizer == "gssoft" and pose_dim == D and not net.training
    assert lang_model.n_words == 13 and lang_model.get_word_index("fox") == lang_model.word2index["fox"]
    _eval_check(net, fx, B, T, D)
