#!/usr/bin/env python3
"""Golden vectors for Autoencoder_VQVAE AS THE REFERENCE SHIPS IT, i.e. with the VQ_Payam_GSSoft override left in place
(model/Autoencoder_VQVAE_model.py:816-820), by IMPORTING the reference (build container only): two iterations of
train_iter_Autoencoder_VQ_seq2seq (train_eval/train_seq2seq.py:664-758) with recorded Dropout(0.95) masks, an eval-mode
forward, and a checkpoint file written the way train_autoencoder_VQVAE.py:234-242 writes it (torch.save of args Namespace,
epoch, a model.vocab.Vocab, pose_dim, gen_dict) for the checkpoint-interop test.

Outputs: vqvae_shipped.npz, vqvae_shipped_ckpt.bin (both data: tensors / pickled attribute dicts, no reference source)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf  # noqa: E402


def main():
    vq, dae, ts = mf._import_reference()
    from model.vocab import Vocab
    torch.set_num_threads(1)
    B, T, D, H, L, K = 16, 12, 40, 32, 2, 48
    args = mf.make_args(rep_learning_dim=D, hidden_size=H, n_layers=L, autoencoder_vq_components=K, n_poses=T, dropout_prob=0.0)
    torch.manual_seed(11)
    net = vq.Autoencoder_VQVAE(args, D, T)             # as shipped: vq_layer is VQ_Payam_GSSoft
    assert type(net.vq_layer).__name__ == "VQ_Payam_GSSoft"
    net.train(True)
    x = torch.randn(B, T, D, generator=torch.Generator().manual_seed(1234))
    optim = torch.optim.Adam(net.parameters(), lr=args.learning_rate, betas=(0.5, 0.999))
    fx = dict(mf.sd_np(net, "w0/"))
    fx["x"] = x.numpy().copy()
    fx["cfg"] = np.array([B, T, D, H, L, K, 2], dtype=np.int64)
    fx["cfg_f"] = np.array([float(args.autoencoder_vq_commitment_cost), 0.0, args.learning_rate, args.loss_l1_weight,
                            args.loss_cont_weight, args.loss_var_weight], dtype=np.float64)
    for step in (1, 2):
        cap = {}
        orig = ts.custom_loss

        def spy(output, target, a):
            cap["outputs"] = output.detach().numpy().copy()
            val = orig(output, target, a)
            cap["custom_loss"] = float(val)
            return val

        def vq_hook(mod, inp, out):
            cap["loss_vq"] = float(out[0])
            cap["quantized"] = out[1].detach().numpy().copy()

        h = net.vq_layer.register_forward_hook(vq_hook)
        ts.custom_loss = spy
        torch.manual_seed(700 + step)
        with mf.MaskRecorder() as rec:
            loss, perp = ts.train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
        ts.custom_loss = orig
        h.remove()
        assert len(rec.masks) == T - 1
        fx[f"s{step}/mask_dec"] = np.packbits(np.stack([m.reshape(B, D) for m in rec.masks]), axis=None)
        fx[f"s{step}/loss"] = np.float64(loss["loss"])
        fx[f"s{step}/perplexity"] = np.float64(float(perp))
        fx[f"s{step}/loss_vq"] = np.float64(cap["loss_vq"])
        fx[f"s{step}/custom_loss"] = np.float64(cap["custom_loss"])
        fx[f"s{step}/outputs"] = cap["outputs"]
        fx[f"s{step}/quantized"] = cap["quantized"]
        if step == 1:
            for n_, p_ in net.named_parameters():
                if p_.grad is not None:
                    fx[f"s1/grad/{n_}"] = p_.grad.detach().numpy().copy()
                else:
                    fx[f"s1/gradnone/{n_}"] = np.zeros(0, dtype=np.float32)
    fx.update(mf.sd_np(net, "wN/"))
    net.train(False)
    torch.manual_seed(4242)
    with torch.no_grad(), mf.MaskRecorder() as rec:
        outs, first_hidden, loss_vq, perp = net(x, x)
    fx["eval/mask_dec"] = np.packbits(np.stack([m.reshape(B, D) for m in rec.masks]), axis=None)
    fx["eval/outputs"] = outs.numpy().copy()
    fx["eval/first_hidden"] = first_hidden.numpy().copy()
    fx["eval/loss_vq"] = np.float64(float(loss_vq))
    fx["eval/perplexity"] = np.float64(float(perp))
    np.savez_compressed(os.path.join(HERE, "vqvae_shipped.npz"), **fx)

    lang = Vocab("words")
    for w in "the quick brown fox jumps over the lazy dog the end".split():
        lang.index_word(w)
    lang.word_embedding_weights = np.random.RandomState(3).randn(lang.n_words, 300).astype(np.float32)
    torch.save({"args": args, "epoch": 2, "lang_model": lang, "pose_dim": D, "gen_dict": net.state_dict()},
               os.path.join(HERE, "vqvae_shipped_ckpt.bin"))
    print("vqvae_shipped: losses", [float(fx[f"s{s}/loss"]) for s in (1, 2)], "perp", [float(fx[f"s{s}/perplexity"]) for s in (1, 2)],
          "gradnone", [k for k in fx if "gradnone" in k], "n_words", lang.n_words)


if __name__ == "__main__":
    main()
