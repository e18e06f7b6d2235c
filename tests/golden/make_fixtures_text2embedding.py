#!/usr/bin/env python3
"""Golden vectors for Part d (text -> gesture-code seq2seq) by IMPORTING the reference (build container only).

Exercises model/text2embedding_model.py::text2embedding_model (:488-746, EncoderRNN path) and
train_eval/train_seq2seq.py::train_iter_text2embedding (:462-538).  Adjustments from outside the reference:
`use_TCN = False` on the module before construction (as checked in, use_TCN=True makes forward crash);
recorded / replayed dropout masks exactly as in make_fixtures.py."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf  # noqa: E402


def main():
    vq, dae, ts = mf._import_reference()
    import model.text2embedding_model as t2e
    t2e.use_TCN = False
    torch.set_num_threads(1)
    for name, att, p in (("t2e_noatt", "False", 0.2), ("t2e_att", "True", 0.2)):
        H, L, K, NW, EMB, S, Tw, B = 32, 2, 64, 120, 300, 6, 12, 16
        args = mf.make_args(hidden_size=H, n_layers=L, dropout_prob=p, autoencoder_vq_components=K, autoencoder_att=att,
                            n_pre_poses=1, n_poses=20, sentence_frame_length=120, text2_embedding_discrete="True")
        torch.manual_seed(3)
        emb = torch.randn(NW, EMB).numpy()
        net = t2e.text2embedding_model(args, 135, args.n_poses, NW, EMB, emb, None)
        net.train(True)
        g = torch.Generator().manual_seed(77)
        lengths = torch.randint(4, Tw + 1, (B,), generator=g).sort(descending=True).values
        lengths[0] = Tw
        ids = torch.zeros(B, Tw, dtype=torch.int64)
        for b in range(B):
            ids[b, : lengths[b]] = torch.randint(4, NW, (int(lengths[b]),), generator=g)
        codes = torch.randint(0, K, (B, S), generator=g)
        fx = dict(mf.sd_np(net, "w0/"))
        fx.update(ids=ids.numpy(), lengths=lengths.numpy(), codes=codes.numpy(),
                  cfg=np.array([B, Tw, S, H, L, K, NW, EMB], dtype=np.int64), cfg_f=np.array([p, 5e-4]))
        optim = torch.optim.Adam(net.parameters(), lr=5e-4, betas=(0.5, 0.999))
        n_steps = 2
        for step in range(1, n_steps + 1):
            seed = 5000 + step
            cap = {}
            orig_fwd = net.forward

            def spy(*a, **k):
                out = orig_fwd(*a, **k)
                cap["outputs"] = out[0].detach().numpy().copy()
                return out

            net.forward = spy
            torch.manual_seed(seed)
            with mf.MaskRecorder() as rec:
                loss = ts.train_iter_text2embedding(args, 1, ids, lengths, None, None, codes, None, net, optim)
            net.forward = orig_fwd
            # F.dropout order: per decode step t = 1..S-1 the Dropout(0.5) on the code embedding (1,B,H)
            assert len(rec.masks) == S - 1
            fx[f"s{step}/mask_emb"] = np.stack([m.reshape(B, H) for m in rec.masks])
            # ATen-internal draws: encoder inter-layer dropout on the PACKED data (sum(lengths), 2H) first, then per
            # step the decoder GRU's inter-layer dropout (1,B,H) after that step's embedding dropout
            plan = [((int(lengths.sum()), 2 * H), p)]
            for _ in range(S - 1):
                plan += [((1, B, H), 0.5), ((1, B, H), p)]
            rp = mf.replay_gru_masks(seed, plan)
            for t in range(S - 1):
                assert np.array_equal(rp[1 + 2 * t].reshape(B, H), fx[f"s{step}/mask_emb"][t]), "RNG replay misaligned"
            fx[f"s{step}/mask_dec_l0"] = np.stack([rp[2 + 2 * t].reshape(B, H) for t in range(S - 1)])
            # encoder inter-layer mask: packed rows are time-major over the still-valid batch rows (lengths sorted
            # descending), i.e. packed index of (t, b) = sum_{t' < t} #{lengths > t'} + b.  Padded layout, pads = keep.
            enc = np.ones((Tw, B, 2 * H), dtype=bool)
            pk, off = rp[0].reshape(-1, 2 * H), 0
            for t in range(Tw):
                nb = int((lengths > t).sum())
                enc[t, :nb] = pk[off:off + nb]
                off += nb
            assert off == pk.shape[0]
            fx[f"s{step}/mask_enc_l0"] = enc
            fx[f"s{step}/loss"] = np.float64(loss["loss"])
            fx[f"s{step}/outputs"] = cap["outputs"]
            if step == 1:
                for n_, p_ in net.named_parameters():
                    if p_.grad is not None:
                        fx[f"s1/grad/{n_}"] = p_.grad.detach().numpy().copy()
                    else:
                        fx[f"s1/gradnone/{n_}"] = np.zeros(0, dtype=np.float32)
        fx.update(mf.sd_np(net, "wN/"))
        net.train(False)
        with torch.no_grad():
            out, _ = net(ids, lengths, None, codes, None, None)
        fx["eval/outputs"] = out.numpy().copy()
        # the inference branch (:685-692): one extra decode step fed with vid_indices produces outputs[0]
        vid = torch.randint(0, K, (B,), generator=torch.Generator().manual_seed(99))
        with torch.no_grad():
            out_v, _ = net(ids, lengths, None, codes, None, vid)
        fx["eval_vid/vid_indices"] = vid.numpy()
        fx["eval_vid/outputs"] = out_v.numpy().copy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
        print(name, "losses", [float(fx[f"s{s}/loss"]) for s in (1, 2)],
              "grad keys", sum(k.startswith("s1/grad/") for k in fx), "none", [k for k in fx if k.startswith("s1/gradnone/")])


def gen_new_classes():
    """text2embedding_model_New (+ EncoderRNN_New / DecoderRNN_New), reference :754-1002: forward in both teacher-forcing
    branches (python `random` seeded so that the coin flip is known), loss = sum of logits * fixed weights, every gradient.
    The classes hard-code batch 128 (initHidden) and vocabulary 3863 (:919); the 3863 x 300 embedding table is generated
    from a seed on both sides instead of being stored, its gradient is stored for the rows that were used."""
    import random
    vq, dae, ts = mf._import_reference()
    import model.text2embedding_model as t2e
    t2e.device = torch.device("cpu")
    torch.set_num_threads(1)
    H, K, B, Tw, S, NW = 32, 512, 128, 4, 3, 3863
    args = mf.make_args(hidden_size=H, autoencoder_vq_components=K)
    emb = np.random.RandomState(0).randn(NW, 300).astype(np.float32)
    torch.manual_seed(31)
    net = t2e.text2embedding_model_New(args, 135, 20, NW, 300, emb, None)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(4, NW, (B, Tw), generator=g)
    codes = torch.randint(0, K, (B, S), generator=g)
    wts = torch.randn(S, B, K + 2, generator=torch.Generator().manual_seed(6))     # regenerated from the seed by the tests
    fx = {k: v for k, v in mf.sd_np(net, "w0/").items() if "encoder.embedding" not in k}
    fx.update(ids=ids.numpy(), codes=codes.numpy(), cfg=np.array([H, K, B, Tw, S, NW], dtype=np.int64))
    used = np.unique(ids.numpy())
    fx["used_rows"] = used
    for tag, seed in (("tf", None), ("free", None)):
        # find a python-random seed whose first draw lands in the wanted branch (< 0.5 = teacher forcing)
        want_tf = tag == "tf"
        seed = next(s for s in range(100) if (random.Random(s).random() < 0.5) == want_tf)
        for p_ in net.parameters():
            p_.grad = None
        random.seed(seed)
        out = net(ids, None, codes, None)
        (out * wts).sum().backward()
        fx[f"{tag}/seed"] = np.int64(seed)
        fx[f"{tag}/out_even_rows"] = out.detach().numpy()[:, ::2].copy()      # every second batch row (fixture size)
        for n_, p_ in net.named_parameters():
            if p_.grad is None:
                fx[f"{tag}/gradnone/{n_}"] = np.zeros(0, dtype=np.float32)
            elif n_ == "encoder.embedding.weight":
                gfull = p_.grad.numpy()
                assert np.abs(np.delete(gfull, used, axis=0)).max() == 0
                fx[f"{tag}/grad/{n_}@used"] = gfull[used].copy()
            else:
                fx[f"{tag}/grad/{n_}"] = p_.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "t2e_new.npz"), **fx)
    print("t2e_new", {t: int(fx[f"{t}/seed"]) for t in ("tf", "free")}, "keys", len(fx),
          "none", [k for k in fx if "gradnone" in k])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "new":      # only the *_New tutorial classes
        gen_new_classes()
    else:
        main()
        gen_new_classes()
