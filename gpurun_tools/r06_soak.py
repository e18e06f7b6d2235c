"""soak: Part d graph replays at large batch (resident GRU kernels, side branches, in-kernel gather) -- the loss must fall and stay finite"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np, torch
from gesture2vec_amd.flat import FlatClipAdam
from gesture2vec_amd.model.text2embedding_model import text2embedding_model
from gesture2vec_amd.train_eval.train_seq2seq import GraphedText2EmbeddingStep
from train_text2embedding import SyntheticSentences
for att, B, n in (("False", 4096, 600), ("True", 2048, 400)):
    args = argparse.Namespace(hidden_size=200, n_layers=2, dropout_prob=0.2, autoencoder_vq_components=512, autoencoder_att=att,
                              n_pre_poses=1, n_poses=20, sentence_frame_length=120, text2_embedding_discrete="True", batch_size=B)
    torch.manual_seed(0)
    net = text2embedding_model(args, 512, 20, 3863, 300, np.random.RandomState(0).randn(3863, 300).astype(np.float32), None).to("cuda:0")
    net.train(True)
    opt = FlatClipAdam(net.parameters(), lr=5e-4)
    data = list(SyntheticSentences(args, 3863, 1, seed=1))[0]
    ids, lengths, codes = data[0].to("cuda:0"), data[1], data[6].to("cuda:0")
    g = GraphedText2EmbeddingStep(args, net, opt, ids, lengths, codes, static_lengths=True, check_every=64)
    losses = []
    for k in range(n):
        g.replay()
        if k % 50 == 0 or k == n - 1:
            losses.append(round(g.read_loss(), 4))
    ok = all(np.isfinite(losses)) and losses[-1] < losses[0] and all(torch.isfinite(p).all().item() for p in net.parameters())
    print(json.dumps({"att": att, "B": B, "replays": n, "losses": losses, "lost_replays": g.lost_replays, "ok": bool(ok)}), flush=True)
    del g, net, opt
