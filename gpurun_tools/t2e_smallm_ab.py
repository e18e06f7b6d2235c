"""A/B: Part-d train iterations at batch argv[1] with the small-M dense-layer kernel admitted up to argv[2:] rows (measurement hook
g2v_linear_set_smallm_rows); prints ms per iteration and the last loss for each threshold."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np, torch
from gesture2vec_amd import _lib
from gesture2vec_amd.flat import FlatClipAdam
from gesture2vec_amd.model.text2embedding_model import text2embedding_model
from gesture2vec_amd.train_eval.train_seq2seq import train_iter_text2embedding
from train_text2embedding import SyntheticSentences
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for rows in [int(v) for v in sys.argv[2:]] or [1024, 4096]:
    _lib.load().g2v_linear_set_smallm_rows(rows)
    args = argparse.Namespace(hidden_size=200, n_layers=2, dropout_prob=0.2, autoencoder_vq_components=512, autoencoder_att="False",
                              n_pre_poses=1, n_poses=20, sentence_frame_length=120, text2_embedding_discrete="True", batch_size=B)
    torch.manual_seed(0)
    net = text2embedding_model(args, 512, 20, 3863, 300, np.random.RandomState(0).randn(3863, 300).astype(np.float32), None).to("cuda:0")
    net.train(True)
    opt = FlatClipAdam(net.parameters(), lr=5e-4)
    data = list(SyntheticSentences(args, 3863, 1, seed=1))[0]
    ids, lengths, codes = data[0].to("cuda:0"), data[1], data[6].to("cuda:0")
    for _ in range(5):
        out = train_iter_text2embedding(args, 1, ids, lengths, None, None, codes, None, net, opt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        out = train_iter_text2embedding(args, 1, ids, lengths, None, None, codes, None, net, opt)
    torch.cuda.synchronize()
    print(f"smallm_rows {rows}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/iter  {out}", flush=True)
