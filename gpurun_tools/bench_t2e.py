"""text2embedding (Part d) train-iteration throughput, samples/s. Diagnostic / DESIGN.md only.
Eager = train_iter_text2embedding as the reference calls it (one host sync per step for loss.item()); graph = the same
kernel sequence (zero_grad -> forward -> CE -> backward -> clip+Adam) replayed from one hipGraph."""
import argparse, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np, torch
from gesture2vec_amd.flat import FlatClipAdam
from gesture2vec_amd.model.text2embedding_model import text2embedding_model
from gesture2vec_amd.train_eval.train_seq2seq import train_iter_text2embedding, GraphedText2EmbeddingStep
from train_text2embedding import SyntheticSentences
out = []
for att in ("False", "True"):
    for B in (128, 4096):
        args = argparse.Namespace(hidden_size=200, n_layers=2, dropout_prob=0.2, autoencoder_vq_components=512, autoencoder_att=att,
                                  n_pre_poses=1, n_poses=20, sentence_frame_length=120, text2_embedding_discrete="True", batch_size=B)
        torch.manual_seed(0)
        net = text2embedding_model(args, 512, 20, 3863, 300, np.random.RandomState(0).randn(3863, 300).astype(np.float32), None).to("cuda:0")
        net.train(True)
        opt = FlatClipAdam(net.parameters(), lr=5e-4)
        data = list(SyntheticSentences(args, 3863, 1, seed=1))[0]
        ids, lengths, codes = data[0].to("cuda:0"), data[1], data[6].to("cuda:0")
        for _ in range(3):
            train_iter_text2embedding(args, 1, ids, lengths, None, None, codes, None, net, opt)
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
        for _ in range(n):
            train_iter_text2embedding(args, 1, ids, lengths, None, None, codes, None, net, opt)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        rec = dict(att=att, B=B, eager_ms=round(dt / n * 1e3, 3), eager_samples_per_s=round(B * n / dt, 1))
        print("eager done", rec, file=sys.stderr, flush=True)
        try:
            g = GraphedText2EmbeddingStep(args, net, opt, ids, lengths, codes, static_lengths=True, check_every=0)
            print("captured", file=sys.stderr, flush=True)
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n):
                g.replay()
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            rec.update(graph_ms=round(dt / n * 1e3, 3), graph_samples_per_s=round(B * n / dt, 1), loss=round(float(g.loss), 4))
        except Exception as e:
            rec["graph_error"] = f"{type(e).__name__}: {e}"[:300]
        out.append(rec)
        g = None
print(json.dumps(out))
