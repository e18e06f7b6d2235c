"""Part d graph replay at B = 2048 / 4096 with the resident BPTT on / off (G2V_OPT_GRU_RESIDENT_BWD), both attention settings"""
import argparse, gc, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np, torch
from gesture2vec_amd import _lib
from gesture2vec_amd.flat import FlatClipAdam
from gesture2vec_amd.model.text2embedding_model import text2embedding_model
from gesture2vec_amd.train_eval.train_seq2seq import GraphedText2EmbeddingStep
from train_text2embedding import SyntheticSentences
lib = _lib.load()
for att in ("False", "True"):
    for B in (2048, 4096):
        row = {"att": att, "B": B}
        for rep in (0, 1):
            for bwd in (1, 0):
                lib.g2v_ctx_set_option(None, _lib.OPT_GRU_RESIDENT_BWD, bwd)
                args = argparse.Namespace(hidden_size=200, n_layers=2, dropout_prob=0.2, autoencoder_vq_components=512, autoencoder_att=att,
                                          n_pre_poses=1, n_poses=20, sentence_frame_length=120, text2_embedding_discrete="True", batch_size=B)
                torch.manual_seed(0)
                net = text2embedding_model(args, 512, 20, 3863, 300, np.random.RandomState(0).randn(3863, 300).astype(np.float32), None).to("cuda:0")
                net.train(True)
                opt = FlatClipAdam(net.parameters(), lr=5e-4)
                data = list(SyntheticSentences(args, 3863, 1, seed=1))[0]
                ids, lengths, codes = data[0].to("cuda:0"), data[1], data[6].to("cuda:0")
                g = GraphedText2EmbeddingStep(args, net, opt, ids, lengths, codes, static_lengths=True, check_every=0)
                for _ in range(5):
                    g.replay()
                torch.cuda.synchronize(); t0 = time.perf_counter(); n = 40
                for _ in range(n):
                    g.replay()
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                row.setdefault(f"resident_bwd{bwd}", []).append(round(dt / n * 1e3, 4))
                del g, net, opt
                gc.collect()
        print(json.dumps(row), flush=True)
