"""graph replay with side branches: which combination crashes (each case in a child process)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import argparse, os, sys, faulthandler
faulthandler.enable()
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np, torch
from gesture2vec_amd import ops
from gesture2vec_amd.flat import FlatClipAdam
from gesture2vec_amd.model.text2embedding_model import text2embedding_model
from gesture2vec_amd.train_eval.train_seq2seq import train_iter_text2embedding, GraphedText2EmbeddingStep
from train_text2embedding import SyntheticSentences
side, mode = int(sys.argv[1]), sys.argv[2]
ops.SIDE_BRANCHES = bool(side)
import gc
def run(B, att, V=3863, eager=3, warm=3):
    args = argparse.Namespace(hidden_size=200, n_layers=2, dropout_prob=0.2, autoencoder_vq_components=512, autoencoder_att=att,
                              n_pre_poses=1, n_poses=20, sentence_frame_length=120, text2_embedding_discrete="True", batch_size=B)
    torch.manual_seed(0)
    net = text2embedding_model(args, 512, 20, V, 300, np.random.RandomState(0).randn(V, 300).astype(np.float32), None).to("cuda:0")
    net.train(True)
    opt = FlatClipAdam(net.parameters(), lr=5e-4)
    data = list(SyntheticSentences(args, V, 1, seed=1))[0]
    ids, lengths, codes = data[0].to("cuda:0"), data[1], data[6].to("cuda:0")
    for _ in range(eager):
        train_iter_text2embedding(args, 1, ids, lengths, None, None, codes, None, net, opt)
    g = GraphedText2EmbeddingStep(args, net, opt, ids, lengths, codes, warmup=warm, static_lengths=True, check_every=0)
    print("captured", B, att, flush=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("ok", float(g.loss), flush=True)
    return g
if mode == "same":
    g = run(128, "False"); g = run(128, "False")
elif mode == "del":
    g = run(128, "False"); del g; gc.collect(); torch.cuda.synchronize(); g = run(128, "False")
elif mode == "big":
    g = run(128, "False"); g = run(4096, "False")
elif mode == "seq4":
    for att in ("False", "True"):
        for B in (128, 4096):
            g = run(B, att)
elif mode == "seq4fresh":
    for att in ("False", "True"):
        for B in (128, 4096):
            ops._side_state["streams"].clear()
            g = run(B, att)
elif mode == "seq4gc":
    for att in ("False", "True"):
        for B in (128, 4096):
            g = run(B, att); del g; gc.collect()
elif mode == "seq4cache":
    for att in ("False", "True"):
        for B in (128, 4096):
            g = run(B, att); torch.cuda.synchronize(); torch.cuda.empty_cache()
elif mode == "seq4sync":
    for att in ("False", "True"):
        for B in (128, 4096):
            g = run(B, att); torch.cuda.synchronize()
elif mode == "seq4del":
    for att in ("False", "True"):
        for B in (128, 4096):
            g = run(B, att); del g; gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
elif mode == "seq6big":      # six multi-stream graphs (B = 2048 forks the encoder's side branch), each captured while its predecessor is alive
    for k in range(6):
        g = run(2048, "True" if k %% 2 else "False", eager=1, warm=1)
elif mode == "att2":
    g = run(128, "True"); g = run(4096, "True")
elif mode == "big2":
    g = run(4096, "False"); g = run(4096, "True")
elif mode == "keepboth":
    g1 = run(128, "False"); g2 = run(128, "False"); g1.replay(); g2.replay(); torch.cuda.synchronize(); print("both ok")
''' % ROOT
modes = [(1, m) for m in sys.argv[1:]] or [(1, "seq4"), (1, "seq6big"), (1, "seq4fresh"), (1, "seq4gc"), (0, "seq4")]
bad = 0
for side, mode in modes:
    r = subprocess.run([sys.executable, "-c", CHILD, str(side), mode], capture_output=True, text=True, timeout=600)
    bad += r.returncode != 0
    print((side, mode), "rc", r.returncode, r.stdout.strip().replace("\n", " | "), "||", r.stderr.strip()[-500:].replace("\n", " | "))
sys.exit(1 if bad else 0)
