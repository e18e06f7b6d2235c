#!/usr/bin/env python3
"""Part d trainer -- drop-in for the reference's `scripts/train_text2embedding.py` on the MI355X kernels.

    python train_text2embedding.py --config=../config/seq2seq_synthetic.yml --synthetic

Same surface: `init_model(args, lang_model, pose_dim, _device)` (lang_model needs `.n_words` and
`.word_embedding_weights`), `train_epochs`, `evaluate_testset` (cross-entropy + code-usage perplexity), `main`;
Adam(lr, betas=(0.5, 0.999)) (:179-181), evaluation every epoch (:195), checkpoint every 10 epochs with the keys
`args, epoch, lang_model, pose_dim, gen_dict` (:202-221).  Without `--synthetic` the sentence-level cache is read
(`cached_sentence_loaders`: code ids from the frozen VQ-VAE in one device launch per batch); `--synthetic` feeds batches of
the collate function's 8-tuple shape (lmdb_data_loader.py:111-120) with random word ids / lengths / code ids (SURVEY.md §8d
config 4)."""
from __future__ import annotations

import logging
import os
import pprint
import random
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
for _p in (_HERE, _ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from config.parse_args import parse_args  # noqa: E402
from model.text2embedding_model import text2embedding_model  # noqa: E402
from train_eval.train_seq2seq import train_iter_text2embedding  # noqa: E402
import utils.train_utils  # noqa: E402
from utils.average_meter import AverageMeter  # noqa: E402
from gesture2vec_amd import ops  # noqa: E402
from gesture2vec_amd.flat import FlatClipAdam  # noqa: E402

device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
debug = False


def init_model(args, lang_model, pose_dim: int, _device):
    n_frames = args.n_poses
    if args.text2_embedding_discrete == "True":
        pose_dim = int(args.autoencoder_vq_components)
    generator = text2embedding_model(args, pose_dim, n_frames, lang_model.n_words, args.wordembed_dim,
                                     lang_model.word_embedding_weights).to(_device)
    return generator, None


class SyntheticSentences:
    """Batches shaped like word_seq_collate_fn's 8-tuple: (in_text, text_lengths, target_vec, in_audio, aux_info,
    sentence_level_latents, cluster_ids, GPT3_embeddings); only ids / lengths / cluster ids carry information."""

    def __init__(self, args, n_words: int, n_batches: int, seed: int, max_len: int = 20):
        self.B, self.n_words, self.n_batches, self.seed, self.max_len = args.batch_size, n_words, n_batches, seed, max_len
        self.S = args.sentence_frame_length // args.n_poses
        self.K = int(args.autoencoder_vq_components)

    def __len__(self):
        return self.n_batches

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for _ in range(self.n_batches):
            lengths = torch.randint(4, self.max_len + 1, (self.B,), generator=g).sort(descending=True).values
            Tw = int(lengths[0])
            ids = torch.zeros(self.B, Tw, dtype=torch.int64)                       # PAD = 0
            for b in range(self.B):
                ids[b, : lengths[b]] = torch.randint(4, self.n_words, (int(lengths[b]),), generator=g)
            codes = torch.randint(0, self.K, (self.B, self.S), generator=g)
            dummy = torch.zeros(self.B, 1)
            yield ids, lengths, dummy, dummy, {}, dummy, codes, dummy


def evaluate_testset(test_data_loader, generator, loss_fn, args):
    """-> (mean cross-entropy over ALL S decode slots, mean code-usage perplexity), reference :300-421."""
    generator.train(False)
    losses, perplexities = AverageMeter("loss"), AverageMeter("perplexity")
    start = time.time()
    with torch.no_grad():
        for data in test_data_loader:
            in_text, text_lengths, target_vec, in_audio, aux_info, latents, cluster_ids, gpt3 = data
            batch_size = target_vec.size(0)
            in_text, cluster_ids = in_text.to(device), cluster_ids.to(device)
            out_latents, _ = generator(in_text, text_lengths, None, cluster_ids, None, None)
            K = out_latents.shape[2]
            flat = out_latents.reshape(-1, K).contiguous()
            loss, _ = ops.cross_entropy_fwd_bwd(flat, cluster_ids.reshape(-1).to(torch.int64).contiguous(), want_grad=False)
            losses.update(float(loss[0]), batch_size)
            pred = ops.argmax_rows(flat)
            stats = ops.vq_stats(pred, torch.zeros((pred.numel(), 1), device=device), K)       # code-usage histogram
            sc = ops.vq_ema_update(stats, None, None, None, None, None, pred.numel(), pred.numel(), 1, K, 0.0, 0.0, 0.0, False)
            perplexities.update(float(sc[1]), batch_size)
    generator.train(True)
    logging.info("[VAL] loss: {:.3f} / {:.1f}s".format(losses.avg, time.time() - start))
    return losses.avg, perplexities.avg


def train_epochs(args, train_data_loader, test_data_loader, lang_model, pose_dim, trial_id=None):
    start = time.time()
    loss_meters = [AverageMeter("loss"), AverageMeter("var_loss")]
    print_interval = int(len(train_data_loader))
    save_model_epoch_interval = 10
    generator, loss_fn = init_model(args, lang_model, pose_dim, device)
    gen_optimizer = FlatClipAdam(generator.parameters(), lr=args.learning_rate, betas=(0.5, 0.999))
    val_metrics_list, loss_list = [], []
    for epoch in range(1, args.epochs + 1):
        val_metrics_list.append(evaluate_testset(test_data_loader, generator, loss_fn, args))
        if epoch % save_model_epoch_interval == 0 and epoch > 0:
            save_name = "{}/{}_checkpoint_{:03d}.bin".format(args.model_save_path, args.name, epoch)
            utils.train_utils.save_checkpoint(
                {"args": args, "epoch": epoch, "lang_model": lang_model, "pose_dim": pose_dim,
                 "gen_dict": generator.state_dict()}, save_name)
        iter_start_time = time.time()
        loss_epoch = AverageMeter("loss")
        for iter_idx, data in enumerate(train_data_loader, 0):
            in_text, text_lengths, target_vec, in_audio, aux_info, latents, cluster_ids, gpt3 = data
            batch_size = target_vec.size(0)
            in_text, cluster_ids = in_text.to(device), cluster_ids.to(device)
            loss = train_iter_text2embedding(args, epoch, in_text, text_lengths, None, None, cluster_ids, None, generator,
                                             gen_optimizer)
            loss_epoch.update(loss["loss"], batch_size)
            for m in loss_meters:
                if m.name in loss:
                    m.update(loss[m.name], batch_size)
            if (iter_idx + 1) % print_interval == 0:
                summary = "EP {} ({:3d}) | {:>8s}, {:.0f} samples/s | ".format(
                    epoch, iter_idx + 1, utils.train_utils.time_since(start), batch_size / (time.time() - iter_start_time))
                for m in loss_meters:
                    if m.count > 0:
                        summary += "{}: {:.3f}, ".format(m.name, m.avg)
                        m.reset()
                logging.info(summary)
            iter_start_time = time.time()
        loss_list.append(loss_epoch.avg)
    return generator, val_metrics_list, loss_list


def main(config: dict):
    args = config["args"]
    if not getattr(args, "synthetic", False):
        # as the reference's main does (train_text2embedding.py:506-508): checkpoints, the log and the sentence-level cache
        # (TrinityDataset_sentencelevel.cache_dir) all live next to the chunk VQ-VAE's checkpoint
        args.model_save_path = os.path.dirname(args.autoencoder_checkpoint) + "/text2mbedding/"
        os.makedirs(args.model_save_path, exist_ok=True)
    if args.random_seed >= 0:
        torch.manual_seed(args.random_seed)
        np.random.seed(args.random_seed)
        random.seed(args.random_seed)
    utils.train_utils.set_logger(args.model_save_path, os.path.basename(__file__).replace(".py", ".log"))
    logging.info(pprint.pformat(vars(args)))
    if not getattr(args, "synthetic", False):
        train_loader, test_loader, lang_model = cached_sentence_loaders(args)
        return train_epochs(args, train_loader, test_loader, lang_model, pose_dim=int(args.autoencoder_vq_components))
    n_words = 3863                                      # the vocabulary size hard-coded at text2embedding_model.py:919
    g = torch.Generator().manual_seed(7)
    lang_model = SimpleNamespace(n_words=n_words,
                                 word_embedding_weights=torch.randn(n_words, args.wordembed_dim, generator=g).numpy())
    nb = getattr(args, "synthetic_batches", 8)
    train_loader = SyntheticSentences(args, n_words, nb, seed=1234)
    test_loader = SyntheticSentences(args, n_words, max(1, nb // 4), seed=4321)
    return train_epochs(args, train_loader, test_loader, lang_model, pose_dim=int(args.autoencoder_vq_components))


def cached_sentence_loaders(args):
    """The reference's data side of Part d (train_text2embedding.py:510-570): `TrinityDataset_sentencelevel` over
    `<model_save_path>lmdb/<basename(train_data_path[0])>_sentence_level_cache` (and the same for `val_data_path[0]`;
    lmdb_data_loader.py:1107-1127), the frozen chunk VQ-VAE of `args.autoencoder_checkpoint` assigning the
    code ids, and the vocabulary pickled next to the data (`vocab_cache.pkl`, :552-556; built by the reference's `build_vocab` with
    fastText, which is outside the hot path).  Code assignment runs as ONE device launch per batch (gesture2vec_amd/data/dataset.py)
    instead of per item on the CPU."""
    import pickle
    from gesture2vec_amd.data.dataset import CacheLoader, TrinityDataset_sentencelevel
    vocab_path = os.path.join(os.path.split(args.train_data_path[0])[0], "vocab_cache.pkl")
    if not os.path.exists(vocab_path):
        raise SystemExit(f"{vocab_path}: the vocabulary cache is written by the reference's build_vocab (fastText); not found")
    with open(vocab_path, "rb") as f:
        lang_model = pickle.load(f)
    _a, vq_net, _l, _lang, _dim = utils.train_utils.load_checkpoint_and_model(args.autoencoder_checkpoint, device, "autoencoder_vq")
    vq_net.train(False)
    loaders = []
    for k, (paths, shuffle) in enumerate(((args.train_data_path, True), (args.val_data_path, False))):
        ds = TrinityDataset_sentencelevel(args, lmdb_dir=paths[0], n_poses=args.n_poses, subdivision_stride=args.subdivision_stride,
                                          pose_resampling_fps=args.motion_resampling_framerate, data_mean=args.data_mean,
                                          data_std=args.data_std, lang_model=lang_model, vq_net=vq_net)
        loaders.append(CacheLoader(len(ds), args.batch_size,
                                   lambda bs, sh, seed, dl, ds=ds: ds.batches(bs, device, shuffle=sh, seed=seed, drop_last=dl),
                                   shuffle=shuffle, drop_last=True, seed=99 + k))
    return loaders[0], loaders[1], lang_model


if __name__ == "__main__":
    _args = parse_args()
    os.makedirs(_args.model_save_path, exist_ok=True)
    main({"args": _args})
