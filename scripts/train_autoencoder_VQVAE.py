#!/usr/bin/env python3
"""Part b trainer -- drop-in for the reference's `scripts/train_autoencoder_VQVAE.py` on the MI355X kernels.

    python train_autoencoder_VQVAE.py --config=../config/VQ-VAE_synthetic.yml --synthetic

Same surface: `init_model(args, lang_model, pose_dim, _device)`, `train_epochs(...)`, `evaluate_testset(...)`,
`main(config)`, `save_config`; Adam(lr, betas=(0.5, 0.999)); evaluation before training and every epoch; the epoch log
line `EP <epoch> (<iter>) | <elapsed>, <n> samples/s | loss: <avg>, `; checkpoint dict keys / file naming of
`train_autoencoder_VQVAE.py:223-244`.  Plotting / t-SNE / BVH inference shell-outs are visualisation only and omitted.
The LMDB datasets need third-party packages that are not available here; `--synthetic` feeds N(0,1) pose chunks
of the configured shape."""
from __future__ import annotations

import logging
import os
import pprint
import random
import sys
import time

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
for _p in (_HERE, _ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from config.parse_args import parse_args  # noqa: E402
from model.Autoencoder_VQVAE_model import Autoencoder_VQVAE  # noqa: E402
from train_eval.train_seq2seq import (FusedClipAdam, train_iter_Autoencoder_VQ_seq2seq,  # noqa: E402
                                      train_iter_Autoencoder_VQ_seq2seq_dp)
import utils.train_utils  # noqa: E402
from utils.average_meter import AverageMeter  # noqa: E402

# One process per GPU under `python -m torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE in the environment): the
# batch shards over the ranks, gradients + codebook EMA statistics are all-reduced over RCCL once per step
# (gesture2vec_amd/dp.py); rank 0 logs and writes checkpoints.  Without those variables: the single-GPU path.
_DIST = "RANK" in os.environ and "WORLD_SIZE" in os.environ
_RANK = int(os.environ.get("RANK", "0"))
_WORLD = int(os.environ.get("WORLD_SIZE", "1"))
_LOCAL_RANK = int(os.environ.get("LOCAL_RANK", "0"))
device = torch.device("cuda", _LOCAL_RANK) if torch.cuda.is_available() else torch.device("cpu")
debug = False


class _EvalMSE:
    """loss_fn returned by init_model (the reference returns torch.nn.MSELoss()); evaluated by the loss kernel."""

    def __call__(self, out_poses, target_poses):
        from gesture2vec_amd import ops
        terms, _ = ops.custom_loss_fwd_bwd(out_poses.transpose(0, 1).contiguous(), target_poses.contiguous(), 0.0, 0.0, 0.0,
                                           want_grad=False)
        return terms[4]


def init_model(args, lang_model, pose_dim: int, _device):
    n_frames = args.n_poses
    generator = Autoencoder_VQVAE(args, pose_dim, n_frames).to(_device)
    return generator, _EvalMSE()


class SyntheticChunks:
    """Iterable of (encoded_input, encoded_output) = (x, x) batches of N(0,1) pose chunks (B, n_poses, rep_learning_dim),
    the shape `TrinityDataset_DAEed_Autoencoder.__getitem__` yields (lmdb_data_loader.py:674)."""

    def __init__(self, args, n_batches: int, seed: int):
        self.shape = (args.batch_size, args.n_poses, args.rep_learning_dim)
        self.n_batches, self.seed = n_batches, seed

    def __len__(self):
        return self.n_batches

    def __iter__(self):
        # drawn on the training device: a CPU randn of one B=4096 batch (18.8 M floats) plus its 75 MB host-to-device copy
        # costs ~60 ms, 25x the train step it feeds
        g = torch.Generator(device=device).manual_seed(self.seed)
        for _ in range(self.n_batches):
            x = torch.randn(self.shape, generator=g, device=device)
            yield x, x


def cached_chunk_loaders(args):
    """The reference's data side of this trainer (train_autoencoder_VQVAE.py:617-653): `TrinityDataset_DAEed_Autoencoder` over
    `<train_data_path[0]>_cache` / `<val_data_path[0]>_cache` with the frozen frame DAE of `args.rep_learning_checkpoint` as
    `rep_model` (no checkpoint / rep_learning_dim == raw pose dim: the reference's ablation, lmdb_data_loader.py:650-651).  The
    caches are read by gesture2vec_amd/data (pure-Python LMDB + legacy-pyarrow readers, parity unpinned: DESIGN.md); the DAE
    encode runs as one device GEMM per batch instead of per item in DataLoader workers."""
    from gesture2vec_amd.data.dataset import CacheLoader, TrinityDataset_DAEed_Autoencoder
    rep_model = None
    ckpt = getattr(args, "rep_learning_checkpoint", "") or ""
    if ckpt:
        if not os.path.exists(ckpt):      # the reference fails here too (load_checkpoint_and_model): never train on un-encoded poses by accident
            raise FileNotFoundError(f"rep_learning_checkpoint {ckpt!r} does not exist (leave the option empty for the raw-pose "
                                    "ablation, lmdb_data_loader.py:650-651)")
        _a, rep_model, _l, _lang, _dim = utils.train_utils.load_checkpoint_and_model(ckpt, device, "DAE")
        rep_model.train(False)
    loaders = []
    for k, (paths, shuffle) in enumerate(((args.train_data_path, True), (args.val_data_path, False))):
        ds = TrinityDataset_DAEed_Autoencoder(args, lmdb_dir=paths[0], n_poses=args.n_poses, subdivision_stride=args.subdivision_stride,
                                              pose_resampling_fps=args.motion_resampling_framerate, data_mean=args.data_mean,
                                              data_std=args.data_std, rep_model=rep_model)
        # data parallel: every rank walks its own shuffle of the cache (rank-dependent seed), like the synthetic shards
        loaders.append(CacheLoader(len(ds), args.batch_size,
                                   lambda bs, sh, seed, dl, ds=ds: ds.batches(bs, device, shuffle=sh, seed=seed, drop_last=dl),
                                   shuffle=shuffle, drop_last=True, seed=1234 + 1000 * k + _RANK))
    return loaders[0], loaders[1]


def evaluate_testset(test_data_loader, generator, loss_fn, args) -> float:
    generator.train(False)
    losses = AverageMeter("loss")
    start = time.time()
    with torch.no_grad():
        for in_poses, target_poses in test_data_loader:
            batch_size = in_poses.size(0)
            in_poses, target_poses = in_poses.to(device), target_poses.to(device)
            out_poses, _latent, _loss_vq, _perp = generator(in_poses, target_poses)
            losses.update(float(loss_fn(out_poses, target_poses)), batch_size)
    generator.train(True)
    logging.info("[VAL] loss: {:.3f} / {:.1f}s".format(losses.avg, time.time() - start))
    return losses.avg


def train_epochs(args, train_data_loader, train_sim_dataset, test_data_loader, lang_model, pose_dim, trial_id=None):
    start = time.time()
    loss_meters = [AverageMeter("loss"), AverageMeter("var_loss")]
    print_interval = int(len(train_data_loader))
    save_model_epoch_interval = getattr(args, "save_every", 0) or args.epochs
    generator, loss_fn = init_model(args, lang_model, pose_dim, device)
    gen_optimizer = FusedClipAdam(generator, lr=args.learning_rate, betas=(0.5, 0.999))
    reduce_fn = None
    if _DIST:
        from gesture2vec_amd.dp import GradStatsAllReduce, broadcast_state
        generator.rng_seed = 1234 + _RANK                               # independent dropout masks per shard
        eng, vq = generator.engine(), generator.vq_layer
        if generator.quantizer == "ema":
            broadcast_state([eng.flat, vq._embedding.weight.data, vq._ema_w.data, vq._ema_cluster_size,
                             vq.pre_linear.weight.data, vq.pre_linear.bias.data, eng.bn_rm, eng.bn_rv])
        else:       # the soft quantiser (what the reference ships): every trainable tensor lives in the flat buffer; gradients only
            broadcast_state([eng.flat, vq.pre_linear.weight.data, vq.pre_linear.bias.data, eng.bn_rm, eng.bn_rv])
        reduce_fn = GradStatsAllReduce()
    val_metrics_list, loss_list = [], []
    first_epoch = 1
    if getattr(args, "resume", ""):
        # True resume: the reference's checkpoints carry weights only; ours add, under the extra key
        # "resume", the Adam moments / step and the dropout RNG counter, so that a continued run is bit-identical to an
        # uninterrupted one.  A checkpoint is written at the START of epoch `epoch` (reference order: evaluate, save, train).
        ck = torch.load(args.resume, map_location="cpu", weights_only=False)
        generator.load_state_dict(ck["gen_dict"], strict=True)
        eng = generator.engine()
        gen_optimizer.load_state_dict({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in ck["resume"]["optim"].items()})
        eng.rng_counter.copy_(ck["resume"]["rng_counter"].to(device))
        val_metrics_list, loss_list = list(ck.get("val_metrics_list", [])), list(ck.get("loss_list", []))
        first_epoch = int(ck["epoch"])
        logging.info("resumed from {} at epoch {}".format(args.resume, first_epoch))
    else:
        evaluate_testset(test_data_loader, generator, loss_fn, args)
    global_iter = 0
    for epoch in range(first_epoch, args.epochs + 1):
        if not (getattr(args, "resume", "") and epoch == first_epoch):     # the resumed epoch was evaluated before its save
            val_metrics_list.append(evaluate_testset(test_data_loader, generator, loss_fn, args))
        if (epoch % save_model_epoch_interval == 0 and epoch > 0 and _RANK == 0
                and not (getattr(args, "resume", "") and epoch == first_epoch)):
            save_name = "{}/{}_checkpoint_{:03d}.bin".format(args.model_save_path, args.name, epoch)
            utils.train_utils.save_checkpoint(
                {"args": args, "epoch": epoch, "lang_model": lang_model, "pose_dim": pose_dim,
                 "gen_dict": generator.state_dict(), "val_metrics_list": val_metrics_list, "loss_list": loss_list,
                 "resume": {"optim": {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in gen_optimizer.state_dict().items()},
                            "rng_counter": generator.engine().rng_counter.cpu().clone()}},
                save_name)
        iter_start_time = time.time()
        loss_epoch = AverageMeter("loss")
        for iter_idx, (encoded_input, encoded_output) in enumerate(train_data_loader, 0):
            global_iter += 1
            batch_size = encoded_output.size(0)
            encoded_input, encoded_output = encoded_input.to(device), encoded_output.to(device)
            if _DIST:
                loss, perplexity = train_iter_Autoencoder_VQ_seq2seq_dp(args, epoch, encoded_input, encoded_output, generator,
                                                                        gen_optimizer, reduce_fn, _WORLD)
            else:
                loss, perplexity = train_iter_Autoencoder_VQ_seq2seq(args, epoch, encoded_input, encoded_output, generator,
                                                                     gen_optimizer)
            loss_epoch.update(loss["loss"], batch_size)
            for m in loss_meters:
                if m.name in loss:
                    m.update(loss[m.name], batch_size)
            if (iter_idx + 1) % print_interval == 0 and _RANK == 0:
                summary = "EP {} ({:3d}) | {:>8s}, {:.0f} samples/s | ".format(
                    epoch, iter_idx + 1, utils.train_utils.time_since(start),
                    _WORLD * batch_size / (time.time() - iter_start_time))
                for m in loss_meters:
                    if m.count > 0:
                        summary += "{}: {:.3f}, ".format(m.name, m.avg)
                        m.reset()
                logging.info(summary)
            iter_start_time = time.time()
        loss_list.append(loss_epoch.avg)
    return generator, val_metrics_list, loss_list


def save_config(_args) -> None:
    os.makedirs(_args.model_save_path, exist_ok=True)
    with open(os.path.join(_args.model_save_path, "conf"), "w") as conf:
        for k, v in sorted(vars(_args).items()):
            conf.write("{}={}\n".format(k, v))


def main(config: dict):
    args = config["args"]
    if args.random_seed >= 0:
        torch.manual_seed(args.random_seed)
        np.random.seed(args.random_seed)
        random.seed(args.random_seed)
        os.environ["PYTHONHASHSEED"] = str(args.random_seed)
    if _DIST:
        import torch.distributed as dist
        torch.cuda.set_device(_LOCAL_RANK)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=_RANK, world_size=_WORLD, device_id=device)      # "nccl" is RCCL on ROCm
    utils.train_utils.set_logger(args.model_save_path if _RANK == 0 else None, os.path.basename(__file__).replace(".py", ".log"))
    logging.info("PyTorch version: {}".format(torch.__version__))
    logging.info("HIP version: {}".format(torch.version.hip))
    logging.info(pprint.pformat(vars(args)))
    if getattr(args, "synthetic", False):
        nb = getattr(args, "synthetic_batches", 8)
        train_loader = SyntheticChunks(args, nb, seed=1234 + _RANK)          # every rank draws its own shard
        test_loader = SyntheticChunks(args, max(1, nb // 4), seed=4321)
    else:
        train_loader, test_loader = cached_chunk_loaders(args)
    out = train_epochs(args, train_loader, None, test_loader, None, pose_dim=args.rep_learning_dim)
    if _DIST:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    _args = parse_args()
    if _args.use_derivative == "True":
        _args.rep_learning_dim = _args.rep_learning_dim * 2
    if _RANK == 0:
        save_config(_args)
    main({"args": _args})
