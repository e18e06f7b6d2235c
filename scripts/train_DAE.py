#!/usr/bin/env python3
"""Part a trainer -- drop-in for the reference's `scripts/train_DAE.py` on the MI355X kernels.

    python train_DAE.py --config=../config/DAE_synthetic.yml --synthetic

Same surface: `init_model(args, lang_model, pose_dim, _device)` -> (DAE_Network(input_motion_dim, hidden_size), MSE),
`train_epochs` (Adam(lr, betas=(0.5, 0.999)) :190-192, checkpoint every 20 epochs with the keys
`args, epoch, lang_model, pose_dim, gen_dict` and the name `<name>_H<hidden>_checkpoint_<epoch>.bin` :203-222),
`evaluate_testset` (mean MSE of the eval-mode reconstruction :314-371), `main`.  Only the plain denoising autoencoder
branch (`autoencoder_vq == autoencoder_vae == "False"`, config/DAE.yml) is on the accelerated path; `VQ_Frame` /
`VAE_Network` raise.  The LMDB frame dataset is not ported: `--synthetic` feeds (noisy, original) frame batches of shape
(B, input_motion_dim, 1) like `TrinityDataset_DAE` does (original = N(0,1) frames, noisy = original + 0.1 N(0,1))."""
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
from model.DAE_model import DAE_Network  # noqa: E402
from train_eval.train_seq2seq import train_iter_DAE  # noqa: E402
import utils.train_utils  # noqa: E402
from utils.average_meter import AverageMeter  # noqa: E402
from gesture2vec_amd import functional as Fn  # noqa: E402
from gesture2vec_amd.flat import FlatClipAdam  # noqa: E402

device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
Global_loss_train = {"loss": []}
Global_loss_eval = []


def init_model(args, lang_model, pose_dim: int, _device):
    if args.autoencoder_vq == "True" or args.autoencoder_vae == "True":
        raise NotImplementedError("VQ_Frame / VAE_Network (frame-level VQ / VAE) are outside the accelerated hot path")
    network = DAE_Network(args.input_motion_dim, args.hidden_size).to(_device)
    return network, Fn.mse_loss


class SyntheticFrames:
    """(noisy, original) frame batches shaped like the reference's DAE dataset items: (B, motion_dim, 1) fp32."""

    def __init__(self, args, n_batches: int, seed: int):
        self.B, self.D, self.n_batches, self.seed = args.batch_size, args.input_motion_dim, n_batches, seed

    def __len__(self):
        return self.n_batches

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for _ in range(self.n_batches):
            original = torch.randn(self.B, self.D, 1, generator=g)
            yield original + 0.1 * torch.randn(self.B, self.D, 1, generator=g), original


def evaluate_testset(test_data_loader, generator, loss_fn, args) -> float:
    generator.train(False)
    losses = AverageMeter("loss")
    start = time.time()
    with torch.no_grad():
        for noisy, original in test_data_loader:
            noisy, original = noisy.to(device), original.to(device)
            out_poses = generator(noisy)
            loss = loss_fn(out_poses, original)
            losses.update(loss.item(), original.size(0))
    generator.train(True)
    logging.info("[VAL] loss: {:.3f} / {:.1f}s".format(losses.avg, time.time() - start))
    Global_loss_eval.append(losses.avg)
    return losses.avg


def train_epochs(args, train_data_loader, test_data_loader, lang_model, pose_dim: int, trial_id=None) -> None:
    start = time.time()
    loss_meters = [AverageMeter("loss")]
    print_interval = max(1, int(len(train_data_loader) / 5))
    save_model_epoch_interval = 20
    generator, loss_fn = init_model(args, lang_model, pose_dim, device)
    gen_optimizer = FlatClipAdam(generator.parameters(), lr=args.learning_rate, betas=(0.5, 0.999))   # clip_grad_norm_(5) fused
    global_iter = 0
    for epoch in range(1, args.epochs + 1):
        evaluate_testset(test_data_loader, generator, loss_fn, args)
        if epoch % save_model_epoch_interval == 0 and epoch > 0:
            save_name = "{}/{}_H{}_checkpoint_{:03d}.bin".format(args.model_save_path, args.name, args.hidden_size, epoch)
            utils.train_utils.save_checkpoint({"args": args, "epoch": epoch, "lang_model": lang_model, "pose_dim": pose_dim,
                                               "gen_dict": generator.state_dict()}, save_name)
        iter_start_time = time.time()
        for iter_idx, (noisy, original) in enumerate(train_data_loader, 0):
            global_iter += 1
            batch_size = original.size(0)
            loss = train_iter_DAE(args, epoch, noisy.to(device), original.to(device), generator, gen_optimizer)
            for m in loss_meters:
                if m.name in loss:
                    m.update(loss[m.name], batch_size)
            Global_loss_train["loss"].append(loss["loss"])
            if (iter_idx + 1) % print_interval == 0:
                print_summary = "EP {} ({:3d}) | {:>8s}, {:.0f} samples/s | ".format(
                    epoch, iter_idx + 1, utils.train_utils.time_since(start), batch_size / (time.time() - iter_start_time))
                for m in loss_meters:
                    if m.count > 0:
                        print_summary += "{}: {:.3f}, ".format(m.name, m.avg)
                        m.reset()
                logging.info(print_summary)
            iter_start_time = time.time()


def main(config: dict) -> None:
    args = config["args"]
    if args.random_seed >= 0:
        torch.manual_seed(args.random_seed)
        np.random.seed(args.random_seed)
        random.seed(args.random_seed)
    os.makedirs(args.model_save_path, exist_ok=True)
    utils.train_utils.set_logger(args.model_save_path, os.path.basename(__file__).replace(".py", ".log"))
    logging.info("PyTorch version: {}".format(torch.__version__))
    logging.info(pprint.pformat(vars(args)))
    if getattr(args, "synthetic", False):
        n_batches = int(getattr(args, "synthetic_batches", 20))
        train = SyntheticFrames(args, n_batches, seed=args.random_seed if args.random_seed >= 0 else 0)
        test = SyntheticFrames(args, max(1, n_batches // 5), seed=4321)
    else:
        # the reference's frame dataset (train_DAE.py:258-285: TrinityDataset_DAE over <train_data_path[0]>_cache), read by
        # gesture2vec_amd/data (pure-Python LMDB + legacy-pyarrow readers, parity unpinned: DESIGN.md)
        from gesture2vec_amd.data.dataset import CacheLoader, TrinityDataset_DAE
        loaders = []
        for k, (paths, shuffle) in enumerate(((args.train_data_path, True), (args.val_data_path, False))):
            ds = TrinityDataset_DAE(args, lmdb_dir=paths[0], n_poses=args.n_poses, subdivision_stride=args.subdivision_stride,
                                    pose_resampling_fps=args.motion_resampling_framerate, data_mean=args.data_mean, data_std=args.data_std,
                                    all_frames=bool(getattr(args, "dae_all_frames", False)))
            loaders.append(CacheLoader(len(ds), args.batch_size,
                                       lambda bs, sh, seed, dl, ds=ds: ds.batches(bs, device, shuffle=sh, seed=seed, drop_last=dl),
                                       shuffle=shuffle, drop_last=True, seed=77 + k))
        train, test = loaders
    train_epochs(args, train, test, None, pose_dim=args.input_motion_dim)


if __name__ == "__main__":
    main({"args": parse_args()})
