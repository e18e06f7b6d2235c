"""Logging / checkpoint helpers with the reference's contract (scripts/utils/train_utils.py:43-175):
checkpoint = torch.save({"args", "epoch", "lang_model", "pose_dim", "gen_dict"[, "val_metrics_list", "loss_list"]}),
`load_checkpoint_and_model(path, device, what)` rebuilds the model through the matching train script's init_model.
The three model kinds on the accelerated path exist here ("autoencoder_vq", "DAE", "text2embedding"); others raise."""
from __future__ import annotations

import logging
import math
import os
import time
from logging.handlers import RotatingFileHandler

import torch


def set_logger(log_path: str = None, log_filename: str = "log") -> None:
    for h in logging.root.handlers[:]:
        logging.root.removeHandler(h)
    handlers = [logging.StreamHandler()]
    if log_path is not None:
        os.makedirs(log_path, exist_ok=True)
        handlers.append(RotatingFileHandler(os.path.join(log_path, log_filename), maxBytes=10 * 1024 * 1024, backupCount=5))
    logging.basicConfig(level=logging.DEBUG, format="%(asctime)s: %(message)s", handlers=handlers)


def as_minutes(s: float) -> str:
    m = math.floor(s / 60)
    return "%dm %ds" % (m, s - m * 60)


def time_since(since: float) -> str:
    return as_minutes(time.time() - since)


def save_checkpoint(state: dict, filename: str) -> None:
    torch.save(state, filename)
    logging.info("Saved the checkpoint")


def load_checkpoint_and_model(checkpoint_path, _device="cpu", what: str = ""):
    """-> (args, generator, loss_fn, lang_model, pose_dim); the model is returned in eval mode."""
    print("loading checkpoint {}".format(checkpoint_path))
    # the reference pickles an argparse.Namespace (and a Vocab) next to the weights: weights_only must be off
    checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    args, epoch = checkpoint["args"], checkpoint["epoch"]
    lang_model, pose_dim = checkpoint["lang_model"], checkpoint["pose_dim"]
    print("epoch {}".format(epoch))
    if what == "autoencoder_vq":
        from train_autoencoder_VQVAE import init_model as VQVAE_init
        # the reference as shipped trains the VQ_Payam_GSSoft override (Autoencoder_VQVAE_model.py:816-820): such
        # checkpoints carry mean_layer / logvar_layer and no EMA buffers; the EMA quantiser has _ema_w / _ema_cluster_size
        if "vq_layer.mean_layer.weight" in checkpoint["gen_dict"]:
            args.autoencoder_vq_quantizer = "gssoft"
        elif "vq_layer._ema_w" in checkpoint["gen_dict"]:
            args.autoencoder_vq_quantizer = "ema"
        generator, loss_fn = VQVAE_init(args, lang_model, pose_dim, "cpu")
        generator.load_state_dict(checkpoint["gen_dict"], strict=True)
        generator = generator.to(_device)
    elif what == "DAE":
        from train_DAE import init_model as DAE_init
        generator, loss_fn = DAE_init(args, lang_model, pose_dim, "cpu")
        generator.load_state_dict(checkpoint["gen_dict"], strict=True)
        generator = generator.to(_device)
    elif what == "text2embedding":
        from train_text2embedding import init_model as t2e_init
        generator, loss_fn = t2e_init(args, lang_model, pose_dim, "cpu")
        generator.load_state_dict(checkpoint["gen_dict"], strict=True)
        generator = generator.to(_device)
    else:
        raise NotImplementedError(f"load_checkpoint_and_model(what={what!r}) is outside the accelerated hot path")
    generator.train(False)
    return args, generator, loss_fn, lang_model, pose_dim
