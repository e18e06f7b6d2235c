"""Config / flag surface of the reference (`config/parse_args.py:16-96`): the same keys, types and defaults, read from a
YAML file given with -c/--config plus `--key value` overrides on the command line.

`configargparse` is not available in this environment, so this is a small table-driven parser of its own.  Two
deliberate deviations: the three `Modality_*` keys are optional (four of the reference's six YAMLs
omit them and cannot be parsed by the reference's own parser), and a `--synthetic` switch selects the synthetic
(T, D) pose-chunk source used for benchmarking.  Boolean-like options stay STRINGS ("True"/"False"): the model code
compares them textually, exactly like the reference does."""
from __future__ import annotations

import argparse
import sys

import yaml

# (key, type, default, required)   -- list-valued keys are handled below
_SPEC = [
    ("name", str, "main", False), ("model_save_path", str, None, True), ("random_seed", int, -1, False),
    ("wordembed_path", str, None, False), ("wordembed_dim", int, 200, False), ("sentence_level", str, None, True),
    ("sentence_frame_length", int, 120, False),
    ("model", str, None, True), ("epochs", int, 10, False), ("batch_size", int, 50, False),
    ("dropout_prob", float, 0.3, False), ("n_layers", int, 2, False), ("hidden_size", int, 200, False),
    ("autoencoder_denoising", str, None, True), ("autoencoder_att", str, None, True),
    ("autoencoder_fixed_weight", str, None, True), ("autoencoder_conditioned", str, None, True),
    ("use_derivative", str, None, True), ("autoencoder_checkpoint", str, None, True),
    ("autoencoder_vae", str, None, True), ("autoencoder_freeze_encoder", str, None, True),
    ("autoencoder_vq", str, None, True), ("autoencoder_vq_components", str, None, True),
    ("autoencoder_vq_commitment_cost", str, None, True), ("text2_embedding_discrete", str, None, True),
    ("use_similarity", str, None, True), ("similarity_labels", str, None, True), ("data_for_sim", str, None, True),
    ("loss_label_weight", float, None, False),
    ("motion_resampling_framerate", int, 24, False), ("n_poses", int, 50, False), ("n_pre_poses", int, 5, False),
    ("subdivision_stride", int, 5, False), ("subdivision_stride_sentence", int, 30, False),
    ("loader_workers", int, 4, False), ("input_motion_dim", int, 135, False),
    ("Modality_Audio", str, "False", False), ("Modality_Text", str, "False", False), ("Modality_Gesture", str, "True", False),
    ("learning_rate", float, 0.001, False), ("loss_l1_weight", float, 50, False), ("loss_cont_weight", float, 0.1, False),
    ("loss_var_weight", float, 0.01, False),
    ("rep_learning_checkpoint", str, "", False), ("rep_learning_dim", int, -1, False), ("noise_dim", int, 200, False),
    # not a reference key: "ema" (north star, fused step) or "gssoft" (the quantiser the reference's model actually ships
    # with, Autoencoder_VQVAE_model.py:816-820; module-level step)
    ("autoencoder_vq_quantizer", str, "ema", False),
]
_LIST_PATH_KEYS = ("train_data_path", "val_data_path", "test_data_path")     # action="append" in the reference
_LIST_FLOAT_KEYS = ("data_mean", "data_std")                                  # append + nargs="*": a list of lists
_REQUIRED_LISTS = ("train_data_path", "val_data_path")


def _coerce(key, typ, value):
    if value is None:
        return None
    if typ is str:
        return str(value)            # YAML `True` -> "True": downstream code compares strings
    return typ(value)


def parse_args(argv=None) -> argparse.Namespace:
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser(description="Gesture2Vec (MI355X) options")
    ap.add_argument("-c", "--config", required=True, help="YAML config file")
    ap.add_argument("--synthetic", action="store_true", help="train on synthetic N(0,1) pose chunks")
    ap.add_argument("--synthetic_batches", type=int, default=8, help="batches per epoch with --synthetic")
    ap.add_argument("--save_every", type=int, default=0, help="checkpoint interval in epochs (0 = the script's reference default)")
    ap.add_argument("--resume", type=str, default="", help="checkpoint written by this trainer to continue from (weights, EMA, "
                                                          "BatchNorm stats, Adam moments and the dropout RNG counter)")
    ap.add_argument("--dae_all_frames", action="store_true",
                    help="train_DAE.py: an epoch visits every frame of every cached chunk (the reference's epoch visits only the "
                         "first n_chunks frames: its __len__ counts LMDB entries, lmdb_data_loader.py:357-364)")
    for key, typ, _default, _req in _SPEC:
        ap.add_argument("--" + key, type=typ, default=None)
    for key in _LIST_PATH_KEYS:
        ap.add_argument("--" + key, action="append", default=None)
    for key in _LIST_FLOAT_KEYS:
        ap.add_argument("--" + key, action="append", type=float, nargs="*", default=None)
    cli = ap.parse_args(argv)
    with open(cli.config) as f:
        cfg = yaml.safe_load(f) or {}
    out = argparse.Namespace(config=cli.config, synthetic=cli.synthetic, synthetic_batches=cli.synthetic_batches,
                             save_every=cli.save_every, resume=cli.resume, dae_all_frames=cli.dae_all_frames)
    missing = []
    for key, typ, default, req in _SPEC:
        v = getattr(cli, key)
        if v is None and key in cfg:
            v = _coerce(key, typ, cfg[key])
        if v is None:
            v = default
            if req:
                missing.append(key)
        setattr(out, key, v)
    for key in _LIST_PATH_KEYS:
        v = getattr(cli, key)
        if v is None and key in cfg:
            c = cfg[key]
            v = [str(x) for x in c] if isinstance(c, (list, tuple)) else [str(c)]
        if v is None and key in _REQUIRED_LISTS and not cli.synthetic:
            missing.append(key)
        setattr(out, key, v)
    for key in _LIST_FLOAT_KEYS:
        v = getattr(cli, key)
        if v is None and key in cfg:
            v = [[float(x) for x in cfg[key]]]       # configargparse append+nargs semantics: one inner list
        setattr(out, key, v)
    if missing:
        ap.error("the following options are required (config file or command line): " + ", ".join(missing))
    return out
