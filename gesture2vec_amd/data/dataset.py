"""`TrinityDataset_DAEed_Autoencoder` (reference data_loader/lmdb_data_loader.py:533-674) on the pure-Python readers.

The reference's dataset opens `<lmdb_dir>_cache`, and per ITEM deserialises the sample, normalises the poses with
`(x - mean) / clip(std, 0.01)` (:640-642) and runs the frozen `rep_model.encoder` (the frame DAE's Linear+ReLU encoder) on
the CPU inside DataLoader worker processes (:647-653), returning `(encoded, encoded)` (:674).  Here the per-item work stops
at "normalised float32 chunk"; `batches()` stacks B of them, moves the batch to the GPU once and runs the DAE encoder as
ONE device GEMM per batch (`DAE_Network.encode`, HIP) -- same numbers, no per-item CPU encode.
`use_derivative` (:656-671) appends the frame-to-frame difference, as the reference does.

The cache must exist (it is produced from raw BVH / audio / subtitles by the reference's DataPreprocessor, which is outside
the hot path); `write_cache` builds one from in-memory samples for tests and synthetic runs."""
from __future__ import annotations

import os
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .arrow_legacy import deserialize, serialize
from .lmdb_format import LMDBReader, write_lmdb


def sample_key(idx: int) -> bytes:
    return "{:010}".format(idx).encode("ascii")        # :632


class TrinityChunks:
    """Raw cached samples: `[words, poses (n_poses, D) float16, audio, aux_info]` (data_preprocessor.py:326-333)."""

    def __init__(self, cache_dir: str):
        self.env = LMDBReader(cache_dir)
        self.n_samples = len(self.env)                 # txn.stat()["entries"] (:603-604)

    def __len__(self) -> int:
        return self.n_samples

    def __getitem__(self, idx: int):
        raw = self.env.get(sample_key(idx))
        if raw is None:
            raise IndexError(idx)
        sample = deserialize(raw)
        word_seq, pose_seq, audio, aux_info = sample[:4]
        return word_seq, np.asarray(pose_seq), audio, aux_info


class TrinityDataset_DAEed_Autoencoder:
    def __init__(self, args, lmdb_dir: str, n_poses: int, subdivision_stride: int, pose_resampling_fps: int,
                 data_mean: Sequence[float], data_std: Sequence[float], rep_model=None):
        self.lmdb_dir, self.n_poses = lmdb_dir, n_poses
        self.subdivision_stride, self.skeleton_resampling_fps = subdivision_stride, pose_resampling_fps
        self.data_mean = np.array(data_mean, dtype=np.float64).squeeze()
        self.data_std = np.array(data_std, dtype=np.float64).squeeze()
        self.use_derivative = str(getattr(args, "use_derivative", "False")) == "True"
        preloaded_dir = lmdb_dir + "_cache"
        if not os.path.exists(preloaded_dir):
            raise FileNotFoundError(f"{preloaded_dir}: the sample cache is produced by the reference's DataPreprocessor "
                                    "(BVH / audio / subtitle preprocessing is outside the hot path)")
        self.chunks = TrinityChunks(preloaded_dir)
        self.n_samples = len(self.chunks)
        self.rep_model = rep_model                      # a gesture2vec_amd DAE_Network (or None = ablation, :650-651)
        self.rep_learning_dim = getattr(args, "rep_learning_dim", None)

    def __len__(self) -> int:
        return self.n_samples

    def normalised(self, idx: int) -> torch.Tensor:
        """(n_poses, pose_dim) float32: (x - mean) / clip(std, 0.01)   (:640-646)"""
        _, pose_seq, _, _ = self.chunks[idx]
        std = np.clip(self.data_std, a_min=0.01, a_max=None)
        pose = (pose_seq - self.data_mean) / std          # float16 - float64 -> float64, as in the reference
        return torch.from_numpy(pose).float()

    def __getitem__(self, idx: int) -> torch.Tensor:
        return self.normalised(idx)

    def encode_batch(self, x: torch.Tensor) -> torch.Tensor:
        """(B,T,D_raw) normalised poses on the GPU -> (B,T,rep_dim [*2 with use_derivative]) through the frozen DAE encoder."""
        B, T, D = x.shape
        dae = self.rep_model
        if dae is not None and dae.encoder is not None:
            with torch.no_grad():
                enc = dae.encode(x.reshape(B * T, D).contiguous()).view(B, T, -1)
        else:
            enc = x
        if self.use_derivative:
            diff = torch.zeros_like(enc)
            diff[:, 1:] = enc[:, 1:] - enc[:, :-1]
            enc = torch.cat((enc, diff), dim=2)
        return enc

    def batches(self, batch_size: int, device, shuffle: bool = True, seed: int = 0, drop_last: bool = True
                ) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        """yields (encoded_poses, encoded_poses) like the reference's DataLoader over this dataset (:674)"""
        order = np.arange(self.n_samples)
        if shuffle:
            np.random.default_rng(seed).shuffle(order)
        for s in range(0, self.n_samples, batch_size):
            ids = order[s:s + batch_size]
            if drop_last and len(ids) < batch_size:
                break
            x = torch.stack([self.normalised(int(i)) for i in ids]).to(device, non_blocking=True)
            enc = self.encode_batch(x)
            yield enc, enc


def write_cache(cache_dir: str, samples: List[list]) -> None:
    """samples: list of `[words, poses, audio, aux_info]` -> an LMDB cache directory in the reference's layout"""
    write_lmdb(cache_dir, {sample_key(i): serialize(s) for i, s in enumerate(samples)})
