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


class CacheLoader:
    """What the trainers iterate over: `len()` = batches per epoch, `iter()` = a fresh pass over `make_batches(epoch_seed)`.
    Stands where the reference builds `DataLoader(dataset, batch_size, shuffle=True, drop_last=True, num_workers=...)`
    (train_autoencoder_VQVAE.py:639-653): the per-item work left on the host is a memcpy-sized normalisation, the device work is
    one launch per batch, so no worker processes are needed."""

    def __init__(self, n_items: int, batch_size: int, make_batches, shuffle: bool = True, drop_last: bool = True, seed: int = 0):
        self.n_items, self.batch_size, self.make_batches = n_items, batch_size, make_batches
        self.shuffle, self.drop_last, self.seed, self.epoch = shuffle, drop_last, seed, 0

    def __len__(self) -> int:
        return self.n_items // self.batch_size if self.drop_last else -(-self.n_items // self.batch_size)

    def __iter__(self):
        self.epoch += 1
        return iter(self.make_batches(self.batch_size, self.shuffle, self.seed + self.epoch, self.drop_last))


class TrinityDataset_DAE:
    """Part-a frame dataset (reference lmdb_data_loader.py:272-508): the frames of the cached chunks, normalised with
    `(x - mean) / clip(std, 0.01)` (:380-382), as (noisy, original) = (x, x) of shape (D, 1) -- the DAE's own dropout is the
    noise (:388-389).  All frames are kept in ONE float32 array (the reference keeps a Python list of per-frame dicts).

    Epoch length.  The reference's `__len__` returns the number of LMDB ENTRIES (chunks, :357-364) while `__getitem__` indexes the
    per-FRAME list `all_poses[idx]` (:366-390): one of its epochs visits only the first `n_chunks` frames.  `all_frames=False`
    (the default here, and what `scripts/train_DAE.py` uses) reproduces exactly that -- same samples per epoch, same updates per
    epoch, same 20-epoch checkpoint cadence; `all_frames=True` iterates every frame of every chunk (n_poses times longer
    epochs; a deliberate deviation, for training the DAE on all the data)."""

    def __init__(self, args, lmdb_dir: str, n_poses: int, subdivision_stride: int, pose_resampling_fps: int,
                 data_mean: Sequence[float], data_std: Sequence[float], all_frames: bool = False):
        preloaded_dir = lmdb_dir + "_cache"
        if not os.path.exists(preloaded_dir):
            raise FileNotFoundError(f"{preloaded_dir}: the sample cache is produced by the reference's DataPreprocessor")
        chunks = TrinityChunks(preloaded_dir)
        mean = np.array(data_mean, dtype=np.float64).squeeze()
        std = np.clip(np.array(data_std, dtype=np.float64).squeeze(), a_min=0.01, a_max=None)
        frames = [((chunks[i][1] - mean) / std).astype(np.float32) for i in range(len(chunks))]
        self.frames = torch.from_numpy(np.concatenate(frames, axis=0)) if frames else torch.zeros(0, mean.size)
        self.n_frames_total = self.frames.shape[0]
        self.n_samples = self.n_frames_total if all_frames else min(len(chunks), self.n_frames_total)

    def __len__(self) -> int:
        return self.n_samples

    def __getitem__(self, idx: int):
        x = self.frames[idx].reshape(-1, 1)
        return x, x

    def batches(self, batch_size: int, device, shuffle: bool = True, seed: int = 0, drop_last: bool = True):
        order = torch.randperm(self.n_samples, generator=torch.Generator().manual_seed(seed)) if shuffle else torch.arange(self.n_samples)
        for s in range(0, self.n_samples, batch_size):
            ids = order[s:s + batch_size]
            if drop_last and len(ids) < batch_size:
                break
            x = self.frames[ids].unsqueeze(2).to(device, non_blocking=True)
            yield x, x


class TrinityDataset_sentencelevel:
    """Part-d sentence dataset (reference lmdb_data_loader.py:1045-1313) + `word_seq_collate_fn` (:29-122) in one place.

    Cached sample = `[word_seq, pose_seq, audio_raws, audio_mels, aux_info, sentence_leve_latents (S, L*H), GPT3_Embedding]`
    (:1209-1221).  Per item the reference normalises the poses, maps words to vocabulary ids up to `aux_info["end_time"]`
    (:1223-1236: no SOS / EOS) and -- on the CPU, inside `__getitem__` -- runs the frozen VQ-VAE's `vq_layer` on the item's (S, E)
    latent rows and takes `argmax(encodings)` (:1274-1281).  Here the per-item part stops before the quantiser; `batches()`
    collates B items (sorted by word count, descending, ids padded with 0 -- `pad_sequence`), moves the (B, S, E) latents to the
    device once and assigns ALL B*S rows with ONE launch of the VQ-VAE's own assignment kernel (`VQ_Payam_EMA.assign`: the
    same codes), then yields the collate function's 8-tuple
    `(word_seq, words_lengths, poses_seq, audio, aux_info, sentence_leve_latents, cluster_ids, GPT3_Embedding)`."""

    N_FIELDS = 7          # [word_seq, pose_seq, audio_raws, audio_mels, aux_info, sentence_leve_latents, GPT3_Embedding] (:1209-1221)

    @staticmethod
    def cache_dir(args, lmdb_dir: str) -> str:
        """Where the reference's dataset opens its cache (lmdb_data_loader.py:1107-1127): NOT `<lmdb_dir>_cache` (that is the 4-field
        chunk cache of Parts a / b) but `args.model_save_path + "lmdb/" + basename(lmdb_dir) + "_sentence_level" + "_cache"`
        (`_cache` alone when `args.sentence_level != "True"`), with `model_save_path` already reset by the trainer's main to
        `dirname(autoencoder_checkpoint) + "/text2mbedding/"` (train_text2embedding.py:506-508; the string concatenation, not a
        path join, is the reference's).  An `args` without `model_save_path` (direct construction in tests / notebooks) falls back
        to `<lmdb_dir>_sentence_level_cache` next to the data."""
        suffix = "_sentence_level_cache" if str(getattr(args, "sentence_level", "True")) == "True" else "_cache"
        msp = getattr(args, "model_save_path", None)
        if msp:
            return msp + "lmdb/" + os.path.basename(lmdb_dir) + suffix
        return lmdb_dir + suffix

    def __init__(self, args, lmdb_dir: str, n_poses: int, subdivision_stride: int, pose_resampling_fps: int,
                 data_mean: Sequence[float], data_std: Sequence[float], lang_model=None, vq_net=None):
        preloaded_dir = self.cache_dir(args, lmdb_dir)
        if not os.path.exists(preloaded_dir):
            raise FileNotFoundError(f"{preloaded_dir}: the sentence-level cache is produced by the reference's DataPreprocessor "
                                    "(clip slicing + per-chunk VQ-VAE latents; outside the hot path)")
        self.env = LMDBReader(preloaded_dir)
        self.n_samples = len(self.env)
        self.data_mean = np.array(data_mean, dtype=np.float64).squeeze()
        self.data_std = np.array(data_std, dtype=np.float64).squeeze()
        self.lang_model, self.vq_net = lang_model, vq_net

    def set_lang_model(self, lang_model) -> None:
        self.lang_model = lang_model

    def __len__(self) -> int:
        return self.n_samples

    def __getitem__(self, idx: int):
        """-> (word ids (Tw,) int64, poses (n, D) float32, audio float32, aux_info, latents (S, E) float32, GPT3 embedding)"""
        raw = self.env.get(sample_key(idx))
        if raw is None:
            raise IndexError(idx)
        sample = deserialize(raw)
        if len(sample) < self.N_FIELDS:
            raise ValueError(f"sample {idx} has {len(sample)} fields, a sentence-level sample has {self.N_FIELDS} (words, poses, audio, "
                             "mels, aux_info, per-chunk latents, GPT3 embedding): is this the 4-field CHUNK cache of Parts a / b?")
        word_seq, pose_seq, _audio_raws, audio_mels, aux_info, latents, gpt3 = sample[:7]
        std = np.clip(self.data_std, a_min=0.01, a_max=None)
        pose = np.asarray(pose_seq)
        pose = torch.from_numpy(((pose - self.data_mean) / std)).reshape(pose.shape[0], -1).float()
        ids = []
        for w in word_seq:
            if aux_info.get("end_time") is not None and w[1] > aux_info["end_time"]:
                break
            ids.append(self.lang_model.get_word_index(w[0]))
        lat = torch.from_numpy(np.array(latents, dtype=np.float32)).reshape(np.asarray(latents).shape[0], -1)
        audio = torch.from_numpy(np.array(audio_mels, dtype=np.float32))
        try:
            gpt3 = torch.from_numpy(np.array(gpt3, dtype=np.float32))
        except Exception:
            gpt3 = torch.zeros(1)
        return torch.tensor(ids, dtype=torch.int64), pose, audio, aux_info, lat, gpt3

    def batches(self, batch_size: int, device, shuffle: bool = True, seed: int = 0, drop_last: bool = True):
        order = np.arange(self.n_samples)
        if shuffle:
            np.random.default_rng(seed).shuffle(order)
        for s in range(0, self.n_samples, batch_size):
            ids = order[s:s + batch_size]
            if drop_last and len(ids) < batch_size:
                break
            items = [self[int(i)] for i in ids]
            items.sort(key=lambda it: len(it[0]), reverse=True)            # pack_padded_sequence wants descending lengths (:75)
            lengths = torch.tensor([len(it[0]) for it in items], dtype=torch.int64)
            words = torch.nn.utils.rnn.pad_sequence([it[0] for it in items], batch_first=True).long()
            poses = torch.stack([it[1] for it in items])
            audio = torch.stack([it[2] for it in items]) if all(it[2].shape == items[0][2].shape for it in items) else items[0][2]
            aux = {k: [it[3][k] for it in items] for k in items[0][3]}
            lat = torch.stack([it[4] for it in items])                     # (B, S, E)
            gpt3 = torch.stack([it[5] for it in items]) if all(it[5].shape == items[0][5].shape for it in items) else items[0][5]
            lat_d = lat.to(device, non_blocking=True)
            B, S, E = lat_d.shape
            rows = lat_d.reshape(B * S, E).contiguous()
            assign = getattr(self.vq_net.vq_layer, "assign", None)         # one launch sequence for the batch
            if assign is not None:
                codes = assign(rows).view(B, S)
            else:                                                          # any other quantiser: the reference's own route (:1274-1281)
                with torch.no_grad():
                    codes = torch.argmax(self.vq_net.vq_layer(rows)[3], dim=1).view(B, S)
            yield words, lengths, poses, audio, aux, lat_d, codes, gpt3
