"""CPU tests of the real-data reader: LMDB page format, legacy pyarrow serialisation, the cached
sample layout and the normalisation of TrinityDataset_DAEed_Autoencoder (lmdb_data_loader.py:600-674)."""
import argparse
import os
import random

import numpy as np
import pytest
import torch

from gesture2vec_amd.data import LMDBReader, deserialize, serialize, write_lmdb
from gesture2vec_amd.data.dataset import TrinityDataset_DAEed_Autoencoder, sample_key, write_cache


def test_lmdb_three_level_tree_and_overflow_pages(tmp_path):
    random.seed(3)
    items = {sample_key(i): os.urandom(random.choice([8, 40, 300])) for i in range(60000)}       # branch -> branch -> leaf
    items.update({b"big%04d" % i: os.urandom(random.choice([2100, 9180, 50000])) for i in range(40)})   # overflow pages
    d = str(tmp_path / "db")
    write_lmdb(d, items)
    r = LMDBReader(d)
    st = r.stat()
    assert st["entries"] == len(items) and st["depth"] >= 3 and st["overflow_pages"] > 0 and st["psize"] == 4096
    for k in random.sample(sorted(items), 500) + [b"big0000", b"big0039", sample_key(0), sample_key(59999)]:
        assert r.get(k) == items[k]
    for k in (b"", b"zzzz", sample_key(60000), b"big0040"):
        assert r.get(k) is None
    keys = [k for k, _ in r.items()]
    assert keys == sorted(items)


def test_meta_page_with_larger_txnid_wins(tmp_path):
    d = str(tmp_path / "db")
    write_lmdb(d, {b"a": b"1"})
    path = os.path.join(d, "data.mdb")
    raw = bytearray(open(path, "rb").read())
    # corrupt the entries count of meta 0 (txnid 0): the reader must use meta 1 (txnid 1)
    raw[16 + 24 + 48 + 40:16 + 24 + 48 + 48] = (12345).to_bytes(8, "little")
    open(path, "wb").write(raw)
    assert len(LMDBReader(d)) == 1


def test_legacy_pyarrow_roundtrip_of_the_reference_sample_layout():
    words = [["so", 0.12, 0.5], ["we", 0.5, 0.9], ["went", 1.0, 1.4], ["there", 1.5, 1.7]]
    poses = np.random.default_rng(0).standard_normal((34, 135)).astype(np.float16)
    aux = {"vid": "Recording_001", "start_frame_no": 0, "end_frame_no": 34, "start_time": 0.0, "end_time": 1.7}
    buf = serialize([words, poses, [0, 0], aux])
    assert int.from_bytes(buf[8:12], "little") == 1                     # header: one ndarray follows the IPC stream
    w2, p2, a2, x2 = deserialize(buf)
    assert w2 == words and a2 == [0, 0] and x2 == aux
    assert p2.dtype == np.float16 and np.array_equal(p2, poses)
    # the raw-clip layout of trinity_data_to_lmdb.py:117-137 (dict with a list of dicts holding two arrays)
    clip = {"vid": "v", "clips": [{"words": words, "poses": poses, "audio_raw": np.arange(7, dtype=np.int16)}]}
    c2 = deserialize(serialize(clip))
    assert c2["vid"] == "v" and np.array_equal(c2["clips"][0]["audio_raw"], np.arange(7, dtype=np.int16))
    assert deserialize(serialize((1, 2.5, None, True, b"by", "s"))) == (1, 2.5, None, True, b"by", "s")


def test_dataset_normalisation_and_batches(tmp_path):
    rng = np.random.default_rng(1)
    T, D, n = 20, 12, 10
    mean, std = rng.standard_normal(D), np.abs(rng.standard_normal(D)) * 0.5
    std[3] = 0.001                                                       # below the clip threshold (:641)
    samples = []
    for i in range(n):
        poses = (rng.standard_normal((T, D)) * 2).astype(np.float16)
        samples.append([[["w", 0.0, 0.1]] * 4, poses, [0], {"vid": "v", "start_frame_no": i, "end_frame_no": i + T,
                                                             "start_time": 0.0, "end_time": 1.0}])
    write_cache(str(tmp_path / "trn_cache"), samples)
    args = argparse.Namespace(use_derivative="False", rep_learning_dim=D)
    ds = TrinityDataset_DAEed_Autoencoder(args, str(tmp_path / "trn"), T, 10, 20, mean, std, rep_model=None)
    assert len(ds) == n
    ref = (samples[4][1] - mean) / np.clip(std, 0.01, None)
    got = ds[4]
    assert got.dtype == torch.float32 and got.shape == (T, D)
    np.testing.assert_array_equal(got.numpy(), torch.from_numpy(ref).float().numpy())
    xs = list(ds.batches(4, "cpu", shuffle=False))                       # ablation path (no DAE encoder): pure host logic
    assert len(xs) == 2 and xs[0][0].shape == (4, T, D) and xs[0][0] is xs[0][1]
    ds.use_derivative = True
    e, _ = next(ds.batches(4, "cpu", shuffle=False))
    assert e.shape == (4, T, 2 * D) and torch.equal(e[:, 0, D:], torch.zeros(4, D))
    torch.testing.assert_close(e[:, 1:, D:], e[:, 1:, :D] - e[:, :-1, :D])
    with pytest.raises(FileNotFoundError):
        TrinityDataset_DAEed_Autoencoder(args, str(tmp_path / "missing"), T, 10, 20, mean, std)


def _chunk_samples(n, T, D, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        poses = (rng.standard_normal((T, D)) * 2).astype(np.float16)
        out.append([[["w", 0.0, 0.1]] * 4, poses, [0], {"vid": "v", "start_frame_no": i, "end_frame_no": i + T,
                                                        "start_time": 0.0, "end_time": 1.0}])
    return out


def test_frame_dataset_of_the_dae_trainer(tmp_path):
    """TrinityDataset_DAE (reference lmdb_data_loader.py:272-508): every frame of every chunk, normalised, (noisy, original) = (x, x)
    of shape (D, 1); the loader wrapper reshuffles per epoch and drops the ragged tail.  (parity unpinned: the cache is written by
    this repo's own writer)"""
    from gesture2vec_amd.data.dataset import CacheLoader, TrinityDataset_DAE
    T, D, n = 6, 5, 7
    rng = np.random.default_rng(2)
    mean, std = rng.standard_normal(D), np.abs(rng.standard_normal(D)) + 0.1
    samples = _chunk_samples(n, T, D, 3)
    write_cache(str(tmp_path / "trn_cache"), samples)
    # the reference's epoch: __len__ = LMDB entries (chunks), __getitem__ indexes the per-frame list (:357-390) -> the first n frames
    ref_ds = TrinityDataset_DAE(argparse.Namespace(), str(tmp_path / "trn"), T, 10, 20, mean, std)
    assert len(ref_ds) == n and ref_ds.n_frames_total == n * T
    assert torch.equal(torch.cat([b[0] for b in ref_ds.batches(4, "cpu", shuffle=False, drop_last=False)])[:, :, 0], ref_ds.frames[:n])
    ds = TrinityDataset_DAE(argparse.Namespace(), str(tmp_path / "trn"), T, 10, 20, mean, std, all_frames=True)
    assert len(ds) == n * T
    noisy, orig = ds[T + 2]                                              # chunk 1, frame 2
    ref = torch.from_numpy(((samples[1][1] - mean) / np.clip(std, 0.01, None))[2].astype(np.float32))
    assert noisy.shape == (D, 1) and torch.equal(noisy, orig) and torch.equal(noisy[:, 0], ref)
    loader = CacheLoader(len(ds), 8, lambda bs, sh, seed, dl: ds.batches(bs, "cpu", shuffle=sh, seed=seed, drop_last=dl), shuffle=True)
    assert len(loader) == (n * T) // 8
    e1 = [b[0] for b in loader]
    e2 = [b[0] for b in loader]
    assert len(e1) == len(loader) and e1[0].shape == (8, D, 1) and not torch.equal(torch.cat(e1), torch.cat(e2))     # reshuffled
    allf = torch.cat([b[0] for b in ds.batches(8, "cpu", shuffle=False, drop_last=False)])
    assert torch.equal(allf[:, :, 0], ds.frames)


def test_sentence_dataset_collates_like_word_seq_collate_fn(tmp_path):
    """TrinityDataset_sentencelevel + the collate function (reference lmdb_data_loader.py:1045-1313, :29-122): word ids up to
    aux_info['end_time'] (no SOS / EOS), sorted by length descending, padded with 0; latents (B, S, E); code ids from ONE call of the
    frozen quantiser's assign() over all B * S rows.  (parity unpinned, see above; the quantiser here is a CPU stand-in)"""
    from gesture2vec_amd.data.dataset import TrinityDataset_sentencelevel
    from gesture2vec_amd.data import write_lmdb
    rng = np.random.default_rng(5)
    S, E, D, n = 3, 8, 4, 5
    vocab = {"a": 4, "b": 5, "c": 6, "d": 7}
    lang = argparse.Namespace(get_word_index=lambda w: vocab.get(w, 3))
    samples, lens = [], [2, 4, 1, 3, 4]
    for i in range(n):
        words = [[("abcd"[j % 4]), 0.1 * j, 0.1 * j + 0.05] for j in range(lens[i])] + [["zzz", 9.0, 9.1]]     # past end_time: dropped
        samples.append([words, (rng.standard_normal((12, D))).astype(np.float16), [0], [[0.0, 1.0]],
                        {"vid": "v", "start_time": 0.0, "end_time": 5.0}, rng.standard_normal((S, E)).astype(np.float32),
                        np.zeros(3, dtype=np.float32)])
    write_lmdb(str(tmp_path / "trn_sentence_level_cache"), {sample_key(i): serialize(s) for i, s in enumerate(samples)})
    calls = []

    class FakeVQ:
        def assign(self, rows):
            calls.append(tuple(rows.shape))
            return (rows.sum(1) > 0).long() + 2 * (rows[:, 0] > 0).long()
    net = argparse.Namespace(vq_layer=FakeVQ())
    ds = TrinityDataset_sentencelevel(argparse.Namespace(), str(tmp_path / "trn"), 4, 10, 20, np.zeros(D), np.ones(D), lang_model=lang,
                                      vq_net=net)
    assert len(ds) == n
    ids0, pose0, _, aux0, lat0, _ = ds[1]
    assert ids0.tolist() == [4, 5, 6, 7] and pose0.shape == (12, D) and lat0.shape == (S, E) and aux0["vid"] == "v"
    (words, lengths, poses, audio, aux, lat, codes, gpt3), = list(ds.batches(5, "cpu", shuffle=False))
    assert calls == [(5 * S, E)]                                          # one assignment call for the whole batch
    assert lengths.tolist() == [4, 4, 3, 2, 1] and words.shape == (5, 4) and words.dtype == torch.int64
    assert words[3].tolist() == [4, 5, 0, 0] and words[4].tolist() == [4, 0, 0, 0]
    assert poses.shape == (5, 12, D) and lat.shape == (5, S, E) and codes.shape == (5, S) and len(aux["vid"]) == 5
    order = [1, 4, 3, 0, 2]                                               # stable sort by length, descending
    ref_lat = torch.from_numpy(np.stack([samples[i][5] for i in order]))
    assert torch.equal(lat, ref_lat) and torch.equal(codes, FakeVQ().assign(ref_lat.reshape(-1, E)).view(5, S))


def test_sentence_cache_location_arity_and_quantisers_without_assign(tmp_path):
    """Round-3 advisor findings.  (1) The dataset opens the cache where the reference's does (lmdb_data_loader.py:1107-1127):
    `model_save_path + "lmdb/" + basename(lmdb_dir) + "_sentence_level_cache"` (`_cache` with sentence_level != "True"), not the
    4-field chunk cache `<lmdb_dir>_cache`; (2) a chunk-cache sample is refused with a clear message; (3) a quantiser module with no
    assign() (any other nn.Module with the reference's 4-tuple forward) takes the reference's own route, argmax(encodings)."""
    from gesture2vec_amd.data.dataset import TrinityDataset_sentencelevel
    from gesture2vec_amd.data import write_lmdb
    S, E, D = 2, 4, 3
    rng = np.random.default_rng(9)
    msp = str(tmp_path / "ckpt") + "/text2mbedding/"
    args = argparse.Namespace(model_save_path=msp, sentence_level="True")
    assert TrinityDataset_sentencelevel.cache_dir(args, "/data/trn/lmdb_train") == msp + "lmdb/lmdb_train_sentence_level_cache"
    args.sentence_level = "False"
    assert TrinityDataset_sentencelevel.cache_dir(args, "/data/trn/lmdb_train") == msp + "lmdb/lmdb_train_cache"
    args.sentence_level = "True"
    with pytest.raises(FileNotFoundError, match="lmdb_train_sentence_level_cache"):
        TrinityDataset_sentencelevel(args, "/data/trn/lmdb_train", 4, 10, 20, np.zeros(D), np.ones(D))
    sent = [[["a", 0.0, 0.1]], rng.standard_normal((6, D)).astype(np.float16), [0], [[0.0]], {"vid": "v", "end_time": 5.0},
            rng.standard_normal((S, E)).astype(np.float32), np.zeros(2, dtype=np.float32)]
    os.makedirs(msp + "lmdb")
    write_lmdb(msp + "lmdb/lmdb_train_sentence_level_cache", {sample_key(0): serialize(sent), sample_key(1): serialize(sent)})
    lang = argparse.Namespace(get_word_index=lambda w: 4)

    class NoAssign(torch.nn.Module):          # the reference's quantiser contract only: forward -> (loss, quantized, perplexity, encodings)
        def forward(self, rows):
            enc = torch.zeros(rows.shape[0], 5)
            enc[torch.arange(rows.shape[0]), (rows[:, 0] > 0).long() * 3] = 1.0
            return torch.tensor(0.0), rows, torch.tensor(0.0), enc
    ds = TrinityDataset_sentencelevel(args, "/data/trn/lmdb_train", 4, 10, 20, np.zeros(D), np.ones(D), lang_model=lang,
                                      vq_net=argparse.Namespace(vq_layer=NoAssign()))
    (_w, _l, _p, _a, _x, lat, codes, _g), = list(ds.batches(2, "cpu", shuffle=False))
    assert torch.equal(codes, ((lat[:, :, 0] > 0).long() * 3))
    # a chunk cache (4 fields) at that place: a clear error instead of an unpack failure
    write_cache(msp + "lmdb/lmdb_val_sentence_level_cache", _chunk_samples(1, 6, D, 1))
    bad = TrinityDataset_sentencelevel(args, "/data/val/lmdb_val", 4, 10, 20, np.zeros(D), np.ones(D), lang_model=lang)
    with pytest.raises(ValueError, match="4 fields"):
        bad[0]
