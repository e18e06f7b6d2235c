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
