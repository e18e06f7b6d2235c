"""Word <-> index table with the reference's pickled layout (scripts/model/vocab.py:21-260).

Reference checkpoints (`torch.save({"args", "epoch", "lang_model", "pose_dim", "gen_dict"})`, train_autoencoder_VQVAE.py
:234-242, train_text2embedding.py:202-221) pickle their `lang_model` as an instance of `model.vocab.Vocab`; unpickling only
needs a class of that import path whose instances accept the same attribute dict:
    name, trimmed, word_embedding_weights (np.float32 (n_words, dim) or None), word2index, word2count, index2word, n_words
and the class constants PAD/SOS/EOS/UNK = 0/1/2/3.  This module provides it, plus the small method surface the hot path's
callers use (index_word, add_vocab, trim, get_word_index).  Word vectors: the reference reads a FastText `.bin` through the
`fasttext` package (:166-190); that package is optional here (`load_word_vectors` says so if it is missing) and
`load_word_vectors_text` reads the plain-text `word v1 v2 ...` format instead."""
from __future__ import annotations

import logging
from typing import Dict, Optional

import numpy as np


class Vocab:
    PAD_token = 0
    SOS_token = 1
    EOS_token = 2
    UNK_token = 3

    def __init__(self, name: str, insert_default_tokens: bool = True):
        self.name = name
        self.trimmed = False
        self.word_embedding_weights: Optional[np.ndarray] = None
        self.reset_dictionary(insert_default_tokens)

    # ------------------------------------------------------------------ table
    def reset_dictionary(self, insert_default_tokens: bool = True) -> None:
        self.word2index: Dict[str, int] = {}
        self.word2count: Dict[str, int] = {}
        specials = ((self.PAD_token, "<PAD>"), (self.SOS_token, "<SOS>"), (self.EOS_token, "<EOS>"), (self.UNK_token, "<UNK>"))
        self.index2word: Dict[int, str] = dict(specials) if insert_default_tokens else {self.UNK_token: "<UNK>"}
        self.n_words = len(self.index2word)

    def index_word(self, word: str) -> None:
        """count one occurrence of `word`, giving it the next free index the first time it is seen"""
        idx = self.word2index.get(word)
        if idx is None:
            idx = self.n_words
            self.word2index[word] = idx
            self.index2word[idx] = word
            self.word2count[word] = 0
            self.n_words += 1
        self.word2count[word] += 1

    def add_vocab(self, other_vocab: "Vocab") -> None:
        for word in other_vocab.word2count:
            self.index_word(word)

    def trim(self, min_count: int) -> None:
        """drop the words seen fewer than min_count times (once); indices are re-assigned in first-seen order"""
        if self.trimmed:
            return
        self.trimmed = True
        kept = [w for w, c in self.word2count.items() if c >= min_count]
        total = max(len(self.word2index), 1)
        logging.info("    word trimming, kept %s / %s = %.4f" % (len(kept), len(self.word2index), len(kept) / total))
        self.reset_dictionary()
        for w in kept:
            self.index_word(w)

    def get_word_index(self, word: str) -> int:
        return self.word2index.get(word, self.UNK_token)

    # ------------------------------------------------------------------ word vectors
    def _random_table(self, embedding_dim: int) -> np.ndarray:
        return np.random.normal(0, scale=1 / np.sqrt(embedding_dim), size=[self.n_words, embedding_dim]).astype(np.float32)

    def load_word_vectors(self, pretrained_path: str, embedding_dim: int = 300) -> None:
        """rows of known words from a FastText model file, N(0, 1/sqrt(dim)) elsewhere (:166-190)"""
        try:
            import fasttext
        except ImportError as e:
            raise ImportError("Vocab.load_word_vectors reads FastText .bin files through the `fasttext` package, which is "
                              "not installed; use load_word_vectors_text for plain-text vectors") from e
        logging.info("  loading word vectors from '{}'...".format(pretrained_path))
        weights = self._random_table(embedding_dim)
        model = fasttext.load_model(pretrained_path)
        for word, idx in self.word2index.items():
            weights[idx] = model.get_word_vector(word)
        self.word_embedding_weights = weights

    def load_word_vectors_text(self, path: str, embedding_dim: int = 300) -> int:
        """same table from a `word v1 ... v_dim` text file; returns how many vocabulary words were found"""
        weights = self._random_table(embedding_dim)
        found = 0
        with open(path, encoding="utf-8") as f:
            for line in f:
                parts = line.split()
                if len(parts) != embedding_dim + 1:
                    continue
                idx = self.word2index.get(parts[0])
                if idx is not None:
                    weights[idx] = np.asarray(parts[1:], dtype=np.float32)
                    found += 1
        self.word_embedding_weights = weights
        return found
