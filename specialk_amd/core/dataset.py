"""Batch collation and the dataset container — the input contract of the encoder/decoder.
Mirrors the reference's core/dataset.py:14-105: sequences are right-padded with PAD to the
batch maximum and positions are index+1 on real tokens, 0 on padding."""
import numpy as np
import torch
import torch.utils.data

from . import constants as Constants


def collate_fn(insts):
    """list[list[int]] -> (seq, pos) int64 tensors of shape (B, Lmax)  (core/dataset.py:21-46)."""
    max_len = max(len(inst) for inst in insts)
    seq = np.full((len(insts), max_len), Constants.PAD, dtype=np.int64)
    for i, inst in enumerate(insts):
        seq[i, : len(inst)] = inst
    pos = np.where(seq != Constants.PAD, np.arange(1, max_len + 1, dtype=np.int64)[None, :], 0)
    return torch.from_numpy(seq), torch.from_numpy(pos)


def paired_collate_fn(insts):
    """[(src, tgt), ...] -> (src_seq, src_pos, tgt_seq, tgt_pos)  (core/dataset.py:14-18)."""
    src_insts, tgt_insts = list(zip(*insts))
    return (*collate_fn(src_insts), *collate_fn(tgt_insts))


class TranslationDataset(torch.utils.data.Dataset):
    """Same constructor, properties and item protocol as core/dataset.py:49-105."""

    def __init__(self, src_word2idx, tgt_word2idx, src_insts=None, tgt_insts=None):
        assert src_insts
        assert not tgt_insts or (len(src_insts) == len(tgt_insts))
        self._src_word2idx = src_word2idx
        self._src_idx2word = {idx: word for word, idx in src_word2idx.items()}
        self._src_insts = src_insts
        self._tgt_word2idx = tgt_word2idx
        self._tgt_idx2word = {idx: word for word, idx in tgt_word2idx.items()}
        self._tgt_insts = tgt_insts

    n_insts = property(lambda self: len(self._src_insts))
    src_vocab_size = property(lambda self: len(self._src_word2idx))
    tgt_vocab_size = property(lambda self: len(self._tgt_word2idx))
    src_word2idx = property(lambda self: self._src_word2idx)
    tgt_word2idx = property(lambda self: self._tgt_word2idx)
    src_idx2word = property(lambda self: self._src_idx2word)
    tgt_idx2word = property(lambda self: self._tgt_idx2word)

    def __len__(self):
        return self.n_insts

    def __getitem__(self, idx):
        if self._tgt_insts:
            return self._src_insts[idx], self._tgt_insts[idx]
        return self._src_insts[idx]
