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


def device_collate_fn(insts, device=None):
    """collate_fn with the padding done on the GPU: the sentences travel as one flat id array plus offsets
    (two host -> device copies) and a kernel writes the padded (seq, pos) pair.  Same result as collate_fn."""
    from .. import _lib as L
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    lens = np.fromiter((len(x) for x in insts), dtype=np.int64, count=len(insts))
    off = np.zeros(len(insts) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    flat = np.fromiter((t for x in insts for t in x), dtype=np.int64, count=int(off[-1]))
    host = torch.from_numpy(np.concatenate([off, flat])).pin_memory()
    dev = host.to(device, non_blocking=True)
    return L.collate(dev[len(off):], dev[:len(off)], len(insts), int(lens.max()), Constants.PAD)


def paired_device_collate_fn(insts, device=None):
    src_insts, tgt_insts = list(zip(*insts))
    return (*device_collate_fn(src_insts, device), *device_collate_fn(tgt_insts, device))


class TokenBudgetBatchSampler(torch.utils.data.Sampler):
    """Length-bucketed batches under a token budget, for `DataLoader(batch_sampler=...)`.

    The reference draws a fixed number of sentences per batch from a shuffled dataset
    (lib/nmtModel.py:356-367), so a batch is as long as its longest sentence and the work per
    step varies with the draw.  Here every epoch the indices are shuffled, cut into pools of
    `pool_batches` x `max_sentences` sentences, sorted by length inside each pool (neighbours in a
    batch have similar lengths: little padding in the id tensors, similar sequence lengths in the
    attention kernels) and cut greedily into batches of at most `max_tokens` tokens (a sentence
    counts max(len(src), len(tgt))) and `max_sentences` sentences; the batches are then shuffled.
    Every index appears exactly once per epoch; the order is a function of (seed, epoch) only, so
    all ranks of a data-parallel run iterate the same batches (each takes its shard of them)."""

    def __init__(self, lengths, max_tokens, max_sentences=1024, pool_batches=64, shuffle=True, seed=0):
        self.lengths = np.asarray(lengths, dtype=np.int64)
        if max_tokens < 1 or max_sentences < 1:
            raise ValueError("max_tokens and max_sentences must be positive")
        self.max_tokens, self.max_sentences = int(max_tokens), int(max_sentences)
        self.pool, self.shuffle, self.seed, self.epoch = int(pool_batches) * int(max_sentences), shuffle, seed, 0
        self._cache = None

    @classmethod
    def for_dataset(cls, ds, max_tokens, **kw):
        tgt = ds._tgt_insts or ds._src_insts
        return cls([max(len(a), len(b)) for a, b in zip(ds._src_insts, tgt)], max_tokens, **kw)

    def set_epoch(self, epoch):
        self.epoch, self._cache = int(epoch), None

    def _batches(self):
        if self._cache is not None and self._cache[0] == self.epoch:
            return self._cache[1]
        n = len(self.lengths)
        rng = np.random.default_rng([self.seed, self.epoch])
        order = rng.permutation(n) if self.shuffle else np.arange(n)
        batches = []
        for p0 in range(0, n, self.pool):
            pool = order[p0:p0 + self.pool]
            pool = pool[np.argsort(self.lengths[pool], kind="stable")]
            cur, tokens = [], 0
            for i in pool.tolist():
                c = int(self.lengths[i])
                if cur and (tokens + c > self.max_tokens or len(cur) == self.max_sentences):
                    batches.append(cur)
                    cur, tokens = [], 0
                cur.append(i)
                tokens += c
            if cur:
                batches.append(cur)
        if self.shuffle:
            batches = [batches[i] for i in rng.permutation(len(batches)).tolist()]
        self._cache = (self.epoch, batches)
        return batches

    def __iter__(self):
        batches = self._batches()
        self.set_epoch(self.epoch + 1)  # a plain `for batch in loader` per epoch reshuffles by itself
        return iter(batches)

    def __len__(self):
        return len(self._batches())


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
