"""core.bpe.Encoder: hand-worked cases of the behaviour the reference's core/bpe.py defines (CPU only).
Parity unpinned — the reference module needs packages that are absent and ships no expected outputs."""
import json

import pytest

from specialk_amd.core import constants as C
from specialk_amd.core.bpe import Encoder, wordpunct_tokenize

TEXT = ["the cat sat on the mat .", "the dog sat on the log .", "a cat and a dog , the end !"]


def fitted(vocab_size=24, pct_bpe=0.5, **kw):
    enc = Encoder(vocab_size=vocab_size, pct_bpe=pct_bpe, ngram_min=1, ngram_max=2, **kw)
    enc.fit(TEXT)
    return enc


def test_tokenizer_splits_words_and_punctuation_runs():
    assert wordpunct_tokenize("Good muffins cost $3.88, ok?!") == ["Good", "muffins", "cost", "$", "3", ".", "88", ",", "ok", "?!"]


def test_vocabulary_layout():
    enc = fitted()
    assert enc.word_vocab_size == 12 and enc.bpe_vocab_size == 12
    # required tokens lead in id order, then words by frequency (ties: first seen)
    assert [w for w, _ in sorted(enc.word_vocab.items(), key=lambda p: p[1])][:5] == C.get_tokens()
    assert enc.word_vocab["the"] == 5 and len(enc.word_vocab) == 12
    assert sorted(enc.word_vocab.values()) == list(range(12))
    # pieces follow the words; the word-boundary markers come first
    assert sorted(enc.bpe_vocab.values()) == list(range(12, 24))
    assert {enc.bpe_vocab["__sow"], enc.bpe_vocab["__eow"]} == {12, 13}


def test_subword_spelling_is_greedy_longest_first():
    enc = Encoder(vocab_size=16, pct_bpe=0.5, ngram_max=2)
    enc.word_vocab = {w: i for i, w in enumerate(C.get_tokens() + ["the"])}
    enc.bpe_vocab = {"__sow": 8, "__eow": 9, "ab": 10, "a": 11, "b": 12, "c": 13}
    enc.inverse_word_vocab = {v: k for k, v in enc.word_vocab.items()}
    enc.inverse_bpe_vocab = {v: k for k, v in enc.bpe_vocab.items()}
    assert enc.subword_tokenize("abc") == ["__sow", "ab", "c", "__eow"]
    assert enc.subword_tokenize("bza") == ["__sow", "b", C.UNK_WORD, "a", "__eow"]
    assert enc.tokenize("The abc") == ["the", "__sow", "ab", "c", "__eow"]
    ids = list(enc.transform(["the abc", "bza"]))
    assert ids == [[5, 8, 10, 13, 9], [8, 12, C.UNK, 11, 9]]
    assert list(enc.inverse_transform(ids)) == ["the abc", "b<unk>a"]
    assert list(enc.transform(["the abc"], reverse=True)) == [[9, 13, 10, 8, 5]]
    assert list(enc.transform(["the"], fixed_length=3)) == [[5, C.PAD, C.PAD]]
    assert list(enc.transform(["the abc"], fixed_length=2)) == [[5, 8]]


def test_round_trip_and_persistence(tmp_path):
    enc = fitted(vocab_size=60, pct_bpe=0.8)  # 12 words, 48 pieces
    ids = list(enc.transform(TEXT))
    assert list(enc.inverse_transform(ids)) == TEXT  # every character of the corpus has a piece
    path = tmp_path / "vocab.json"
    enc.save(str(path))
    d = json.loads(path.read_text())
    assert set(d) == {"byte_pairs", "words", "kwargs"} and d["kwargs"]["ngram_max"] == 2
    again = Encoder.load(str(path))
    assert list(again.transform(TEXT)) == ids and again.vocabs_to_dict() == enc.vocabs_to_dict()


def test_strict_mode_and_unknown_ids():
    enc = fitted(strict=True)
    sow, eow = enc.bpe_vocab["__sow"], enc.bpe_vocab["__eow"]
    with pytest.raises(ValueError):
        list(enc.inverse_transform([[sow, sow]]))
    with pytest.raises(ValueError):
        list(enc.inverse_transform([[eow]]))
    with pytest.raises(ValueError):
        list(enc.inverse_transform([[999]]))
    with pytest.raises(ValueError):
        Encoder(vocab_size=0)


def test_reclip_drops_trailing_words_until_the_encoding_fits():
    from specialk_amd.lib.nmtModel import reclip
    enc = fitted()
    raw = "the cat sat on the mat ."
    full = next(iter(enc.transform([raw])))
    assert reclip(raw, full, enc, len(full)) == full
    short = reclip(raw, full, enc, 3)
    assert len(short) <= 3 and short == next(iter(enc.transform(["the cat sat"])))[:3]
