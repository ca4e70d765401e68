"""Word + byte-pair vocabulary with the interface of the reference's core/bpe.py (Encoder: fit, tokenize,
transform, inverse_transform, vocabs_to_dict, from_dict, save, load) and its dictionary format
({"byte_pairs", "words", "kwargs"}), written without the reference's dependencies (`toolz`, `nltk`).

Host-side text processing, outside the accelerated path.  Behaviour restated from core/bpe.py:
  * the most frequent words get ids 0 .. word_vocab_size-1, the required tokens first (:124-141);
  * words outside that vocabulary are spelt as  __sow  piece ... piece  __eow, the pieces found greedily,
    longest first, at most ngram_max characters each, an unknown single character becoming UNK (:193-214);
  * the piece vocabulary holds the most frequent character n-grams (ngram_min .. ngram_max) of the words
    outside the word vocabulary, ids following the words' (:96-116, :143-163);
  * inverse_transform glues the pieces between __sow and __eow back into one word (:263-306).
The default word tokenizer is the regular expression nltk's wordpunct_tokenize is defined by
(runs of word characters, runs of other non-space characters).
Parity unpinned: the reference module cannot be imported here (its dependencies are absent) and its
repository holds no vocabulary file or expected output for it; tests/test_bpe.py checks the behaviour listed
above on hand-worked cases.
"""
import json
import re
from collections import Counter

from . import constants as Constants

DEFAULT_EOW = "__eow"
DEFAULT_SOW = "__sow"
DEFAULT_UNK = Constants.UNK_WORD
DEFAULT_PAD = Constants.PAD_WORD

_WORDPUNCT = re.compile(r"\w+|[^\w\s]+")


def wordpunct_tokenize(text):
    return _WORDPUNCT.findall(text)


def _required_tokens():
    get = getattr(Constants, "get_tokens", None)
    if get is not None:
        return list(get())
    return [Constants.PAD_WORD, Constants.UNK_WORD, Constants.SOS_WORD, Constants.EOS_WORD]


class Encoder:
    """Encodes white-space separated text with a word vocabulary backed by byte pairs."""

    def __init__(self, vocab_size=8192, pct_bpe=0.2, word_tokenizer=None, silent=True, ngram_min=2, ngram_max=2,
                 required_tokens=None, strict=False, EOW=DEFAULT_EOW, SOW=DEFAULT_SOW, UNK=DEFAULT_UNK,
                 PAD=DEFAULT_PAD):
        if vocab_size < 1:
            raise ValueError("vocab_size must be at least 1")
        self.EOW, self.SOW, self.UNK, self.PAD = EOW, SOW, UNK, PAD
        self.eow_len, self.sow_len = len(EOW), len(SOW)
        self.required_tokens = required_tokens if required_tokens else _required_tokens()
        self.vocab_size = vocab_size
        self.pct_bpe = pct_bpe
        self.word_vocab_size = max(int(vocab_size * (1 - pct_bpe)), len(self.required_tokens or []))
        self.bpe_vocab_size = vocab_size - self.word_vocab_size
        self.word_tokenizer = word_tokenizer if word_tokenizer is not None else wordpunct_tokenize
        self.custom_tokenizer = word_tokenizer is not None
        self.word_vocab, self.bpe_vocab = {}, {}
        self.inverse_word_vocab, self.inverse_bpe_vocab = {}, {}
        self.silent = silent
        self.ngram_min, self.ngram_max = ngram_min, ngram_max
        self.strict = strict

    def mute(self):
        self.silent = True

    def unmute(self):
        self.silent = False

    # ------------------------------------------------------------------ learning
    def _all_words(self, sentences):
        for s in sentences:
            yield from self.word_tokenizer(s)

    def learn_word_vocab(self, sentences):
        counts = Counter(self._all_words(sentences))
        for i, token in enumerate(self.required_tokens):
            counts[token] = int(2 ** 63) - i  # required tokens lead, in their own order
        ranked = sorted(counts.items(), key=lambda p: -p[1])  # stable: ties keep first-seen order
        return {word: idx for idx, (word, _) in enumerate(ranked[: self.word_vocab_size])}

    def byte_pair_counts(self, words):
        """Per distinct word: counts of its characters and of its n-grams of ngram_min .. ngram_max characters,
        each weighted by the word's frequency."""
        for word, count in Counter(words).items():
            chars = list(word)
            spaced_len = 2 * len(chars) - 1  # the reference measures the space-joined spelling
            c = Counter()
            for ch in chars:
                c[ch] += count
            for n in range(self.ngram_min, min(self.ngram_max, spaced_len) + 1):
                for i in range(len(chars) - n + 1):
                    c["".join(chars[i:i + n])] += count
            yield c

    @staticmethod
    def trim_vocab(n, vocab):
        for pair, _ in sorted(vocab.items(), key=lambda p: -p[1])[n:]:
            del vocab[pair]

    def learn_bpe_vocab(self, words):
        vocab = Counter()
        for token in {self.SOW, self.EOW}:
            vocab[token] = int(2 ** 63)
        for idx, counts in enumerate(self.byte_pair_counts(words)):
            for piece, count in counts.items():
                vocab[piece] += count
            if (idx + 1) % 10000 == 0:
                self.trim_vocab(10 * self.bpe_vocab_size, vocab)
        ranked = sorted(vocab.items(), key=lambda p: -p[1])[: self.bpe_vocab_size]
        return {piece: idx + self.word_vocab_size for idx, (piece, _) in enumerate(ranked)}

    def fit(self, text):
        lines = [l.lower().strip() for l in text]
        self.word_vocab = self.learn_word_vocab(lines)
        rest = [w for w in self._all_words(lines) if w not in self.word_vocab]
        self.bpe_vocab = self.learn_bpe_vocab(rest)
        self.inverse_word_vocab = {i: t for t, i in self.word_vocab.items()}
        self.inverse_bpe_vocab = {i: t for t, i in self.bpe_vocab.items()}

    # ------------------------------------------------------------------ encoding
    def subword_tokenize(self, word):
        out = [self.SOW]
        start = 0
        while start < len(word):
            end = min(len(word), start + self.ngram_max)
            while end - start > 1 and word[start:end] not in self.bpe_vocab:
                end -= 1
            piece = word[start:end]
            out.append(piece if piece in self.bpe_vocab else self.UNK)
            start = end
        out.append(self.EOW)
        return out

    def tokenize(self, sentence):
        tokens = []
        for w in self.word_tokenizer(sentence.lower().strip()):
            if w in self.word_vocab:
                tokens.append(w)
            else:
                tokens.extend(self.subword_tokenize(w))
        return tokens

    def transform(self, sentences, reverse=False, fixed_length=None):
        unk, step = self.word_vocab[self.UNK], (-1 if reverse else 1)
        for sentence in sentences:
            ids, in_subword = [], False
            for token in self.tokenize(sentence.lower().strip()):
                if in_subword:
                    if token in self.bpe_vocab:
                        in_subword = token != self.EOW
                        ids.append(self.bpe_vocab[token])
                    else:
                        ids.append(unk)
                elif token == self.SOW:
                    in_subword = True
                    ids.append(self.bpe_vocab[token])
                else:
                    ids.append(self.word_vocab.get(token, unk))
            if fixed_length is not None:
                ids = ids[:fixed_length] + [self.word_vocab[self.PAD]] * max(0, fixed_length - len(ids))
            yield ids[::step]

    def _decode_row(self, row):
        """Ids of one sentence -> its words: pieces between the word markers are glued into one word."""
        pieces, whole = self.inverse_bpe_vocab, self.inverse_word_vocab
        out, buf, inside = [], [], False
        for i in row:
            text = pieces.get(i)
            if text == self.SOW or text == self.EOW:
                opening = text == self.SOW
                if self.strict and opening == inside:
                    raise ValueError("word start marker inside an unfinished word" if opening else
                                     "word end marker without a word start")
                if not opening:
                    out.append("".join(buf))
                    buf = []
                inside = opening
                continue
            if text is None:
                text = whole.get(i)
                if text is None:
                    raise ValueError("id {} is in neither vocabulary".format(i))
            elif not inside:
                if i in whole:  # an id both vocabularies hold reads as a word outside a word's pieces
                    text = whole[i]
                elif self.strict:
                    raise ValueError("piece id {} outside a word".format(i))
            (buf if inside else out).append(text)
        return [w for w in out if w]

    def inverse_transform(self, rows):
        for row in rows:
            yield " ".join(self._decode_row(row))

    # ------------------------------------------------------------------ persistence
    def vocabs_to_dict(self, dont_warn=False):
        if self.custom_tokenizer and not dont_warn:
            print("[Warning] the word tokenizer is not the default one and is not saved: pass it again after loading")
        ctor = {name: getattr(self, name) for name in ("vocab_size", "pct_bpe", "silent", "ngram_min", "ngram_max",
                                                        "required_tokens", "strict", "EOW", "SOW", "UNK", "PAD")}
        return {"byte_pairs": self.bpe_vocab, "words": self.word_vocab, "kwargs": ctor}

    def save(self, outpath, dont_warn=False):
        with open(outpath, "w") as f:
            json.dump(self.vocabs_to_dict(dont_warn), f)

    @classmethod
    def from_dict(cls, vocabs):
        enc = cls(**vocabs["kwargs"])
        enc.word_vocab, enc.bpe_vocab = vocabs["words"], vocabs["byte_pairs"]
        enc.inverse_word_vocab = dict(map(reversed, enc.word_vocab.items()))
        enc.inverse_bpe_vocab = dict(map(reversed, enc.bpe_vocab.items()))
        return enc

    @classmethod
    def load(cls, in_path):
        with open(in_path) as f:
            return cls.from_dict(json.load(f))
