"""Model wrapper base class with the reference's interface (lib/nmtModel.py:24-456): device
choice, run directory, CSV logs, dataset / test-data loaders, id -> text decoding.

Outside the accelerated path (host-side plumbing only).  Differences from the reference:
  * the device is always a GPU (this package has no CPU path); `-cuda_device` selects it;
  * the Telegram notifier is optional: it is imported on use, and a clear error is raised when its
    package is not installed; the BPE vocabulary class is this package's own core.bpe.Encoder;
  * BPE test sentences that encode to more than max_token_seq_len - 2 ids lose words from the end until
    they fit — what the reference's `reclip` (preprocess.py:141-157) sets out to do; as written it hands
    `transform` a string and so keeps the encoding of the sentence's first character only;
  * under torch.distributed only rank 0 creates directories and writes logs / checkpoints.
"""
import atexit
import datetime
import math
import os

import torch
import torch.utils.data

from ..core import constants
from ..core.bpe import Encoder as BPE
from ..core.serialization import load_trusted_types_only
from ..core.dataset import (TokenBudgetBatchSampler, TranslationDataset, collate_fn, paired_collate_fn,
                            paired_device_collate_fn)


def _rank():
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def load_text_file(filepath, case_sensitive=True, max_seq=None):
    """One whitespace-tokenised sentence per line."""
    sequences = []
    with open(filepath) as f:
        for count, line in enumerate(f):
            if max_seq and count >= max_seq:
                break
            line = line if case_sensitive else line.lower()
            sequences.append(line.strip().split())
    return sequences


def load_text_lines(filepath, case_sensitive=True):
    """One raw sentence per line (the form a BPE vocabulary tokenizes itself)."""
    with open(filepath) as f:
        return [(line if case_sensitive else line.lower()).strip() for line in f]


def reclip(ref_seq, bpe_seq, enc, max_len):
    """Ids of the longest prefix (in whitespace words) of `ref_seq` whose encoding has at most max_len ids."""
    words = ref_seq.split()
    while len(bpe_seq) > max_len and words:
        words.pop()
        bpe_seq = next(iter(enc.transform([" ".join(words)])))
    return bpe_seq[:max_len]


def seq2idx(sequences, w2i):
    return [[w2i.get(w, constants.UNK) for w in s] for s in sequences]


class NMTModel:
    def __init__(self, opt, models_folder="models"):
        self.opt = opt
        idx = getattr(opt, "cuda_device", None)
        self.device = torch.device("cuda", idx if idx is not None else torch.cuda.current_device()
                                   if torch.cuda.is_available() else 0)
        self.constants = constants
        self.opt.directory = self.init_dir(stores=models_folder)
        self.reset_metrics()
        self.bot = None
        if getattr(self.opt, "telegram", None):
            self.init_telegram(self.opt.telegram)
        self.src_bpe = None
        self.tgt_bpe = None
        self.log_train_file = self.log_valid_file = None
        atexit.register(self.exit_handler)

    # ---- to be provided by the model-specific wrapper
    def load(self, *a, **k):
        print("[Warning]: load() is not implemented.")
        return self

    def initiate(self):
        print("[Warning]: initiate() is not implemented.")
        return self

    def setup_optimiser(self):
        print("[Warning] setup_optimiser() is not implemented.")

    def train(self, *a, **k):
        print("[Warning]: train() is not implemented.")
        return self

    def translate(self, *a, **k):
        print("[Warning]: translate() is not implemented.")
        return self

    def save(self, *a, **k):
        print("[Warning]: save() is not implemented.")
        return self

    # ---- shared behaviour
    def reset_metrics(self):
        self.train_losses, self.valid_losses, self.train_accs, self.valid_accs = [], [], [], []

    def init_dir(self, stores):
        """Run directory: next to a loaded encoder checkpoint unless -new_directory, else
        <cwd>/<stores>/<-directory_name | model-timestamp> (created only when saving)."""
        if getattr(self.opt, "checkpoint_encoder", None) and not getattr(self.opt, "new_directory", False):
            return os.path.split(self.opt.checkpoint_encoder)[0]
        basepath = os.path.join(os.path.abspath(os.getcwd()), stores)
        name = getattr(self.opt, "directory_name", "") or (
            str(self.opt.model) + "-" + datetime.datetime.now().strftime("%y-%m-%d-%H-%M-%S"))
        directory = os.path.join(basepath, name)
        if not os.path.isdir(directory) and getattr(self.opt, "save_model", False) and _rank() == 0:
            os.makedirs(directory, exist_ok=True)
        return directory

    def init_logs(self):
        """train.log / valid.log (CSV: epoch,loss,ppl,accuracy); an existing file gets a numbered sibling."""
        if _rank() != 0:
            return
        def fresh(stem):
            path, step = os.path.join(self.opt.directory, stem + ".log"), 1
            while os.path.isfile(path):
                path = os.path.join(self.opt.directory, f"{stem}_{step}.log")
                step += 1
            return path
        self.log_train_file, self.log_valid_file = fresh("train"), fresh("valid")
        print("[Info] Training performance will be written to file: {} and {}".format(
            self.log_train_file, self.log_valid_file))
        for path in (self.log_train_file, self.log_valid_file):
            with open(path, "w") as f:
                f.write("epoch,loss,ppl,accuracy\n")

    def update_logs(self, epoch_i):
        if _rank() != 0:
            return
        if not (self.log_train_file and self.log_valid_file):
            print("[Warning] log files are not initiated. No updates are kept into storage.")
            return
        rows = ((self.log_train_file, self.train_losses[-1], self.train_accs[-1]),
                (self.log_valid_file, self.valid_losses[-1], self.valid_accs[-1]))
        for path, loss, acc in rows:
            with open(path, "a") as f:
                f.write("{epoch},{loss: 8.5f},{ppl: 8.5f},{accu:3.3f}\n".format(
                    epoch=epoch_i, loss=loss, ppl=math.exp(min(loss, 100)), accu=100 * acc))

    def load_dataset(self):
        """Loads the pickled dataset written by the reference's preprocess.py:
        {settings, dict: {src, tgt}, train: {src, tgt}, valid: {src, tgt}}."""
        data = load_trusted_types_only(self.opt.data, "a dataset file")
        self.dataset_settings = data["settings"]
        self.opt.max_token_seq_len = data["settings"].max_token_seq_len
        if "byte_pairs" in data["dict"]["src"] and "__sow" in data["dict"]["src"]["byte_pairs"]:
            self.src_bpe = BPE.from_dict(data["dict"]["src"])  # lib/nmtModel.py:219-223
            self.tgt_bpe = BPE.from_dict(data["dict"]["tgt"])
        self.training_data, self.validation_data = self.init_dataloaders(data, self.opt)
        self.opt.src_vocab_size = self.training_data.dataset.src_vocab_size
        self.opt.tgt_vocab_size = self.training_data.dataset.tgt_vocab_size
        return self

    @staticmethod
    def _flatten_vocab(word2idx):
        if "byte_pairs" in word2idx and "__sow" in word2idx["byte_pairs"]:
            pairs = {x + "_": y for x, y in word2idx["byte_pairs"].items()}
            return {**pairs, **word2idx["words"]}, True
        return word2idx, False

    @staticmethod
    def init_dataloaders(data, opt):
        src_word2idx, _ = NMTModel._flatten_vocab(data["dict"]["src"])
        tgt_word2idx, _ = NMTModel._flatten_vocab(data["dict"]["tgt"])
        # -device_collate: pad the batches on the GPU (loaders run in this process: num_workers = 0)
        pair_fn = paired_device_collate_fn if getattr(opt, "device_collate", False) else paired_collate_fn
        def loader(split, shuffle):
            ds = TranslationDataset(src_word2idx=src_word2idx, tgt_word2idx=tgt_word2idx,
                                    src_insts=data[split]["src"], tgt_insts=data[split]["tgt"])
            budget = int(getattr(opt, "batch_tokens", 0) or 0)
            if budget > 0:  # length-bucketed batches of at most `budget` tokens and -batch_size sentences
                sampler = TokenBudgetBatchSampler.for_dataset(ds, budget, max_sentences=opt.batch_size, shuffle=shuffle,
                                                              seed=int(getattr(opt, "seed", 0) or 0))
                return torch.utils.data.DataLoader(ds, num_workers=0, batch_sampler=sampler, collate_fn=pair_fn)
            return torch.utils.data.DataLoader(ds, num_workers=0, batch_size=opt.batch_size,
                                               collate_fn=pair_fn, shuffle=shuffle)
        train_loader, valid_loader = loader("train", True), loader("valid", False)
        if opt.embs_share_weight:
            assert train_loader.dataset.src_word2idx == train_loader.dataset.tgt_word2idx, \
                "The src/tgt word2idx table are different but you asked to share the word embeddings."
        return train_loader, valid_loader

    def load_testdata(self, test_datapath, test_vocab):
        """Text file + dataset vocabulary -> (DataLoader of (seq, pos), max_token_seq_len, is_bpe, decoder)."""
        data = load_trusted_types_only(test_vocab, "a dataset (vocabulary) file")
        settings = data["settings"]
        is_bpe = str(getattr(settings, "format", "word")).lower() == "bpe"
        override = getattr(self.opt, "override_max_token_seq_len", None)
        if override and override > 0:
            settings.max_token_seq_len = override
            self.change_max_seq_len(override + 5)
        if is_bpe:  # lib/nmtModel.py:267-305
            bpe_src, decoder = BPE.from_dict(data["dict"]["src"]), BPE.from_dict(data["dict"]["tgt"])
            lines = load_text_lines(test_datapath, getattr(settings, "case_sensitive", True))
            insts = [[constants.SOS] + reclip(raw, enc, bpe_src, settings.max_token_seq_len - 2) + [constants.EOS]
                     for raw, enc in zip(lines, bpe_src.transform(lines))]
            src_w2i, _ = self._flatten_vocab(data["dict"]["src"])
            tgt_w2i, _ = self._flatten_vocab(data["dict"]["tgt"])
            ds = TranslationDataset(src_word2idx=src_w2i, tgt_word2idx=tgt_w2i, src_insts=insts)
            loader = torch.utils.data.DataLoader(ds, num_workers=0, batch_size=self.opt.batch_size, collate_fn=collate_fn)
            return loader, settings.max_token_seq_len, is_bpe, decoder
        sents = load_text_file(test_datapath, getattr(settings, "case_sensitive", True))
        insts = seq2idx(sents, data["dict"]["src"])
        insts = [[constants.SOS] + s[: settings.max_word_seq_len] + [constants.EOS] for s in insts]
        ds = TranslationDataset(src_word2idx=data["dict"]["src"], tgt_word2idx=data["dict"]["tgt"], src_insts=insts)
        loader = torch.utils.data.DataLoader(ds, num_workers=0, batch_size=self.opt.batch_size, collate_fn=collate_fn)
        return loader, settings.max_token_seq_len, is_bpe, data["dict"]["tgt"]

    @staticmethod
    def translate_decode_bpe(hypotheses, bpe_enc):
        lines = []
        for x in bpe_enc.inverse_transform([h[0] for h in hypotheses]):
            words = x.split()
            index = 0
            for j, w in enumerate(words):
                if w == constants.EOS_WORD and j > 0:
                    index = j
                    break
            line = " ".join(words[:index])
            lines.append(line if line.strip() else constants.UNK_WORD)
        return lines

    @staticmethod
    def translate_decode_dict(hypotheses, idx2w):
        """hypotheses[i][j] token lists -> one line per hypothesis, EOS dropped.  `idx2w` maps id -> word
        (a word -> id dictionary, as stored in the dataset file, is inverted first)."""
        if idx2w and isinstance(next(iter(idx2w)), str):
            idx2w = {i: w for w, i in idx2w.items()}
        lines = []
        for sequence in hypotheses:
            for tokens in sequence:
                lines.append(" ".join(idx2w[i] for i in tokens if i != constants.EOS))
        return lines

    def exit_handler(self):
        if getattr(self.opt, "save_model", False) and _rank() == 0:
            d = self.opt.directory
            if os.path.isdir(d) and len(os.listdir(d)) < 1:
                os.rmdir(d)

    def init_telegram(self, api_json_path):
        import json
        try:
            import telebot
        except ImportError as e:
            raise RuntimeError("-telegram needs the `telebot` package") from e
        with open(api_json_path) as f:
            api_q = json.load(f)
        self.bot = telebot.TeleBot(api_q["api_private_key"])
        self.bot_chatid = api_q["chat_id"]

    def t_msg(self, msg):
        if self.bot:
            self.bot.send_message(self.bot_chatid, msg)

    @staticmethod
    def parse(x):
        return x.split()
