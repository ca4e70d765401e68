"""Trainer / translator wrapper for the Transformer with the reference's interface
(lib/TransformerModel.py:34-450): initiate / load / setup_optimiser / train / translate / save /
change_max_seq_len / performance / calculate_loss / compute_epoch, the split encoder / decoder
checkpoint format and the `opt` namespace conventions.

What runs where: the model, loss, backward and optimiser step are HIP kernels (specialk_amd.ops,
specialk_amd.optim); this file is host-side orchestration.  Differences from the reference:
  * `-dtype {fp32,bf16}` (extension; default fp32 = the reference's arithmetic);
  * statistics (loss sum, correct, words) accumulate on the device and are read ONCE per epoch
    instead of three `.item()` syncs per step (lib/TransformerModel.py:428-430);
  * under torch.distributed (one process per GPU) `translate` decodes a contiguous shard of the
    sentences per rank (no communication; rank 0 joins the output files in rank order); in training
    every rank takes a contiguous shard of each
    batch and gradients are SUM-all-reduced in buckets over RCCL while backward runs — the
    multi-process replacement of nn.DataParallel (train.py:385-388).  The loss is a token sum,
    so the update equals the single-GPU update on the whole batch.
"""
import math
import os
import time
from collections import OrderedDict

import torch
import torch.nn as nn

from ..core.serialization import load_trusted_types_only
from .. import ops
from ..optim import FusedAdam
from ..parallel import GradBucketReducer, all_reduce_scalars
from .nmtModel import NMTModel, _rank
from .transformer.Models import Transformer, get_sinusoid_encoding_table
from .transformer.Optim import ScheduledOptim
from .transformer.Translator import Translator

_LOAD_BLACKLIST = {
    "checkpoint_encoder", "checkpoint_decoder", "cuda", "directory", "directory_name", "data", "log", "model",
    "save_mode", "telegram", "save_model", "train_from_state_dict", "batch_size", "epochs", "epoch", "device",
    # extensions of this package that describe the run, not the network
    "dtype", "seed", "cuda_device", "src", "vocab", "output", "beam_size", "n_best", "override_max_token_seq_len",
}


def _world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _strip_module_prefix(sd):
    """Checkpoints written under nn.DataParallel carry a `module.` prefix (lib/TransformerModel.py:83-88)."""
    if sd and next(iter(sd)).startswith("module."):
        return OrderedDict((k[7:], v) for k, v in sd.items())
    return sd


class TransformerModel(NMTModel):
    def __init__(self, opt):
        super().__init__(opt)
        self.save_trip = False
        self.packed = True  # skip padding rows while training (collate_fn's batches are right-padded)
        self.reducer = None
        self.model = None

    # ------------------------------------------------------------------ construction
    def _compute_dtype(self):
        return torch.bfloat16 if str(getattr(self.opt, "dtype", "fp32")).lower() in ("bf16", "bfloat16") else torch.float32

    def initiate(self):
        """Build the network and apply xavier-uniform to every matrix, as the reference does
        (lib/TransformerModel.py:57-75) — which, like there, also overwrites the sinusoid table."""
        o = self.opt
        self.model = Transformer(
            o.src_vocab_size, o.tgt_vocab_size, o.max_token_seq_len,
            tgt_emb_prj_weight_sharing=o.proj_share_weight, emb_src_tgt_weight_sharing=o.embs_share_weight,
            d_k=o.d_k, d_v=o.d_v, d_model=o.d_model, d_word_vec=o.d_word_vec, d_inner=o.d_inner_hid,
            n_layers=o.layers, n_head=o.n_head, dropout=o.dropout).to(self.device)
        for p in self.model.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.model.set_compute_dtype(self._compute_dtype())
        return self

    def load(self, encoder_path, decoder_path=None):
        """Load encoder (and decoder + generator) checkpoints; saved settings override `opt`
        except for run-specific fields (lib/TransformerModel.py:77-153)."""
        if encoder_path:
            enc = load_trusted_types_only(encoder_path, "an encoder checkpoint")
            enc["model"] = _strip_module_prefix(enc["model"])
            saved = enc["settings"]
            for arg in dir(saved):
                if arg[0] == "_" or arg in _LOAD_BLACKLIST:
                    continue
                setattr(self.opt, arg, getattr(saved, arg))
            self.initiate()
            self.model.encoder.load_state_dict(enc["model"])
            if self.opt.verbose:
                print("[Info] Loaded encoder model.")
        if decoder_path:
            dec = load_trusted_types_only(decoder_path, "a decoder checkpoint")
            dec["model"] = _strip_module_prefix(dec["model"])
            self.model.decoder.load_state_dict(dec["model"])
            self.model.generator.load_state_dict(dec["generator"])
            if self.opt.verbose:
                print("[Info] Loaded decoder model.")
        self.model.to(self.device)
        return self

    def setup_optimiser(self):
        """Adam(beta=(0.9,0.98), eps=1e-9) under the Noam schedule (lib/TransformerModel.py:155-168);
        the learning-rate flag is overwritten every step, as in the reference."""
        params = [p for p in self.model.parameters() if p.requires_grad]
        self.optimiser = ScheduledOptim(
            FusedAdam(params, betas=(0.9, 0.98), eps=1e-09, lr=self.opt.learning_rate),
            self.opt.d_model, self.opt.n_warmup_steps)
        if _world() > 1:
            self.reducer = GradBucketReducer(params)
        if self.opt.verbose:
            print("[Info] optimiser configured.")

    # ------------------------------------------------------------------ training
    def train(self, epoch, evaluate=True):
        start = time.time()
        train_loss, train_acc = self.compute_epoch(self.training_data, False)
        self.train_losses.append(train_loss)
        self.train_accs.append(train_acc)
        if self.opt.verbose and _rank() == 0:
            print("  - (Training)   perplexity: {p: 8.5f}, accuracy: {a:3.3f} %, elapse: {e:3.3f} min".format(
                p=math.exp(min(train_loss, 100)), a=100 * train_acc, e=(time.time() - start) / 60))
        if evaluate:
            with torch.no_grad():
                valid_loss, valid_acc = self.compute_epoch(self.validation_data, True)
            self.valid_losses.append(valid_loss)
            self.valid_accs.append(valid_acc)
            if self.opt.verbose and _rank() == 0:
                print("  - (Validation) perplexity: {p: 8.5f}, accuracy: {a:3.3f} %, elapse: {e:3.3f} min".format(
                    p=math.exp(min(valid_loss, 100)), a=100 * valid_acc, e=(time.time() - start) / 60))
        return self

    def performance(self, pred, gold, smoothing=False):
        """(loss, n_correct) of logits pred (T,V) against gold ids (lib/TransformerModel.py:340-351);
        one fused kernel computes the loss, its gradient and the accuracy count."""
        loss, stats = ops.xent(pred, gold, smoothing)
        return loss, int(stats[1].item())

    def calculate_loss(self, pred, gold, smoothing=False):
        """Summed (label-smoothed, eps = 0.1) cross entropy over non-pad targets
        (lib/TransformerModel.py:353-375)."""
        return ops.xent(pred, gold, smoothing)[0]

    def _shard(self, batch):
        world = _world()
        if world == 1:
            return batch
        n = batch[0].shape[0]
        per = (n + world - 1) // world
        lo, hi = min(n, _rank() * per), min(n, (_rank() + 1) * per)
        return tuple(x[lo:hi] for x in batch)

    def compute_epoch(self, dataset, validation=False):
        """One pass over `dataset`; returns (loss per word, accuracy) (lib/TransformerModel.py:377-435)."""
        m, o = self.model, self.opt
        if validation:
            m.eval()
        else:
            m.encoder.train(not o.freeze_encoder)
            m.decoder.train(not o.freeze_decoder)
            m.generator.train(not o.freeze_decoder)
        totals = torch.zeros(3, dtype=torch.float64, device=self.device)
        want_packed = not (validation and o.log)
        ops.settle_host_garbage()  # no 100 ms collection pause inside the steps that follow

        def staged():
            """Batches one step ahead of their use: batch k + 1 is copied to the device and its packing plan
            started (two small kernels, one host copy that nobody waits for yet) before batch k runs, so a step
            begins without waiting for the device and the host keeps queueing work ahead of it."""
            ahead = None
            for batch in dataset:
                shard = self._shard(batch)
                # A rank whose shard of a short last batch is empty still has to join every
                # collective: it runs sentence 0 with weight zero.
                idle = shard[0].shape[0] == 0
                if idle:
                    shard = tuple(x[:1] for x in batch)
                dev = tuple(x.to(self.device, non_blocking=True) for x in shard)
                plans = m.plan_ahead(*dev) if self.packed and want_packed else None
                if ahead is not None:
                    yield ahead
                ahead = (dev, idle, plans)
            if ahead is not None:
                yield ahead

        for (src_seq, src_pos, tgt_seq, tgt_pos), idle, plans in staged():
            if not validation:
                self.reducer.zero_grad() if self.reducer else self.optimiser.zero_grad()
            pred = None
            if self.packed and want_packed:
                try:  # the network on the batch without its padding rows: same loss, counts and gradients
                    pred, gold = m.forward_packed(src_seq, src_pos, tgt_seq, tgt_pos, plans=plans)
                except ValueError:  # not a right-padded batch
                    self.packed = False
            if pred is None:
                pred = m(src_seq, src_pos, tgt_seq, tgt_pos)
                gold = tgt_seq[:, 1:]
            loss, stats = ops.xent(pred.view(-1, pred.size(-1)), gold, o.label_smoothing, unit_grad=not idle)
            if idle:
                loss, stats = loss * 0.0, stats * 0.0
            if not validation:
                # weight gradients of the layers run as grouped launches and land directly in .grad
                with ops.batched_weight_grads(on_ready=self.reducer.mark_ready if self.reducer else None):
                    loss.backward()
                if self.reducer:
                    self.reducer.finish()
                self.optimiser.step_and_update_lr()
            elif o.log:
                self.save_eval_outputs(pred)
            totals += stats
        totals = all_reduce_scalars(totals, self.device).cpu()
        n_word = max(float(totals[2]), 1.0)
        return float(totals[0]) / n_word, float(totals[1]) / n_word

    def save_eval_outputs(self, pred, output_dir="eval_outputs"):
        if self.tgt_bpe is None or _rank() != 0:
            return
        outputs = torch.argmax(pred, dim=2).tolist()
        results = self.tgt_bpe.inverse_transform(outputs)
        directory = os.path.join(self.opt.directory, output_dir)
        os.makedirs(directory, exist_ok=True)
        path = os.path.join(directory, "eval_attempt_" + str(self.opt.current_epoch) + ".txt")
        with open(path, "a") as f:
            for seq in results:
                f.write(seq.split(self.constants.EOS_WORD)[0] + "\n")

    # ------------------------------------------------------------------ inference
    def translate(self):
        """Beam-search every line of opt.src and write opt.output (lib/TransformerModel.py:225-262)."""
        test_loader, max_token_seq_len, is_bpe, decoder = self.load_testdata(self.opt.src, self.opt.vocab)
        self.model.word_prob_prj = nn.LogSoftmax(dim=1)
        self.model.eval()
        translator = Translator(self.opt, new=False)
        translator.model = self.model
        translator.max_token_seq_len = max_token_seq_len
        hypotheses = []
        # Under torch.distributed every rank decodes a contiguous shard of the sentences on its own GPU — the
        # sentences are independent, nothing is exchanged — and rank 0 joins the shards' lines in rank order.
        world, rank = _world(), _rank()
        out_path = self.opt.output
        if world > 1:
            ds = test_loader.dataset
            per = (len(ds) + world - 1) // world
            shard = torch.utils.data.Subset(ds, range(min(len(ds), rank * per), min(len(ds), (rank + 1) * per)))
            test_loader = torch.utils.data.DataLoader(shard, num_workers=0, batch_size=self.opt.batch_size,
                                                      collate_fn=test_loader.collate_fn)
            out_path = "{}.rank{}".format(self.opt.output, rank)
        with open(out_path, "w") as f:
            for batch in test_loader:
                all_hyp, _ = translator.translate_batch(*batch)
                lines = self.translate_decode_bpe(all_hyp, decoder) if is_bpe else \
                    self.translate_decode_dict(all_hyp, decoder)
                for line in lines:
                    f.write(line + "\n")
                hypotheses += all_hyp
        if world > 1:
            import torch.distributed as dist
            dist.barrier()  # every shard's file is complete
            if rank == 0:
                with open(self.opt.output, "w") as f:
                    for r in range(world):
                        part = "{}.rank{}".format(self.opt.output, r)
                        with open(part) as g:
                            f.write(g.read())
                        os.remove(part)
            dist.barrier()
        if self.opt.verbose:
            print("[Info] Finished.")
        return hypotheses

    # ------------------------------------------------------------------ checkpoints
    def save(self, epoch=None, note=None):
        """encoder{_note}.chkpt = {type, model, epoch, settings}; decoder{_note}.chkpt adds the
        generator state (lib/TransformerModel.py:264-318).  save_mode: all | best."""
        if _rank() != 0:
            return
        if getattr(self.opt, "telegram", None):
            del self.opt.telegram
        enc = {"type": "transformer", "model": self.model.encoder.state_dict(), "epoch": epoch, "settings": self.opt}
        dec = {"type": "transformer", "model": self.model.decoder.state_dict(),
               "generator": self.model.generator.state_dict(), "epoch": epoch, "settings": self.opt}
        if not self.opt.save_model:
            if not self.save_trip:
                if self.opt.verbose:
                    print("    - [Warning]: the model is not specified to save.")
                self.save_trip = True
            return
        if self.opt.save_mode == "all":
            name = "_" + str(note or "")
        elif self.valid_accs and self.valid_accs[-1] >= max(self.valid_accs):
            name = ""
            if self.opt.verbose:
                print("    - [Info] The checkpoint file has been updated.")
        else:
            return
        os.makedirs(self.opt.directory, exist_ok=True)
        torch.save(enc, os.path.join(self.opt.directory, "encoder" + name + ".chkpt"))
        torch.save(dec, os.path.join(self.opt.directory, "decoder" + name + ".chkpt"))

    # ------------------------------------------------------------------ transformer specific
    def change_max_seq_len(self, n_position):
        """Rebuild both (frozen) position tables for longer inputs (lib/TransformerModel.py:324-338)."""
        for part, emb in ((self.model.encoder, self.model.encoder.src_word_emb),
                          (self.model.decoder, self.model.decoder.tgt_word_emb)):
            part.position_enc = nn.Embedding.from_pretrained(
                get_sinusoid_encoding_table(n_position, emb.weight.shape[1], padding_idx=0), freeze=True).to(self.device)
