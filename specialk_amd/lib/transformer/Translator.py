"""Beam-search translation with the reference's interface
(lib/transformer/Translator.py:14-226): Translator(opt, new=True), an assignable `.model`
and `.max_token_seq_len`, and translate_batch(src_seq, src_pos) -> (hypotheses, scores)."""
import torch
import torch.nn as nn

from ...core.serialization import load_trusted_types_only
from ...decode import BeamDecoder
from .Models import Transformer


class Translator(object):
    """Load with trained model and handle the beam search"""

    def __init__(self, opt, new=True):
        self.opt = opt
        if getattr(opt, "cuda_device", None):
            self.device = torch.device("cuda:" + str(opt.cuda_device))
        else:
            self.device = torch.device("cuda")  # this package has no CPU path
        if new:
            checkpoint = load_trusted_types_only(opt.model, "a model checkpoint")
            model_opt = checkpoint["settings"]
            self.model_opt = model_opt
            self.max_token_seq_len = model_opt.max_token_seq_len
            model = Transformer(
                model_opt.src_vocab_size, model_opt.tgt_vocab_size, model_opt.max_token_seq_len,
                tgt_emb_prj_weight_sharing=model_opt.proj_share_weight,
                emb_src_tgt_weight_sharing=model_opt.embs_share_weight,
                d_k=model_opt.d_k, d_v=model_opt.d_v, d_model=model_opt.d_model, d_word_vec=model_opt.d_word_vec,
                d_inner=model_opt.d_inner_hid, n_layers=model_opt.n_layers, n_head=model_opt.n_head,
                dropout=model_opt.dropout)
            model.load_state_dict(checkpoint["model"])
            print("[Info] Trained model state loaded.")
            model.word_prob_prj = nn.LogSoftmax(dim=1)
            self.model = model.to(self.device)
            self.model.eval()

    def translate_batch(self, src_seq, src_pos):
        """Translation work in one batch: returns (batch_hyp, batch_scores) where
        batch_hyp[i][j] is the j-th best token list of sentence i (without SOS) and
        batch_scores[i] the matching n_best summed log-probabilities."""
        src_seq, src_pos = src_seq.to(self.device), src_pos.to(self.device)
        # one decoder for as long as the model, the beam size and the length limit stay what they are (all three are
        # assignable): it keeps the prepared weights, and per batch shape its buffers and captured graph, between calls
        key = (self.model, self.opt.beam_size, self.max_token_seq_len)
        held = getattr(self, "_decoder", None)
        if held is None or held[0][0] is not key[0] or held[0][1:] != key[1:]:
            held = self._decoder = (key, BeamDecoder(*key))
        return held[1].run(src_seq, src_pos, self.opt.n_best)
