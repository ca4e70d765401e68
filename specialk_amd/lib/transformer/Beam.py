"""One sentence's beam with the interface of the reference's Beam (lib/transformer/Beam.py:14-110): `size`,
`scores`, `all_scores`, `prev_ks`, `next_ys`, `done`, `advance`, `sort_scores`, `get_current_state`,
`get_current_origin`, `get_the_best_score_and_idx`, `get_tentative_hypothesis`, `get_hypothesis`.

The translator of this package holds the beams of a whole batch in device tensors
(specialk_amd.decode.BeamDecoder); this class serves callers that step a single beam by hand, and
`advance` runs the same fused top-k kernel.  Semantics kept from the reference: the first step expands only
row 0 (the other rows start as PAD), later steps rank `word_prob + scores[:, None]` over beam x vocabulary;
the back-pointer is the FLOOR quotient of the flat index by the vocabulary size (the reference's `/` at
Beam.py:70 was integer division on the torch it was written for); a beam is done once its best entry is EOS;
hypotheses are read by walking the back-pointers.
"""
import torch

from ... import _lib as L
from ...core import constants as Constants


class Beam:
    def __init__(self, size, device=False):
        where = device or "cuda"
        start = torch.full((size,), Constants.PAD, dtype=torch.long, device=where)
        start[0] = Constants.SOS
        self.size = size
        self.scores = torch.zeros(size, dtype=torch.float, device=where)
        self.next_ys = [start]   # tokens chosen at every step
        self.prev_ks = []        # beam row each of them extends
        self.all_scores = []
        self._done = False

    done = property(lambda self: self._done)

    def advance(self, word_prob):
        """word_prob (size, n_words) log-probabilities of the next token -> True once the best entry is EOS."""
        logp = word_prob.to(torch.float32).contiguous()
        new_scores, origin, token = L.beam_topk(logp, self.scores.view(1, -1).contiguous(), 1, self.size,
                                                not self.prev_ks, normalize=False)
        self.all_scores.append(self.scores)
        self.scores = new_scores.view(-1)
        self.prev_ks.append(origin.view(-1).long())
        self.next_ys.append(token.view(-1).long())
        if int(self.next_ys[-1][0]) == Constants.EOS:
            self.all_scores.append(self.scores)
            self._done = True
        return self._done

    def sort_scores(self):
        return torch.sort(self.scores, 0, True)

    def get_the_best_score_and_idx(self):
        # the reference returns entry 1 of the sorted beam (Beam.py:85-88); kept as it is
        ranked, order = self.sort_scores()
        return ranked[1], order[1]

    def get_current_origin(self):
        return self.prev_ks[-1]

    def get_current_state(self):
        return self.get_tentative_hypothesis()

    def get_tentative_hypothesis(self):
        """(size, t) token ids of the live hypotheses, best first, each starting with SOS."""
        if not self.prev_ks:
            return self.next_ys[0].unsqueeze(1)
        order = self.sort_scores()[1]
        return torch.LongTensor([[Constants.SOS] + self.get_hypothesis(k) for k in order])

    def get_hypothesis(self, k):
        """Tokens of the hypothesis that ends in beam row k (without SOS)."""
        row = int(k)
        tokens = [y.tolist() for y in self.next_ys[1:]]
        origins = [p.tolist() for p in self.prev_ks]
        out = []
        for step in reversed(range(len(origins))):
            out.append(tokens[step][row])
            row = origins[step][row]
        out.reverse()
        return out
