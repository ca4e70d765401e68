"""Per-sentence beam state with the reference's interface (lib/transformer/Beam.py:14-110).

The translator in this package keeps all beams of a batch in device tensors
(specialk_amd.decode.BeamDecoder); this class exists for callers that drive a single beam by
hand.  advance() runs the same fused top-k kernel.  The back-pointer `prev_k` is the integer
(floor) quotient of the flat id by the vocabulary size — what the reference's `/` meant on the
torch version it was written for (Beam.py:70).
"""
import torch

from ... import _lib as L
from ...core import constants as Constants


class Beam:
    """Beam search"""

    def __init__(self, size, device=False):
        self.size = size
        self._done = False
        dev = device if device else "cuda"
        self.scores = torch.zeros((size,), dtype=torch.float, device=dev)
        self.all_scores = []
        self.prev_ks = []
        self.next_ys = [torch.full((size,), Constants.PAD, dtype=torch.long, device=dev)]
        self.next_ys[0][0] = Constants.SOS

    def get_current_state(self):
        "Get the outputs for the current timestep."
        return self.get_tentative_hypothesis()

    def get_current_origin(self):
        "Get the backpointers for the current timestep."
        return self.prev_ks[-1]

    @property
    def done(self):
        return self._done

    def advance(self, word_prob):
        "Update beam status with log-probabilities (size, n_words) and check if finished or not."
        word_prob = word_prob.to(torch.float32).contiguous()
        first = len(self.prev_ks) == 0
        scores, prev, tok = L.beam_topk(word_prob, self.scores.view(1, -1).contiguous(), 1, self.size, first,
                                        normalize=False)
        self.all_scores.append(self.scores)
        self.scores = scores.view(-1)
        self.prev_ks.append(prev.view(-1).long())
        self.next_ys.append(tok.view(-1).long())
        # End condition is when top-of-beam is EOS.
        if self.next_ys[-1][0].item() == Constants.EOS:
            self._done = True
            self.all_scores.append(self.scores)
        return self._done

    def sort_scores(self):
        "Sort the scores."
        return torch.sort(self.scores, 0, True)

    def get_the_best_score_and_idx(self):
        "Get the score of the best in the beam."
        scores, ids = self.sort_scores()
        return scores[1], ids[1]

    def get_tentative_hypothesis(self):
        "Get the decoded sequence for the current timestep."
        if len(self.next_ys) == 1:
            return self.next_ys[0].unsqueeze(1)
        _, keys = self.sort_scores()
        hyps = [[Constants.SOS] + self.get_hypothesis(k) for k in keys]
        return torch.LongTensor(hyps)

    def get_hypothesis(self, k):
        """Walk back to construct the full hypothesis."""
        k = int(k)
        ys = [y.tolist() for y in self.next_ys]
        ks = [p.tolist() for p in self.prev_ks]
        hyp = []
        for j in range(len(ks) - 1, -1, -1):
            hyp.append(ys[j + 1][k])
            k = ks[j][k]
        return hyp[::-1]
