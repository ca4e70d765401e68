"""Special token ids and strings — same values and names as the reference's
core/constants.py:6-16 (they are baked into its datasets and checkpoints)."""

PAD = 0
UNK = 1
SOS = 2
EOS = 3
BLO = 4

PAD_WORD = "<blank>"
UNK_WORD = "<unk>"
SOS_WORD = "<s>"
EOS_WORD = "</s>"
BLO_WORD = "<p>"  # paragraph block


def get_tokens():
    """Special-token strings ordered by id (reference: core/constants.py:19-24)."""
    words = {"PAD": PAD_WORD, "UNK": UNK_WORD, "SOS": SOS_WORD, "EOS": EOS_WORD, "BLO": BLO_WORD}
    ids = {"PAD": PAD, "UNK": UNK, "SOS": SOS, "EOS": EOS, "BLO": BLO}
    return [words[k] for k in sorted(words, key=lambda k: ids[k])]
