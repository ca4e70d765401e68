"""Loading of user-supplied checkpoint and dataset files without executing anything from them.

The reference reads its `.chkpt` / `.pt` files with plain `torch.load` (lib/TransformerModel.py:83,118;
lib/nmtModel.py:213,236; lib/transformer/Translator.py:21), i.e. with the unpickler, which runs whatever
callables the file names.  Those files hold only tensors, dicts, lists, strings, numbers and one
`argparse.Namespace` (the saved options), so here they go through torch's weights-only unpickler with
that single class allowed; a file that needs anything else is refused with a message naming it."""
import argparse

import torch


def load_trusted_types_only(path, what, map_location="cpu"):
    """torch.load(path) restricted to tensors, containers, scalars and argparse.Namespace."""
    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            return torch.load(path, map_location=map_location, weights_only=True)
    except FileNotFoundError:
        raise
    except Exception as e:  # the weights-only loader names the global it refused
        raise RuntimeError(
            f"{path}: {what} must hold only tensors, dicts, lists, strings, numbers and an argparse.Namespace; "
            f"the weights-only loader refused it: {e}") from e
