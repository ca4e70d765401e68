"""Alias of core.constants (the reference keeps a second copy at lib/transformer/Constants.py)."""
from ...core.constants import *  # noqa: F401,F403
