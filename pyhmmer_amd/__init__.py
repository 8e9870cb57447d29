"""pyhmmer_amd -- MI355X-native drop-in for pyhmmer's ``plan7.Pipeline`` / ``hmmer.hmmsearch`` path.

Only the hot path (HMMER3's ``p7_Pipeline`` filter cascade) and the callers / data formats either
side of it are implemented; see DESIGN.md for scope and INTEGRATION.md for the reference-side binding.
"""
from . import easel, errors, plan7  # noqa: F401

__version__ = "0.1.0"
