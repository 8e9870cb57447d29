"""pyhmmer_amd -- MI355X-native drop-in for pyhmmer's ``plan7.Pipeline`` / ``hmmer.hmmsearch`` path.

Only the hot path (HMMER3's ``p7_Pipeline`` filter cascade) and the callers / data formats either
side of it are implemented; see DESIGN.md for scope and INTEGRATION.md for the reference-side binding.
"""
import os as _os

# The kernel classes of a batch of queries run on up to eight streams per search in flight: ask the HIP runtime for
# more hardware queues than its default of four.  Only effective when set before the process makes its first HIP
# call (import this package before torch, or export the variable); never overrides the user's setting.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import easel, errors, plan7  # noqa: E402,F401

__version__ = "0.1.0"
