"""spadot_amd -- MI355X-native implementation of SpaDOT's training hot path.

`train` mirrors SpaDOT.train (reference SpaDOT/__init__.py:1-5); the preprocess/analyze stages are
out of scope.  Importing this package does not load the HIP libraries; the first
numeric call does, and fails loudly if they have not been built (python -m spadot_amd.csrc.build)."""
from .train import train  # noqa: F401  (binds the function over the submodule name, as the reference does)

__all__ = ["train"]
