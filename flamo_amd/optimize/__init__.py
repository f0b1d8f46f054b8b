"""The criteria of flamo.optimize.loss that sit directly on the hot path's output, evaluated by the library's kernels."""
from .loss import mse_loss, sparsity_loss  # noqa: F401
