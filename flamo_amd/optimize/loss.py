"""``mse_loss`` and ``sparsity_loss`` of the reference (flamo/optimize/loss.py:12-103) on the library's kernels -- the two
criteria of the colorless-FDN training (examples/e8_colorless_fdn.py:137-138).

The reference's training loop calls ``criterion(estimations, targets)`` (flamo/optimize/trainer.py:179-189); with
``flamo_amd.optimize.mse_loss`` in that list the loop runs unedited and the criterion costs one streaming pass over the
prediction each way (``ops.mse``) instead of torch's sum / sub / pow / mean kernels and their backward."""
import torch
from torch import nn

from .. import ops


class mse_loss(nn.Module):
    """Wrapper for the mean squared error loss: nn.MSELoss()(y_pred.sum(-1), y_true.squeeze(-1)), as
    flamo/optimize/loss.py:101-102.  Same constructor and attributes (nfft, device, mse_loss, name)."""

    def __init__(self, nfft: int = None, device: str = "cpu"):
        super().__init__()
        self.nfft = nfft
        self.device = device
        self.mse_loss = nn.MSELoss()
        self.name = "MSE"

    def forward(self, y_pred, y_true):
        if (torch.is_tensor(y_pred) and y_pred.is_cuda and y_pred.dtype in (torch.float32, torch.float64) and y_pred.dim() >= 1
                and y_true.is_cuda and tuple(y_true.squeeze(-1).shape) == tuple(y_pred.shape[:-1]) and y_pred.numel() > 0
                and not y_true.requires_grad and y_true.dtype == y_pred.dtype and y_pred.shape[-1] <= ops.MSE_MAX_COLS):
            return ops.mse(y_pred, y_true, sum_last=True)
        # anything else (host tensors, complex predictions, a target that takes a gradient or has another dtype, more
        # summed columns than the kernel takes): the reference's own lines
        y_pred_sum = torch.sum(y_pred, dim=-1)
        return self.mse_loss(y_pred_sum, y_true.squeeze(-1))


# where the reference looks for the mixing matrix of an FDN core, in its order (optimize/loss.py:41-49)
_MIXING_MATRIX_PATHS = (
    ("feedback_loop", "feedback"),
    ("feedback_loop", "feedback", "mixing_matrix"),
    ("branchA", "feedback_loop", "feedback", "mixing_matrix"),
)


def _mapped_mixing_matrix(core):
    """(module, map(param)) of the first place of ``_MIXING_MATRIX_PATHS`` that has a mapped parameter; the last place's
    own exception propagates when none has (the reference's nested try / except ends the same way)."""
    for n, path in enumerate(_MIXING_MATRIX_PATHS):
        try:
            module = core
            for name in path:
                module = getattr(module, name)
            return module, module.map(module.param)
        except Exception:
            if n == len(_MIXING_MATRIX_PATHS) - 1:
                raise


class sparsity_loss(nn.Module):
    """Sparsity of the feedback matrix of an FDN model's core, (sum|A| - N sqrt N) / (N (1 - sqrt N)) -- flamo/optimize/loss.py:12-63.
    Same signature (``y_pred`` and ``y_target`` are accepted and ignored, as flamo.optimize.trainer.Trainer passes them) and the same
    places the mixing matrix is looked for; a ``HouseholderMatrix`` holds the unit vector u and the criterion is that of
    ``I - 2 u u^T`` (loss.py:51-53); a (C, N, N) stack gives the mean over C.  On real square device matrices the criterion is one
    launch each way (``ops.sparsity``) instead of torch's abs / sum / sub / div / neg launches and their backward; complex
    matrices (the Householder form: its map returns a complex vector), host tensors and anything else take the torch lines."""

    def forward(self, y_pred, y_target, model):
        from ..processor.dsp import HouseholderMatrix

        mixing_matrix, A = _mapped_mixing_matrix(model.get_core())
        if isinstance(mixing_matrix, HouseholderMatrix):
            u = A
            A = torch.eye(u.shape[0], device=u.device, dtype=u.dtype) - 2 * u @ u.T
        N = A.shape[-1]
        root = N ** 0.5
        if (A.is_cuda and A.dtype in (torch.float32, torch.float64) and A.dim() in (2, 3) and A.shape[-2] == N and N >= 2
                and A.numel() > 0):
            return ops.sparsity(A)
        if A.dim() == 3:
            return torch.mean((torch.sum(torch.abs(A), dim=(-2, -1)) - N * root) / (N * (1 - root)))
        return -(torch.sum(torch.abs(A)) - N * root) / (N * (root - 1))
