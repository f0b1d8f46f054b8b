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
                and not y_true.requires_grad):
            return ops.mse(y_pred, y_true, sum_last=True)
        # anything else (host tensors, complex predictions, a target that takes a gradient): the reference's own lines
        y_pred_sum = torch.sum(y_pred, dim=-1)
        return self.mse_loss(y_pred_sum, y_true.squeeze(-1))


class sparsity_loss(nn.Module):
    """Sparsity of the feedback matrix of an FDN model's core, (sum|A| - N sqrt N) / (N (1 - sqrt N)) -- flamo/optimize/loss.py:12-63.
    Same signature (``y_pred`` and ``y_target`` are accepted and ignored, as flamo.optimize.trainer.Trainer passes them) and the same
    places the mixing matrix is looked for; a (C, N, N) stack gives the mean over C.  On device tensors the criterion is one launch
    each way (``ops.sparsity``) instead of torch's abs / sum / sub / div / neg launches and their backward."""

    def forward(self, y_pred, y_target, model):
        core = model.get_core()
        try:
            mixing_matrix = core.feedback_loop.feedback
            A = mixing_matrix.map(mixing_matrix.param)
        except Exception:
            try:
                mixing_matrix = core.feedback_loop.feedback.mixing_matrix
                A = mixing_matrix.map(mixing_matrix.param)
            except Exception:
                mixing_matrix = core.branchA.feedback_loop.feedback.mixing_matrix
                A = mixing_matrix.map(mixing_matrix.param)
        N = A.shape[-1]
        if (A.is_cuda and A.dtype in (torch.float32, torch.float64) and A.dim() in (2, 3) and A.shape[-2] == N and N >= 2
                and A.numel() > 0):
            return ops.sparsity(A)
        if A.dim() == 3:
            return torch.mean((torch.sum(torch.abs(A), dim=(-2, -1)) - N * N ** 0.5) / (N * (1 - N ** 0.5)))
        return -(torch.sum(torch.abs(A)) - N * N ** 0.5) / (N * (N ** 0.5 - 1))
