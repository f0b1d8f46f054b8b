"""Small helpers shared by the processor modules (counterpart of flamo/utils.py)."""
import torch


def get_device():
    """'cuda' (ROCm) when a GPU is visible, else 'cpu' (flamo/utils.py:7-9)."""
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


def to_complex(x: torch.Tensor) -> torch.Tensor:
    """Real tensor -> complex tensor with zero imaginary part (flamo/utils.py:12-22)."""
    if x.is_complex():
        return x
    return x.to(torch.complex64 if x.dtype == torch.float32 else (torch.complex128 if x.dtype == torch.float64 else torch.complex32))


def get_frequency_samples(num: int, rho: float = 1.0, device="cpu", dtype=torch.float64):
    """num points z = rho * exp(j*theta), theta linearly spaced on [0, pi] (flamo/utils.py:33-51)."""
    theta = torch.linspace(0, 1, num, device=device, dtype=dtype) * torch.pi
    return rho * torch.exp(1j * theta)
