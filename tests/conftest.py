import os as _os_env
_os_env.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # see flamo_amd/__init__.py: must precede HIP runtime init

import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    arrays = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    return meta, arrays


def golden_names(prefix=""):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and f.startswith(prefix))


def relerr(a, b):
    a = torch.as_tensor(a)
    b = torch.as_tensor(b)
    den = torch.linalg.vector_norm(b.to(torch.complex128 if b.is_complex() else torch.float64))
    num = torch.linalg.vector_norm((a.to(b.device) - b).to(torch.complex128 if b.is_complex() else torch.float64))
    return (num / den.clamp_min(1e-300)).item()


@pytest.fixture(scope="session")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
