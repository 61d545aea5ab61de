import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_sessionstart(session):
    # the torch fp32 restatement is the checker: pin its thread count so that its own reduction order (and with it the
    # last bits of the reference values the GPU path is compared against) does not change with the host's core count
    try:
        import os
        import torch
        torch.set_num_threads(int(os.environ.get("LZ_TEST_TORCH_THREADS", "8")))
    except Exception:
        pass
