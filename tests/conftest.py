import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The GPU tests share the process with torch (device buffers for the bound-output and exchange tests).  torch brings its
    own HIP runtime; initialising it before the library's first context keeps the order of the tests from deciding which
    runtime enumerates the device first (seen: 'No HIP GPUs are available' from a late torch.cuda init)."""
    markexpr = getattr(session.config.option, "markexpr", "") or ""
    if "not gpu" in markexpr:
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:  # no torch, no GPU: the tests that need them say so themselves
        pass
