import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'g-nerf_amd')
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# The drop-in root (mirrors the reference's g_nerf/ directory: it is put on sys.path and
# `training.*` / `torch_utils.*` resolve inside it) and the repo root (for `oracle`).
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name)))
    return load


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
