import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    from timeviper_amd.build import ensure_built
    ensure_built()          # hipcc cross-compiles without a GPU; no-op when the .so is in the tree


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests are skipped automatically when no GPU is visible so that a
    # plain `pytest tests/` stays green on the CPU-only build container.
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    z = np.load(GOLDEN / f"{name}.npz")
    return {k: z[k] for k in z.files}


def golden_state_dict(g, prefix="w."):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden():
    return load_golden
