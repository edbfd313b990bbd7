import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _default_edge_rows():
    """GANStep.step raises the process-wide edge-level threshold to B N^2 / 2 of its batch (dg_set_edge_rows): every test
    starts from the library default, so that kernel-level tests do not depend on which step test ran before them."""
    import torch
    if torch.cuda.is_available():
        from druggen_amd import _lib
        _lib.set_edge_rows(0)
    yield


@pytest.fixture(autouse=True)
def _default_options():
    """Every test starts from, and leaves, the shipped options (druggen_amd/options.py): a test that switches one (the
    equivalence tests of a fused launch against the launches it replaces) cannot leak into the next."""
    from druggen_amd.options import options
    before = options.as_dict()
    yield
    options.set(**before)


@pytest.fixture
def hidden_mode():
    """``hidden_mode("f32")``: options.hidden for the rest of the test (restored by the fixture above)."""
    from druggen_amd.options import options

    def set_mode(mode):
        options.hidden = mode
    return set_mode
