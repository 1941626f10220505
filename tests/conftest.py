import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture
def bf16x2():
    """the opt-in two-piece precision (ops.set_precision("bf16x2")) for the tests that pin that variant; everything else -- kernel-level
    and model-level files alike -- runs the shipped default, the "f32" precision: fp32-level products (two fp16 pieces of the scaled operand
    where magnitude words exist, three bf16 pieces elsewhere)."""
    from gaot_amd import ops
    assert ops.precision() == "f32" and set(ops._PIECES.values()) == {3}            # exact products ARE the default
    old = ops.set_precision("bf16x2")
    try:
        yield
    finally:
        ops.set_gemm_pieces(**old)
