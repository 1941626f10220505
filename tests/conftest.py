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


@pytest.fixture(autouse=True)
def _exact_gemm_products_in_kernel_level_tests(request):
    """tests/test_ops_gpu.py pins the kernels against float64 math at fp32-rounding bars: it runs the tile GEMMs with exact
    three-piece products (ops.set_gemm_pieces(3)).  The shipped default -- two rounded pieces per operand -- is what every
    model-level file (test_model_gpu, test_configs_gpu, test_ddp_gpu, ...) runs, and test_gemm_two_piece_products pins its own bars."""
    if request.node.fspath.basename != "test_ops_gpu.py" or "two_piece" in request.node.name:
        yield
        return
    from gaot_amd import ops
    old = ops.set_gemm_pieces(3)
    try:
        yield
    finally:
        ops.set_gemm_pieces(**old)
