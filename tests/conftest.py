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
    three-piece products (ops.set_gemm_pieces(3)) and the head_dim-32 attention with exact three-way splits.  The shipped default --
    two rounded pieces per operand -- is what every model-level file (test_model_gpu, test_configs_gpu, test_ddp_gpu, ...) runs, and
    test_gemm_two_piece_products / test_attention_two_piece_default / test_kernel_mlp_two_piece_default pin its own bars."""
    if request.node.fspath.basename != "test_ops_gpu.py" or "two_piece" in request.node.name:
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    from gaot_amd import ops, _lib
    lib = _lib.load()
    old = ops.set_gemm_pieces(3)
    old_p, old_op = lib.gaot_debug_set_attention_p_pieces(33), lib.gaot_debug_set_attention_operand_pieces(3)
    try:
        yield
    finally:
        ops.set_gemm_pieces(**old)
        lib.gaot_debug_set_attention_p_pieces(old_p)
        lib.gaot_debug_set_attention_operand_pieces(old_op)
