"""-m gpu: bench.py's same-node baseline leg (`torch_gpu_steps`: the reference's algorithm as plain eager PyTorch-ROCm ops on cuda:0,
with F.scaled_dot_product_attention and torch.optim.AdamW as the reference calls them) computes the reference's function: its first
prediction / loss on the GPU against the CPU oracle at BASELINE configs[0]'s shape, fx and vx (per-sample lists).  The leg is a
reported baseline, never the product: nothing under gaot_amd/ is involved here."""
import pytest
import torch

from tests._golden import rel_l2
from tests._workloads import grid, uniform_points

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("vx", [False, True])
def test_torch_gpu_baseline_leg_is_the_reference_function(vx):
    import bench
    from oracle import gaot_oracle as O
    from tests.test_configs_gpu import make_model
    dev = torch.device("cuda:0")
    _, sd, ocfg = make_model(1, 1, [64, 64], seed=5)
    g = torch.Generator().manual_seed(5)
    lat = grid([64, 64])
    B, N = 4, 1024
    p, tgt = torch.randn(B, N, 1, generator=g), torch.randn(B, N, 1, generator=g)
    if vx:
        x = torch.stack([uniform_points(N, 2, g) for _ in range(B)])
        enc = [[O.radius_csr(x[b], lat, 0.033)] for b in range(B)]
        dec = [[O.radius_csr(lat, x[b], 0.033)] for b in range(B)]
    else:
        x = uniform_points(N, 2, g)
        enc, dec = [O.radius_csr(x, lat, 0.033)], [O.radius_csr(lat, x, 0.033)]
    okw = dict(latent=lat, xcoord=x, pndata=p, encoder_nbrs=enc, decoder_nbrs=dec)
    loss, _, _, _, pred = O.train_step(sd, ocfg, dict(okw, target=tgt), return_pred=True)
    rate, ms, pred0, loss0 = bench.torch_gpu_steps(sd, ocfg, okw, tgt, dev, B, steps=2, warmup=1)
    assert pred0.is_cuda and rate > 0 and ms > 0
    assert rel_l2(pred0.cpu(), pred) < 1e-5
    assert abs(loss0 - float(loss)) < 1e-5 * abs(float(loss))
