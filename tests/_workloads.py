"""Synthetic geometries shaped like BASELINE.json's configs (shared by the -m gpu parity tests and tools/bench_configs.py).
Pure torch on the CPU, seeded: the same call gives the same points here and on the GPU box."""
import torch


def grid(sizes, lo: float = -1.0, hi: float = 1.0) -> torch.Tensor:
    """latent token coordinates: meshgrid(linspace) 'ij', flattened (the reference's data_processor.py:289-294)"""
    axes = [torch.linspace(lo, hi, n) for n in sizes]
    return torch.stack(torch.meshgrid(*axes, indexing="ij"), -1).reshape(-1, len(sizes))


def naca_points(n: int, g: torch.Generator, spread: float = 0.25) -> torch.Tensor:
    """[n, 2] points in [-1,1]^2 whose density falls off exponentially (scale `spread`) with the distance to a
    NACA0012 contour: the degree skew of an airfoil mesh (C3, "irregular neighbourhood stress").  At n = 8192 against a
    64x64 latent grid with radius 0.033: spread 0.25 -> max encoder degree ~230, ~1 500 empty latent rows;
    spread 0.15 -> max degree ~340, ~2 300 empty rows; decoder degree <= 5 either way."""
    t = torch.rand(n * 6, generator=g)
    xc = t ** 2
    yt = 0.6 * (0.2969 * xc.sqrt() - 0.1260 * xc - 0.3516 * xc ** 2 + 0.2843 * xc ** 3 - 0.1015 * xc ** 4)
    side = (torch.rand(n * 6, generator=g) < 0.5).float() * 2 - 1
    r = torch.empty(n * 6).exponential_(1.0, generator=g) * spread
    ang = torch.rand(n * 6, generator=g) * 6.2832
    px = (xc - 0.5) * 0.9 + r * torch.cos(ang)
    py = side * yt * 0.9 + r * torch.sin(ang)
    keep = (px.abs() <= 1) & (py.abs() <= 1)
    pts = torch.stack([px[keep], py[keep]], -1)[:n]
    assert pts.shape[0] == n
    return pts.contiguous()


def shell_points(n: int, g: torch.Generator) -> torch.Tensor:
    """[n, 3] car-ish surface cloud: union of three ellipsoid shells in [-1,1]^3 (C5, DrivAerNet++-shaped)"""
    v = torch.randn(n, 3, generator=g)
    v = v / v.norm(dim=1, keepdim=True)
    which = torch.randint(0, 3, (n,), generator=g)
    ax = torch.tensor([[0.9, 0.4, 0.3], [0.5, 0.35, 0.25], [0.3, 0.3, 0.2]])[which]
    ctr = torch.tensor([[0.0, 0.0, -0.1], [-0.1, 0.0, 0.2], [0.5, 0.0, 0.15]])[which]
    return (v * ax + ctr).clamp(-1, 1).contiguous()


def uniform_points(n: int, d: int, g: torch.Generator) -> torch.Tensor:
    return (torch.rand(n, d, generator=g) * 2 - 1).contiguous()
