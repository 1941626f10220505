"""segment_csr with the call signature of the reference wrapper (utils/segment_csr.py:14-55), on the HIP
segment-reduce kernel.  sum / mean over CSR segments of [E, C] or [B, E, C]; empty segment -> 0."""
import torch

from .... import ops
from ....plan import GeometryPlan


def segment_csr(src: torch.Tensor, indptr: torch.Tensor, reduce: str = "sum", use_scatter: bool = True):
    if reduce not in ("sum", "mean"):
        raise ValueError("reduce must be one of 'mean', 'sum' (max lives inside the fused segment softmax)")
    ip = indptr.reshape(-1, indptr.shape[-1])[0]
    E = src.shape[-2] if src.dim() >= 2 else src.shape[0]
    plan = GeometryPlan(torch.zeros(E, dtype=torch.long, device=src.device), ip, n_src=1)
    x = src if src.dim() == 3 else (src[None] if src.dim() == 2 else src[None, :, None])
    scale = (1.0 / plan.deg.clamp(min=1).to(torch.float32)) if reduce == "mean" else None
    out = ops.segment_sum(x, plan, scale)
    return out if src.dim() == 3 else (out[0] if src.dim() == 2 else out[0, :, 0])
