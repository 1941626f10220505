"""segment_csr with the call signature of the reference wrapper (utils/segment_csr.py:14-55), on the HIP segment-reduce kernels.
sum / mean / max over CSR segments of [E], [E, C] or [B, E, C]; an empty segment gives 0 (torch_scatter's convention for all three)."""
import torch

from .... import ops
from ....plan import GeometryPlan


def segment_csr(src: torch.Tensor, indptr: torch.Tensor, reduce: str = "sum", use_scatter: bool = True):
    # the reference's native branch knows mean and sum only (segment_csr.py:46-47); 'max' is torch_scatter's (agno.py:131-133)
    if reduce not in ("sum", "mean", "max") or (not use_scatter and reduce == "max"):
        raise ValueError("reduce must be one of 'mean', 'sum'")
    ip = indptr.reshape(-1, indptr.shape[-1])[0]
    E = src.shape[-2] if src.dim() >= 2 else src.shape[0]
    plan = GeometryPlan(torch.zeros(E, dtype=torch.long, device=src.device), ip, n_src=1)
    if reduce == "max":
        if src.dim() == 3:          # [B, E, C] -> rows of [E, B * C]: one launch of the per-row maximum kernel (gaot_segment_max_fwd)
            B, _, Cc = src.shape
            out = ops.segment_max(src.permute(1, 0, 2).reshape(E, B * Cc), plan)
            return out.reshape(plan.Q, B, Cc).permute(1, 0, 2).contiguous()
        out = ops.segment_max(src if src.dim() == 2 else src[:, None], plan)
        return out if src.dim() == 2 else out[:, 0]
    x = src if src.dim() == 3 else (src[None] if src.dim() == 2 else src[None, :, None])
    scale = (1.0 / plan.deg.clamp(min=1).to(torch.float32)) if reduce == "mean" else None
    out = ops.segment_sum(x, plan, scale)
    return out if src.dim() == 3 else (out[0] if src.dim() == 2 else out[0, :, 0])
