"""Training-time neighbour sub-sampling (edge_drop.py:8-106 semantics): the reference's FUNCTION, for callers that want a CSR dict back.

Off by default (MAGNOConfig.sampling_strategy=None).  The function returns tensors whose sizes depend on the draw, so it reads sizes back to
the host; the MODEL does not call it on the GPU: MAGNO's training passes draw on the device into static buffers with a device-resident seed
(plan.DropPlan, csrc/edge_drop.hip: no host value depends on the draw, a captured step draws a fresh subset on every replay).  This function
serves host tensors (CPU plumbing tests) and vx batches that cannot use static unions (node_embedding)."""
from typing import Dict, Optional

import torch


def apply_edge_drop_csr(neighbors: Dict[str, torch.Tensor], sampling_strategy: Optional[str],
                        max_neighbors: Optional[int] = None, sample_ratio: Optional[float] = None,
                        training: bool = True) -> Dict[str, torch.Tensor]:
    if not training or sampling_strategy is None:
        return neighbors
    index, splits = neighbors["neighbors_index"], neighbors["neighbors_row_splits"]
    E = index.numel()
    if E == 0:
        return neighbors
    Q = splits.numel() - 1
    deg = splits[1:] - splits[:-1]
    if bool((deg < 0).any()) or int(deg.sum()) != E:
        raise ValueError("Invalid CSR structure before edge sampling.")
    dev = index.device
    qid = torch.repeat_interleave(torch.arange(Q, device=dev), deg)
    if sampling_strategy == 'ratio':
        if sample_ratio is None or sample_ratio >= 1.0:
            return neighbors
        keep = torch.rand(E, device=dev) < sample_ratio
    elif sampling_strategy == 'max_neighbors':
        if max_neighbors is None or not bool((deg > max_neighbors).any()):
            return neighbors
        # rank of a random key inside each segment < max_neighbors  ==  uniform subset of that size
        key = torch.rand(E, device=dev)
        order = torch.argsort(qid.double() * 2.0 + key.double())       # segment-major, random inside a segment
        rank = torch.empty(E, dtype=torch.long, device=dev)
        rank[order] = torch.arange(E, device=dev) - splits[:-1][qid[order]]
        keep = rank < max_neighbors
    else:
        return neighbors
    new_deg = torch.bincount(qid[keep], minlength=Q)
    new_splits = torch.zeros(Q + 1, dtype=splits.dtype, device=dev)
    torch.cumsum(new_deg, dim=0, out=new_splits[1:])
    return {"neighbors_index": index[keep], "neighbors_row_splits": new_splits}
