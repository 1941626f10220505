"""Checkpoint files interchangeable with the reference trainers (src/core/trainer_utils.py:23-92; called from
base_trainer.py:181-189 with `model=` only): ONE torch.save of {name: state_dict} for any stateful objects.

gaot_amd.model.GAOT has the reference's state_dict keys and key order (tests/test_host_cpu.py), and
trainer.FlatAdamW.state_dict() has torch.optim.AdamW's layout, so a file written on either side loads on the other.
Loading reconciles a leading 'module.' (DataParallel / DistributedDataParallel wrappers) in either direction and is
non-strict, like the reference's."""
from typing import Any, Dict, List

import torch


def save_ckpt(path: str, **stateful) -> None:
    torch.save({name: obj.state_dict() for name, obj in stateful.items()}, path)


def _reconcile_prefix(saved: Dict[str, Any], wanted_keys) -> Dict[str, Any]:
    pre = "module."
    saved_all = len(saved) > 0 and all(isinstance(k, str) and k.startswith(pre) for k in saved)
    saved_none = not any(isinstance(k, str) and k.startswith(pre) for k in saved)
    want_all = len(wanted_keys) > 0 and all(k.startswith(pre) for k in wanted_keys)
    want_none = not any(k.startswith(pre) for k in wanted_keys)
    if saved_all and want_none:
        return {k[len(pre):]: v for k, v in saved.items()}
    if saved_none and want_all:
        return {pre + k: v for k, v in saved.items()}
    return saved


def load_ckpt(path: str, map_location=None, **stateful) -> List[Any]:
    ckpt = torch.load(path, map_location=map_location)
    for name, obj in stateful.items():
        sd = ckpt[name]
        if isinstance(obj, torch.nn.Module):
            obj.load_state_dict(_reconcile_prefix(sd, list(obj.state_dict().keys())), strict=False)
        else:
            obj.load_state_dict(sd)
    return list(stateful.values())
