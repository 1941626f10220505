#!/usr/bin/env python3
"""Per-family HBM traffic files (the `traffic` entry of bench.py's roofline blocks) from a pmc_report.py summary.
usage: python tools/traffic_from_pmc.py profiles/<tag>_pmc_c2.json <tag> <commit>   (every transform kernel runs once per step:
the least-launched one counts the steps the counters saw)
writes profiles/<tag>_gemm_traffic.json, profiles/<tag>_gno_traffic.json, profiles/<tag>_attn_bwd_traffic.json and points
profiles/current_traffic.json at them."""
import json, os, sys
src, tag, commit = sys.argv[1], sys.argv[2], sys.argv[3]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
d = json.load(open(src))
prof = os.path.dirname(src)
def dump(name, obj):
    with open(os.path.join(prof, f"{tag}_{name}_traffic.json"), "w") as f:
        json.dump(obj, f, indent=1)
# GEMM tile kernels: launch-weighted mean bytes per launch
gk = {k: v for k, v in d.items() if k.startswith(("gemm_split_kernel", "gemm_ad_kernel", "gemm_tn_grouped_kernel"))}       # the launches bench.py's `roofline` covers
n = sum(v["launches_seen"] for v in gk.values())
dump("gemm", {"kernel_family": f"matrix-pipe GEMM kernels (gemm_split_kernel / gemm_ad_kernel tiles, gemm_tn_grouped_kernel) over {steps} eager steps of the bench configuration",
              "launches": n, "hbm_bytes_per_launch": sum(v["hbm_bytes_per_launch"] * v["launches_seen"] for v in gk.values()) / max(n, 1),
              "source": f"{src} (FETCH_SIZE KiB x2 + WRITE_SIZE KiB per launch, launch-weighted mean)"})
# integral-transform kernels: the family's launches of ONE step
nk = {k: v for k, v in d.items() if k.startswith(("lift_", "proj_", "ep_fixup", "gno_")) and not k.startswith("proj_fold")}      # (proj_fold_*: the projection fold, not a transform)
dump("gno", {"kernel_family": "integral-transform kernels of one step: " + ", ".join(sorted(nk)),
             "hbm_bytes_per_launch": sum(v["hbm_bytes_per_launch"] * v["launches_seen"] / min(w["launches_seen"] for w in nk.values()) for v in nk.values()),
             "note": "sum over the family's launches of ONE step", "source": src})
ab = [(k, v) for k, v in d.items() if k.startswith(("attn_bwd_h16", "attn_bwd_split8"))]
if ab:
    dump("attn_bwd", {"kernel_family": ab[0][0], "launches": ab[0][1]["launches_seen"], "hbm_bytes_per_launch": ab[0][1]["hbm_bytes_per_launch"], "source": src})
cur = {"gemm": {"file": f"{tag}_gemm_traffic.json", "commit": commit}, "gno": {"file": f"{tag}_gno_traffic.json", "commit": commit}}
if ab:
    cur["attn_bwd"] = {"file": f"{tag}_attn_bwd_traffic.json", "commit": commit}
with open(os.path.join(prof, "current_traffic.json"), "w") as f:
    json.dump(cur, f, indent=1)
print(json.dumps(cur))
