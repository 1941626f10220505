import sys, os, json, torch
sys.path.insert(0, os.getcwd())
import tools.fuzz_parity as F
from gaot_amd import ops
c = F.draw(int(sys.argv[1]))
rec = []
orig = ops.mlp_chain
def spy(x, weights, biases, acts):
    y = orig(x, weights, biases, acts)
    if list(acts) == ["relu", "relu"] and not rec:
        rec.append((x.detach().cpu(), [w.detach().cpu() for w in weights], [b.detach().cpu() for b in biases], y.detach().cpu()))
    return y
ops.mlp_chain = spy
ok, info = F.run(c, torch.device("cuda:0"))
x, ws, bs, y = rec[0]
z1 = x @ ws[0].t() + bs[0]; h1 = torch.relu(z1); z2 = h1 @ ws[1].t() + bs[1]            # the fp32 reference's arithmetic (CPU)
z1d = x.double() @ ws[0].double().t() + bs[0].double(); h1d = torch.relu(z1d); z2d = h1d @ ws[1].double().t() + bs[1].double()
flips2 = ((y > 0) != (z2 > 0))
print(json.dumps({"rows": list(x.shape), "layer2_gate_flips_hip_vs_fp32_reference": int(flips2.sum()), "of": flips2.numel(),
                  "abs_z64_at_flips": [float(v) for v in z2d[flips2].abs().tolist()][:8],
                  "layer1_gates_fp32_vs_fp64": int(((z1 > 0) != (z1d > 0)).sum()), "layer2_gates_fp32_vs_fp64": int(((z2 > 0) != (z2d > 0)).sum()),
                  "layer2_hip_vs_fp64": int(((y > 0) != (z2d > 0)).sum()), "min_abs_z2_64": float(z2d.abs().min()), "min_abs_z1_64": float(z1d.abs().min())}))
