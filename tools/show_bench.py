import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print({k: d[k] for k in ("metric","value","ms_per_step","n_gpus","dtype")})
print("roofline", d["roofline"])
print("cpu_baseline", d["cpu_baseline"])
for c in ("C3","C4","C5"):
    x = d["configs"][c]; print(c, x["samples_per_s"], x["ms_per_step"], x.get("rollout_ms"), x.get("rel_l2_vs_oracle",{}).get("output"))
c3 = d["configs"]["C3"]
for k in ("fixed_batch","shuffled","reference_loop_vx","reference_loop_vx_shuffled","reference_loop_vx_uploaded"):
    print(k, json.dumps(c3[k])[:260])
print("reference_loop", json.dumps(d["reference_loop"])[:300])
