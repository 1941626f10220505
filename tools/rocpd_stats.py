#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into per-kernel stats (like --stats CSV).
usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db [steps_in_trace] > profiles/xyz.txt"""
import collections, re, sqlite3, sys

db = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
c = sqlite3.connect(db)
suf = [r[0] for r in c.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace('rocpd_kernel_dispatch', '')
rows = c.execute(f"select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch{suf} d join rocpd_info_kernel_symbol{suf} s on d.kernel_id=s.id").fetchall()
agg = collections.defaultdict(lambda: [0, 0, 10**18, 0])
for n, s, e in rows:
    a = agg[n]; d = e - s
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(v[1] for v in agg.values())
if steps is None or steps <= 0:          # one optimizer launch per step: count them instead of trusting the caller
    n_opt = [v[0] for k, v in agg.items() if 'adamw_kernel' in k]
    steps = float(n_opt[0]) if n_opt else None
def demangle(n):
    n = re.sub(r'\.kd$', '', n)
    m = re.match(r'_ZN4gaot(\d+)([A-Za-z_0-9]+)', n)
    if m:
        name = m.group(2)[:int(m.group(1))]
        t = re.search(r'I(Li\d+E|Lb[01]E)+E', n)
        targs = ''
        if t:
            targs = '<' + ','.join(x[2:-1] if x.startswith('Li') else ('T' if x[2] == '1' else 'F') for x in re.findall(r'Li\d+E|Lb[01]E', t.group(0))) + '>'
        return 'gaot::' + name + targs
    return n[:100]
print(f"# {db}: {len(rows)} dispatches, {tot/1e6:.2f} ms of kernel time" + (f", {tot/1e6/steps:.3f} ms/step over {steps:g} steps" if steps else ""))
print(f"{'%':>6} {'total_ms':>9} {'calls':>7} {'avg_us':>9} {'min_us':>8} {'max_us':>8}  kernel")
for n, (cnt, ns, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if ns / tot < 0.0005: continue
    print(f"{ns/tot*100:6.2f} {ns/1e6:9.2f} {cnt:7d} {ns/cnt/1e3:9.1f} {mn/1e3:8.1f} {mx/1e3:8.1f}  {demangle(n)}")
