#!/usr/bin/env python3
"""Average PMC counters per kernel from a rocprofv3 rocpd db.  usage: pmc_summary.py db [kernel-substring]"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1]); sub = sys.argv[2] if len(sys.argv) > 2 else "gemm_kernel"
t = {r[0].split('_0')[0] if False else r[0]: r[0] for r in c.execute("select name from sqlite_master where type='table'")}
def tab(prefix): return [n for n in t if n.startswith(prefix)][0]
kd, ks, pe, ip = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
cols = [r[1] for r in c.execute(f"pragma table_info({pe})")]
q = f"select s.kernel_name, i.name, e.value from {pe} e join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id join {ip} i on e.pmc_id=i.id"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for kn, cn, v in c.execute(q):
    if sub in kn: agg[kn][cn].append(v)
for kn, d in agg.items():
    print(kn[:90])
    for cn, vs in sorted(d.items()):
        print(f"   {cn:32s} n={len(vs):3d} avg={sum(vs)/len(vs):16.1f}")
