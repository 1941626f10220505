#!/usr/bin/env python3
"""One step of the reference-shaped loop (tools/ref_loop_trace.py) from a kernel trace: the launches between the last two mean-reduction
(MSELoss) kernels, gaot kernels folded into runs.  usage: loop_sequence.py trace.db"""
import re, sqlite3, sys
db = sys.argv[1]
c = sqlite3.connect(db)
suf = [r[0] for r in c.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace('rocpd_kernel_dispatch', '')
rows = c.execute(f"select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch{suf} d join rocpd_info_kernel_symbol{suf} s on d.kernel_id=s.id order by d.start").fetchall()
marks = [i for i, r in enumerate(rows) if 'MeanOps' in r[0]]
a, b = marks[-2], marks[-1]
step = rows[a:b]
t0, prev_end = step[0][1], step[0][1]
busy = sum(e - s for _, s, e in step)
print(f"# {len(step)} dispatches, {(step[-1][2] - t0) / 1e3:.1f} us wall, kernel time {busy / 1e3:.1f} us, idle {((step[-1][2] - t0) - busy) / 1e3:.1f} us")
run = None
def flush():
    global run
    if run: print(f"      {run[0]:3d} gaot kernels, {run[1] / 1e3:8.1f} us busy, gaps {run[2] / 1e3:6.1f} us")
    run = None
for n, s, e in step:
    gap = s - prev_end
    if '_ZN4gaot' in n and gap < 20e3:
        run = run or [0, 0, 0]
        run[0] += 1; run[1] += e - s; run[2] += max(gap, 0)
    else:
        flush()
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} gap {gap / 1e3:7.1f}  {('gaot::' + n[8:60]) if '_ZN4gaot' in n else n[:100]}")
    prev_end = e
flush()
