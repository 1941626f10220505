#!/usr/bin/env python3
"""The kernels of ONE training step in launch order, from a rocprofv3 kernel-trace database: everything between the last two
adamw launches.  usage: python tools/step_sequence.py trace.db  ->  index, start offset (us), duration (us), gap to predecessor, name"""
import re, sqlite3, sys

db = sys.argv[1]
c = sqlite3.connect(db)
suf = [r[0] for r in c.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace('rocpd_kernel_dispatch', '')
rows = c.execute(f"select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch{suf} d join rocpd_info_kernel_symbol{suf} s on d.kernel_id=s.id order by d.start").fetchall()


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
    return n[:90]


marks = [i for i, r in enumerate(rows) if 'adamw_kernel' in r[0]]
a, b = marks[-2], marks[-1]
step = rows[a + 1:b + 1]
t0 = step[0][1]
prev_end = rows[a][2]
print(f"# {len(step)} dispatches, {(step[-1][2] - t0) / 1e3:.1f} us from first start to last end, kernel time {sum(e - s for _, s, e in step) / 1e3:.1f} us")
for i, (n, s, e) in enumerate(step):
    print(f"{i:4d} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}  {demangle(n)}")
    prev_end = e
