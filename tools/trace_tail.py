#!/usr/bin/env python3
"""The last N kernel launches of a rocprofv3 kernel-trace database in order: index, start offset (us), duration, gap, name.
usage: python tools/trace_tail.py trace.db N"""
import sqlite3, sys, importlib.util, os
spec = importlib.util.spec_from_file_location("rs", os.path.join(os.path.dirname(os.path.abspath(__file__)), "rocpd_stats.py"))
db, n = sys.argv[1], int(sys.argv[2])
c = sqlite3.connect(db)
suf = [r[0] for r in c.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace('rocpd_kernel_dispatch', '')
rows = c.execute(f"select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch{suf} d join rocpd_info_kernel_symbol{suf} s on d.kernel_id=s.id order by d.start").fetchall()[-n:]
import re
def demangle(nm):
    nm = re.sub(r'\.kd$', '', nm)
    m = re.match(r'_ZN4gaot(\d+)([A-Za-z_0-9]+)', nm)
    if m:
        name = m.group(2)[:int(m.group(1))]
        t = re.search(r'I(Li\d+E|Lb[01]E)+E', nm)
        targs = ''
        if t:
            targs = '<' + ','.join(x[2:-1] if x.startswith('Li') else ('T' if x[2] == '1' else 'F') for x in re.findall(r'Li\d+E|Lb[01]E', t.group(0))) + '>'
        return 'gaot::' + name + targs
    return nm[:90]
t0, prev = rows[0][1], rows[0][1]
print(f"# last {len(rows)} dispatches: {(rows[-1][2] - t0) / 1e3:.1f} us first start to last end, kernel time {sum(e - s for _, s, e in rows) / 1e3:.1f} us")
for i, (nm, s, e) in enumerate(rows):
    print(f"{i:4d} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev) / 1e3:7.1f}  {demangle(nm)}")
    prev = e
