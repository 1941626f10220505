#!/usr/bin/env python3
"""Per-kernel PMC report from rocprofv3 rocpd databases (separate --pmc passes, one counter group each, as the profiling
guide prescribes): HBM traffic per launch (FETCH_SIZE KiB x2 on gfx950 for wide coalesced reads + WRITE_SIZE KiB), MFMA-busy
share, resident waves, LDS bank-conflict share, wait shares.

usage: pmc_report.py out.json db1 [db2 ...]      (every db is scanned for whichever counters it holds)"""
import collections, json, re, sqlite3, sys


def demangle(n):
    n = re.sub(r'\.kd$', '', n)
    m = re.match(r'_ZN4gaot(\d+)([A-Za-z_0-9]+)', n)
    if m:
        name = m.group(2)[:int(m.group(1))]
        t = re.search(r'I(Li\d+E|Lb[01]E)+E', n)
        targs = ''
        if t:
            targs = '<' + ','.join(x[2:-1] if x.startswith('Li') else ('T' if x[2] == '1' else 'F') for x in re.findall(r'Li\d+E|Lb[01]E', t.group(0))) + '>'
        return name + targs
    return n[:80]


def scan(db, agg, dur):
    c = sqlite3.connect(db)
    names = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    tab = lambda p: [n for n in names if n.startswith(p)][0]
    kd, ks = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol")
    for kn, s, e in c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id"):
        dur[demangle(kn)].append(e - s)
    try:
        pe, ip = tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
    except IndexError:
        return
    q = (f"select d.dispatch_id, s.kernel_name, i.name, sum(e.value) from {pe} e join {kd} d on e.event_id=d.event_id "
         f"join {ks} s on d.kernel_id=s.id join {ip} i on e.pmc_id=i.id group by d.dispatch_id, i.name")
    for _, kn, cn, v in c.execute(q):
        agg[demangle(kn)][cn].append(v)


out, dbs = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for db in dbs:
    scan(db, agg, dur)
rep = {}
for k, d in agg.items():
    avg = {cn: sum(v) / len(v) for cn, v in d.items()}
    r = {"launches_seen": max(len(v) for v in d.values())}
    if k in dur:
        r["avg_us_under_pmc"] = sum(dur[k]) / len(dur[k]) / 1e3      # PMC passes serialise kernels: for reference only
    if "FETCH_SIZE" in avg:
        r["fetch_bytes_per_launch_x2"] = 2 * 1024 * avg["FETCH_SIZE"]
    if "WRITE_SIZE" in avg:
        r["write_bytes_per_launch"] = 1024 * avg["WRITE_SIZE"]
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        r["hbm_bytes_per_launch"] = r["fetch_bytes_per_launch_x2"] + r["write_bytes_per_launch"]
    busy = avg.get("SQ_BUSY_CYCLES") or avg.get("GRBM_GUI_ACTIVE")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and avg.get("SQ_BUSY_CU_CYCLES"):
        r["mfma_busy_frac"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / avg["SQ_BUSY_CU_CYCLES"] / 4.0     # 4 SIMDs per CU
    if "SQ_WAVE_CYCLES" in avg and avg.get("SQ_BUSY_CU_CYCLES"):
        r["resident_waves_per_cu"] = avg["SQ_WAVE_CYCLES"] / avg["SQ_BUSY_CU_CYCLES"]
    if "SQ_LDS_BANK_CONFLICT" in avg and avg.get("SQ_LDS_IDX_ACTIVE"):
        r["lds_bank_conflict_frac"] = avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"]
    elif "SQ_LDS_BANK_CONFLICT" in avg:
        r["lds_bank_conflict_cycles"] = avg["SQ_LDS_BANK_CONFLICT"]
    for cn in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS"):
        if cn in avg and avg.get("SQ_WAVE_CYCLES"):
            r[cn.lower() + "_per_wave_cycle"] = avg[cn] / avg["SQ_WAVE_CYCLES"]
    r["raw_avg"] = {cn: round(v, 1) for cn, v in sorted(avg.items())}
    rep[k] = r
json.dump(dict(sorted(rep.items())), open(out, "w"), indent=1)
print(f"{out}: {len(rep)} kernels")
