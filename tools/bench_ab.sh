#!/bin/bash
# same-box A/B of bench.py with a debug hook flipped: bench_ab.sh <hook name> <value A> <value B> [rounds]
HOOK=$1; A=$2; B=$3; R=${4:-2}
for r in $(seq $R); do for v in $A $B; do
python - <<PY
import sys, runpy, io, json, contextlib
from gaot_amd import _lib
getattr(_lib.load(), "$HOOK")($v)
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-reference-loop", "--no-configs"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    try: runpy.run_path("bench.py", run_name="__main__")
    except SystemExit: pass
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print("$HOOK=$v", round(d["value"], 1), "samples/s", round(d["ms_per_step"], 4), "ms  frac", round(d["roofline"]["frac"], 3), " sustained", round(d["sustained"]["ms_per_step"], 4), flush=True)
PY
done; done
