import os, sys
sys.path.insert(0, "/root/repo")
import runpy, torch
from gaot_amd import _lib
lib = _lib.load()
sys.argv = ["eager_steps.py", "c3", "3"]
lib.gaot_debug_split_redo_count(1)
runpy.run_path("/root/repo/tools/eager_steps.py", run_name="__main__")
torch.cuda.synchronize()
print("redo tiles over 6 eager C3 steps:", lib.gaot_debug_split_redo_count(1))
