#!/usr/bin/env python3
"""bench.py's C3 block alone (vx, shuffled dataset): python tools/c3_line.py [--no-oracle] [C4 C5 ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
which = tuple(a for a in sys.argv[1:] if not a.startswith("--")) or ("C3",)
out = bench.secondary_configs(torch.device("cuda:0"), which=which, oracle="--no-oracle" not in sys.argv)
print(json.dumps(out))
