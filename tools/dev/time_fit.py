"""Development aid: the fit phase at the reference's minibatch (256) alone, for a kernel trace.
    python tools/dev/time_fit.py [system]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench

r = bench.optimiser_step(1, None, sys.argv[1] if len(sys.argv) > 1 else "cartpole")
print(json.dumps({k: r[k] for k in ("ms_per_update", "fit_phase")}))
