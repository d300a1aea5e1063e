import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "nearhover"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
print(json.dumps(bench.param_gradient_kernels(name, B)))
