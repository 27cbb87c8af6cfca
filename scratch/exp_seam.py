"""The calc_likelihood seam per call (bench.py's abi_seam), three times."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for rep in range(3):
    s = bench.abi_seam()
    print({k: (round(v["us_per_call_1_state"], 2), round(v["us_per_call_1024_states"], 1), round(v["numpy_binned_us_per_call"], 2))
           for k, v in s.items() if isinstance(v, dict)}, flush=True)
