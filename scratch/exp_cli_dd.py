"""Wall clock of DDRate.py end to end at cfg5's size (50k synthetic lineages, --chains 256, a sample every 1000 iterations)."""
import os, sys, time, subprocess, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from literate_amd import synth
n_iter = sys.argv[1] if len(sys.argv) > 1 else "1000000"
chains = sys.argv[2] if len(sys.argv) > 2 else "256"
n_lin = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
extra = sys.argv[4:]
ts, te, _ = synth.make_lineages(n_lin, 128, 20, 0)
tmp = tempfile.mkdtemp()
data = os.path.join(tmp, "dd.tsv")
np.savetxt(data, np.column_stack([np.arange(len(ts)), ts, te - 0.5]), fmt="%d\t%g\t%g", header="id\tts\tte", comments="")
cmd = [sys.executable, os.path.join(ROOT, "DDRate.py"), "-d", data, "-n", n_iter, "-s", "1000", "-seed", "31", "--chains", chains] + extra
t0 = time.perf_counter()
out = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, cwd=tmp, text=True).stdout
dt = time.perf_counter() - t0
print([l for l in out.splitlines() if "iterations x" in l][-1:])
logs = [f for f in os.listdir(tmp) if f.endswith(".log")]
size = sum(os.path.getsize(os.path.join(tmp, f)) for f in logs)
print("%d lineages, %s iterations x %s chains %s: %.1f s wall, %.1f MB of logs in %d files" % (len(ts), n_iter, chains, " ".join(extra), dt, size / 1e6, len(logs)), flush=True)
shutil.rmtree(tmp)
