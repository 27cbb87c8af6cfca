"""Wall clock of the CLI end to end (the tutorial's run: LiteRateForward.py -n 10,000,000 -s 1000 on the example data), with
--chains 128: process start to logs on disk.  Data rebuilt from the golden arrays (no reference files on the GPU box)."""
import os, sys, time, subprocess, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import numpy as np
G = np.load(os.path.join(ROOT, "tests", "golden", "binning_lik.npz"))
n_iter = sys.argv[1] if len(sys.argv) > 1 else "10000000"
chains = sys.argv[2] if len(sys.argv) > 2 else "128"
extra = sys.argv[3:]
tmp = tempfile.mkdtemp()
data = os.path.join(tmp, "example.tsv")
tbp = ["-TBP"]
if extra and extra[0] == "synth":            # cfg4's synthetic lineages (years since the origin: no -TBP)
    sys.path.insert(0, ROOT)
    from literate_amd import synth
    ts, te, _ = synth.make_lineages(int(extra[1]), 128, 20, 0)
    extra, tbp = extra[2:], []
    np.savetxt(data, np.column_stack([np.arange(len(ts)), ts, te - 0.5]), fmt="%d\t%g\t%g", header="id\tts\tte", comments="")
else:
    ts, te = G["example_TBP/ts"], G["example_TBP/te"] - 0.5
    with open(data, "w") as f:
        f.write("id\tts\tte\n")
        for i, (a, b) in enumerate(zip(ts, te)):
            f.write("%d\t%g\t%g\n" % (i, 24.0 - a, 24.0 - b))
cmd = [sys.executable, os.path.join(ROOT, "LiteRateForward.py"), "-d", data] + tbp + ["-n", n_iter, "-s", "1000", "-p", "1000000",
       "-seed", "31", "--chains", chains] + extra
t0 = time.perf_counter()
out = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, cwd=tmp, text=True).stdout
dt = time.perf_counter() - t0
print([l for l in out.splitlines() if "iterations x" in l or "kernel" in l.lower()][-2:])
logs = os.path.join(tmp, "literate_mcmc_logs")
size = sum(os.path.getsize(os.path.join(logs, f)) for f in os.listdir(logs))
print("%d lineages, %s iterations x %s chains %s: %.1f s wall = %.0f iterations/s per chain, %.1f MB of logs in %d files" % (
    len(ts), n_iter, chains, " ".join(extra), dt, int(n_iter) / dt, size / 1e6, len(os.listdir(logs))), flush=True)
shutil.rmtree(tmp)
