"""Shader clock and power while the four-chain kernel runs cfg4 (rocm-smi sampled from a thread), and s_memtime (100 MHz)
against the shader-clock counter inside a kernel is not needed: rocm-smi reports sclk directly."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from literate_amd import synth
from literate_amd.engine import ChainEngine
ts, te, _ = synth.make_lineages(100000, 128, 20, 0)
eng = ChainEngine(ts, te, 1024, model=0, seed=1, s_freq=1 << 30, n_trace_slots=2, engine="persistent4")
eng.init(); eng.steps(3000); torch.cuda.synchronize()
samples, stop = [], False
def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=10).stdout
            samples.append([l.strip() for l in out.splitlines() if ("sclk" in l or "mclk" in l or "fclk" in l or "Power" in l or "Temperature (Sensor junction)" in l)])
        except Exception as ex:
            samples.append([repr(ex)])
        time.sleep(0.3)
print("idle:", subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout[-600:])
t = threading.Thread(target=poll); t.start()
t0 = time.time()
n_done = 0
while time.time() - t0 < 6.0:
    eng.steps(4096 * 8); n_done += 4096 * 8
    torch.cuda.synchronize()
el = time.time() - t0
stop = True; t.join()
print("%.3f us per iteration over %.1f s" % (el / n_done * 1e6, el))
for s in samples[:: max(1, len(samples) // 8)]:
    print(s)
