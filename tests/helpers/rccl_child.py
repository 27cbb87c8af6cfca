"""Child process of tests/test_hip_surface.py::test_rccl_on_the_one_gpu - started under
`python -m torch.distributed.run --nproc-per-node 1` (the launcher runs before anything touches the GPU).

A world-size-1 "nccl" group IS RCCL on ROCm: the communicator is created on cuda:0 and the engine's sharded path -
TraceStreamer with gather=True: the status all-reduce and the gather of every window's rows on the side stream while
the next window runs (literate_amd/engine.py, dist.py) - executes on the device as it does with eight ranks."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", world_size=int(os.environ.get("WORLD_SIZE", "1")), rank=int(os.environ.get("RANK", "0")),
                            device_id=torch.device("cuda", 0))
    from literate_amd import synth
    from literate_amd.dist import gather_traces
    from literate_amd.engine import ChainEngine, TraceStreamer
    ts, te, _ = synth.make_lineages(5000, n_bins=48, n_shifts=6, seed=3)
    eng = ChainEngine(ts, te, 8, model=0, seed=11, s_freq=10, n_trace_slots=40)
    eng.init()
    st = TraceStreamer(eng, total_chains=8, gather=True)
    eng.steps(200)
    st.mark()
    eng.steps(200)
    st.mark()
    rows1, snap1, win1 = st.collect()
    rows2, snap2, win2 = st.collect()
    tr = eng.trace_rows()
    assert win1 == (0, 20, 200) and win2 == (20, 40, 400), (win1, win2)
    assert np.array_equal(np.concatenate([rows1, rows2]), tr, equal_nan=True)
    assert np.all(snap2["it"] == 400)
    # the collectives themselves, on device tensors
    t = torch.arange(4, dtype=torch.float64, device="cuda")
    dist.all_reduce(t)
    dist.barrier()
    g = gather_traces(eng.trace[0:3, :, :13].contiguous(), total_chains=8)
    assert g.is_cuda and torch.equal(g, eng.trace[0:3, :, :13])
    torch.cuda.synchronize()
    maps = open("/proc/self/maps").read()
    out = dict(rccl_loaded="librccl" in maps, backend=dist.get_backend(), rows=int(len(tr)),
               all_reduce=[float(x) for x in t.cpu()])
    print("RCCL_CHILD " + json.dumps(out))
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
