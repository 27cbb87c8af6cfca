"""Child of test_bin_unit_publication_under_out_of_step_blocks: lr_bin_unit_kernel's hand-off to the block that takes the
last ticket (returning agent-scope atomics + s_waitcnt + barrier, no release / acquire fence; csrc/lr_stats.hip) under
LR_UB_BLOCKS_PER_CU = argv[1] (latched by the library at first use, hence a process of its own): shuffled continuous
times, n chosen so that the LAST block holds 1/64 of what the others hold (it takes its ticket long before they do),
windows few enough that several blocks share a CU, `reps` launches - every one bit-identical to the first, which is
checked against lr_bin_events (the general-window kernel: another accumulation altogether) and torch.bincount."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    bpc, reps = int(sys.argv[1]), int(sys.argv[2])
    os.environ["LR_UB_BLOCKS_PER_CU"] = str(bpc)
    import torch
    from literate_amd import ops
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    blocks = min(n_cu * bpc, 1024)                         # LR_UB_MAX_BLOCKS
    chunk = 64 * 2048                                      # lineages per block: 64 trips of a 1024-thread block's two pairs
    n = (blocks - 1) * chunk + chunk // 64
    out = {"blocks_per_cu": bpc, "blocks": blocks, "lineages": n, "cases": []}
    g = torch.Generator(device="cuda")
    g.manual_seed(17 + bpc)
    for W in (24, 128):                                    # 24 windows: 22 KB of histograms, blocks share a CU; 128: one per CU
        ts = torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * (W - 1.0)
        te = torch.minimum(ts + 0.01 + torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 9.0,
                           torch.full((), W + 0.5, device="cuda", dtype=torch.float64))
        sp, ex, br = [x.clone() for x in ops.bin_unit_events(ts, te, 0.0, W)]
        lo = torch.arange(W, dtype=torch.float64, device="cuda")
        gsp, gex, gbr = ops.bin_events(ts, te, lo, lo + 1.0)
        ok_ref = bool(torch.equal(sp, gsp) and torch.equal(ex, gex) and torch.allclose(br, gbr, rtol=1e-12, atol=0)
                      and torch.equal(sp, torch.bincount(torch.floor(ts).long(), minlength=W)[:W]))
        bad = 0
        for _ in range(reps):
            a, b, c = ops.bin_unit_events(ts, te, 0.0, W)
            bad += int(not (torch.equal(a, sp) and torch.equal(b, ex) and torch.equal(c, br)))
        torch.cuda.synchronize()
        out["cases"].append({"windows": W, "first_run_matches_lr_bin_events": ok_ref, "reps": reps, "differing_runs": bad})
        del ts, te
    print("BIN_UNIT_STRESS " + json.dumps(out))


if __name__ == "__main__":
    main()
