#!/usr/bin/env python3
"""Footprint against speed (round-4 verdict item 5): the hg38-scale index of BASELINE.json configs[2] opened under a series of
device-memory budgets (vg_index_open_ex) -- which views each budget buys (vg_index_plan), what the replica then holds, reads/s and
the main-tier kernel's roofline fraction on the default workload's batches, and parity of 1 M reads with the oracle at every budget.
One process: the index files are built once (bench.py's work directory), the oracle is loaded once.
    python3 profiles/budget_sweep.py [budget GB ...]       (default: 0 = whole device, 200 160 128 96 60)   -> one JSON line per budget
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402


def main():
    budgets = [float(x) for x in sys.argv[1:]] or [0, 200, 160, 128, 96, 60]
    sys.argv = [sys.argv[0]]
    args = bench.parse_args()
    from vargeno_amd import synth

    tag = "g%d_s%d_c%d" % (args.genome, args.snps, args.chroms)
    d = os.path.join(os.environ.get("VG_BENCH_DIR") or "/tmp/vg_bench", tag)
    prefix = os.path.join(d, "idx")
    g, s, _ = synth.genome_and_snps(genome_len=args.genome, n_snps=args.snps, n_chroms=args.chroms)
    bench.build_index_files(args, g, s, d, prefix)
    import torch

    from oracle import oracle as O
    from vargeno_amd.api import GenoIndex, gate_words

    dev = torch.device("cuda", 0)
    src = synth.DeviceReadSource(g, s, dev)
    del g, s
    batches = [src.batch(b, args.reads, lowq=args.lowq) for b in range(2)]
    src.release()
    del src
    batches = [tuple(b) + (gate_words(b[1], b[2]),) for b in batches]
    n_check = 1_000_000
    r0 = synth.reads_to_host(*batches[0][:3], 0, n_check)
    ox = O.OracleIndex.load(prefix)
    ox.process(r0.bases, r0.quals, r0.offsets, nthreads=min(os.cpu_count() or 1, 64))
    so = ox.sites()
    ox.close()
    b1 = int(batches[0][2][n_check].item())
    sub = (batches[0][0][:b1], batches[0][1][:b1], batches[0][2][:n_check + 1].contiguous())
    for gb in budgets:
        time.sleep(20)                      # (memory a handle has just freed is cleared at its next allocation: profiles/cold_start_r05.txt)
        t0 = time.time()
        try:
            gx = GenoIndex.open(prefix, device=0, max_device_bytes=int(gb * 1e9) if gb else None)
        except Exception as e:
            print(json.dumps({"budget_GB": gb, "failed": repr(e)}), flush=True)
            continue
        t_open = time.time() - t0
        rep_ = gx.open_report
        out = {"budget_GB": gb or "whole device", "index_open_s": t_open, "device_GB": gx.device_bytes / 1e9, "views": list(gx.views), "plan": gx.plan,
               "memory": rep_[rep_.find("memory:"):] if "memory:" in rep_ else None}
        assert not gb or gx.device_bytes <= gb * 1e9, ("the handle holds more than its budget", gb, gx.device_bytes)
        # parity: 1 M reads, timed build (the views) and counting build
        for stats in (True, False):
            gx.set_stats(stats)
            gx.reset()
            gx.process_device(sub[0], sub[1], sub[2], n_check)
            rc, ac = gx.counts()
            assert np.array_equal(rc, so["ref_cnt"]) and np.array_equal(ac, so["alt_cnt"]), ("parity", gb, stats)
        out["parity"] = {"equal": True, "reads": n_check, "builds": "counting build, timed build"}
        gx.set_stats(True)
        gx.reset()
        gx.process_device(batches[0][0], batches[0][1], batches[0][2], args.reads)
        alg = gx.stats()["alg_bytes"]
        gx.set_stats(False)
        gx.reset()
        for i in range(5):
            gx.process_device(*batches[i % 2][:3], args.reads)
        gx.sync()
        gx.timing()
        gx.reset()
        t0 = time.perf_counter()
        for i in range(20):
            gx.process_device(*batches[i % 2][:3], args.reads)
        gx.counts(copy=False)
        dt = time.perf_counter() - t0
        tm = gx.timing()
        out.update({"reads_per_s": 20 * args.reads / dt, "ms_per_step": 1e3 * dt / 20, "kernel_ms": tm["ms_main"], "pack_ms": tm["ms_pack"],
                    "roofline_frac": alg / (tm["ms_main"] * 1e-3) / 8e12, "kernel": bench.main_kernel_name(gx.views)})
        gx.close()
        torch.cuda.empty_cache()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
