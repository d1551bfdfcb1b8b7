#!/usr/bin/env python3
"""bench.py -- reads/s of the `vargeno geno` read loop on MI355X (BASELINE.json metric).

One step = one pass of the hot path (encode -> dictionary lookups -> gated neighbour search -> vote
-> pile-up counter updates) over one batch of synthetic 150 bp reads that is ALREADY RESIDENT in
HBM as ASCII bases + quality characters, through the C-ABI (vg_reads_process_device); for N > 1
ranks each step ends with the path's one exchange, an RCCL all-reduce of the per-site counters.

Workload (config.workload): BASELINE.json configs[1] by default -- chr22-scale: one 40 Mbp synthetic
chromosome with planted repeats, ~1 M SNPs, 1 M x 150 bp reads per step at 0.5 % error, 8 % low-quality
characters, seed 20261002 -- about a minute end to end.  `--workload hg38` runs the
configs[2] shape (3.1 Gbp in 24 sequences, 10 M SNPs, 8 M-read steps of its 30x reads; ~6 minutes, most of
it building and loading the index; profiles/bench_hg38_scale_r01.json holds the committed run).

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R]
        python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["chr22", "hg38"], default="chr22",
                    help="chr22 = BASELINE.json configs[1] (default, about a minute end to end); hg38 = configs[2] shape: 3.1 Gbp in 24 "
                         "sequences, 10 M SNPs, 8 M-read batches of its 30x reads (index build + load take ~4 minutes, ~190 GB of HBM)")
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU per step")
    ap.add_argument("--genome", type=int, default=None)
    ap.add_argument("--snps", type=int, default=None)
    ap.add_argument("--chroms", type=int, default=None, help="number of sequences the genome is split into")
    ap.add_argument("--lowq", type=float, default=0.08, help="fraction of low-quality (gate-open) characters; 0.5 = the stress profile of SURVEY.md §8d")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000, help="reads timed on the CPU oracle (0 = skip)")
    ap.add_argument("--workdir", default=os.environ.get("VG_BENCH_DIR", "/tmp/vg_bench"))
    ap.add_argument("--no-check", action="store_true", help="skip the parity check against the oracle")
    ap.add_argument("--no-cpu-reference", action="store_true", help="do not also time the reference binary (oracle/_ref/vargeno) on the host")
    args = ap.parse_args()
    preset = {"chr22": dict(genome=40_000_000, snps=1_000_000, chroms=1, reads=1_000_000),
              "hg38": dict(genome=3_100_000_000, snps=10_000_000, chroms=24, reads=8_000_000)}[args.workload]
    for k, v in preset.items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    args.cpu_sample = min(args.cpu_sample, args.reads) if args.cpu_sample else 0

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)               # one rank per GPU; (VG_BENCH_BACKEND=gloo lets ranks share a GPU for plumbing tests)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        backend = os.environ.get("VG_BENCH_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", dev_index)

    from vargeno_amd import synth
    from vargeno_amd.api import GenoIndex, all_reduce_counts

    # ---- data set + index files (rank 0 builds, everyone loads a replica) ------------------------
    tag = "g%d_s%d_c%d" % (args.genome, args.snps, args.chroms)
    d = os.path.join(args.workdir, tag)
    prefix = os.path.join(d, "idx")
    t0 = time.time()
    g, s, r = synth.chr22_scale(genome_len=args.genome, n_snps=args.snps, n_reads=args.reads, n_chroms=args.chroms, lowq=args.lowq)
    if rank == 0:
        log("[bench] synthetic data: %.1fs (%d bp, %d SNPs, %d reads)" % (time.time() - t0, g.total_len, len(s.pos), r.n))
        if not os.path.exists(prefix + ".ref.dict"):
            os.makedirs(d, exist_ok=True)
            t0 = time.time()
            synth.write_fasta(os.path.join(d, "ref.fa"), g)
            synth.write_vcf(os.path.join(d, "snps.vcf"), g, s)
            env = dict(os.environ, VARGENO_NO_LITE="1")
            subprocess.check_call([os.path.join(ROOT, "vargeno_amd", "csrc", "vargeno"), "index", "ref.fa", "snps.vcf", "idx"],
                                  cwd=d, env=env, stdout=subprocess.DEVNULL)
            log("[bench] vargeno index: %.1fs" % (time.time() - t0))
    if world > 1:
        dist.barrier()
    t0 = time.time()
    gx = GenoIndex.open(prefix, device=dev_index)
    if rank == 0:
        log("[bench] index resident in HBM: %.1fs, %.1f GB, %d sites" % (time.time() - t0, gx.device_bytes / 1e9, gx.num_sites))

    # ---- reads resident in HBM --------------------------------------------------------------------
    d_bases = torch.from_numpy(r.bases).to(dev)
    d_quals = torch.from_numpy(r.quals).to(dev)
    d_offs = torch.from_numpy(r.offsets.astype(np.int64)).to(dev)
    torch.cuda.synchronize(dev)

    # ---- one counted pass: event counts -> algorithmic bytes; parity against the oracle -----------
    gx.set_stats(True)
    gx.reset()
    gx.process_device(d_bases, d_quals, d_offs, r.n)
    st = gx.stats()
    alg_bytes_per_launch = st["alg_bytes"]
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:        # the CPU legs (and the parity check they feed) run at N = 1 only
        from oracle import oracle as O

        t0 = time.time()
        ox = O.OracleIndex.load(prefix)
        log("[bench] oracle index load: %.1fs" % (time.time() - t0))
        ns = min(args.cpu_sample, r.n)
        sub = r.slice(0, ns)
        # bounded sample: whole passes over the first `ns` reads until about 10 s of single-thread work; the first pass is
        # the one the parity check uses, the others follow it
        t0 = time.time()
        ox.process(sub.bases, sub.quals, sub.offsets, nthreads=1)
        t_cpu, passes = time.time() - t0, 1
        if not args.no_check:
            if ns == r.n:
                so = ox.sites()
                rc, ac = gx.counts()
                bad = int((rc != so["ref_cnt"]).sum() + (ac != so["alt_cnt"]).sum())
                assert bad == 0, "HIP counters != oracle at %d of %d site counters" % (bad, 2 * len(rc))
                want = ox.stats.as_dict()
                for k, v in want.items():
                    assert st[k] == v, "event counter %s: hip %d oracle %d" % (k, st[k], v)
                # the timed build (event counting off; it answers the high-half neighbour queries from the
                # LO32-ordered view) must give the same counters
                gx.set_stats(False)
                gx.reset()
                gx.process_device(d_bases, d_quals, d_offs, r.n)
                rc2, ac2 = gx.counts()
                assert np.array_equal(rc2, so["ref_cnt"]) and np.array_equal(ac2, so["alt_cnt"]), "timed build != oracle"
                log("[bench] parity: %d site counters (both builds) and %d event counters identical to the oracle" % (2 * len(rc), len(want)))
        while t_cpu < 10.0 and passes < 8:
            ox.reset()
            t0 = time.time()
            ox.process(sub.bases, sub.quals, sub.offsets, nthreads=1)
            t_cpu += time.time() - t0
            passes += 1
        cpu = {"value": passes * ns / t_cpu, "unit": "reads/s", "cores": 1, "kind": "port",
               "sample": "%d pass(es) over the first %d reads of the same batch, oracle/vg_oracle.c, 1 thread, %.1f s" % (passes, ns, t_cpu)}
        ncores = os.cpu_count() or 1
        nt = min(ncores, 64)
        ox.reset()
        t0 = time.time()
        ox.process(sub.bases, sub.quals, sub.offsets, nthreads=nt)
        t_all = time.time() - t0
        cpu["all_cores"] = {"value": ns / t_all, "threads": nt, "host_cores": ncores}
        ox.close()
        # the reference itself, when its binary came along (oracle/_ref/vargeno, built from /root/reference by oracle/Makefile in
        # the build container): `geno` wall time on the sample minus wall time on an empty FASTQ = its read loop, one thread
        ref_bin = os.path.join(ROOT, "oracle", "_ref", "vargeno")
        if os.path.exists(ref_bin) and not args.no_cpu_reference and args.workload == "chr22":
            try:
                synth.write_fastq(os.path.join(d, "cpu_sample.fq"), sub)
                open(os.path.join(d, "cpu_empty.fq"), "w").close()
                wall = {}
                for name in ("cpu_empty", "cpu_sample"):
                    t0 = time.time()
                    subprocess.run([ref_bin, "geno", "idx", name + ".fq", "snps.vcf", name + ".vcf"], cwd=d, check=True, timeout=900,
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                    wall[name] = time.time() - t0
                loop = wall["cpu_sample"] - wall["cpu_empty"]
                if loop > 0:
                    cpu["reference_binary"] = {"value": ns / loop, "unit": "reads/s", "cores": 1, "kind": "reference",
                                               "sample": "oracle/_ref/vargeno geno on the same %d reads: %.1f s wall, minus %.1f s wall on an empty FASTQ (its start-up)" % (ns, wall["cpu_sample"], wall["cpu_empty"])}
            except Exception as e:                                  # the baseline of record is the port above
                log("[bench] reference binary not timed: %r" % (e,))

    # ---- timed region: K batches back to back, then (N > 1) the job's one exchange -----------------------
    gx.set_stats(False)
    gx.reset()
    for _ in range(args.warmup):
        gx.process_device(d_bases, d_quals, d_offs, r.n)
    if world > 1:
        all_reduce_counts(gx)
    gx.sync()
    gx.timing()                                         # drop the warm-up batches from the event averages
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gx.process_device(d_bases, d_quals, d_offs, r.n)
    if world > 1:
        all_reduce_counts(gx)                           # one RCCL all-reduce of the per-site counters over xGMI
    gx.sync()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    tm = gx.timing()                                    # HIP events on the library's own streams, averaged over the K batches
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        k_ms = tm["ms_main"]
        achieved = alg_bytes_per_launch / (k_ms * 1e-3) / 1e9
        # HBM traffic of the dominant kernel cannot be counted inside a timed run: it comes from the separate
        # rocprofv3 --pmc passes of this same command, committed under profiles/ (null for any other workload)
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_r01.json")))
            if tj["workload"] == {"genome": args.genome, "snps": args.snps, "reads": args.reads} and args.lowq == 0.08:
                traffic = tj["traffic_bytes_per_launch"]
        except Exception:
            pass
        out = {
            "metric": "reads/sec genotyped (whole node), hg38+dbSNP 30\u00d7; achieved HBM GB/s vs peak",
            "value": world * r.n * args.steps / elapsed,
            "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "%s: %d bp synthetic genome in %d sequence(s), %d SNPs, %d x 150 bp reads per GPU per step, "
                                   "0.5%% error, %g%% low-quality chars, seed 20261002" % (
                                       "chr22-scale (BASELINE.json configs[1])" if g.total_len < 10 ** 9 else "hg38-scale (BASELINE.json configs[2], one batch of its 30x reads)",
                                       g.total_len, len(g.seqs), len(s.pos), r.n, 100 * args.lowq),
                       "reads_per_step_per_gpu": r.n, "genome_bp": args.genome, "snps_requested": args.snps, "index_bytes_hbm": gx.device_bytes,
                       "parallelism": "reads sharded over %d GPU(s), index replicated, one RCCL all-reduce of the site counters after the K batches" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                         "kernel": "vg_wave_kernel", "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                         "algorithmic_bytes_per_read": alg_bytes_per_launch / r.n},
            "cpu_baseline": cpu,
            "device_ms_per_step": {"pack": tm["ms_pack"], "wave": k_ms, "spill_tiers_overlapped": tm["ms_tail"], "of_which_deep_list_wave_tier": tm["ms_deep_lists"], "batches": tm["batches"]},
            "reads_per_step_spilled_to_lane_tier": st["overflow_reads"], "reads_per_step_deep_scratch": st["overflow_deep"],
            "events_per_read": {k: st[k] / r.n for k in ("passes", "chunks", "gate_open", "ref_query", "snp_query", "ctx", "walks", "incr")},
        }
        print(json.dumps(out), flush=True)
    gx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
