#!/usr/bin/env python3
"""bench.py -- reads/s of the `vargeno geno` read loop on MI355X (BASELINE.json metric).

One step = one pass of the hot path (encode -> dictionary lookups -> gated neighbour search -> vote -> pile-up
counter updates) over one batch of synthetic 150 bp reads that is ALREADY RESIDENT in HBM as ASCII bases + quality
characters, through the C-ABI (vg_reads_process_device).  The steps rotate over NB distinct resident batches of the
workload's read stream (default 4), so a step does not replay the cache lines of the step before.  For N > 1 ranks
every rank works on its own batches (disjoint seeds of the same 30x stream, index replicated) and the job ends with
the path's one exchange, an RCCL all-reduce of the per-site counters.

Workload (config.workload): BASELINE.json configs[2] by default -- hg38-scale: 3.1 Gbp synthetic genome in 24
sequences with planted repeats, ~10 M SNPs, 8 M x 150 bp reads per step at 0.5 % error, 8 % low-quality characters
(SURVEY.md §8d), seed 20261002; the index (~240 GB with its re-laid-out views) is resident in HBM.  `--workload chr22`
runs configs[1] (40 Mbp, 1 M SNPs, 1 M-read steps; about a minute end to end).

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload hg38|chr22]
`--gpus N` with N > 1 starts the N ranks itself (one per GPU, `python -m torch.distributed.run`, before anything in this
process touches a GPU); under an external torchrun (WORLD_SIZE set) it runs as one rank of that job.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "vargeno_amd", "csrc", "vargeno")

PRESETS = {"chr22": dict(genome=40_000_000, snps=1_000_000, chroms=1, reads=1_000_000, cpu_sample=1_000_000),
           "hg38": dict(genome=3_100_000_000, snps=10_000_000, chroms=24, reads=8_000_000, cpu_sample=2_000_000),
           # the index of BASELINE.json configs[4] (hg38 + full dbSNP, ~100 M SNPs) on ONE replica: ~100 GB of index files (point
           # VG_BENCH_DIR at a file system with the room, e.g. /dev/shm), ~195 GB of HBM, no merged view / direct table
           "hg38f": dict(genome=3_100_000_000, snps=100_000_000, chroms=24, reads=8_000_000, cpu_sample=500_000)}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(PRESETS), default="hg38",
                    help="hg38 = BASELINE.json configs[2] (default: 3.1 Gbp in 24 sequences, 10 M SNPs, 8 M-read steps of its 30x reads; ~240 GB of HBM); "
                         "chr22 = configs[1] (40 Mbp, 1 M SNPs, 1 M-read steps); hg38f = the index of configs[4] (100 M SNPs) on one replica")
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU per step")
    ap.add_argument("--batches", type=int, default=4, help="distinct resident batches the steps rotate over")
    ap.add_argument("--genome", type=int, default=None)
    ap.add_argument("--snps", type=int, default=None)
    ap.add_argument("--chroms", type=int, default=None, help="number of sequences the genome is split into")
    ap.add_argument("--repeats", type=float, default=0.0, help="fraction of the genome in planted exact repeats (2-10 copies and > 10 copies) -- the repeat-rich stress genome; 0 = the default genome (2 %% diverged repeats)")
    ap.add_argument("--lowq", type=float, default=0.08, help="fraction of low-quality (gate-open) characters; 0.5 = the stress profile of SURVEY.md §8d")
    ap.add_argument("--cpu-sample", type=int, default=None, help="reads timed on one host thread of the CPU oracle (0 = skip the CPU legs and the parity check)")
    ap.add_argument("--workdir", default=os.environ.get("VG_BENCH_DIR"), help="where the index files go (default: $VG_BENCH_DIR, else /tmp/vg_bench, else -- when /tmp lacks the room -- /dev/shm/vg_bench)")
    ap.add_argument("--no-check", action="store_true", help="skip the parity check against the oracle")
    ap.add_argument("--cpu-reference", choices=["auto", "yes", "no"], default="auto",
                    help="time the reference binary (oracle/_ref/vargeno) on the host, beside the GPU legs, as the cpu_baseline of record: auto = yes unless the "
                         "workload is hg38f (the reference cannot run it: int index into a > 2^31-entry SNP dictionary, qv.cc:447) or the host is short of memory")
    ap.add_argument("--ascii-quals", action="store_true", help="hand the read loop the quality STRINGS (vg_reads_process_device) instead of one gate word per read")
    ap.add_argument("--no-gather-probe", action="store_true", help="do not measure the chip's random-gather ceiling (tools/gather_probe, ~5 s)")
    ap.add_argument("--no-ingest", action="store_true", help="skip the secondary end-to-end number (FASTQ text in pinned host memory -> counters)")
    args = ap.parse_args()
    for k, v in PRESETS[args.workload].items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    args.cpu_sample = min(args.cpu_sample, args.reads)
    return args


def self_launch(args):
    """--gpus N without a launcher: start N ranks as a CHILD job (this process has not touched a GPU and never will)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def build_index_files(args, g, s, d, prefix):
    """rank 0: FASTA + VCF + `vargeno index` (host tool), unless the work directory already holds them."""
    from vargeno_amd import synth

    if os.path.exists(prefix + ".done"):
        return
    os.makedirs(d, exist_ok=True)
    t0 = time.time()
    synth.write_fasta(os.path.join(d, "ref.fa"), g)
    synth.write_vcf(os.path.join(d, "snps.vcf"), g, s)
    log("[bench] FASTA + VCF written: %.1fs" % (time.time() - t0))
    t0 = time.time()
    subprocess.check_call([BIN, "index", "ref.fa", "snps.vcf", "idx"], cwd=d, env=dict(os.environ, VARGENO_NO_LITE="1"), stdout=subprocess.DEVNULL)
    log("[bench] vargeno index: %.1fs" % (time.time() - t0))
    open(prefix + ".done", "w").close()


class ReferenceTimer:
    """The reference itself (oracle/_ref/vargeno, built from the reference sources by oracle/Makefile in the build container) on the
    GPU box's host, one thread -- it has no other mode -- next to the GPU legs: two `geno` child processes, one over the
    sample's FASTQ and one over an empty FASTQ, started as soon as the index files exist and left alone until the GPU legs
    are done.  The reference prints "Processing..." right before its read loop (qv.cc:753) and creates the output VCF right
    after the calling scan that follows the loop (qv.cc:1573-1637), so
        read loop = (sample: VCF created - "Processing...") - (empty: VCF created - "Processing...")
    whatever its minutes of start-up (16 GiB jump table) took on a host that is busy with the rest of the bench."""

    def __init__(self, ref_bin, d, sample_fq, n_reads):
        import threading

        import atexit

        self.n = n_reads
        self.res = {}
        self.threads = []
        self.procs = []
        atexit.register(self.kill)                     # a bench that dies early must not leave two 65 GB processes behind
        for name, fq in (("empty", os.path.join(d, "cpu_empty.fq")), ("sample", sample_fq)):
            if name == "empty":
                open(fq, "w").close()
            out_vcf = os.path.join(d, "cpu_%s.vcf" % name)
            if os.path.exists(out_vcf):
                os.remove(out_vcf)
            t = threading.Thread(target=self._run, args=(name, [ref_bin, "geno", "idx", os.path.basename(fq), "snps.vcf", os.path.basename(out_vcf)], d, out_vcf), daemon=True)
            t.start()
            self.threads.append(t)

    def _run(self, name, cmd, cwd, out_vcf):
        try:
            t_start = time.time()
            p = subprocess.Popen(cmd, cwd=cwd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            self.procs.append(p)
            t_proc = None
            for line in p.stderr:                        # stderr is unbuffered on the reference's side
                if line.startswith(b"Processing"):
                    t_proc = time.time()
                    break
            t_vcf = None
            while p.poll() is None:
                if t_vcf is None and os.path.exists(out_vcf):
                    t_vcf = time.time()
                time.sleep(0.02)
            t_end = time.time()
            if p.returncode == 0 and t_proc is not None:
                self.res[name] = {"startup": t_proc - t_start, "loop_and_scan": (t_vcf or t_end) - t_proc, "wall": t_end - t_start}
        except Exception as e:
            log("[bench] reference binary (%s FASTQ) not timed: %r" % (name, e))

    def kill(self):
        for p in self.procs:
            if p.poll() is None:
                p.kill()

    def result(self, timeout=1500):
        t0 = time.time()
        for t in self.threads:
            t.join(max(1.0, timeout - (time.time() - t0)))
        self.kill()                                     # (only on a timeout is anything still alive)
        if "sample" not in self.res or "empty" not in self.res:
            return None
        a, b = self.res["sample"], self.res["empty"]
        loop = a["loop_and_scan"] - b["loop_and_scan"]
        if loop <= 0:
            return None
        return {"value": self.n / loop, "unit": "reads/s", "cores": 1, "kind": "reference",
                "sample": "oracle/_ref/vargeno geno (the reference's own binary, its only mode: one thread) on the first %d reads of batch 0: %.1f s from \"Processing...\" to the "
                          "output VCF's creation, minus %.1f s of the same span on an empty FASTQ (the calling scan); start-up %.0f s, not counted; "
                          "ran beside the GPU legs of this bench" % (self.n, a["loop_and_scan"], b["loop_and_scan"], a["startup"])}


def gather_ceiling():
    """The chip's random-gather ceiling (SURVEY.md §8d asks for it beside the 8 TB/s line): best rate of tools/gather_probe
    over its lane/ILP shapes on a 16 GiB table.  Runs as a child process before this one opens the index."""
    exe = os.path.join(ROOT, "vargeno_amd", "csrc", "tools", "gather_probe")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, "16"], capture_output=True, text=True, timeout=120).stdout
        rows = [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]
        best = max(rows, key=lambda r: r["Gloads_per_s"])
        return {"gathers_per_s": best["Gloads_per_s"] * 1e9, "table_GiB": best["table_GiB"], "lanes": best["lanes"], "ilp": best["ilp"], "dependent": best["dependent"]}
    except Exception as e:
        log("[bench] gather probe failed: %r" % (e,))
        return None


def measure_ingest(gx, batch, log, chunk_mb=64, reps=3):
    """Batch 0 as FASTQ text in page-locked host memory, streamed to the device in 64 MiB chunks `reps` times; the counters of
    the first pass must equal those of the resident batch.  Returns the secondary bench number."""
    import torch

    from vargeno_amd import synth
    from vargeno_amd.api import pinned_buffer

    tb, tq, to = batch[:3]
    n = len(to) - 1
    L = int(to[1].item())
    gx.set_stats(False)
    gx.reset()
    gx.process_device(tb, tq, to, n)
    want = gx.counts_tensor().clone()
    # fixed-width records: "@r%08d\n" bases "\n+\n" quals "\n"
    rec = 10 + 1 + L + 3 + L + 1
    text, owner = pinned_buffer(n * rec)
    m = text.reshape(n, rec)
    ids = np.arange(n, dtype=np.int64)
    m[:, 0] = ord("@"); m[:, 1] = ord("r")
    for k in range(8):
        m[:, 2 + k] = 48 + (ids // 10 ** (7 - k)) % 10
    m[:, 10] = 10
    m[:, 11:11 + L] = tb.cpu().numpy().reshape(n, L)
    m[:, 11 + L] = 10; m[:, 12 + L] = ord("+"); m[:, 13 + L] = 10
    m[:, 14 + L:14 + 2 * L] = tq.cpu().numpy().reshape(n, L)
    m[:, 14 + 2 * L] = 10
    step = chunk_mb << 20
    chunks = [text[a:a + step] for a in range(0, len(text), step)]
    gx.reset()
    got = gx.fastq_stream(chunks)
    assert got[0] == n and got[1] == len(text) and not got[3], "FASTQ stream framed %r of %d records" % (got, n)
    assert torch.equal(gx.counts_tensor(), want), "counters through the FASTQ stream != counters of the resident batch"
    gx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        gx.fastq_stream(chunks)
    dt = time.perf_counter() - t0
    del m, text, owner
    out = {"value": reps * n / dt, "unit": "reads/s", "text_GB_per_s": reps * n * rec / dt / 1e9, "reads": reps * n, "bytes_per_read": rec,
           "path": "FASTQ text in pinned host memory -> vg_fastq_stream_push (%d MiB chunks: H2D over PCIe, record framing on the device) -> read loop -> counters; "
                   "counters of the first pass identical to the resident batch" % chunk_mb}
    log("[bench] ingest end to end: %.3g reads/s (%.1f GB/s of FASTQ text)" % (out["value"], out["text_GB_per_s"]))
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    from vargeno_amd import synth

    # ---- data set + index files: host only (rank 0 builds, the others wait for its marker file) ----------------------------
    tag = "g%d_s%d_c%d" % (args.genome, args.snps, args.chroms) + ("_r%g" % args.repeats if args.repeats else "")
    if args.workdir is None:
        # index files: ~16 bytes per base of the genome + ~550 bytes per SNP (hg38 + 10 M SNPs: 48 GB; + 100 M SNPs: 104 GB).
        # /tmp unless it lacks the room for an index that is not there yet and /dev/shm (memory-backed) has it
        need = 1.25 * (16.0 * args.genome + 550.0 * args.snps)

        def free(path):
            try:
                st = os.statvfs(path)
                return st.f_bavail * st.f_frsize
            except OSError:
                return 0
        args.workdir = "/tmp/vg_bench"
        if not os.path.exists(os.path.join(args.workdir, tag, "idx.done")) and free("/tmp") < need and free("/dev/shm") >= need:
            args.workdir = "/dev/shm/vg_bench"
    d = os.path.join(args.workdir, tag)
    prefix = os.path.join(d, "idx")
    t0 = time.time()
    g, s, _ = synth.genome_and_snps(genome_len=args.genome, n_snps=args.snps, n_chroms=args.chroms, genotypes="hwe" if args.workload == "hg38f" else "uniform", repeats=args.repeats)
    if rank == 0:
        log("[bench] synthetic genome + SNP list: %.1fs (%d bp, %d SNPs)" % (time.time() - t0, g.total_len, len(s.pos)))
        build_index_files(args, g, s, d, prefix)
    else:
        while not os.path.exists(prefix + ".done"):
            time.sleep(1.0)
    ceiling = None
    if rank == 0 and not args.no_gather_probe:
        ceiling = gather_ceiling()                        # child process, before this one holds 240 GB of the device

    # ---- GPU from here on --------------------------------------------------------------------------------------------------
    import torch
    import torch.distributed as dist

    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)               # one rank per GPU (VG_BENCH_BACKEND=gloo lets ranks share a GPU for plumbing tests)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev                                      # where the small bookkeeping collectives live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        backend = os.environ.get("VG_BENCH_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
            coll_dev = torch.device("cpu")
        one = torch.ones(1, dtype=torch.int64, device=coll_dev)
        dist.all_reduce(one)
        n_seen = int(one.item())                          # ranks the collective library actually connected
    else:
        n_seen = 1

    from vargeno_amd.api import GenoIndex, all_reduce_counts, shard_range

    # reads resident in HBM: NB batches of this rank's part of the read stream + (N > 1) one stream every rank knows
    t0 = time.time()
    src = synth.DeviceReadSource(g, s, dev)
    del g, s
    batches = [src.batch(rank * 1000 + b, args.reads, lowq=args.lowq) for b in range(args.batches)]
    common = src.batch(999_999, min(args.reads, 1_000_000), lowq=args.lowq) if world > 1 else None
    src.release()
    del src
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()
    if rank == 0:
        log("[bench] %d batches of %d reads generated on the device: %.1fs" % (args.batches, args.reads, time.time() - t0))

    # ---- the reference binary on the host, beside everything that follows (N = 1 only) ------------------------------------------
    ref_timer = None
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "vargeno")
    if rank == 0 and world == 1 and args.cpu_sample > 0 and os.path.exists(ref_bin) and (args.cpu_reference == "yes" or (args.cpu_reference == "auto" and args.workload != "hg38f")):
        avail_gb = 0.0
        try:
            avail_gb = [int(ln.split()[1]) for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")][0] / 1e6
        except Exception:
            pass
        # two reference processes (each: 16 GiB jump table + both dictionaries + a pile-up table of up to 17 GB) next to the oracle's copy
        need_gb = 40.0 if args.genome < 10 ** 9 else 260.0
        if avail_gb >= need_gb or args.cpu_reference == "yes":
            t0 = time.time()
            sub = synth.reads_to_host(*batches[0][:3]).slice(0, args.cpu_sample)
            fq = os.path.join(d, "cpu_sample.fq")
            synth.write_fastq(fq, sub)
            del sub
            ref_timer = ReferenceTimer(ref_bin, d, fq, args.cpu_sample)
            log("[bench] reference binary started on the host (sample FASTQ of %d reads written in %.1fs; %.0f GB of host memory available)" % (args.cpu_sample, time.time() - t0, avail_gb))
        else:
            log("[bench] reference binary NOT timed: %.0f GB of host memory available, %.0f wanted" % (avail_gb, need_gb))

    t0 = time.time()
    gx = GenoIndex.open(prefix, device=dev_index)
    if rank == 0:
        log("[bench] index resident in HBM: %.1fs, %.1f GB, %d sites" % (time.time() - t0, gx.device_bytes / 1e9, gx.num_sites))

    # The resident batches: ASCII bases + offsets + one gate word per read (bit c = quality character c < '8': all the path reads of
    # a quality string, qv.cc:836, 943; vg_reads_process_device_gated -- the form the device-side FASTQ framing hands the read loop too).
    # --ascii-quals hands over the quality strings themselves instead (vg_reads_process_device); both forms are timed, see below.
    from vargeno_amd.api import gate_words

    batches = [tuple(b) + (gate_words(b[1], b[2]),) for b in batches]
    torch.cuda.synchronize(dev)

    def run(b, strings=args.ascii_quals):
        if strings:
            gx.process_device(b[0], b[1], b[2], len(b[2]) - 1)
        else:
            gx.process_device_gated(b[0], b[3], b[2], len(b[2]) - 1)

    # ---- one counted pass over batch 0: event counts -> algorithmic bytes; parity against the oracle ------------------------
    gx.set_stats(True)
    gx.reset()
    run(batches[0])
    st = gx.stats()
    alg_bytes_per_launch = st["alg_bytes"]
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:        # the CPU legs (and the parity check they feed) run at N = 1 only
        from oracle import oracle as O

        t0 = time.time()
        ox = O.OracleIndex.load(prefix)
        log("[bench] oracle index load: %.1fs" % (time.time() - t0))
        r0 = synth.reads_to_host(*batches[0][:3])
        ncores = os.cpu_count() or 1
        nt = min(ncores, 64)
        # every host core first: the whole batch, which is also what the parity check compares
        t0 = time.time()
        ox.process(r0.bases, r0.quals, r0.offsets, nthreads=nt)
        t_all = time.time() - t0
        if not args.no_check:
            so = ox.sites()
            rc, ac = gx.counts()
            bad = int((rc != so["ref_cnt"]).sum() + (ac != so["alt_cnt"]).sum())
            assert bad == 0, "HIP counters != oracle at %d of %d site counters" % (bad, 2 * len(rc))
            want = ox.stats.as_dict()
            for k, v in want.items():
                assert st[k] == v, "event counter %s: hip %d oracle %d" % (k, st[k], v)
            # the timed build (event counting off; it reads the re-laid-out views) must give the same counters
            gx.set_stats(False)
            gx.reset()
            run(batches[0])
            rc2, ac2 = gx.counts()
            assert np.array_equal(rc2, so["ref_cnt"]) and np.array_equal(ac2, so["alt_cnt"]), "timed build != oracle"
            log("[bench] parity: %d site counters (both builds) and %d event counters identical to the oracle on the %d reads of batch 0" % (2 * len(rc), len(want), r0.n))
        # one thread on a bounded sample (the reference is single-threaded: this is the baseline of record)
        ns = args.cpu_sample
        sub = r0.slice(0, ns)
        t_cpu, passes = 0.0, 0
        while passes == 0 or (t_cpu < 10.0 and passes < 8):
            ox.reset()
            t0 = time.time()
            ox.process(sub.bases, sub.quals, sub.offsets, nthreads=1)
            t_cpu += time.time() - t0
            passes += 1
        # the many-thread figure: 64 threads above (the parity run), and every hardware thread of the host when it has more;
        # the faster of the two is the one reported, both are listed
        tried = {str(nt): r0.n / t_all}
        best_nt, best_t = nt, t_all
        if ncores > nt:
            ox.reset()
            t0 = time.time()
            ox.process(r0.bases, r0.quals, r0.offsets, nthreads=ncores)
            t_more = time.time() - t0
            tried[str(ncores)] = r0.n / t_more
            if t_more < best_t:
                best_nt, best_t = ncores, t_more
        cpu = {"value": passes * ns / t_cpu, "unit": "reads/s", "cores": 1, "kind": "port",
               "sample": "%d pass(es) over the first %d reads of batch 0, oracle/vg_oracle.c, 1 thread, %.1f s" % (passes, ns, t_cpu),
               "all_cores": {"value": r0.n / best_t, "threads": best_nt, "host_cores": ncores, "reads_per_s_by_threads": tried,
                             "sample": "batch 0 (%d reads), %.1f s" % (r0.n, best_t)}}
        ox.close()
        del r0, sub

    # ---- secondary number (N = 1): end to end from FASTQ text in pinned HOST memory -- H2D over PCIe, framing on the device, the
    #      read loop -- through vg_fastq_stream_push.  Never `value`: the metric is quoted on batches resident in HBM. -------------
    ingest = None
    if rank == 0 and world == 1 and not args.no_ingest:
        ingest = measure_ingest(gx, batches[0], log)

    # ---- N > 1: the sharded path must reproduce one rank.  Every rank takes its shard of one common stream; the all-reduced
    #      counters must equal what rank 0 gets from the whole stream alone. ---------------------------------------------------
    verification = None
    if world > 1:
        gx.set_stats(False)
        cb, cq, co = common
        n_c = len(co) - 1
        whole = None
        if rank == 0:
            gx.reset()
            gx.process_device(cb, cq, co, n_c)
            whole = gx.counts_tensor().clone()
        gx.reset()
        lo, hi = shard_range(n_c, rank, world)
        if hi > lo:
            b0, b1 = int(co[lo].item()), int(co[hi].item())
            so = (co[lo:hi + 1] - co[lo]).contiguous()
            gx.process_device(cb[b0:b1], cq[b0:b1], so, hi - lo)
        all_reduce_counts(gx)
        if rank == 0:
            got = gx.counts_tensor().clone()
            assert torch.equal(got, whole), "all-reduced counters of %d read shards != one rank on the whole stream" % world
            verification = {"sharded_equals_single_rank": True, "reads": n_c, "site_counters": int(got.numel()), "increments": int(got.sum().item())}
            log("[bench] %d ranks, %d reads sharded: all-reduced counters identical to one rank on the whole stream" % (world, n_c))

    # ---- timed region: K steps back to back over the rotating batches, then (N > 1) the job's one exchange --------------------
    gx.set_stats(False)
    gx.reset()
    for i in range(args.warmup):
        run(batches[i % args.batches])
    if world > 1:
        all_reduce_counts(gx)
    gx.sync()
    gx.timing()                                         # drop the warm-up batches from the event averages
    gx.reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        run(batches[i % args.batches])
    local_sum = None
    if world > 1:
        gx.sync()
        local_sum = gx.counts_tensor().sum(dtype=torch.int64).reshape(1).to(coll_dev)      # this rank's increments, before the exchange (checked below)
        all_reduce_counts(gx)                           # one RCCL all-reduce of the per-site counters over xGMI
    fetched = gx.counts()                               # "first submit -> counters reduced and fetched": fold, clamp at 63, device -> host
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    del fetched
    tm = gx.timing()                                    # HIP events on the library's own streams, averaged over the K batches
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # checksum of checksums: the reduced counters hold exactly the increments of all ranks
        dist.all_reduce(local_sum)
        total_after = int(gx.counts_tensor().sum(dtype=torch.int64).item())
        assert total_after == int(local_sum.item()), "reduced counters hold %d increments, the ranks made %d" % (total_after, int(local_sum.item()))
        if rank == 0:
            verification["timed_region_increments"] = total_after

    # ---- the other input form, for the record (N = 1): the same K steps with the quality strings handed over (or, under --ascii-quals, the gate words)
    other_form = None
    if rank == 0 and world == 1:
        gx.reset()
        for i in range(min(args.warmup, 2)):
            run(batches[i % args.batches], strings=not args.ascii_quals)
        gx.sync()
        gx.timing()
        t0 = time.perf_counter()
        for i in range(args.steps):
            run(batches[i % args.batches], strings=not args.ascii_quals)
        gx.counts()
        dt = time.perf_counter() - t0
        tm2 = gx.timing()
        other_form = {"input": "quality strings (vg_reads_process_device)" if not args.ascii_quals else "gate words (vg_reads_process_device_gated)",
                      "value": args.reads * args.steps / dt, "unit": "reads/s", "ms_per_step": 1e3 * dt / args.steps, "pack_ms": tm2["ms_pack"], "wave_ms": tm2["ms_main"]}
    if rank == 0 and ref_timer is not None:
        # the reference's own binary is the baseline of record; the port (oracle) stays beside it
        t0 = time.time()
        refres = ref_timer.result()
        log("[bench] waited %.0fs more for the reference binary: %s" % (time.time() - t0, refres and "%.4g reads/s" % refres["value"]))
        if refres is not None:
            port = cpu or {}
            cpu = dict(refres)
            if port:
                cpu["port"] = {k: port[k] for k in ("value", "unit", "cores", "kind", "sample")}
                cpu["all_cores"] = port.get("all_cores")
    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        k_ms = tm["ms_main"]
        achieved = alg_bytes_per_launch / (k_ms * 1e-3) / 1e9
        # HBM traffic and L2 misses of the dominant kernel cannot be counted inside a timed run: they come from the separate
        # rocprofv3 --pmc passes of this same command, committed under profiles/ (null for any other workload)
        # -- and only of THIS build of the library: a traffic file is stamped with the vg_build_id() it was measured on
        from vargeno_amd._lib import lib as _vg_lib

        build_id = _vg_lib().vg_build_id().decode()
        traffic, misses, traffic_note = None, None, "no profiles/traffic_*.json for this workload"
        import glob

        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json")), reverse=True):
            try:
                tj = json.load(open(path))
                if tj["workload"] != {"genome": args.genome, "snps": args.snps, "reads": args.reads} or args.lowq != 0.08 or args.repeats != 0.0:
                    continue
                if tj.get("build_id") != build_id:
                    traffic_note = "%s was measured on build %s, this is build %s: not quoted" % (os.path.basename(path), tj.get("build_id"), build_id)
                    continue
                traffic, misses = tj["traffic_bytes_per_launch"], tj.get("TCC_MISS_sum")
                traffic_note = "%s (separate rocprofv3 --pmc passes of this command on this build; %s)" % (os.path.basename(path), tj.get("traffic_formula", "FETCH_SIZE + WRITE_SIZE"))
                break
            except Exception:
                pass
        gc = None
        if ceiling:
            # An L2 miss of a random gather moves one 128-byte line (profiles/line_probe_r03_counters.txt), so the chip's measured
            # random-gather rate IS the HBM roofline for this access shape: lines/s x 128 B
            gc = {"peak": ceiling["gathers_per_s"], "unit": "random 128-byte lines/s (tools/gather_probe: 8-byte gathers from a 16 GiB table, one line each; measured in this run)",
                  "peak_GB_per_s": ceiling["gathers_per_s"] * 128 / 1e9,
                  "l2_misses_per_launch": misses, "achieved": (misses / (k_ms * 1e-3)) if misses else None}
            gc["frac"] = (gc["achieved"] / gc["peak"]) if misses else None
        out = {
            "metric": "reads/sec genotyped (whole node), hg38+dbSNP 30×; achieved HBM GB/s vs peak",
            "value": world * args.reads * args.steps / elapsed,
            "unit": "reads/s",
            "n_gpus": n_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "%s: %d bp synthetic genome in %d sequence(s), %d SNPs requested, %d x 150 bp reads per GPU per step rotating over %d "
                                   "distinct resident batches of the read stream, 0.5%% error, %g%% low-quality chars, seed 20261002%s" % (
                                       "hg38 + full-dbSNP-scale index (BASELINE.json configs[4], one replica)" if args.snps >= 5 * 10 ** 7 else
                                       "hg38-scale (BASELINE.json configs[2])" if args.genome >= 10 ** 9 else "chr22-scale (BASELINE.json configs[1])",
                                       args.genome, args.chroms, args.snps, args.reads, args.batches, 100 * args.lowq,
                                       "" if not args.repeats else "; REPEAT-RICH genome: %g%% of it in planted families of near-identical copies (2-10 and 11-200 copies), 50 microsatellites per Mbp" % (100 * args.repeats)),
                       "reads_per_step_per_gpu": args.reads, "resident_batches": args.batches, "genome_bp": args.genome, "snps_requested": args.snps,
                       "index_bytes_hbm": gx.device_bytes, "index_views": gx.views, "lib_build_id": build_id,
                       "parallelism": "reads sharded over %d GPU(s), index replicated, one RCCL all-reduce of the site counters after the K steps" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_note,
                         "traffic_GB_per_s": (traffic / (k_ms * 1e-3) / 1e9) if traffic else None, "traffic_frac_of_peak": (traffic / (k_ms * 1e-3) / 1e9 / 8000.0) if traffic else None,
                         "kernel": "vg_wave_kernel", "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                         "algorithmic_bytes_per_read": alg_bytes_per_launch / args.reads, "gather_ceiling": gc},
            "cpu_baseline": cpu,
            "device_ms_per_step": {"pack": tm["ms_pack"], "wave": k_ms, "spill_tiers_overlapped": tm["ms_tail"], "of_which_deep_list_wave_tier": tm["ms_deep_lists"], "batches": tm["batches"],
                                   "note": "spill tiers: elapsed time from the end of a batch's main-tier kernel to the end of its last tier, on the tail stream, under the NEXT batches' "
                                           "kernels -- mostly waiting (a tier's workgroups are placed when main-tier workgroups of the following batch retire), not work: see reads_per_step_redone_by_deep_list_tier"},
            "reads_per_step_redone_by_deep_list_tier": st["overflow_reads"], "reads_per_step_sent_on_to_lane_tier": st["overflow_deep"],
            "events_per_read": {k: st[k] / args.reads for k in ("passes", "chunks", "gate_open", "ref_query", "snp_query", "ctx", "walks", "incr")},
            "input_form": "ASCII bases + offsets + " + ("quality strings" if args.ascii_quals else "one gate word per read (bit c = quality character c < '8')") + ", resident in HBM",
            "other_input_form": other_form,
            "ingest_end_to_end": ingest,
            "multi_gpu_verification": verification,
        }
        print(json.dumps(out), flush=True)
    gx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
