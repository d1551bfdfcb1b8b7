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

Input form.  `value` is timed on the form the reference consumes: ASCII bases + the reads' quality STRINGS
(vg_reads_process_device).  The reduced form -- one gate word per read, what the device-side FASTQ framing hands the read loop
(vg_reads_process_device_gated) -- is timed in the same run and reported as `other_input_form` (`--gate-words` swaps the two).

Secondary configurations (N = 1, default workload only; `--secondary none` switches them off).  After the main line's legs the
same run measures, each with its own roofline and a parity check against the oracle: the 50 % low-quality stress profile on the
index that is already open; then, as child processes of this one (each prints its own JSON line, which becomes an entry of
`secondary`): chr22-scale (configs[1]), the repeat-rich hg38-size genome (--repeats 0.3), and the index of configs[4] (hg38 +
100 M SNPs) on one replica.  A leg is skipped, with the reason stated, when the run's time budget ($VG_BENCH_BUDGET_S, default
1500 s) would not hold it; a failing leg never fails the main line.

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload hg38|chr22|hg38f]
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
    ap.add_argument("--read-len", type=int, default=150, help="read length in bases (150 = BASELINE.json's reads, 4 chunks of 32; the reference's own experiment used 101 bp reads: 3 chunks, src/vartype.h:12, experiment/experiment.md:20-27)")
    ap.add_argument("--softmask", type=float, default=0.0, help="fraction of the FASTA written in lower case (soft-masked runs, as in a UCSC download): the index then has a reference bit vector that is not the dictionary's LO32 set (generate_bf.cc:230 does not fold case) and the kernel reads the vector itself")
    ap.add_argument("--lowq", type=float, default=0.08, help="fraction of low-quality (gate-open) characters; 0.5 = the stress profile of SURVEY.md §8d")
    ap.add_argument("--cpu-sample", type=int, default=None, help="reads timed on one host thread of the CPU oracle (0 = skip the CPU legs and the parity check)")
    ap.add_argument("--workdir", default=os.environ.get("VG_BENCH_DIR"), help="where the index files go (default: $VG_BENCH_DIR, else /tmp/vg_bench, else -- when /tmp lacks the room -- /dev/shm/vg_bench)")
    ap.add_argument("--no-check", action="store_true", help="skip the parity check against the oracle")
    ap.add_argument("--cpu-reference", choices=["auto", "yes", "no"], default="auto",
                    help="time the reference binary (oracle/_ref/vargeno) on the host, beside the GPU legs, as the cpu_baseline of record: auto = yes unless the "
                         "workload is hg38f (the reference cannot run it: int index into a > 2^31-entry SNP dictionary, qv.cc:447) or the host is short of memory")
    ap.add_argument("--gate-words", action="store_true", help="time `value` on the reduced input form (one gate word per read, vg_reads_process_device_gated) and report the quality-string form beside it, instead of the other way round")
    ap.add_argument("--secondary", default="auto", help="secondary configurations measured after the main line: auto (= all, for the default workload at N = 1), none, or a comma list of lowq50,chr22,repeats30,hg38f")
    ap.add_argument("--sustain-seconds", type=float, default=2.0, help="length of the `sustained` leg: blocks of K steps back to back for at least this long (0 = skip)")
    ap.add_argument("--job-reads", type=int, default=None, help="the `job` leg: one whole `vargeno geno` run (index open + FASTQ ingest + caller + VCF) on a FASTQ file of this many distinct reads written by this bench "
                    "(default: 200 000 000 for the default workload at N = 1 -- scaled down to what the work directory's file system holds --, else 0 = skip)")
    ap.add_argument("--stream-reads", type=int, default=None, help="the `job_stream` leg: one whole `vargeno geno` run on this many reads streamed through a FIFO (default: 620 000 000 = 30x, BASELINE.json's metric, "
                    "for the default workload at N = 1 when the `job` leg runs; 0 = skip)")
    ap.add_argument("--cleanup", action="store_true", help="remove this run's index files when done (the secondary legs' child runs do)")
    ap.add_argument("--no-gather-probe", action="store_true", help="do not measure the chip's random-gather ceiling (tools/gather_probe, ~5 s)")
    ap.add_argument("--no-pretouch", action="store_true", help="do not have a child process take the device's free memory once before the genome is generated (see pretouch_start)")
    ap.add_argument("--no-ingest", action="store_true", help="skip the secondary end-to-end number (FASTQ text in pinned host memory -> counters)")
    ap.add_argument("--device-budget", type=int, default=None, help="device-memory budget of the index in bytes (vg_index_open_ex): the plan takes the widest tables and the views it holds")
    ap.add_argument("--detail-out", default=None, help="where the run's full record goes (every leg's phases, plans, notes: what used to be on the line before it outgrew the driver's parser); "
                    "default: bench_detail.json beside the index files, and gpurun_out/bench_detail_<workload>.json when that directory exists.  The LAST stdout line stays a compact object (< 8000 bytes)")
    args = ap.parse_args()
    for k, v in PRESETS[args.workload].items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    args.cpu_sample = min(args.cpu_sample, args.reads)
    args.ascii_quals = not args.gate_words
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
    synth.write_fasta(os.path.join(d, "ref.fa"), g, softmask=args.softmask)
    synth.write_vcf(os.path.join(d, "snps.vcf"), g, s)
    log("[bench] FASTA + VCF written: %.1fs" % (time.time() - t0))
    t0 = time.time()
    subprocess.check_call([BIN, "index", "ref.fa", "snps.vcf", "idx"], cwd=d, env=dict(os.environ, VARGENO_NO_LITE="1"), stdout=subprocess.DEVNULL)
    log("[bench] vargeno index: %.1fs" % (time.time() - t0))
    open(prefix + ".done", "w").close()


def warm_index_files(prefix):
    """The index files read once (8 threads, 64 MiB pieces, nothing kept) before vg_index_open is timed: files that `vargeno index`
    has just written are normally in the page cache (27 GB/s into the loader's ring) -- on one box of eight this round they were not,
    came from the disk at 5.7 GB/s, and the same vg_index_open took 9.9 s instead of 3.1.  The line says what this read found
    (`config.index_files_read_before_open`: GB, seconds; > 15 GB/s means they were cached already)."""
    import threading

    paths = [prefix + ext for ext in (".ref.dict", ".snp.dict", ".ref.bf", ".snp.bf") if os.path.exists(prefix + ext)]
    piece = 64 << 20
    work = []
    for pth in paths:
        sz = os.path.getsize(pth)
        work += [(pth, off, min(piece, sz - off)) for off in range(0, sz, piece)]
    nxt, lock, total = [0], threading.Lock(), sum(w[2] for w in work)
    t0 = time.time()

    def reader():
        buf = bytearray(piece)
        fds = {}
        while True:
            with lock:
                i = nxt[0]
                nxt[0] += 1
            if i >= len(work):
                break
            pth, off, n = work[i]
            if pth not in fds:
                fds[pth] = os.open(pth, os.O_RDONLY)
            got = 0
            mv = memoryview(buf)
            while got < n:
                k = os.preadv(fds[pth], [mv[got:n]], off + got)
                if k <= 0:
                    break
                got += k
        for fd in fds.values():
            os.close(fd)

    ths = [threading.Thread(target=reader) for _ in range(8)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = max(time.time() - t0, 1e-6)
    return {"GB": total / 1e9, "seconds": dt, "GB_per_s": total / 1e9 / dt}


class ReferenceTimer:
    """The reference itself (oracle/_ref/vargeno, built from the reference sources by oracle/Makefile in the build container) on the
    GPU box's host, one thread -- it has no other mode: two `geno` child processes, one over the sample's FASTQ and one over an
    empty FASTQ, started as soon as the index files exist.  The reference prints "Processing..." right before its read loop
    (qv.cc:753) and creates the output VCF right after the calling scan that follows the loop (qv.cc:1573-1637), so
        read loop = (sample: VCF created - "Processing...") - (empty: VCF created - "Processing...")
    whatever its minutes of start-up (16 GiB jump table) took.  The start-up may overlap anything; the LOOP is timed in a quiet
    window: every leg of this bench that loads the host's cores (the many-thread oracle runs, `vargeno index` of a secondary
    configuration) first calls wait_quiet(), which blocks while a reference process is between "Processing..." and its exit,
    and the seconds of such work that overlapped a loop all the same are reported (`contended_s`, 0 by construction)."""

    def __init__(self, ref_bin, d, sample_fq, n_reads):
        import threading

        import atexit

        self.n = n_reads
        self.ref_bin = ref_bin
        self.res = {}
        self.threads = []
        self.procs = []
        self.loops = {}                                # name -> [t "Processing...", t exit or None]
        self.heavy = []                                # [t0, t1] of host-heavy legs of the bench
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
                    self.loops[name] = [t_proc, None]
                    break
            t_vcf = None
            while p.poll() is None:
                if t_vcf is None and os.path.exists(out_vcf):
                    t_vcf = time.time()
                time.sleep(0.02)
            t_end = time.time()
            if name in self.loops:
                self.loops[name][1] = t_end
            if p.returncode == 0 and t_proc is not None:
                self.res[name] = {"startup": t_proc - t_start, "loop_and_scan": (t_vcf or t_end) - t_proc, "wall": t_end - t_start}
        except Exception as e:
            log("[bench] reference binary (%s FASTQ) not timed: %r" % (name, e))

    def in_loop(self):
        return any(v[1] is None for v in list(self.loops.values()))

    def wait_quiet(self, what, timeout=600.0):
        """Call before a leg that loads the host's cores: blocks while a reference process is inside its timed span."""
        t0 = time.time()
        while self.in_loop() and time.time() - t0 < timeout:
            time.sleep(0.05)
        if time.time() - t0 > 0.5:
            log("[bench] %s held back %.1fs: the reference binary was inside its timed read loop" % (what, time.time() - t0))

    def heavy_begin(self):
        self.heavy.append([time.time(), None])

    def heavy_end(self):
        self.heavy[-1][1] = time.time()

    def kill(self):
        for p in self.procs:
            if p.poll() is None:
                p.kill()

    def alive(self):
        return any(t.is_alive() for t in self.threads)

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
        contended = 0.0
        for l0, l1 in self.loops.values():
            for h0, h1 in self.heavy:
                contended += max(0.0, min(l1 or time.time(), h1 or time.time()) - max(l0, h0))
        import hashlib

        sha = hashlib.sha256(open(self.ref_bin, "rb").read()).hexdigest()
        try:
            recipe_sha = open(os.path.join(ROOT, "tests", "golden", "ref_binary.sha256")).read().split()[0]
        except Exception:
            recipe_sha = None
        return {"value": self.n / loop, "unit": "reads/s", "cores": 1, "kind": "reference", "contended_s": round(contended, 2),
                "binary_sha256": sha, "binary_is_what_oracle_Makefile_builds": (sha == recipe_sha) if recipe_sha else None,
                "sample": "oracle/_ref/vargeno (sha256 %s), 1 thread, first %d reads of batch 0: read loop %.1f s (%.1f s to the VCF's creation - %.1f s on an empty FASTQ); start-up %.0f s not counted; contended %.1f s"
                          % (sha[:12], self.n, loop, a["loop_and_scan"], b["loop_and_scan"], a["startup"], contended)}


class NoRef:
    """Stand-in when the reference is not being timed."""
    def wait_quiet(self, what, timeout=0):
        pass

    def heavy_begin(self):
        pass

    def heavy_end(self):
        pass


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


PRETOUCH_CHILD = r"""
import ctypes, sys, time
hip = None
for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
    try:
        hip = ctypes.CDLL(name)
        break
    except OSError:
        pass
dev, world = int(sys.argv[1]), int(sys.argv[2])
n = ctypes.c_int(0)
if hip is None or hip.hipGetDeviceCount(ctypes.byref(n)) != 0 or n.value < 1 or world > n.value:
    print("0 0.0 -1")                    # no device, or ranks that share one (a plumbing rehearsal): nothing to do
    sys.exit(0)
hip.hipSetDevice(dev % n.value)
fr, tot = ctypes.c_size_t(), ctypes.c_size_t()
hip.hipMemGetInfo(ctypes.byref(fr), ctypes.byref(tot))
size = max(0, fr.value - (4 << 30))
p = ctypes.c_void_p()
t0 = time.time()
rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(size))
dt = time.time() - t0
if rc == 0:
    hip.hipFree(p)
print("%d %.3f %d" % (size, dt, rc))
"""


def pretouch_start(dev_index, world):
    """The device's memory in a known state before vg_index_open is timed.  Memory that ANOTHER process has freed is cleared by the
    driver when it is next handed out (~40 GB/s: profiles/cold_start_r05.txt), and a box of the pool comes with whatever the tenant
    before left: the same vg_index_open took 3.3 s on one fresh box and 9.8 s on another (6.5 s of it inside the one hipMalloc of its
    block).  So a child process takes all free memory once and gives it back -- that pays for the clearing, if any is due -- while
    this process generates the genome and builds the index files on the host; memory this lease itself has freed is scrubbed in the
    background within ~15 s.  The job leg and the child legs wait 20 s on an idle device for the same reason."""
    try:
        return subprocess.Popen([sys.executable, "-c", PRETOUCH_CHILD, str(dev_index), str(world)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True), time.time()
    except Exception:
        return None, time.time()


def pretouch_finish(handle, log):
    p, t_start = handle
    if p is None:
        return None
    try:
        out, _ = p.communicate(timeout=120)
        size, dt, rc = out.split()[:3]
        res = {"bytes": int(size), "hipMalloc_s": float(dt), "rc": int(rc),
               "note": "a child process took the device's free memory once and gave it back before the genome was generated: what the driver had to clear of a previous tenant's memory was cleared there (hipMalloc_s), not inside the timed vg_index_open"}
        # (what the child freed is scrubbed in the background; if it had to wait for a clearing, give the scrub its 20 s)
        # (what the child freed is scrubbed in the background within ~15 s: an index whose files were already there is opened no
        # earlier than 20 s after the child gave the memory back)
        time.sleep(max(0.0, 20.0 - (time.time() - t_start - res["hipMalloc_s"] - 0.5)))
        return res
    except Exception as e:
        log("[bench] device pre-touch failed: %r" % (e,))
        return None


def measure_ingest(gx, batch, log, reps=3):
    """Batch 0 as FASTQ text in page-locked host memory, streamed through the library `reps` times, along both of its ingest paths:
    (a) the text itself crosses the link in 64 MiB chunks and is framed on the device (vg_fastq_stream_begin);
    (b) host threads inside the library frame and 2-bit pack 256 MiB chunks, the packed form (48 bytes per read) crosses the link
        (vg_fastq_stream_begin_packed with the CPUs this process may use -- a cgroup quota counts -- less three).
    The counters of each path's first pass must equal those of the resident batch.  Returns the secondary bench number: the
    faster path's, with both listed."""
    import torch

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
    paths = {}
    quota = None
    try:                                                          # a container's CPU quota bounds the host-packing path (cgroup v2 cpu.max: "<quota> <period>")
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q[0] == "max" else float(q[0]) / float(q[1])
    except Exception:
        pass
    usable = int(min(os.cpu_count() or 1, quota or 1e9))
    pack_threads = max(2, min(usable - 3, 96))                    # (the library itself packs on the host only from 32 usable CPUs on; the leg is measured regardless)
    for name, chunk_mb, threads, what in (("device_framing", 64, None, "vg_fastq_stream_begin: the text crosses PCIe in %d MiB chunks, record framing + packing on the device"),
                                          ("host_packing", 256, pack_threads, "vg_fastq_stream_begin_packed: " + str(pack_threads) + " host threads inside the library frame and 2-bit pack %d MiB chunks, 48 bytes per read cross PCIe")):
        step = chunk_mb << 20
        chunks = [text[a:a + step] for a in range(0, len(text), step)]
        try:
            gx.reset()
            got = gx.fastq_stream(chunks, host_threads=threads)
            assert got[0] == n and got[1] == len(text) and not got[3], "FASTQ stream framed %r of %d records" % (got, n)
            assert torch.equal(gx.counts_tensor(), want), "counters through the FASTQ stream (%s) != counters of the resident batch" % name
            gx.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                gx.fastq_stream(chunks, host_threads=threads)
            dt = time.perf_counter() - t0
            paths[name] = {"value": reps * n / dt, "unit": "reads/s", "text_GB_per_s": reps * n * rec / dt / 1e9, "reads": reps * n, "bytes_per_read_of_text": rec,
                           "path": "FASTQ text in pinned host memory -> " + what % chunk_mb + " -> read loop -> counters; counters of the first pass identical to the resident batch"}
            log("[bench] ingest end to end, %s: %.3g reads/s (%.1f GB/s of FASTQ text)" % (name, paths[name]["value"], paths[name]["text_GB_per_s"]))
        except Exception as e:
            paths[name] = {"failed": repr(e)}
            log("[bench] ingest leg %s failed: %r" % (name, e))
    del m, text, owner
    ok = {k: v for k, v in paths.items() if "value" in v}
    if not ok:
        return {"failed": paths}
    best = max(ok, key=lambda k: ok[k]["value"])
    out = dict(ok[best])
    out["chosen"] = best
    out["host_threads_available"] = os.cpu_count()
    out["host_cpu_quota"] = quota
    out["host_packing_threads"] = pack_threads
    out["paths"] = paths
    return out


def big_file_room():
    """Where a file of tens of GB may go on this host, and how large it may be: (directory, bytes).  /dev/shm when the container's
    memory limit has the room (its pages are charged to the cgroup: memory.max - memory.current, less 64 GB for everybody else),
    never the root file system beyond half of what it has free -- a GPU box of this pool has a 79 GB root and a 300 GiB memory
    limit, and a container that fills either one is killed (round 5 lost a box that way)."""
    def rd(path):
        try:
            v = open(path).read().split()[0]
            return None if v == "max" else int(v)
        except Exception:
            return None
    best = (None, 0)
    try:
        st_ = os.statvfs("/dev/shm")
        shm_free = st_.f_bavail * st_.f_frsize
        mx, cur = rd("/sys/fs/cgroup/memory.max"), rd("/sys/fs/cgroup/memory.current")
        if mx is None:
            mx = [int(ln.split()[1]) * 1024 for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")][0]
            cur = 0
        # (page cache -- the index files this process has written and read -- is charged to the container too, but is given up under
        # pressure: memory.stat's `file` less `shmem` does not count against the room)
        cache = 0
        try:
            stat = dict(ln.split()[:2] for ln in open("/sys/fs/cgroup/memory.stat"))
            cache = max(0, int(stat.get("file", 0)) - int(stat.get("shmem", 0)))
        except Exception:
            pass
        room = min(shm_free, mx - max(0, (cur or 0) - cache) - (64 << 30)) // 2
        if room > best[1]:
            best = ("/dev/shm/vg_bench_job", room)
    except Exception:
        pass
    return best


def job_fastq(src, gx, path, n_reads, batch_reads, lowq, log, read_len=150, keep_first=0):
    """The `job` leg's input: n_reads DISTINCT reads of the workload's stream (batches no other leg uses) as one FASTQ file, written
    batch by batch (text put together on the device, page-locked copy, positioned writes by four threads) -- and, on the way, the
    same batches through the resident-batch path of the open index: its counters are what the command line must reproduce from
    the file.  Returns {"reads", "bytes", "counts": (ref, alt), "first": host copy of the first batches for the oracle, ...}."""
    import threading

    import torch

    from vargeno_amd import synth
    from vargeno_amd.api import pinned_buffer

    t_all = time.time()
    gx.set_stats(False)
    gx.reset()
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    pin = None
    off, done, first = 0, 0, []
    try:
        b = 0
        while done < n_reads:
            n = min(batch_reads, n_reads - done)
            tb, tq, to = src.batch(500_000 + b, n, length=read_len, lowq=lowq)[:3]
            gx.process_device(tb, tq, to, n)                            # the resident-batch path, same reads
            if done < keep_first:
                first.append(synth.reads_to_host(tb, tq, to))
            L = int(to[1].item())
            rec = 13 + L + 3 + L + 1                                    # "@r%010d\n" bases "\n+\n" quals "\n"
            m = torch.empty((n, rec), dtype=torch.uint8, device=tb.device)
            ids = torch.arange(done, done + n, device=tb.device, dtype=torch.int64)
            m[:, 0] = 64
            m[:, 1] = 114
            for k in range(10):
                m[:, 2 + k] = (48 + (ids // 10 ** (9 - k)) % 10).to(torch.uint8)
            m[:, 12] = 10
            m[:, 13:13 + L] = tb.view(n, L)
            m[:, 13 + L] = 10
            m[:, 14 + L] = 43
            m[:, 15 + L] = 10
            m[:, 16 + L:16 + 2 * L] = tq.view(n, L)
            m[:, 16 + 2 * L] = 10
            nb = n * rec
            if pin is None or len(pin[0]) < nb:
                pin = pinned_buffer(nb)
            torch.as_tensor(pin[0][:nb]).copy_(m.view(-1))              # device -> page-locked host
            del m, ids, tb, tq, to
            view = memoryview(pin[0][:nb])
            parts = [(a, min(nb, a + (nb + 3) // 4)) for a in range(0, nb, (nb + 3) // 4)]
            ths = [threading.Thread(target=lambda a=a, e=e: os.pwrite(fd, view[a:e], off + a)) for a, e in parts]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            off += nb
            done += n
            b += 1
    finally:
        os.close(fd)
    rc, ac = gx.counts()
    log("[bench] job: FASTQ of %d distinct reads (%.1f GB) written + the same batches through the resident-batch path: %.1fs" % (done, off / 1e9, time.time() - t_all))
    return {"reads": done, "bytes": off, "counts": (rc, ac), "first": first, "write_s": time.time() - t_all}


def job_run(d, job_dir, job, log):
    """One whole job through the drop-in command line: `vargeno geno idx job.fq snps.vcf job.vcf` as a child process -- index open,
    FASTQ ingest (packing starts beside the open), caller, VCF -- timed from outside; its counters (VARGENO_DUMP_COUNTS) against
    the resident-batch path's on the same reads; calls and GQ histogram out of the VCF it wrote."""
    import re

    dump = os.path.join(job_dir, "job.counts")
    env = dict(os.environ, VARGENO_VERBOSE="1", VARGENO_DUMP_COUNTS=dump, VARGENO_PREPACK_GB="16", VARGENO_VCF_CLOCKS="1")
    # the device idle for a while, as a job's would be: this process has just freed 250 GB of it, and memory that another process has
    # JUST freed is cleared by the driver at allocation (~43 GB/s; vg_index_open 9 s instead of 3: profiles/cold_start_r05.txt)
    time.sleep(IDLE_BEFORE_CHILD_S)
    t0 = time.time()
    p = subprocess.run([BIN, "geno", "idx", os.path.join(job_dir, "job.fq"), "snps.vcf", os.path.join(job_dir, "job.vcf")], cwd=d, env=env, capture_output=True, text=True, timeout=600)
    wall = time.time() - t0
    out = {"reads": job["reads"], "fastq_GB": job["bytes"] / 1e9, "device_idle_before_s": IDLE_BEFORE_CHILD_S, "wall_s": wall, "whole_job_reads_per_s": job["reads"] / wall, "rc": p.returncode}
    if p.returncode != 0:
        out["failed"] = (p.stderr or "")[-600:]
        return out
    mt = re.search(r"wall: ([\d.]+) s = index load ([\d.]+) \+ FASTQ->counters ([\d.]+) \(([\d.]+) M reads/s\) \+ call/VCF ([\d.]+)(?: \+ close ([\d.]+))?", p.stderr)
    if mt:
        out.update({"cli_wall_s": float(mt.group(1)), "index_open_s": float(mt.group(2)), "ingest_after_open_s": float(mt.group(3)), "call_vcf_s": float(mt.group(5)), "close_s": float(mt.group(6) or 0),
                    "wall_minus_open_s": wall - float(mt.group(2))})
    for ln in p.stderr.splitlines():
        if ln.startswith("ingest, replica 0:"):
            out["ingest_route"] = ln[len("ingest, replica 0:"):].strip()
        if ln.startswith("index start-up:"):
            out["index_open_phases"] = ln[len("index start-up:"):].strip()
        if ln.startswith("vcf:"):
            out["call_vcf_phases"] = ln[4:].strip()
        if ln.startswith("process: alive for"):
            # the child's own clocks: before main() (loader, runtime start-up) and after its last line (the operating system
            # taking the process apart: page-locked memory unpinned, 240 GB of device memory unmapped)
            alive = float(ln.split()[3])
            out["before_main_s"] = max(0.0, alive - out.get("cli_wall_s", alive))
            out["exit_teardown_s"] = max(0.0, wall - alive)
    cnt = np.fromfile(dump, dtype=np.uint8)
    rc, ac = job["counts"]
    ns = len(rc)
    out["counters_equal_resident_batch_path"] = bool(len(cnt) == 2 * ns and np.array_equal(cnt[:ns], rc) and np.array_equal(cnt[ns:], ac))
    assert out["counters_equal_resident_batch_path"], "the command line's counters differ from the resident-batch path's on the same reads"
    text = open(os.path.join(job_dir, "job.vcf"), "rb").read()
    calls = re.findall(rb"\t([01]/[01]):(\d+)\n", text)
    gq = np.array([int(q) for _, q in calls], dtype=np.int64) if calls else np.zeros(0, np.int64)
    gts = {}
    for g_, _ in calls:
        gts[g_.decode()] = gts.get(g_.decode(), 0) + 1
    hist, edges = np.histogram(gq, bins=[0, 10, 20, 30, 50, 100, 200, 400, 10 ** 6]) if len(gq) else ([], [])
    out.update({"called": len(calls), "genotypes": gts, "gq_histogram": {"%d-%d" % (edges[i], edges[i + 1] - 1): int(hist[i]) for i in range(len(hist))}, "gq_median": float(np.median(gq)) if len(gq) else None,
                "mean_coverage_per_site": float((rc.astype(np.int64).sum() + ac.astype(np.int64).sum()) / max(ns, 1))})
    log("[bench] job: %d reads, wall %.2f s (open %.2f + ingest %.2f + call/VCF %.2f), %.4g reads/s whole job, %d called, counters equal the resident path's" % (
        job["reads"], wall, out.get("index_open_s", 0), out.get("ingest_after_open_s", 0), out.get("call_vcf_s", 0), out["whole_job_reads_per_s"], len(calls)))
    return out


class FifoFeed:
    """The writing end of the `job_stream` leg's FIFO.  A pipe is two copies (write(): user -> pipe pages, read(): pipe pages -> user)
    and a page allocation per 4 KiB; `vmsplice` lends the pipe the writer's own pages instead, so that only the reader's copy is left
    (1.6 x the rate in the build container: 2.5 -> 4.0 GB/s).  Lent pages must not change before they are read: the last
    2 x pipe-size bytes of every buffer go through write() -- when that returns every slot of the pipe's ring holds a copied page, so no
    page of the buffer is in the pipe any more and the caller may refill it.  (That holds for a reader that copies out of this pipe.  The
    command line MOVES the pipe's pages on to private pipes -- splice, host/main.cpp PipeIngest -- where up to a few MiB of them wait
    for their copier thread: `job_stream` therefore rotates THREE buffers of a whole batch, 2.5 GB, and refills one only after a whole
    later batch has gone through the pipe completely -- every lent page of the first is then gigabytes behind the reader's copiers.)  Memory the
    kernel cannot lend (device-driver mappings such as hipHostMalloc's) or a kernel that refuses the call: plain write() from the
    first refusal on."""

    def __init__(self, fd, lend=True):
        import ctypes
        import fcntl

        self.fd = fd
        self.lent_bytes = 0
        self.copied_bytes = 0
        self.refusal = None
        try:
            fcntl.fcntl(fd, 1031, 1 << 20)                            # F_SETPIPE_SZ: the largest pipe an unprivileged process may ask for
        except Exception:
            pass
        try:
            self.pipe_bytes = int(fcntl.fcntl(fd, 1032))              # F_GETPIPE_SZ
        except Exception:
            self.pipe_bytes = 1 << 16
        self.libc = None
        if lend:
            try:
                self.libc = ctypes.CDLL(None, use_errno=True)
                self.libc.vmsplice.restype = ctypes.c_ssize_t
                self.libc.vmsplice.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_ulong, ctypes.c_uint]
            except Exception as e:
                self.libc, self.refusal = None, repr(e)

    def write_all(self, view):
        """All of `view` (a writable contiguous buffer: memoryview / numpy uint8 array) into the pipe; returns when none of its pages is
        in the pipe any more."""
        import ctypes

        mv = memoryview(view).cast("B")
        n = len(mv)
        tail = min(n, 2 * self.pipe_bytes + 8192)
        at = 0
        if self.libc is not None and n > tail:
            base = ctypes.addressof(ctypes.c_char.from_buffer(mv))

            class IoVec(ctypes.Structure):
                _fields_ = [("base", ctypes.c_void_p), ("len", ctypes.c_size_t)]
            iov = IoVec()
            while at < n - tail:
                iov.base, iov.len = base + at, min(n - tail - at, 16 << 20)
                w = self.libc.vmsplice(self.fd, ctypes.byref(iov), 1, 0)
                if w < 0:
                    err = ctypes.get_errno()
                    if err == 4:                                       # EINTR
                        continue
                    if err == 32:                                      # EPIPE: the reader has gone
                        raise BrokenPipeError(err, os.strerror(err))
                    self.libc, self.refusal = None, "vmsplice: %s" % os.strerror(err)
                    break
                at += w
            self.lent_bytes += at
        while at < n:
            w = os.write(self.fd, mv[at:at + (64 << 20)])
            at += w
            self.copied_bytes += w


def lendable_pinned_buffer(nbytes):
    """(numpy uint8 view, owner, kind): anonymous memory -- which a pipe can borrow, FifoFeed -- page-locked for the device's copy engine
    through hipHostRegister; when that is not to be had, the library's page-locked allocation (hipHostMalloc: a driver mapping, copies
    at the same rate, but a pipe cannot borrow it)."""
    import ctypes
    import mmap

    try:
        import torch

        m = mmap.mmap(-1, int(nbytes))                                # page-aligned, zero-filled on first touch
        try:
            m.madvise(mmap.MADV_HUGEPAGE)
        except Exception:
            pass
        arr = np.frombuffer(m, dtype=np.uint8)
        arr[::4096] = 0                                               # every page exists before it is locked
        addr = ctypes.addressof(ctypes.c_char.from_buffer(m))
        rc = torch.cuda.cudart().cudaHostRegister(addr, int(nbytes), 0)
        if int(rc) != 0:
            raise RuntimeError("hipHostRegister: %r" % (rc,))

        class _Owner:
            def __init__(self):
                self.m, self.addr = m, addr

            def __del__(self):
                try:
                    torch.cuda.cudart().cudaHostUnregister(self.addr)
                except Exception:
                    pass
        return arr, _Owner(), "anonymous memory, hipHostRegister"
    except Exception as e:
        from vargeno_amd.api import pinned_buffer

        a, own = pinned_buffer(nbytes)
        return a, own, "hipHostMalloc (%r)" % (e,)


def job_stream(src, d, job_dir, prefix, dev_index, n_reads, batch_reads, lowq, read_len, log, deadline_s=90.0):
    """The metric's own job size through the command line: BASELINE.json quotes "hg38 + dbSNP 30x" = 620 M reads, 195 GB of FASTQ -- more
    than the container may keep in a file beside everything else, so the reads are never a file: this process generates them batch
    by batch on the device (the seeds of the `job` leg, continued), puts the FASTQ text together there, and writes it into a FIFO that
    `vargeno geno idx <fifo> snps.vcf out.vcf` reads -- the once-only route of the command line (one descriptor, host packer,
    read store while the index opens, then straight into the read loop: host/main.cpp, PipeIngest).  The reference reads such a
    stream the same way (fopen + fgets, qv.cc:2182, 760-763).  Wall time from the child's start to its exit; what bounds it is on the
    line (the generator's and the pipe's rates beside the command line's own numbers).  Afterwards this process opens the index
    again and runs the SAME batches through the resident-batch path: the command line's counters must be identical.
    `deadline_s` ends the stream early (a complete record, EOF) when the feed is slower than planned: the line says how many reads went through."""
    import re
    import threading

    import torch

    from vargeno_amd.api import GenoIndex, pinned_buffer

    fifo = os.path.join(job_dir, "job.fifo")
    dump = os.path.join(job_dir, "job_stream.counts")
    out_vcf = os.path.join(job_dir, "job_stream.vcf")
    for pth in (fifo, dump, out_vcf):
        if os.path.exists(pth):
            os.remove(pth)
    os.mkfifo(fifo)
    env = dict(os.environ, VARGENO_VERBOSE="1", VARGENO_DUMP_COUNTS=dump, VARGENO_PREPACK_GB="16")
    time.sleep(IDLE_BEFORE_CHILD_S)                                   # (the device idle for a while: see job_run)
    t0 = time.time()
    p = subprocess.Popen([BIN, "geno", "idx", fifo, "snps.vcf", out_vcf], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    import errno
    import fcntl

    fd = None
    while fd is None:                                                 # (a FIFO opens for writing once somebody reads it: the child does so first thing)
        try:
            fd = os.open(fifo, os.O_WRONLY | os.O_NONBLOCK)
        except OSError as e:
            if e.errno != errno.ENXIO or p.poll() is not None or time.time() - t0 > 60:
                p.kill()
                return {"skipped": "the command line did not open the FIFO: %r / %s" % (e, (p.communicate()[1] or "")[-300:])}
            time.sleep(0.01)
    fcntl.fcntl(fd, fcntl.F_SETFL, fcntl.fcntl(fd, fcntl.F_GETFL) & ~os.O_NONBLOCK)
    feed = FifoFeed(fd, lend=os.environ.get("VG_BENCH_FIFO_LEND", "1") != "0")
    pins = [None, None, None]                                         # (three: see below)
    pin_kind = None
    done, gen_s, write_s, nb = 0, 0.0, 0.0, 0
    t_feed = time.time()
    wr = {"th": None, "err": None}

    def write_all(view):
        try:
            feed.write_all(view)
        except Exception as e:                                        # (EPIPE: the child has gone)
            wr["err"] = e
    try:
        while done < n_reads and wr["err"] is None and p.poll() is None:
            if time.time() - t_feed > deadline_s:
                break
            tg = time.time()
            n = min(batch_reads, n_reads - done)
            tb, tq, to = src.batch(500_000 + nb, n, length=read_len, lowq=lowq)[:3]
            L = int(to[1].item())
            rec = 13 + L + 3 + L + 1                                  # "@r%010d\n" bases "\n+\n" quals "\n"
            m = torch.empty((n, rec), dtype=torch.uint8, device=tb.device)
            ids = torch.arange(done, done + n, device=tb.device, dtype=torch.int64)
            m[:, 0] = 64
            m[:, 1] = 114
            for k in range(10):
                m[:, 2 + k] = (48 + (ids // 10 ** (9 - k)) % 10).to(torch.uint8)
            m[:, 12] = 10
            m[:, 13:13 + L] = tb.view(n, L)
            m[:, 13 + L] = 10
            m[:, 14 + L] = 43
            m[:, 15 + L] = 10
            m[:, 16 + L:16 + 2 * L] = tq.view(n, L)
            m[:, 16 + 2 * L] = 10
            nbytes = n * rec
            # the buffer of THREE batches ago: batch nb - 2 has gone through the pipe completely since its last lent page (its writer was
            # joined in the previous iteration), so none of its pages can still be anywhere between this process and the reader's copiers --
            # whatever the scheduler does (with two buffers only the start of batch nb - 1 lay between, a margin of time, not of bytes)
            k2 = nb % 3
            if pins[k2] is None or len(pins[k2][0]) < nbytes:
                pins[k2] = None
                pins[k2] = lendable_pinned_buffer(nbytes) if feed.libc is not None else pinned_buffer(nbytes) + ("hipHostMalloc",)
                pin_kind = pins[k2][2]
            torch.as_tensor(pins[k2][0][:nbytes]).copy_(m.view(-1))   # device -> page-locked host
            if nb == 0 and os.environ.get("VG_BENCH_KEEP_FASTQ"):      # (profiles/pipe_ab.py feeds this text again and again)
                with open(os.environ["VG_BENCH_KEEP_FASTQ"], "wb") as f:
                    f.write(memoryview(pins[k2][0][:nbytes]))
            del m, ids, tb, tq, to
            gen_s += time.time() - tg
            tw = time.time()
            if wr["th"] is not None:
                wr["th"].join()                                       # one batch is written while the next is generated
            write_s += time.time() - tw
            wr["th"] = threading.Thread(target=write_all, args=(memoryview(pins[k2][0][:nbytes]),))
            wr["th"].start()
            done += n
            nb += 1
        if wr["th"] is not None:
            tw = time.time()
            wr["th"].join()
            write_s += time.time() - tw
    finally:
        os.close(fd)
    feed_s = time.time() - t_feed
    try:
        so, se = p.communicate(timeout=300)
    except subprocess.TimeoutExpired:
        p.kill()
        so, se = p.communicate()
    wall = time.time() - t0
    out = {"reads": done, "reads_wanted": n_reads, "fastq_GB": done * (2 * read_len + 17) / 1e9, "wall_s": wall, "whole_job_reads_per_s": done / wall, "rc": p.returncode,
           "feed_s": feed_s, "generator_s": gen_s, "generator_waiting_for_the_pipe_s": write_s, "feed_GB_per_s": done * (2 * read_len + 17) / 1e9 / max(feed_s, 1e-9),
           "input": "a FIFO this process writes (reads generated on the device, FASTQ text put together there, one batch in flight): the command line's once-only route",
           "feed_pages": {"lent_to_the_pipe_GB": feed.lent_bytes / 1e9, "copied_GB": feed.copied_bytes / 1e9, "pipe_bytes": feed.pipe_bytes, "buffers": pin_kind, "refusal": feed.refusal},
           "bound_by": "the feed (generator + pipe): the command line waits for text" if feed_s > 0.8 * wall else "the command line"}
    if p.returncode != 0 or wr["err"] is not None:
        out["failed"] = ((se or "")[-500:] + " | writer: %r" % (wr["err"],))
        return out
    mt = re.search(r"wall: ([\d.]+) s = index load ([\d.]+) \+ FASTQ->counters ([\d.]+) \(([\d.]+) M reads/s\) \+ call/VCF ([\d.]+)", se)
    if mt:
        out.update({"cli_wall_s": float(mt.group(1)), "index_open_s": float(mt.group(2)), "ingest_after_open_s": float(mt.group(3)), "call_vcf_s": float(mt.group(5)), "wall_minus_open_s": wall - float(mt.group(2))})
    for ln in se.splitlines():
        if ln.startswith("ingest, replica 0:"):
            out["ingest_route"] = ln[len("ingest, replica 0:"):].strip()[:400]
    text = open(out_vcf, "rb").read()
    calls = re.findall(rb"\t([01]/[01]):(\d+)\n", text)
    gq = np.array([int(q) for _, q in calls], dtype=np.int64) if calls else np.zeros(0, np.int64)
    hist, edges = np.histogram(gq, bins=[0, 10, 20, 30, 50, 100, 200, 400, 10 ** 6]) if len(gq) else ([], [])
    out.update({"called": len(calls), "gq_histogram": {"%d-%d" % (edges[i], edges[i + 1] - 1): int(hist[i]) for i in range(len(hist))}, "gq_median": float(np.median(gq)) if len(gq) else None})
    del text, calls
    # the same batches through the resident-batch path of a handle of this process: identical counters
    time.sleep(5.0)
    tv = time.time()
    with GenoIndex.open(prefix, device=dev_index) as gx2:
        gx2.set_stats(False)
        at = 0
        for b in range(nb):
            n = min(batch_reads, done - at)
            tb, tq, to = src.batch(500_000 + b, n, length=read_len, lowq=lowq)[:3]
            gx2.process_device(tb, tq, to, n)
            gx2.sync()
            del tb, tq, to
            at += n
        rc, ac = gx2.counts()
    cnt = np.fromfile(dump, dtype=np.uint8)
    ns = len(rc)
    out["counters_equal_resident_batch_path"] = bool(len(cnt) == 2 * ns and np.array_equal(cnt[:ns], rc) and np.array_equal(cnt[ns:], ac))
    out["verification_s"] = time.time() - tv
    out["mean_coverage_per_site"] = float((rc.astype(np.int64).sum() + ac.astype(np.int64).sum()) / max(ns, 1))
    assert out["counters_equal_resident_batch_path"], "the streamed job's counters differ from the resident-batch path's on the same reads"
    log("[bench] job_stream: %d reads through a FIFO, wall %.1f s (open %.2f, feed %.1f s = %.2f GB/s: generator %.1f s, waiting for the pipe %.1f s), %.4g reads/s whole job, %d called, counters equal the resident path's (checked in %.0f s)" % (
        done, wall, out.get("index_open_s", 0), feed_s, out["feed_GB_per_s"], gen_s, write_s, out["whole_job_reads_per_s"], out["called"], out["verification_s"]))
    return out


IDLE_BEFORE_CHILD_S = 20.0        # seconds between this process's release of the device and a child's start (see job_run)
T_START = time.time()
BUDGET_S = float(os.environ.get("VG_BENCH_BUDGET_S", "1500"))
# (estimated wall seconds, bench.py arguments) of the secondary legs that run as child processes, in this order
# (measured on the pool's boxes: 15 s, 145 s, 315 s)
CHILD_LEGS = [("chr22", 60, ["--workload", "chr22", "--steps", "40", "--warmup", "5"]),
              # the same index under a 10 GB budget (vg_index_open_ex through $VG_MAX_DEVICE_BYTES): tables of the index's own size -- ~7.6 GB of HBM
              # instead of 91 -- at the price profiles/ab_chr22_table_bits_r06.txt measures (sparser buckets are faster)
              ("chr22_compact", 60, ["--workload", "chr22", "--steps", "40", "--warmup", "5", "--device-budget", "10000000000"]),
              ("repeats30", 240, ["--workload", "hg38", "--repeats", "0.3"]),
              ("hg38f", 450, ["--workload", "hg38f", "--steps", "20", "--warmup", "3"]),
              ("softmask50", 240, ["--workload", "hg38", "--softmask", "0.5"])]


def main_kernel_name(views):
    """The kernel the main tier of this handle launches (vargeno_hip.hip, enqueue_batch): the instantiation for an index
    without the merged view is a kernel of its own."""
    return "vg_wave_kernel_big" if "mx" not in views else "vg_wave_kernel"


def same_device_code(a, b):
    """Two builds of the library whose kernels are the same machine code (profiles/device_code_sha.sh prints build id and the sha256
    of the gfx950 code object's .text; profiles/device_code_r*.txt keeps the lines): a counter profile of one holds for the other.
    Returns the file and hash that say so, or None."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "device_code_r*.txt"))):
        sha = {}
        for line in open(path):
            f = line.split()
            if len(f) >= 8 and f[0] == "build" and "sha256" in f:
                sha[f[1]] = f[f.index("sha256") + 1]
        if a in sha and b in sha and sha[a] == sha[b]:
            return "%s, .text sha256 %s" % (os.path.basename(path), sha[a][:16])
    return None


def traffic_for(args, build_id):
    """HBM traffic and L2 misses of the dominant kernel cannot be counted inside a timed run: they come from the separate
    rocprofv3 --pmc passes of this same command, committed under profiles/ -- and only of THIS build of the library: a traffic
    file is stamped with the vg_build_id() it was measured on and with the workload it was measured for."""
    import glob

    traffic, misses, note = None, None, "no profiles/traffic_*.json for this workload"
    want = {"genome": args.genome, "snps": args.snps, "reads": args.reads}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json")), reverse=True):
        try:
            tj = json.load(open(path))
            w = dict(tj["workload"])
            if {k: w.get(k) for k in want} != want or float(w.get("lowq", 0.08)) != args.lowq or float(w.get("repeats", 0.0)) != args.repeats or bool(w.get("gate_words", True)) != bool(args.gate_words) \
                    or int(w.get("read_len", 150)) != args.read_len or float(w.get("softmask", 0.0)) != args.softmask or (w.get("device_budget") or None) != (args.device_budget or None):
                continue
            same = same_device_code(tj.get("build_id"), build_id)
            if tj.get("build_id") != build_id and not same:
                note = "%s was measured on build %s, this is build %s: not quoted" % (os.path.basename(path), tj.get("build_id"), build_id)
                continue
            traffic, misses = tj["traffic_bytes_per_launch"], tj.get("TCC_MISS_sum")
            note = "%s (separate rocprofv3 --pmc passes of this command on %s; %s)" % (
                os.path.basename(path), "this build" if tj.get("build_id") == build_id else "build %s, whose kernels are byte-identical to this build's: %s" % (tj.get("build_id"), same),
                tj.get("traffic_formula", "FETCH_SIZE + WRITE_SIZE"))
            break
        except Exception:
            pass
    return traffic, misses, note


def timed_block(gx, run, batches, steps, first=0):
    """K steps back to back over the rotating batches, then fold + clamp + fetch of the counters ("first submit
    -> counters reduced and fetched"); returns seconds."""
    t0 = time.perf_counter()
    for i in range(steps):
        run(batches[(first + i) % len(batches)])
    timed_block.enqueue_s = time.perf_counter() - t0        # the host's part: K batches handed over (a slot ring of three: it waits when the device is behind)
    fetched = gx.counts(copy=False)
    dt = time.perf_counter() - t0
    del fetched
    return dt


def check_against_oracle(gx, ox, run, batch, n_check, st_full, ref):
    """Parity of one batch (its first n_check reads) against the oracle: site counters of the counting build and of the timed
    build, and the event counters.  Returns (parity dict, seconds the many-thread oracle run took, host reads)."""
    from vargeno_amd import synth

    n_all = len(batch[2]) - 1
    n_check = min(n_check, n_all)
    r0 = synth.reads_to_host(*batch[:3], 0, n_check)
    nt = min(os.cpu_count() or 1, 64)
    ref.wait_quiet("the oracle's %d-thread run" % nt)
    ref.heavy_begin()
    ox.reset()
    t0 = time.time()
    ox.process(r0.bases, r0.quals, r0.offsets, nthreads=nt)
    t_all = time.time() - t0
    ref.heavy_end()
    so = ox.sites()
    want = ox.stats.as_dict()
    if n_check == n_all:
        sub, st = batch, st_full
    else:                                                   # a slice of the batch: its own counted pass
        b1 = int(batch[2][n_check].item())
        sub = (batch[0][:b1], batch[1][:b1], batch[2][:n_check + 1].contiguous(), batch[3][:n_check].contiguous())
        gx.set_stats(True)
        gx.reset()
        run(sub)
        st = gx.stats()
    rc, ac = gx.counts()                                    # (of the counted pass just made -- the caller's, for a whole batch)
    bad = int((rc != so["ref_cnt"]).sum() + (ac != so["alt_cnt"]).sum())
    assert bad == 0, "HIP counters != oracle at %d of %d site counters" % (bad, 2 * len(rc))
    for k, v in want.items():
        assert st[k] == v, "event counter %s: hip %d oracle %d" % (k, st[k], v)
    # the timed build (event counting off; it reads the re-laid-out views) must give the same counters, in both input forms
    gx.set_stats(False)
    for strings in (True, False):
        gx.reset()
        run(sub, strings=strings)
        rc2, ac2 = gx.counts()
        assert np.array_equal(rc2, so["ref_cnt"]) and np.array_equal(ac2, so["alt_cnt"]), "timed build (%s) != oracle" % ("quality strings" if strings else "gate words")
    par = {"equal": True, "reads": int(n_check), "site_counters": int(2 * len(rc)), "event_counters": len(want), "increments": int(rc.astype(np.int64).sum() + ac.astype(np.int64).sum()),
           "builds": "counting build, timed build with quality strings, timed build with gate words", "against": "oracle/vg_oracle.c (pinned on the reference binary, tests/test_oracle_golden.py)"}
    log("[bench] parity: %d site counters (counting build + timed build in both input forms) and %d event counters identical to the oracle on %d reads" % (2 * len(rc), len(want), n_check))
    return par, t_all, r0


def roofline_of(alg_bytes, k_ms, reads, kernel):
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "kernel": kernel, "kernel_ms": k_ms,
            "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_per_read": alg_bytes / reads}


LINE_LIMIT = 8000          # bytes; the driver reads the last ~10 KB of stdout and parses the last line (BENCH_r05: a 25 KB line came back `parsed: null`)


def _r(x, sig=5):
    """a float with `sig` significant digits (the compact line carries numbers, not noise)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float("%.*g" % (sig, float(x)))
    except Exception:
        return x


def _short(sv, n):
    sv = "" if sv is None else str(sv)
    return sv if len(sv) <= n else sv[:n - 1] + "\u2026"


def compact_line(out, detail_path):
    """The one JSON line of the contract: metric/value/.../config/roofline/cpu_baseline in full (short strings), one-line summaries of
    everything else; the full record is `out` itself, written to `detail_path` and to stderr.  Asserted < LINE_LIMIT bytes (also by
    tests/test_abi_and_layout.py on a synthetic worst case)."""
    c = out["config"]
    rf = out["roofline"] or {}
    cb = out.get("cpu_baseline")
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: _r(out[k], 7) if k in ("value", "ms_per_step") else out[k] for k in keep}
    line["config"] = {"workload": _short(c["workload"], 140), "reads_per_step_per_gpu": c["reads_per_step_per_gpu"], "resident_batches": c["resident_batches"], "genome_bp": c["genome_bp"],
                      "snps_requested": c["snps_requested"], "read_len": c["read_len"], "lowq": c["lowq"], "repeats": c["repeats"], "softmask": c["softmask"], "gate_words": c["gate_words"], "device_budget": c.get("device_budget"),
                      "index_bytes_hbm": c["index_bytes_hbm"], "index_views": _short(",".join(c.get("index_views") or []) if isinstance(c.get("index_views"), (list, tuple)) else c.get("index_views"), 80), "index_open_s": _r(c["index_open_s"], 4), "lib_build_id": c["lib_build_id"],
                      "parallelism": _short(c["parallelism"], 120)}
    gcl = rf.get("gather_ceiling") or {}
    line["roofline"] = {"bound": rf.get("bound"), "achieved": _r(rf.get("achieved")), "peak": rf.get("peak"), "unit": rf.get("unit"), "frac": _r(rf.get("frac")), "kernel": rf.get("kernel"),
                        "kernel_ms": _r(rf.get("kernel_ms")), "algorithmic_bytes_per_launch": rf.get("algorithmic_bytes_per_launch"), "algorithmic_bytes_per_read": _r(rf.get("algorithmic_bytes_per_read"), 6),
                        "traffic": rf.get("traffic"), "traffic_source": _short(rf.get("traffic_source"), 90),
                        "random_line_ceiling_GB_per_s": _r(gcl.get("peak_GB_per_s"), 4), "frac_of_random_line_ceiling": _r(gcl.get("frac"), 4)}
    if cb:
        line["cpu_baseline"] = {"value": _r(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"), "sample": _short(cb.get("sample"), 200)}
        if cb.get("binary_sha256"):
            line["cpu_baseline"]["binary_sha256_16"] = cb["binary_sha256"][:16]
        if cb.get("port"):
            line["cpu_baseline"]["port_1_thread"] = _r(cb["port"].get("value"))
        if cb.get("all_cores"):
            line["cpu_baseline"]["port_all_cores"] = {"value": _r(cb["all_cores"].get("value")), "threads": cb["all_cores"].get("threads")}
    else:
        line["cpu_baseline"] = None
    par = out.get("parity")
    line["parity"] = {"equal": par.get("equal"), "reads": par.get("reads"), "site_counters": par.get("site_counters"), "event_counters": par.get("event_counters")} if par else None
    dm = out.get("device_ms_per_step") or {}
    line["device_ms_per_step"] = {"pack": _r(dm.get("pack"), 4), "wave": _r(dm.get("wave"), 4), "spill_tiers_overlapped": _r(dm.get("spill_tiers_overlapped"), 4)}
    line["input_form"] = "quality strings" if "quality strings" in (out.get("input_form") or "") else "gate words"
    of = out.get("other_input_form")
    if of:
        line["other_input_form"] = {"input": "gate words" if "gate word" in of.get("input", "") else "quality strings", "value": _r(of.get("value")), "ms_per_step": _r(of.get("ms_per_step"), 4)}
    su = out.get("sustained")
    if su:
        line["sustained"] = {"value": _r(su.get("value")), "seconds": _r(su.get("seconds"), 3), "steps": su.get("steps"), "ms_per_step_median": _r(su["ms_per_step"].get("median"), 4)}
    ing = out.get("ingest_end_to_end")
    if ing:
        line["ingest_end_to_end"] = {"failed": True} if "failed" in ing else {"value": _r(ing.get("value")), "chosen": ing.get("chosen"),
                                     "paths": {k: _r(v.get("value")) for k, v in (ing.get("paths") or {}).items()}}
    jb = out.get("job")
    if jb:
        line["job"] = {"skipped": _short(jb["skipped"], 100)} if "skipped" in jb else \
            {k: _r(jb.get(k), 4) for k in ("reads", "wall_s", "index_open_s", "ingest_after_open_s", "call_vcf_s", "wall_minus_open_s", "whole_job_reads_per_s", "fastq_source_reads_per_s", "called", "gq_median", "counters_equal_resident_batch_path") if k in jb}
        if "first_reads_against_oracle" in jb and jb["first_reads_against_oracle"]:
            line["job"]["first_reads_against_oracle"] = {k: jb["first_reads_against_oracle"].get(k) for k in ("equal", "reads") if k in jb["first_reads_against_oracle"]}
    js = out.get("job_stream")
    if js:
        line["job_stream"] = {"skipped": _short(js["skipped"], 100)} if "skipped" in js else \
            {k: _r(js.get(k), 4) for k in ("reads", "wall_s", "index_open_s", "wall_minus_open_s", "whole_job_reads_per_s", "feed_GB_per_s", "called", "gq_median", "counters_equal_resident_batch_path") if k in js}
        if "bound_by" in js:
            line["job_stream"]["bound_by"] = _short(js["bound_by"], 60)
    if out.get("multi_gpu_verification"):
        line["multi_gpu_verification"] = out["multi_gpu_verification"]
    pr = out.get("multi_gpu_per_rank")
    if pr:
        line["multi_gpu_per_rank"] = {k: ([_r(x, 4) for x in v] if isinstance(v, list) else v) for k, v in pr.items()}
    sec = out.get("secondary")
    if sec:
        line["secondary"] = {}
        for name, e in sec.items():
            if "skipped" in e:
                line["secondary"][name] = {"skipped": _short(e["skipped"], 100)}
                continue
            r2 = e.get("roofline") or {}
            line["secondary"][name] = {"value": _r(e.get("value")), "ms_per_step": _r(e.get("ms_per_step"), 4), "kernel_ms": _r(r2.get("kernel_ms"), 4), "frac": _r(r2.get("frac"), 4),
                                       "traffic": r2.get("traffic"), "alg_bytes_per_read": _r(r2.get("algorithmic_bytes_per_read"), 5), "index_bytes_hbm": e.get("index_bytes_hbm"),
                                       "parity": (e.get("parity") or {}).get("equal"), "parity_reads": (e.get("parity") or {}).get("reads")}
    line["detail"] = detail_path
    line["bench_wall_s"] = _r(out.get("bench_wall_s"), 4)
    text = json.dumps(line)
    if len(text) >= LINE_LIMIT:                                  # never again an unparseable line: shed the optional summaries, largest first
        for k in ("multi_gpu_per_rank", "ingest_end_to_end", "sustained", "other_input_form", "job_stream", "job", "secondary"):
            line.pop(k, None)
            text = json.dumps(line)
            if len(text) < LINE_LIMIT:
                break
    assert len(text) < LINE_LIMIT, "bench line is %d bytes" % len(text)
    return text


def write_detail(out, args, d):
    """The full record: one file beside the index files (or --detail-out), a copy under gpurun_out/ when that exists (merged back from a
    gpurun lease), and one `[bench-detail]` line on stderr.  Returns the path reported on the compact line."""
    paths = [args.detail_out] if args.detail_out else [os.path.join(os.path.dirname(d), "bench_detail_%s.json" % os.path.basename(d))]
    gdir = os.path.join(ROOT, "gpurun_out")
    if not args.detail_out and os.path.isdir(gdir):
        paths.append(os.path.join(gdir, "bench_detail_%s.json" % os.path.basename(d)))
    text = json.dumps(out)
    wrote = None
    for pth in paths:
        try:
            with open(pth, "w") as f:
                f.write(text + "\n")
            wrote = wrote or pth
        except OSError:
            pass
    log("[bench-detail] " + text)
    return wrote


def cgroup_room():
    """Bytes the container may still take (cgroup v2 memory.max - memory.current; tmpfs and the page cache count), or None without a limit.
    A box of the pool is lost when its container reaches the limit: every leg that needs tens of GB asks first and is watched while it runs."""
    def rd(path):
        try:
            v = open(path).read().split()[0]
            return None if v == "max" else int(v)
        except Exception:
            return None
    mx, cur = rd("/sys/fs/cgroup/memory.max"), rd("/sys/fs/cgroup/memory.current")
    if mx is None or cur is None:
        return None
    # (page cache of files on disk is charged too, but given up under pressure: memory.stat's `file` less `shmem` does not count against the room)
    cache = 0
    try:
        stat = dict(ln.split()[:2] for ln in open("/sys/fs/cgroup/memory.stat"))
        cache = max(0, int(stat.get("file", 0)) - int(stat.get("shmem", 0)))
    except Exception:
        pass
    return mx - max(0, cur - cache)


def drop_file_cache(directory):
    """The page cache of the files under `directory` given up (it is charged to the container)."""
    for root, _, files in os.walk(directory):
        for fn in files:
            try:
                fd = os.open(os.path.join(root, fn), os.O_RDONLY)
                try:
                    os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
                finally:
                    os.close(fd)
            except OSError:
                pass


# host memory a child leg needs at its peak, GB: index files on tmpfs or in the page cache + `vargeno index`'s arrays + the oracle's copy
CHILD_LEG_ROOM_GB = {"chr22": 12, "chr22_compact": 12, "repeats30": 140, "hg38f": 235, "softmask50": 140}


def run_child_leg(name, est, extra, args, ref):
    """One secondary configuration as a child process of this one (which holds no index any more): its own bench.py line."""
    left = BUDGET_S - (time.time() - T_START)
    if left < est:
        return {"skipped": "time budget: %.0f s of $VG_BENCH_BUDGET_S = %.0f s left, this leg is estimated at %d s" % (left, BUDGET_S, est)}
    room = cgroup_room()
    if room is not None and room < CHILD_LEG_ROOM_GB.get(name, 100) * 1e9:
        return {"skipped": "container memory: %.0f GB left (cgroup memory.max), this leg needs %d GB" % (room / 1e9, CHILD_LEG_ROOM_GB.get(name, 100))}
    ref.wait_quiet("secondary leg %s" % name)
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--secondary", "none", "--no-gather-probe", "--no-ingest", "--cpu-reference", "no", "--sustain-seconds", "0",
           "--cleanup", "--cpu-sample", "200000", "--job-reads", "0", "--no-pretouch"] + extra
    detail = os.path.join(args.workdir, "bench_detail_child_%s.json" % name)
    cmd += ["--detail-out", detail]
    if args.workdir_given:
        cmd += ["--workdir", args.workdir]
    time.sleep(IDLE_BEFORE_CHILD_S)                        # (the device's memory, just freed by the process before, scrubbed: see job_run)
    t0 = time.time()
    log("[bench] secondary leg %s: %s" % (name, " ".join(cmd[2:])))
    try:
        ref.heavy_begin()
        limit = max(60.0, min(left - 20.0, 2.5 * est))
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        # (watched while it runs: should the container come within 12 GB of its memory limit the leg is ended -- a skipped leg, not a lost box)
        import threading

        verdict = {}

        def watch():
            while p.poll() is None:
                r_ = cgroup_room()
                if r_ is not None and r_ < 12e9:
                    verdict["why"] = "container memory: %.0f GB left while the leg ran" % (r_ / 1e9)
                elif time.time() - t0 > limit:
                    verdict["why"] = "child run exceeded its share of the time budget"
                if verdict:
                    try:
                        os.killpg(p.pid, 9)
                    except OSError:
                        pass
                    import shutil
                    shutil.rmtree("/dev/shm/vg_bench", ignore_errors=True)
                    return
                time.sleep(0.5)
        th = threading.Thread(target=watch, daemon=True)
        th.start()
        out_s, err_s = p.communicate()
        th.join(2.0)
        ref.heavy_end()
        if verdict or p.returncode != 0:
            # (a leg that did not end by itself has not cleaned up: its index files -- up to 48 GB on a 79 GB root -- go now)
            import shutil
            for sub in os.listdir(args.workdir):
                full = os.path.join(args.workdir, sub)
                if os.path.isdir(full) and os.path.abspath(full) != os.path.abspath(args.main_dir):
                    shutil.rmtree(full, ignore_errors=True)
        if verdict:
            return {"skipped": verdict["why"], "wall_s": time.time() - t0}
        if p.returncode != 0 or not os.path.exists(detail):
            return {"skipped": "child run failed (rc %d): %s" % (p.returncode, (err_s or "").strip().splitlines()[-1:] or ["no output"]), "wall_s": time.time() - t0}
        j = json.load(open(detail))                        # the child's full record (its stdout line is the compact one)
        rf = j["roofline"]
        return {"workload": j["config"]["workload"], "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "steps": j["steps"], "input_form": j["input_form"],
                "other_input_form": j.get("other_input_form") and {k: j["other_input_form"][k] for k in ("input", "value", "ms_per_step")},
                "roofline": {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "kernel_ms", "algorithmic_bytes_per_read", "traffic", "traffic_source")},
                "parity": j.get("parity"), "device_ms_per_step": j.get("device_ms_per_step"), "reads_per_step_redone_by_deep_list_tier": j.get("reads_per_step_redone_by_deep_list_tier"),
                "index_bytes_hbm": j["config"].get("index_bytes_hbm"), "index_open_s": j["config"].get("index_open_s"), "index_open_phases": j["config"].get("index_open_phases"), "index_views": j["config"].get("index_views"), "index_plan": j["config"].get("index_plan"), "cpu_port_reads_per_s": (j.get("cpu_baseline") or {}).get("value"), "wall_s": time.time() - t0}
    except Exception as e:
        try:
            ref.heavy_end()
        except Exception:
            pass
        return {"skipped": "child run failed: %r" % (e,), "wall_s": time.time() - t0}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    from vargeno_amd import synth

    default_workload = args.workload == "hg38" and args.repeats == 0.0 and args.lowq == 0.08 and args.read_len == 150 and args.softmask == 0.0 and args.genome == PRESETS["hg38"]["genome"] and args.snps == PRESETS["hg38"]["snps"] and args.reads == PRESETS["hg38"]["reads"]
    if args.secondary == "auto":
        legs = ["lowq50", "len101", "len250", "chr22", "chr22_compact", "repeats30", "hg38f", "softmask50"] if (default_workload and world == 1 and args.cpu_sample > 0) else []
    else:
        legs = [x for x in args.secondary.split(",") if x and x != "none"]

    # ---- data set + index files: host only (rank 0 builds, the others wait for its marker file) ----------------------------
    tag = "g%d_s%d_c%d" % (args.genome, args.snps, args.chroms) + ("_r%g" % args.repeats if args.repeats else "") + ("_m%g" % args.softmask if args.softmask else "")
    args.workdir_given = args.workdir is not None
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
    args.main_dir = d
    prefix = os.path.join(d, "idx")
    t0 = time.time()
    # (the chip's gather ceiling first: a child process whose 16 GiB table is freed -- and scrubbed by the driver -- long before
    # this process allocates; memory that another process has JUST freed is cleared at allocation, ~43 GB/s: profiles/cold_start_r05.txt)
    ceiling = None
    if rank == 0 and not args.no_gather_probe:
        ceiling = gather_ceiling()
    # (hg38-scale indexes only, and not in the child legs: those start on a device that has been idle for 20 s, and a pre-touch
    # shortly before a small index's open would itself be the process that has just freed the memory)
    pretouch = pretouch_start(local_rank, world) if (args.genome >= 10 ** 9 and not args.no_pretouch) else (None, time.time())
    g, s, _ = synth.genome_and_snps(genome_len=args.genome, n_snps=args.snps, n_chroms=args.chroms, genotypes="hwe" if args.workload == "hg38f" else "uniform", repeats=args.repeats)
    if rank == 0:
        log("[bench] synthetic genome + SNP list: %.1fs (%d bp, %d SNPs)" % (time.time() - t0, g.total_len, len(s.pos)))
        build_index_files(args, g, s, d, prefix)
    else:
        while not os.path.exists(prefix + ".done"):
            time.sleep(1.0)
    files_warm = None
    if rank == 0 and args.genome >= 10 ** 9:
        files_warm = warm_index_files(prefix)
        log("[bench] index files read once before the open: %.1f GB in %.1f s (%.1f GB/s)" % (files_warm["GB"], files_warm["seconds"], files_warm["GB_per_s"]))
    pretouched = pretouch_finish(pretouch, log)
    if rank == 0 and pretouched and pretouched["rc"] == 0:
        log("[bench] device memory taken once and given back by a child process: hipMalloc of %.1f GB took %.2f s" % (pretouched["bytes"] / 1e9, pretouched["hipMalloc_s"]))
    # ---- GPU from here on --------------------------------------------------------------------------------------------------
    import torch
    import torch.distributed as dist

    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)               # one rank per GPU (VG_BENCH_BACKEND=gloo lets ranks share a GPU for plumbing tests)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev                                      # where the small bookkeeping collectives live
    backend = None
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

    from vargeno_amd.api import GenoIndex, all_reduce_counts, gate_words, shard_range

    # ---- the index first: resident in HBM before this process has allocated (and freed) anything else on the device, and before anything
    #      loads the host (vg_index_open is part of what a job pays: its time and its phases go into the line; the reference binary's two
    #      processes -- 48 GB of per-field fread each -- start after it)
    t0 = time.time()
    cpu0 = time.process_time()
    # (ranks that share a device -- the gloo rehearsals on a one-GPU box -- share its memory: each plans for an equal part)
    sharers = (world + max(ndev, 1) - 1 - dev_index) // max(ndev, 1) if world > max(ndev, 1) else 1
    budget = None
    if sharers > 1:
        # vg_share_budget = (what is free now - 12 GiB) / sharers: the same number for every rank only if every rank asks before any
        # rank opens (round 5's advisor: a rank that asked after a sibling had taken its block got half a share and failed).  The
        # collective below is the barrier; the minimum is everybody's budget
        from vargeno_amd._lib import lib as _l

        b = torch.tensor([int(_l().vg_share_budget(dev_index, int(sharers)))], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(b, op=dist.ReduceOp.MIN)
        budget = int(b.item()) or None
    if args.device_budget:
        budget = min(budget, args.device_budget) if budget else args.device_budget
    gx = GenoIndex.open(prefix, device=dev_index, max_device_bytes=budget)
    t_open = time.time() - t0
    cpu_open = time.process_time() - cpu0                  # host CPU seconds of this process (all its threads) inside vg_index_open
    open_report = gx.open_report
    if rank == 0:
        log("[bench] index resident in HBM: %.1fs, %.1f GB, %d sites" % (t_open, gx.device_bytes / 1e9, gx.num_sites))
        log("[bench] vg_index_open phases: %s" % open_report)

    # reads resident in HBM: NB batches of this rank's part of the read stream + (N > 1) one stream every rank knows
    t0 = time.time()
    src = synth.DeviceReadSource(g, s, dev)
    del g, s
    batches = [src.batch(rank * 1000 + b, args.reads, length=args.read_len, lowq=args.lowq) for b in range(args.batches)]
    common = src.batch(999_999, min(args.reads, 1_000_000), length=args.read_len, lowq=args.lowq) if world > 1 else None
    # the stress profile's batches (50 % low-quality characters) wait on the host until their leg
    # (likewise the batches of the read-length legs: 101 bp -- the reference's own experiment, 3 chunks -- and 250 bp, 7 chunks)
    lowq_host = None
    if "lowq50" in legs:
        lowq_host = [tuple(t.cpu() for t in src.batch(700_000 + b, args.reads, length=args.read_len, lowq=0.5)) for b in range(2)]
    len_host = {}
    for L_ in (101, 250):
        if "len%d" % L_ in legs:
            len_host[L_] = [tuple(t.cpu() for t in src.batch(800_000 + 10 * L_ + b, args.reads, length=L_, lowq=args.lowq)) for b in range(2)]
    want_job = args.job_reads if args.job_reads is not None else (200_000_000 if (default_workload and world == 1 and rank == 0 and args.cpu_sample > 0) else 0)
    if world > 1 or rank != 0:
        want_job = 0
    if args.stream_reads is None:
        args.stream_reads = 620_000_000 if (want_job and default_workload) else 0
    if not want_job:
        src.release()
        del src
        src = None
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()
    if rank == 0:
        log("[bench] %d batches of %d reads generated on the device: %.1fs" % (args.batches, args.reads, time.time() - t0))

    # ---- the `job` leg (N = 1): its first 16 M reads now, for the oracle (they are generated again, with all the others, when the
    #      FASTQ file is written -- late, when the reference binary's processes and the oracle have left the host's memory)
    job = None
    if src is not None:
        job = {"first": [synth.reads_to_host(*src.batch(500_000 + b, args.reads, length=args.read_len, lowq=args.lowq)[:3]) for b in range(min(2, want_job // args.reads))], "wanted": want_job}
        torch.cuda.empty_cache()

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
    ref = ref_timer or NoRef()

    # The resident batches: ASCII bases + quality strings + offsets (the form the reference consumes, vg_reads_process_device) and, for
    # the reduced form, one gate word per read (bit c = quality character c < '8': all the path reads of a quality string,
    # qv.cc:836, 943; vg_reads_process_device_gated -- what the device-side FASTQ framing hands the read loop).  Both forms are timed.
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
    cpu, parity, ox = None, None, None
    if rank == 0 and world == 1 and args.cpu_sample > 0:        # the CPU legs (and the parity check they feed) run at N = 1 only
        from oracle import oracle as O

        t0 = time.time()
        ox = O.OracleIndex.load(prefix)
        log("[bench] oracle index load: %.1fs" % (time.time() - t0))
        ncores = os.cpu_count() or 1
        nt = min(ncores, 64)
        # every host core first: the whole batch, which is also what the parity check compares
        if not args.no_check:
            parity, t_all, r0 = check_against_oracle(gx, ox, run, batches[0], args.reads, st, ref)
        else:
            r0 = synth.reads_to_host(*batches[0][:3])
            ref.wait_quiet("the oracle's %d-thread run" % nt)
            ref.heavy_begin()
            t0 = time.time()
            ox.process(r0.bases, r0.quals, r0.offsets, nthreads=nt)
            t_all = time.time() - t0
            ref.heavy_end()
        # one thread on a bounded sample (the reference is single-threaded: this is the baseline of record when the reference binary is not timed)
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
            ref.wait_quiet("the oracle's %d-thread run" % ncores)
            ref.heavy_begin()
            ox.reset()
            t0 = time.time()
            ox.process(r0.bases, r0.quals, r0.offsets, nthreads=ncores)
            t_more = time.time() - t0
            ref.heavy_end()
            tried[str(ncores)] = r0.n / t_more
            if t_more < best_t:
                best_nt, best_t = ncores, t_more
        cpu = {"value": passes * ns / t_cpu, "unit": "reads/s", "cores": 1, "kind": "port",
               "sample": "%d pass(es) over the first %d reads of batch 0, oracle/vg_oracle.c, 1 thread, %.1f s" % (passes, ns, t_cpu),
               "all_cores": {"value": r0.n / best_t, "threads": best_nt, "host_cores": ncores, "reads_per_s_by_threads": tried,
                             "sample": "batch 0 (%d reads), %.1f s" % (r0.n, best_t)}}
        del r0, sub
        # the job leg's first reads (up to 16 M) against the oracle, through the resident-batch path (the command line's counters are
        # checked against that path's on ALL the job's reads when it has run)
        if job is not None and job.get("first"):
            try:
                fr = job.pop("first")
                job["first"] = None
                ref.wait_quiet("the oracle's run over the job's first reads")
                ref.heavy_begin()
                ox.reset()
                for r_ in fr:
                    ox.process(r_.bases, r_.quals, r_.offsets, nthreads=nt)
                ref.heavy_end()
                so_j = ox.sites()
                gx.set_stats(False)
                gx.reset()
                for r_ in fr:
                    gx.submit(r_.bases, r_.quals, r_.offsets)
                rc_j, ac_j = gx.counts()
                assert np.array_equal(rc_j, so_j["ref_cnt"]) and np.array_equal(ac_j, so_j["alt_cnt"]), "the job's first reads: HIP counters != oracle"
                job["first_reads_against_oracle"] = {"equal": True, "reads": int(sum(r_.n for r_ in fr)), "site_counters": int(2 * len(rc_j))}
                log("[bench] job: the first %d reads' counters identical to the oracle's" % job["first_reads_against_oracle"]["reads"])
                del fr
            except AssertionError:
                raise
            except Exception as e:
                job["first_reads_against_oracle"] = {"skipped": repr(e)}
        if not any(x in legs for x in ("lowq50", "len101", "len250")):
            ox.close()
            ox = None

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
    gx.counts(copy=False)                               # the fetch is part of the warm-up too: its page-locked buffer is made here, not inside the timed region
    gx.sync()
    gx.timing()                                         # drop the warm-up batches from the event averages
    gx.reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        run(batches[i % args.batches])
    t_enqueued = time.perf_counter() - t0               # the host's part of the K steps (a slot ring of three: it waits when the device is behind)
    local_sum, t_reduce = None, None
    if world > 1:
        gx.sync()
        t_steps_done = time.perf_counter()
        local_sum = gx.counts_tensor().sum(dtype=torch.int64).reshape(1).to(coll_dev)      # this rank's increments, before the exchange (checked below)
        t1 = time.perf_counter()
        all_reduce_counts(gx)                           # one RCCL all-reduce of the per-site counters over xGMI
        t_reduce = time.perf_counter() - t1
    fetched = gx.counts(copy=False)                     # "first submit -> counters reduced and fetched": fold, clamp at 63, device -> (page-locked) host
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    del fetched
    tm = gx.timing()                                    # HIP events on the library's own streams, averaged over the K batches
    per_rank = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # checksum of checksums: the reduced counters hold exactly the increments of all ranks
        dist.all_reduce(local_sum)
        total_after = int(gx.counts_tensor().sum(dtype=torch.int64).item())
        assert total_after == int(local_sum.item()), "reduced counters hold %d increments, the ranks made %d" % (total_after, int(local_sum.item()))
        # what every rank saw, so that a scaling run explains itself: main-tier kernel ms, steps' wall ms, the exchange's ms
        mine = torch.tensor([tm["ms_main"], 1e3 * (t_steps_done - t0) / args.steps, 1e3 * t_reduce, t_open, cpu_open, gx.device_bytes / 1e9], dtype=torch.float64, device=coll_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        if rank == 0:
            verification["timed_region_increments"] = total_after
            per_rank = {"kernel_ms": [float(x[0]) for x in allr], "ms_per_step_before_exchange": [float(x[1]) for x in allr], "all_reduce_ms": [float(x[2]) for x in allr],
                        # every replica's start-up as it ran beside the others (they read the same files at the same time): wall seconds, host
                        # CPU seconds of the rank's process, GB of device memory it holds -- what a first run on real hardware is compared with
                        "index_open_s": [float(x[3]) for x in allr], "index_open_cpu_s": [float(x[4]) for x in allr], "index_device_GB": [float(x[5]) for x in allr],
                        "all_reduce_bytes": int(gx.counts_tensor().numel()) * 4, "backend": "RCCL (torch.distributed nccl)" if backend == "nccl" else backend,
                        "devices_visible": ndev, "ranks_seen_by_the_collective": n_seen}

    # ---- the other input form, for the record (N = 1): the same K steps with the gate words handed over (or, under --gate-words, the strings)
    other_form, sustained = None, None
    if rank == 0 and world == 1:
        gx.reset()
        for i in range(min(args.warmup, 2)):
            run(batches[i % args.batches], strings=not args.ascii_quals)
        gx.sync()
        if min(args.warmup, 2) > 0:
            gx.timing()
        t0 = time.perf_counter()
        for i in range(args.steps):
            run(batches[i % args.batches], strings=not args.ascii_quals)
        gx.counts(copy=False)
        dt = time.perf_counter() - t0
        tm2 = gx.timing()
        other_form = {"input": "quality strings (vg_reads_process_device)" if not args.ascii_quals else "one gate word per read (vg_reads_process_device_gated)",
                      "value": args.reads * args.steps / dt, "unit": "reads/s", "ms_per_step": 1e3 * dt / args.steps, "pack_ms": tm2["ms_pack"], "wave_ms": tm2["ms_main"]}
        # ---- sustained: blocks of K steps (each with its fetch, like the timed region) back to back for args.sustain_seconds, so that clocks
        #      and the atomics' contention at steady state are on record (the timed region above lasts tens of milliseconds)
        if args.sustain_seconds > 0:
            gx.reset()
            blocks, t_s0 = [], time.perf_counter()
            while time.perf_counter() - t_s0 < args.sustain_seconds or len(blocks) < 3:
                blocks.append(1e3 * timed_block(gx, run, batches, args.steps, first=len(blocks) * args.steps) / args.steps)
            t_s = time.perf_counter() - t_s0
            gx.timing()
            sustained = {"seconds": t_s, "blocks": len(blocks), "steps": len(blocks) * args.steps, "reads": len(blocks) * args.steps * args.reads, "value": len(blocks) * args.steps * args.reads / t_s, "unit": "reads/s",
                         "ms_per_step": {"min": float(np.min(blocks)), "median": float(np.median(blocks)), "max": float(np.max(blocks)), "first": blocks[0], "last": blocks[-1]},
                         "note": "blocks of %d steps + fold/clamp/fetch of the counters, back to back without a reset (the counters keep summing; the clamp is applied at fetch)" % args.steps}
            log("[bench] sustained: %.2f s, %d steps, %.4g reads/s, ms/step min %.3f median %.3f max %.3f" % (t_s, sustained["steps"], sustained["value"], sustained["ms_per_step"]["min"], sustained["ms_per_step"]["median"], sustained["ms_per_step"]["max"]))

    from vargeno_amd._lib import lib as _vg_lib

    build_id = _vg_lib().vg_build_id().decode()
    views, dev_bytes, plan_text = gx.views, gx.device_bytes, gx.plan
    kernel = main_kernel_name(views)

    # ---- secondary legs on the open index: the stress profile (50 % low-quality characters: 3 gate-open chunks per read) and the two
    #      other read lengths -- each with its own counted pass (algorithmic bytes), parity of 1 M reads against the oracle, K timed steps
    secondary = {}

    def open_index_leg(name, host_batches, lowq, read_len, what):
        t_leg = time.time()
        lb = [tuple(t.to(dev) for t in hb) for hb in host_batches]
        lb = [tuple(b) + (gate_words(b[1], b[2]),) for b in lb]
        torch.cuda.synchronize(dev)
        gx.set_stats(True)
        gx.reset()
        run(lb[0])
        st_l = gx.stats()
        par_l, _, _ = check_against_oracle(gx, ox, run, lb[0], args.reads, st_l, ref)   # (every read of the batch: the oracle's many-thread run is seconds)
        gx.set_stats(False)
        gx.reset()
        for i in range(3):
            run(lb[i % 2])
        gx.sync()
        gx.timing()
        gx.reset()
        dt = timed_block(gx, run, lb, args.steps)
        tm_l = gx.timing()
        roof_l = roofline_of(st_l["alg_bytes"], tm_l["ms_main"], args.reads, kernel)
        import copy

        a_l = copy.copy(args)
        a_l.lowq, a_l.read_len = lowq, read_len
        roof_l["traffic"], _, roof_l["traffic_source"] = traffic_for(a_l, build_id)
        out = {"workload": what % (st_l["gate_open"] / args.reads, args.reads, read_len),
               "value": args.reads * args.steps / dt, "unit": "reads/s", "bases_per_s": args.reads * args.steps * read_len / dt, "ms_per_step": 1e3 * dt / args.steps, "steps": args.steps,
               "input_form": "quality strings" if args.ascii_quals else "gate words",
               "roofline": roof_l, "parity": par_l,
               "device_ms_per_step": {"pack": tm_l["ms_pack"], "wave": tm_l["ms_main"], "spill_tiers_overlapped": tm_l["ms_tail"]},
               "events_per_read": {k: st_l[k] / args.reads for k in ("passes", "chunks", "gate_open", "ref_query", "snp_query", "walks")},
               "reads_per_step_redone_by_deep_list_tier": st_l["overflow_reads"], "wall_s": time.time() - t_leg}
        log("[bench] secondary %s: %.4g reads/s, %.3f ms/step, kernel %.3f ms, frac %.3f" % (name, out["value"], out["ms_per_step"], tm_l["ms_main"], roof_l["frac"]))
        del lb
        torch.cuda.empty_cache()
        return out

    open_legs = []
    if "lowq50" in legs:
        open_legs.append(("lowq50", lowq_host, 0.5, args.read_len, "the main line's index, 50 %% low-quality characters (SURVEY.md §8d stress profile: %.2f gate-open chunks per read), %d x %d bp reads per step rotating over 2 resident batches"))
    for L_ in (101, 250):
        if "len%d" % L_ in legs:
            open_legs.append(("len%d" % L_, len_host.get(L_), args.lowq, L_, "the main line's index, reads of another length (%.2f gate-open chunks per read), %d x %d bp reads per step rotating over 2 resident batches"))
    if rank == 0 and open_legs:
        del batches[1:]                                        # make room: the main line's batches are done with
        torch.cuda.empty_cache()
    for name, hb, lq, L_, what in open_legs:
        if rank != 0:
            continue
        if hb is None or ox is None:
            secondary[name] = {"skipped": "needs the oracle (--cpu-sample > 0) at N = 1"}
            continue
        try:
            secondary[name] = open_index_leg(name, hb, lq, L_, what)
        except Exception as e:                                 # a secondary leg never fails the main line
            secondary[name] = {"skipped": "failed: %r" % (e,)}
            log("[bench] secondary %s failed: %r" % (name, e))
    lowq_host, len_host = None, {}
    if ox is not None:
        ox.close()
        ox = None

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
    # ---- secondary number (N = 1): end to end from FASTQ text in pinned HOST memory -- over PCIe as text (framed on the device) or
    #      framed + packed by host threads -- through vg_fastq_stream_push.  Never `value`: the metric is quoted on batches resident
    #      in HBM.  After the reference binary has been collected: the host-packing route needs the container's CPUs for itself.
    ingest = None
    if rank == 0 and world == 1 and not args.no_ingest:
        ingest = measure_ingest(gx, batches[0], log)
    # ---- the `job` leg: a FASTQ file of distinct reads (and the same reads through the resident-batch path of the open index), then one
    #      whole run of the drop-in command line on it, once this process holds no index any more
    job_out, job_dir = None, None
    if rank == 0 and job is not None and src is not None:
        left = BUDGET_S - (time.time() - T_START)
        job_dir, room = big_file_room()
        per_read = 2 * args.read_len + 17
        n_job = min(want_job, int(room / per_read) // args.reads * args.reads)
        if left < 240:
            job_out = {"skipped": "time budget: %.0f s left" % left}
        elif job_dir is None or n_job < 2 * args.reads:
            job_out = {"skipped": "no room for the job's FASTQ file (memory limit of the container / free space): %d reads would fit" % n_job}
        else:
            try:
                os.makedirs(job_dir, exist_ok=True)
                first_check = job.get("first_reads_against_oracle")
                job = job_fastq(src, gx, os.path.join(job_dir, "job.fq"), n_job, args.reads, args.lowq, log, read_len=args.read_len)
                job["first_reads_against_oracle"] = first_check
                job["wanted"] = want_job
            except Exception as e:
                job_out = {"skipped": "writing the job's FASTQ failed: %r" % (e,)}
                log("[bench] job leg: %r" % (e,))
    gx.close()
    del batches
    torch.cuda.empty_cache()
    if rank == 0 and job_out is None and job is not None and "counts" in job:
        try:
            job_out = job_run(d, job_dir, job, log)
            job_out["first_reads_against_oracle"] = job.get("first_reads_against_oracle")
            job_out["fastq_written_in_s"] = job.get("write_s")
            job_out["reads_wanted"] = job.get("wanted")
        except AssertionError:
            raise
        except Exception as e:
            job_out = {"skipped": "failed: %r" % (e,)}
    # ---- ... and the metric's own job size, 30x = 620 M reads, streamed through a FIFO (never a file): job_stream
    stream_out = None
    if rank == 0 and src is not None and args.stream_reads:
        left = BUDGET_S - (time.time() - T_START)
        if job_out is not None and "skipped" in job_out and job_dir is None:
            job_dir = "/tmp/vg_bench_job"
        if left < 820:                                                # (the child legs behind this one need ~660 s, this leg ~110)
            stream_out = {"skipped": "time budget: %.0f s left" % left}
        else:
            try:
                os.makedirs(job_dir, exist_ok=True)
                if os.path.exists(os.path.join(job_dir, "job.fq")):
                    os.remove(os.path.join(job_dir, "job.fq"))        # (63 GB of tmpfs back before anything else)
                stream_out = job_stream(src, d, job_dir, prefix, dev_index, args.stream_reads, args.reads, args.lowq, args.read_len, log)
            except AssertionError:
                raise
            except Exception as e:
                stream_out = {"skipped": "failed: %r" % (e,)}
                log("[bench] job_stream: %r" % (e,))
    if src is not None:
        src.release()
        del src
    torch.cuda.empty_cache()
    if job_dir is not None:
        import shutil

        shutil.rmtree(job_dir, ignore_errors=True)

    # ---- secondary legs with an index of their own: child processes, one after the other, now that this one holds no index ------
    if rank == 0 and legs:
        drop_file_cache(d)                                     # (48 GB of index files nobody reads any more: the page cache counts against the container's memory)
    if rank == 0:
        for name, est, extra in CHILD_LEGS:
            if name in legs:
                secondary[name] = run_child_leg(name, est, extra, args, NoRef())
                sk = secondary[name].get("skipped")
                log("[bench] secondary %s: %s" % (name, sk if sk else "%.4g reads/s, %.3f ms/step, frac %.3f, parity %s" % (secondary[name]["value"], secondary[name]["ms_per_step"], secondary[name]["roofline"]["frac"], (secondary[name].get("parity") or {}).get("equal"))))

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        k_ms = tm["ms_main"]
        traffic, misses, traffic_note = traffic_for(args, build_id)
        gc = None
        if ceiling:
            # An L2 miss of a random gather moves one 128-byte line (profiles/line_probe_r03_counters.txt), so the chip's measured
            # random-gather rate IS the HBM roofline for this access shape: lines/s x 128 B
            gc = {"peak": ceiling["gathers_per_s"], "unit": "random 128-byte lines/s (tools/gather_probe: 8-byte gathers from a 16 GiB table, one line each; measured in this run)",
                  "peak_GB_per_s": ceiling["gathers_per_s"] * 128 / 1e9,
                  "l2_misses_per_launch": misses, "achieved": (misses / (k_ms * 1e-3)) if misses else None}
            gc["frac"] = (gc["achieved"] / gc["peak"]) if misses else None
        roof = roofline_of(alg_bytes_per_launch, k_ms, args.reads, kernel)
        roof.update({"traffic": traffic, "traffic_source": traffic_note, "traffic_GB_per_s": (traffic / (k_ms * 1e-3) / 1e9) if traffic else None,
                     "traffic_frac_of_peak": (traffic / (k_ms * 1e-3) / 1e9 / 8000.0) if traffic else None, "gather_ceiling": gc})
        out = {
            "metric": "reads/sec genotyped (whole node), hg38+dbSNP 30×; achieved HBM GB/s vs peak",
            "value": world * args.reads * args.steps / elapsed,
            "unit": "reads/s",
            "n_gpus": n_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step,
            "host_enqueue_ms_per_step": 1e3 * t_enqueued / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "%s: %d bp synthetic genome in %d sequence(s), %d SNPs requested, %d x %d bp reads per GPU per step rotating over %d "
                                   "distinct resident batches of the read stream, 0.5%% error, %g%% low-quality chars, seed 20261002%s" % (
                                       "hg38 + full-dbSNP-scale index (BASELINE.json configs[4], one replica)" if args.snps >= 5 * 10 ** 7 else
                                       "hg38-scale (BASELINE.json configs[2])" if args.genome >= 10 ** 9 else "chr22-scale (BASELINE.json configs[1])",
                                       args.genome, args.chroms, args.snps, args.reads, args.read_len, args.batches, 100 * args.lowq,
                                       "" if not args.repeats else "; REPEAT-RICH genome: %g%% of it in planted families of near-identical copies (2-10 and 11-200 copies), 50 microsatellites per Mbp" % (100 * args.repeats)),
                       "reads_per_step_per_gpu": args.reads, "resident_batches": args.batches, "genome_bp": args.genome, "snps_requested": args.snps, "lowq": args.lowq, "repeats": args.repeats, "read_len": args.read_len, "softmask": args.softmask, "gate_words": bool(args.gate_words), "device_budget": args.device_budget,
                       "index_bytes_hbm": dev_bytes, "index_views": views, "index_plan": plan_text, "index_open_s": t_open, "index_open_cpu_s": cpu_open, "index_open_phases": open_report, "device_memory_pretouch": pretouched, "index_files_read_before_open": files_warm, "lib_build_id": build_id,
                       "parallelism": "reads sharded over %d GPU(s), index replicated, one RCCL all-reduce of the site counters after the K steps" % world},
            "roofline": roof,
            "cpu_baseline": cpu,
            "parity": parity,
            "device_ms_per_step": {"pack": tm["ms_pack"], "wave": k_ms, "spill_tiers_overlapped": tm["ms_tail"], "of_which_deep_list_wave_tier": tm["ms_deep_lists"], "batches": tm["batches"],
                                   "note": "spill tiers: elapsed time from the end of a batch's main-tier kernel to the end of its last tier, on the tail stream, under the NEXT batches' "
                                           "kernels -- mostly waiting (a tier's workgroups are placed when main-tier workgroups of the following batch retire), not work: see reads_per_step_redone_by_deep_list_tier"},
            "reads_per_step_redone_by_deep_list_tier": st["overflow_reads"], "reads_per_step_sent_on_to_lane_tier": st["overflow_deep"],
            "events_per_read": {k: st[k] / args.reads for k in ("passes", "chunks", "gate_open", "ref_query", "snp_query", "ctx", "walks", "incr")},
            "input_form": "ASCII bases + offsets + " + ("quality strings (what the reference reads; vg_reads_process_device)" if args.ascii_quals else "one gate word per read (bit c = quality character c < '8'; vg_reads_process_device_gated)") + ", resident in HBM",
            "fetch": "pinned (fold + clamp on the device, two bytes per site into a reused page-locked buffer, GenoIndex.counts(copy=False); rounds 1-3 fetched into pageable memory)",
            "other_input_form": other_form,
            "sustained": sustained,
            "ingest_end_to_end": ingest,
            "job": job_out,
            "job_stream": stream_out,
            "multi_gpu_verification": verification,
            "multi_gpu_per_rank": per_rank,
            "secondary": secondary if legs else None,
            "bench_wall_s": time.time() - T_START,
        }
        print(compact_line(out, write_detail(out, args, d)), flush=True)
    if args.cleanup and rank == 0:
        import shutil

        shutil.rmtree(d, ignore_errors=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
