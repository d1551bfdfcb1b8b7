#!/usr/bin/env python3
"""A/B of the once-only FASTQ route of `vargeno geno` (host/main.cpp, PipeIngest) and of the bench's feed (bench.FifoFeed) on ONE
box: the same FASTQ text (one batch of the default workload, kept by `VG_BENCH_KEEP_FASTQ=<path> bench.py ...`) is written into a
FIFO `repeat` times per variant, the command line reads it, the wall time and the feed's rate are printed as one JSON line per
variant; the VCFs of all variants of the same `repeat` must be identical.

    python3 profiles/pipe_ab.py <index dir (idx.*, snps.vcf)> <fastq> <repeat> name:KEY=VALUE,KEY=VALUE ...

KEY LEND=0|1 is the harness's (1: the pipe borrows the writer's pages, vmsplice; 0: write()); every other KEY goes into the command
line's environment (VARGENO_CHUNK_MB, VARGENO_PACK_THREADS, VARGENO_PREPACK ...)."""
import errno
import fcntl
import hashlib
import importlib.util
import json
import mmap
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.environ.get("VG_PIPE_AB_BIN") or os.path.join(ROOT, "vargeno_amd", "csrc", "vargeno")


def main():
    d, fq, repeat = sys.argv[1], sys.argv[2], int(sys.argv[3])
    spec = importlib.util.spec_from_file_location("vg_bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    n = os.path.getsize(fq)
    m = mmap.mmap(-1, n)
    with open(fq, "rb") as f:
        at = 0
        while at < n:
            got = f.readinto(memoryview(m)[at:at + (256 << 20)])
            if not got:
                break
            at += got
    head = memoryview(m)[:min(n, 1 << 20)].tobytes()     # (records of one length, as bench.py writes them: the first one's size)
    per, at = 0, -1
    for _ in range(4):
        at = head.index(b"\n", at + 1)
    per = at + 1
    assert n % per == 0, "records of different lengths: %d bytes, first record %d" % (n, per)
    reads = n // per
    shas = {}
    for spec_ in sys.argv[4:]:
        name, _, kv = spec_.partition(":")
        env = dict(os.environ, VARGENO_VERBOSE="1")
        lend = True
        for item in [x for x in kv.split(",") if x]:
            k, _, v = item.partition("=")
            if k == "LEND":
                lend = v != "0"
            else:
                env[k] = v
        fifo = os.path.join(d, "pipe_ab.fifo")
        out_vcf = os.path.join(d, "pipe_ab.%s.vcf" % name)
        for pth in (fifo, out_vcf):
            if os.path.exists(pth):
                os.remove(pth)
        os.mkfifo(fifo)
        time.sleep(float(os.environ.get("VG_PIPE_AB_IDLE_S", "8")))                                 # (the device idle between two command lines: freed memory is scrubbed)
        t0 = time.time()
        p = subprocess.Popen([BIN, "geno", "idx", fifo, "snps.vcf", out_vcf], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        fd = None
        while fd is None:
            try:
                fd = os.open(fifo, os.O_WRONLY | os.O_NONBLOCK)
            except OSError as e:
                if e.errno != errno.ENXIO or p.poll() is not None or time.time() - t0 > 60:
                    p.kill()
                    raise
                time.sleep(0.01)
        fcntl.fcntl(fd, fcntl.F_SETFL, fcntl.fcntl(fd, fcntl.F_GETFL) & ~os.O_NONBLOCK)
        feed = bench.FifoFeed(fd, lend=lend)
        tf = time.time()
        err = None
        try:
            for _ in range(repeat):
                feed.write_all(memoryview(m)[:n])
        except Exception as e:
            err = repr(e)
        os.close(fd)
        feed_s = time.time() - tf
        try:
            so, se = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            so, se = p.communicate()
        wall = time.time() - t0
        rec = {"variant": name, "settings": kv, "reads": reads * repeat, "text_GB": n * repeat / 1e9, "wall_s": round(wall, 2), "feed_s": round(feed_s, 2), "feed_GB_per_s": round(n * repeat / 1e9 / max(feed_s, 1e-9), 2),
               "lent_GB": round(feed.lent_bytes / 1e9, 1), "copied_GB": round(feed.copied_bytes / 1e9, 1), "refusal": feed.refusal, "pipe_bytes": feed.pipe_bytes, "rc": p.returncode, "writer_error": err}
        mt = re.search(r"wall: ([\d.]+) s = index load ([\d.]+) \+ FASTQ->counters ([\d.]+) \(([\d.]+) M reads/s\) \+ call/VCF ([\d.]+)", se or "")
        if mt:
            rec.update({"index_open_s": float(mt.group(2)), "ingest_after_open_s": float(mt.group(3)), "call_vcf_s": float(mt.group(5)), "reads_per_s_whole_job": round(reads * repeat / wall)})
        for ln in (se or "").splitlines():
            if ln.startswith("ingest, replica 0:"):
                rec["ingest_route"] = ln[len("ingest, replica 0:"):].strip()[:300]
        if p.returncode == 0 and os.path.exists(out_vcf):
            rec["vcf_sha256_16"] = hashlib.sha256(open(out_vcf, "rb").read()).hexdigest()[:16]
            shas[name] = rec["vcf_sha256_16"]
            os.remove(out_vcf)
        else:
            rec["stderr_tail"] = (se or "")[-400:]
        os.remove(fifo)
        print(json.dumps(rec), flush=True)
    print(json.dumps({"vcfs_identical": len(set(shas.values())) <= 1, "variants_with_a_vcf": len(shas)}), flush=True)


if __name__ == "__main__":
    main()
