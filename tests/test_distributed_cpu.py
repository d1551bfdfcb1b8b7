"""world_size-2 gloo test of the N > 1 path (reads shard by index, one all-reduce of the per-site
counters, clamp at 63) with the CPU oracle standing in for the per-rank worker."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, prefix, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from vargeno_amd import synth
    from vargeno_amd.api import all_reduce_sum_, clamp_counts, shard_range

    r = synth.f_tiny()[2]
    lo, hi = shard_range(r.n, rank, world)
    sub = r.slice(lo, hi)
    ox = O.OracleIndex.load(prefix)
    for _ in range(7):                              # 7x coverage so that the clamp matters
        ox.process(sub.bases, sub.quals, sub.offsets)
    s = ox.sites()
    t = torch.from_numpy(np.stack([s["ref_cnt"], s["alt_cnt"]], axis=1).astype(np.int32).reshape(-1).copy())
    all_reduce_sum_(t)
    if rank == 0:
        np.save(os.path.join(out_dir, "reduced.npy"), clamp_counts(t.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_reduce_equals_one_rank(ftiny_dir, tmp_path):
    from oracle import oracle as O
    from vargeno_amd import synth
    from vargeno_amd.api import shard_range

    assert [shard_range(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    assert shard_range(2, 3, 4) == (2, 2)
    prefix = os.path.join(ftiny_dir, "idx")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, prefix, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "reduced.npy").reshape(-1, 2)
    r = synth.f_tiny()[2]
    ox = O.OracleIndex.load(prefix)
    for _ in range(7):
        ox.process(r.bases, r.quals, r.offsets)
    s = ox.sites()
    assert np.array_equal(got[:, 0], s["ref_cnt"]) and np.array_equal(got[:, 1], s["alt_cnt"])
    assert got.max() == 63                          # saturation really happened somewhere


def test_shard_arithmetic_at_eight_ranks_with_fewer_items_than_ranks():
    """The target is 8 GPUs: contiguous shards must tile [0, n) exactly for every n, including n < 8 (empty shards)."""
    from vargeno_amd.api import shard_range

    for n in (0, 1, 5, 7, 8, 9, 63, 64, 65, 1_000_003):
        cuts = [shard_range(n, r, 8) for r in range(8)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(cuts[:-1], cuts[1:]))
        sizes = [hi - lo for lo, hi in cuts]
        assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 0


def test_cli_range_cuts_at_eight_replicas(tmp_path):
    """`vargeno fqcuts <fastq> <n>` prints the byte offsets at which n replicas split a FASTQ file (every inner cut a record
    start, a line beginning with '@' whose next-but-one line begins with '+'): 8 ranges over files with more, as many and
    fewer records than ranges, quality lines that begin with '@', and a split point inside the last record."""
    import subprocess

    from conftest import BIN

    def cuts_of(text, n):
        f = tmp_path / "x.fq"
        f.write_bytes(text)
        out = subprocess.run([BIN, "fqcuts", str(f), str(n)], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        return [int(x) for x in out.stdout.split()]

    rec = lambda i, q=b"I": b"@r%d\n" % i + b"ACGT" * 10 + b"\n+\n" + q * 40 + b"\n"
    for nrec in (100, 8, 3, 1, 0):
        text = b"".join(rec(i, b"@" if i % 3 == 0 else b"I") for i in range(nrec))
        c = cuts_of(text, 8)
        assert len(c) == 9 and c[0] == 0 and c[-1] == len(text) and c == sorted(c)
        starts = {0, len(text)}
        off = 0
        for i in range(nrec):
            starts.add(off)
            off += len(rec(i, b"@" if i % 3 == 0 else b"I"))
        assert all(x in starts for x in c), (nrec, c)
        if nrec >= 16:
            assert len(set(c)) == 9                                # every replica got a range


def test_the_all_reduce_of_eight_replicas_against_a_mock_of_eight_devices(tmp_path):
    """vg_counts_allreduce_devices's order of operations (vargeno_amd/csrc/vg_allreduce_plan.h: the product instantiates it with HIP
    + RCCL) run against a mock of the devices (tests/allreduce_mock.cpp, compiled here): 8 replicas on 8 devices -- the target's
    shape, which no test box has --, replicas sharing devices in every mix, one device only.  The mock insists on what RCCL
    does: one rank per device, every rank joining exactly once inside a group; afterwards every replica must hold the sum."""
    import subprocess

    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "allreduce_mock")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-o", exe, os.path.join(here, "allreduce_mock.cpp")])
    for devs, ranks in (([0, 1, 2, 3, 4, 5, 6, 7], 8), ([0, 0, 0, 0, 0, 0, 0, 0], 1), ([0, 1, 0, 1, 2, 2, 3, 0], 4), ([3], 1), ([7, 6, 5, 4, 3, 2, 1, 0], 8), ([0, 0, 1], 2), ([1, 0, 0, 0], 2)):
        p = subprocess.run([exe] + [str(d) for d in devs], capture_output=True, text=True)
        assert p.returncode == 0 and p.stdout.strip() == "ok %d" % ranks, (devs, p.stdout, p.stderr)
