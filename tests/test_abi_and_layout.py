"""CPU-only checks of the drop-in boundary: the shared library loads and exports exactly what
include/vargeno_hip.h declares; without a GPU every entry point fails with a code (never a fallback);
the product never touches oracle/."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from vargeno_amd import _lib


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "vargeno_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vg_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    syms = _header_symbols()
    assert sorted(_lib.SYMBOLS) == syms
    L = _lib.lib()
    for s in syms:
        assert hasattr(L, s), s


def test_ctypes_mirror_has_the_headers_struct_layout(tmp_path):
    """The Python side mirrors three structs of the C header by hand: sizes and field offsets must agree with what a C
    compiler makes of include/vargeno_hip.h."""
    import subprocess

    fields = {"vg_stats": list(_lib.STAT_FIELDS), "vg_timing": [f[0] for f in _lib.VgTiming._fields_],
              "vg_index_arrays": [f[0] for f in _lib.VgIndexArrays._fields_]}
    mirror = {"vg_stats": _lib.VgStats, "vg_timing": _lib.VgTiming, "vg_index_arrays": _lib.VgIndexArrays}
    src = ['#include <stddef.h>', '#include <stdio.h>', '#include "vargeno_hip.h"', "int main(void) {"]
    for st, fs in fields.items():
        src.append('printf("%s %%zu\\n", sizeof(%s));' % (st, st))
        for f in fs:
            src.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (st, f, st, f))
    src.append("return 0; }")
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = str(tmp_path / "layout")
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", exe, str(c)])
    got = dict(line.split() for line in subprocess.check_output([exe]).decode().splitlines())
    for st, fs in fields.items():
        assert int(got[st]) == C.sizeof(mirror[st]), st
        for f in fs:
            assert int(got["%s.%s" % (st, f)]) == getattr(mirror[st], f).offset, (st, f)


def test_header_cites_the_reference_interface_it_replaces():
    txt = open(os.path.join(ROOT, "include", "vargeno_hip.h")).read()
    for cite in ("qv.cc:519-695", "qv.cc:760-1558", "qv.cc:1775-1786", "generate_bf.h" if False else "dictgen.c:63-154"):
        assert cite in txt, cite


def test_no_gpu_means_error_codes_not_a_cpu_path(ftiny_dir):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible; the no-device behaviour is checked on CPU-only hosts")
    L = _lib.lib()
    assert L.vg_device_count() == 0
    h = C.c_void_p()
    rc = L.vg_index_open(os.fsencode(os.path.join(ftiny_dir, "idx")), 0, C.byref(h))
    assert rc == -4 and not h.value                           # VG_ENODEV
    assert b"no CPU fallback" in L.vg_last_error()
    assert L.vg_sync(None) == -1 and L.vg_counts_reset(None) == -1      # VG_EINVAL, no crash
    assert L.vg_index_views(None) == 0 and L.vg_reads_process_device_gated(None, None, None, None, 0) == -1
    from vargeno_amd.api import GenoIndex
    with pytest.raises(_lib.VgError):
        GenoIndex.open(os.path.join(ftiny_dir, "idx"))


def test_product_never_references_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "vargeno_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".c", "Makefile")):
                txt = open(os.path.join(base, fn), errors="replace").read()
                if re.search(r"\boracle\b|vg_oracle|liboracle|vgo_", txt):
                    bad.append(os.path.join(base, fn))
    assert not bad, bad
    inc = open(os.path.join(ROOT, "include", "vargeno_hip.h")).read()
    assert "oracle" not in inc
    # and nothing that runs on the GPU box reads /root/reference
    for fn in ("bench.py", "__graft_entry__.py"):
        txt = open(os.path.join(ROOT, fn)).read()
        for line in txt.splitlines():
            if "/root/reference" in line:
                assert "isdir" in line or "make" in line or line.strip().startswith("#"), (fn, line)


def test_bit_vector_file_whose_size_does_not_match_its_header_is_refused(ftiny_dir, tmp_path):
    """A corrupt .bf header (the bit count the kernels take the hash modulo of) must be refused by the loader before anything is
    sized from it -- on the host, so this needs no GPU."""
    import shutil

    from vargeno_amd._lib import VgError
    from vargeno_amd.api import GenoIndex

    for which, bits in (("snp", (1 << 64) - 5), ("snp", 0), ("ref", 1 << 40), ("snp", 1_120_000_064)):
        d = tmp_path / ("%s_%d" % (which, bits % 1000))
        d.mkdir()
        for fn in ("idx.ref.dict", "idx.snp.dict", "idx.chrlens"):
            shutil.copy(os.path.join(ftiny_dir, fn), d / fn)
        for w in ("ref", "snp"):
            src, dst = os.path.join(ftiny_dir, "idx.%s.bf" % w), d / ("idx.%s.bf" % w)
            os.link(src, dst) if w != which else None
        # a sparse copy of the right size with a wrong header
        size = os.path.getsize(os.path.join(ftiny_dir, "idx.%s.bf" % which))
        with open(d / ("idx.%s.bf" % which), "wb") as f:
            f.write(int(bits).to_bytes(8, "little"))
            f.truncate(size)
        with pytest.raises(VgError) as e:
            GenoIndex.open(str(d / "idx"))
        assert e.value.code == -2 and "size does not match its header" in str(e.value), (which, bits, str(e.value))


def test_python_binding_refuses_a_stale_library(monkeypatch):
    """_lib.lib() compares vg_build_id() with the hash of the sources next to the library."""
    assert _lib.lib().vg_build_id().decode() == _lib.source_build_id()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "source_build_id", lambda: "0123456789abcdef")
    with pytest.raises(RuntimeError, match="stale"):
        _lib.lib()
    monkeypatch.setattr(_lib, "_lib", None)


def test_gate_words_helper_matches_the_quality_strings():
    """api.gate_words (what bench.py and the GPU tests hand to vg_reads_process_device_gated): bit c of a read's word is set iff
    quality character c is below '8', for the read's chunk numbers c < len // 32 -- checked on CPU tensors against a plain loop."""
    import numpy as np
    import torch

    from vargeno_amd.api import gate_words

    rng = np.random.default_rng(7)
    lens = rng.integers(0, 300, size=500)
    lens[:4] = [0, 31, 32, 1022]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    quals = rng.integers(ord("#"), ord("I") + 1, size=int(offs[-1]) + 1, dtype=np.uint8)
    got = gate_words(torch.from_numpy(quals), torch.from_numpy(offs)).numpy().view(np.uint32)
    for r, L in enumerate(lens):
        want = 0
        for c in range(min(int(L) // 32, 32)):
            if quals[offs[r] + c] < ord("8"):
                want |= 1 << c
        assert int(got[r]) == want, (r, int(L))


def test_repeat_rich_genome_option_plants_exact_copies():
    """synth.genome_and_snps(repeats=F): about that fraction of the 32-mers of the genome occurs more than once."""
    import numpy as np

    from vargeno_amd import synth

    def dup_fraction(g):
        seq = np.concatenate(g.seqs)
        code = np.full(256, 4, np.uint8)
        for i, ch in enumerate(b"ACGT"):
            code[ch] = i
        c = code[seq].astype(np.uint64)
        k = np.zeros(len(c) - 31, np.uint64)
        for j in range(32):
            k |= c[j:len(c) - 31 + j] << np.uint64(2 * j)
        bad = np.convolve((c > 3).astype(np.int32), np.ones(32, np.int32), "valid") > 0
        k = k[~bad]
        _, counts = np.unique(k, return_counts=True)
        return float((counts[counts > 1]).sum()) / len(k)

    g0, _, _ = synth.genome_and_snps(genome_len=1_000_000, n_snps=1000, n_chroms=2)
    g3, _, _ = synth.genome_and_snps(genome_len=1_000_000, n_snps=1000, n_chroms=2, repeats=0.3)
    f0, f3 = dup_fraction(g0), dup_fraction(g3)
    assert f0 < 0.10 and 0.20 < f3 < 0.45 and f3 > 3 * f0, (f0, f3)


def test_the_device_memory_arena_against_a_mock_block(tmp_path):
    """vargeno_amd/csrc/vg_arena.h (one block per handle, permanent arrays from the bottom, temporaries from the top; the library
    instantiates it with hipMalloc / hipFree) against a mock block (tests/arena_mock.cpp, compiled here with AddressSanitizer):
    `fuzz`: 20 000 random takes and gives per seed under a shadow model (no overlap, alignment, no refusal while a gap could hold
    the request, the block returned exactly once); `replay`: the loader's own allocation sequence with the array sizes of
    BASELINE.json's configurations -- in a block the size of the FINISHED index every request has to find room: that is what the
    construction order of vg_index_open was chosen for."""
    import subprocess

    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "arena_mock")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-fsanitize=address,undefined", "-o", exe, os.path.join(here, "arena_mock.cpp")])
    p = subprocess.run([exe, "replay"], capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok") and p.stdout.count(" 0 requests without room") == 5, (p.stdout, p.stderr)
    for seed in (1, 2, 3, 4, 6):                       # (seeds 3 and 6: with a floor under the temporaries, what a budget-limited index sets)
        p = subprocess.run([exe, "fuzz", str(seed)], capture_output=True, text=True)
        assert p.returncode == 0 and p.stdout.strip() == "ok", (seed, p.stdout, p.stderr)


def test_bench_line_stays_small_enough_for_the_driver_to_parse():
    """BENCH_r05.parsed was null: the bench's one stdout line had grown to 25 KB.  The line is now a compact object built by
    bench.compact_line from the full record (which goes to a side file + stderr): here the round-5 record with every leg present,
    and a worst case with every string blown up -- both must stay under bench.LINE_LIMIT and keep the contract's keys."""
    import importlib.util
    import json
    import sys

    spec = importlib.util.spec_from_file_location("vg_bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    assert bench.LINE_LIMIT <= 8000
    full = json.load(open(os.path.join(ROOT, "profiles", "bench_default_r05.json")))
    assert len(json.dumps(full)) > 20000                       # (the record that broke the driver)

    def blow(x):
        if isinstance(x, str):
            return x * 20
        if isinstance(x, dict):
            return {k: blow(v) for k, v in x.items()}
        if isinstance(x, list):
            return [blow(v) for v in x]
        return x
    worst = blow(full)
    worst["metric"], worst["unit"] = full["metric"], full["unit"]
    for rec in (full, worst):
        text = bench.compact_line(rec, "/tmp/vg_bench/bench_detail.json")
        assert len(text) < bench.LINE_LIMIT and "\n" not in text
        line = json.loads(text)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in line, k
        assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
        assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"]) and len(line["cpu_baseline"]["sample"]) <= 200
        assert "workload" in line["config"] and "model" not in line["config"]
    small = json.loads(bench.compact_line(full, "x"))
    assert set(small["secondary"]) == set(full["secondary"]) and small["secondary"]["chr22"]["parity"] is True
    assert abs(small["value"] / full["value"] - 1) < 1e-5 and abs(small["roofline"]["frac"] / full["roofline"]["frac"] - 1) < 1e-3


@pytest.mark.parametrize("lend", [True, False])
def test_bench_fifo_feed_never_changes_bytes_the_reader_has_not_seen(lend):
    """bench.FifoFeed (the `job_stream` leg's writer) lends the pipe its own pages (vmsplice) and refills a buffer as soon as
    write_all has returned: a slow reader must still see every byte as it was written.  ONE buffer refilled at once (the bench
    rotates two: this is the stronger claim), odd sizes, a reader that sleeps; also the copying route (lend=False: what a refused vmsplice falls back to)."""
    import importlib.util
    import mmap
    import sys
    import threading
    import time

    import numpy as np

    spec = importlib.util.spec_from_file_location("vg_bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    r, w = os.pipe()
    feed = bench.FifoFeed(w, lend=lend)
    n = 5 * (1 << 20) + 12345
    buf = np.frombuffer(mmap.mmap(-1, n), dtype=np.uint8)
    rounds = 7
    got = []

    def reader():
        k = 0
        while True:
            b = os.read(r, 200_000 + 4096 * (k % 5))
            if not b:
                return
            got.append(b)
            k += 1
            if k % 4 == 0:
                time.sleep(0.03)                                  # (slow enough that a megabyte of lent pages is still in the pipe when the writer refills:
                                                                  #  without the copied tail this test fails)
    th = threading.Thread(target=reader)
    th.start()
    want = []
    base = np.arange(n, dtype=np.uint32)
    pats = [((base * (2 * k + 3) + k) >> 3).astype(np.uint8) for k in range(rounds)]
    for k in range(rounds):
        buf[:] = pats[k]
        feed.write_all(buf[: n - 17 * k])                        # (ends that are not page ends)
        want.append(pats[k][: n - 17 * k])
    os.close(w)
    th.join()
    os.close(r)
    assert b"".join(got) == b"".join(x.tobytes() for x in want)
    total = sum(len(x) for x in want)
    assert feed.lent_bytes + feed.copied_bytes == total
    if lend and feed.refusal is None:
        assert feed.lent_bytes > total // 2                       # (the pages were lent, the tails copied)
    elif not lend:
        assert feed.lent_bytes == 0
