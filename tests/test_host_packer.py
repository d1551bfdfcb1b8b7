"""CPU suite: the library's host-side FASTQ framing + 2-bit packing (vg_packer_*, vargeno_amd/csrc/vg_hostpack.cpp -- what
vg_fastq_stream_begin_packed runs inside the library) against a plain restatement of the reference's rules written here:
four lines per record (qv.cc:760-763), read = second line without its last character (qv.cc:778), trimmed to whole 32-base
chunks, encode_kmer (util.c:89-111), gate bit c = quality character c below '8' (qv.cc:836), N => skipped, other characters
=> the reference aborts (util.c:103).  No device is involved."""
import numpy as np
import pytest

from vargeno_amd.api import HostPacker

CODE = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3, ord("a"): 0, ord("c"): 1, ord("g"): 2, ord("t"): 3}
SKIP_N, INVALID = 1 << 62, 1 << 63


def expect_record(read, qual):
    """(kmers, meta) of one record by the rules above; read / qual are the lines WITHOUT their newline."""
    n = len(read) // 32
    kmers, meta = [], 0
    for c in range(n):
        k = 0
        for j, ch in enumerate(read[32 * c:32 * c + 32]):
            k |= CODE.get(ch, 0) << (2 * j)
        kmers.append(k)
    for c in range(n):
        q = qual[c]
        if (q - 256 if q >= 128 else q) < ord("8"):
            meta |= 1 << c
    flag = 0
    for c in range(n):                                     # first offender in the reference's scan order: chunk by chunk, base 31 down to 0
        for j in range(31, -1, -1):
            ch = read[32 * c + j]
            if ch not in CODE:
                flag = SKIP_N if ch in (ord("N"), ord("n")) else INVALID
                break
        if flag:
            break
    return kmers, meta | flag


def make_text(rng, n_rec, eol=b"\n", plus_id=False, weird=True):
    recs, want_k, want_m, want_n = [], [], [], []
    for i in range(n_rec):
        L = int(rng.integers(0, 400)) if rng.random() < 0.4 else int(rng.choice([31, 32, 33, 64, 127, 128, 150, 151, 160, 250]))
        read = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L)
        if weird and L and rng.random() < 0.2:
            read[rng.random(L) < 0.3] |= 0x20
        if weird and L and rng.random() < 0.1:
            p = int(rng.integers(0, L)); read[p:p + 3] = ord("N")
        if weird and L and rng.random() < 0.03:
            read[int(rng.integers(0, L))] = rng.choice(np.frombuffer(b"XR.-*", np.uint8))
        qual = np.where(rng.random(L) < 0.3, rng.integers(ord("#"), ord("8"), size=L), rng.integers(ord("8"), ord("J"), size=L)).astype(np.uint8)
        if weird and L and rng.random() < 0.02:
            qual[:4] = 0x90                                   # a byte >= 0x80 is a negative char: below '8'
        rb, qb = read.tobytes(), qual.tobytes()
        sep = b"+" + (b"r%d extra" % i if plus_id else b"")
        recs.append(b"@r%d some text" % i + eol + rb + eol + sep + eol + qb + eol)
        # what the path sees: the line up to (not including) its '\n' -- with CRLF the '\r' is part of the read and of the quality line
        line_r, line_q = rb + eol[:-1], qb + eol[:-1]
        k, m = expect_record(line_r, line_q)
        want_k += k; want_m.append(m); want_n.append(len(k))
    return b"".join(recs), np.array(want_k, np.uint64), np.array(want_m, np.uint64), np.array(want_n, np.int64)


@pytest.mark.parametrize("threads", [1, 3, 8])
@pytest.mark.parametrize("eol,plus_id", [(b"\n", False), (b"\r\n", False), (b"\n", True)])
def test_packer_matches_the_rules_whatever_the_cuts(threads, eol, plus_id):
    rng = np.random.default_rng(7 + threads)
    text, wk, wm, wn = make_text(rng, 5000, eol=eol, plus_id=plus_id)
    pk = HostPacker(threads)
    for cuts in ([len(text)], [1, 2, 3, 5, 100, 4097], sorted(rng.integers(0, len(text), size=40).tolist()), list(range(300000, len(text), 300000))):
        pk.begin()
        bounds = [0] + [c for c in cuts if 0 < c < len(text)] + [len(text)]
        ks, ms, ns, invalid = [], [], [], 0
        for a, b in zip(bounds[:-1], bounds[1:]):
            k, m, o, bad = pk.push(text[a:b])
            ks.append(k); ms.append(m); ns.append(np.diff(o.astype(np.int64))); invalid += bad
            assert o[0] == 0 and (len(o) == 1 or o[-1] == len(k))
        n, used, last, refused = pk.end()
        assert not refused and n == len(wm) and used == len(text)
        assert text[last:last + 2] == b"@r" and text[last:].count(b"\n") == 4
        assert np.array_equal(np.concatenate(ns), wn)
        assert np.array_equal(np.concatenate(ms), wm)
        live = np.repeat((wm >> np.uint64(62)) == 0, wn)          # the k-mers of a read that is skipped or aborted on are never looked at
        assert live.sum() > 10000 and np.array_equal(np.concatenate(ks)[live], wk[live])
        assert invalid == int(((wm >> np.uint64(63)) & np.uint64(1)).sum())
    pk.close()


def test_packer_leaves_an_unfinished_record_and_refuses_what_fgets_would_split():
    rng = np.random.default_rng(3)
    text, wk, wm, wn = make_text(rng, 200, weird=False)
    pk = HostPacker(4)
    # a truncated final record: everything before it is framed, `consumed` stops at its start
    cut = text.rfind(b"@r") + 20
    pk.begin()
    k, m, o, _ = pk.push(text[:cut])
    n, used, last, refused = pk.end()
    assert not refused and n == 199 and used == text.rfind(b"@r") and np.array_equal(m, wm[:199])
    # a line of 1023 characters + newline is beyond one fgets(buf, 1024): the chunk and everything after it are refused
    long_rec = b"@r\n" + b"A" * 1023 + b"\n+\n" + b"I" * 1023 + b"\n"
    ok_rec = b"@r\n" + b"A" * 1022 + b"\n+\n" + b"I" * 1022 + b"\n"
    pk.begin()
    k, m, o, _ = pk.push(text + ok_rec)
    assert len(m) == 201 and int(o[-1] - o[-2]) == 31
    k2, m2, o2, _ = pk.push(long_rec + text)
    assert len(m2) == 0
    k3, m3, o3, _ = pk.push(text)
    assert len(m3) == 0
    n, used, last, refused = pk.end()
    assert refused and n == 201 and used == len(text) + len(ok_rec)
    # a quality line shorter than the read's chunk count would show the reference's stale buffer: refused
    pk.begin()
    k, m, o, _ = pk.push(b"@r\n" + b"ACGT" * 16 + b"\n+\nI\n")
    assert len(m) == 0 and pk.end()[3]
    # ... one character per chunk is enough
    pk.begin()
    k, m, o, _ = pk.push(b"@r\n" + b"ACGT" * 16 + b"\n+\n#I\n")
    assert len(m) == 1 and int(m[0]) == 1 and not pk.end()[3]
    pk.close()


def test_packer_is_clean_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """tests/packer_asan.cpp compiled with -fsanitize=address,undefined against both builds of the hot loops (AVX2 where the CPU has
    it, SSE forced, and the exact two-sweep framing): 60 random texts with malformed pieces, cut anywhere into exact-size heap
    chunks, 1 / 3 / 7 threads."""
    import os
    import subprocess

    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(here, "..", "vargeno_amd", "csrc")
    flags = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-march=x86-64-v2", "-I" + csrc]
    objs = []
    for src, extra in (("vg_hostpack.cpp", []), ("vg_hostpack_avx2.cpp", ["-mavx2", "-mbmi2"])):
        o = str(tmp_path / (src + ".o"))
        subprocess.check_call(["g++"] + flags + extra + ["-c", os.path.join(csrc, src), "-o", o])
        objs.append(o)
    exe = str(tmp_path / "packer_asan")
    subprocess.check_call(["g++"] + flags + [os.path.join(here, "packer_asan.cpp")] + objs + ["-o", exe, "-lpthread"])
    for env in ({}, {"VG_PACK_ISA": "sse"}, {"VG_PACK_TWO_SWEEPS": "1"}):
        p = subprocess.run([exe], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1", **env), capture_output=True, text=True)
        assert p.returncode == 0 and "ok" in p.stdout, (env, p.stdout[-500:], p.stderr[-2000:])
