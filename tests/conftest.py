import gzip
import os
import shutil
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


CSRC = os.path.join(ROOT, "vargeno_amd", "csrc")
# the files (in this order) whose sha256 csrc/Makefile compiles into the library / the command-line tool
LIB_SOURCES = [os.path.join(ROOT, "include", "vargeno_hip.h")] + [os.path.join(CSRC, f) for f in ("vg_device.h", "vg_wave.h", "vargeno_hip.hip", "vg_sort.hip", "vg_arena.h", "vg_hostpack.h", "vg_hostpack.cpp", "vg_allreduce_plan.h", "vg_hostpack_impl.h", "vg_hostpack_impl.inc", "vg_hostpack_avx2.cpp")]
HOST_SOURCES = [os.path.join(ROOT, "include", "vargeno_hip.h"), os.path.join(CSRC, "host", "vg_host.h")] + [
    os.path.join(CSRC, "host", f) for f in ("main.cpp", "index_build.cpp", "fastq.cpp", "caller_vcf.cpp")]


def source_id(files):
    import hashlib

    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


_host_binary_checked = []


def host_binary():
    """Path of the `vargeno` command-line tool.  The binary is a build product (not in git; it travels to the GPU box next
    to its sources), so it is only trusted if the build id compiled into it is the hash of the sources as they are now --
    for the tool itself and for the HIP library it links."""
    import subprocess

    if os.environ.get("VARGENO_TEST_BIN"):                           # e.g. csrc/vargeno_asan (`make asan`): the CPU tests of the
        return os.environ["VARGENO_TEST_BIN"]                        # host tools under AddressSanitizer + UBSan
    path = os.path.join(CSRC, "vargeno")
    if not _host_binary_checked:
        assert os.path.exists(path), "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % path
        out = subprocess.run([path, "version"], capture_output=True, text=True)
        ids = dict(ln.split() for ln in out.stdout.splitlines() if len(ln.split()) == 2)
        assert ids.get("host") == source_id(HOST_SOURCES), "stale %s (built from other host sources: %r): rebuild with __graft_entry__.build()" % (path, ids)
        assert ids.get("lib") == source_id(LIB_SOURCES), "stale libvargeno_hip.so (built from other sources: %r): rebuild with __graft_entry__.build()" % (ids,)
        _host_binary_checked.append(True)
    return path


class _HostBinary(os.PathLike):
    """`subprocess.run([BIN, ...])`: resolves to the verified tool the first time it is used."""

    def __fspath__(self):
        return host_binary()

    def __str__(self):
        return host_binary()


BIN = _HostBinary()


# jobs started by one test and finished by a later one of the same session (the reference binary at hg38 scale: test_gpu_cli.py
# starts it, test_gpu_zz_hg38_reference.py compares); whatever is left at the end of the session is killed
BACKGROUND = {}


def pytest_sessionfinish(session, exitstatus):
    for job in BACKGROUND.values():
        p = job.get("ref")
        if p is not None and p.poll() is None:
            p.kill()


# ---- the container's memory -------------------------------------------------------------------------------------------------
# A GPU box of the pool is LOST when its container fills its memory limit (cgroup memory.max, 300 GiB there: tmpfs and the page
# cache count) -- round 5 lost one to bench.py, round 6 one to this suite (hg38 + full-dbSNP index files in /dev/shm, the index
# builder's arrays, the page cache of the hg38 index files and a 60 GB reference process at the same time).  /proc/meminfo shows
# the HOST's memory, not the container's: the tests that need tens of GB ask cgroup_room() and skip, and a watchdog thread ends the
# session (exit status 86, a line on stderr) before the limit is reached -- a failed suite instead of a lost box.
def _cg(path):
    try:
        v = open(path).read().split()[0]
        return None if v == "max" else int(v)
    except Exception:
        return None


def cgroup_room():
    """Bytes the container may still take (memory.max - memory.current), or None when there is no limit to read."""
    mx, cur = _cg("/sys/fs/cgroup/memory.max"), _cg("/sys/fs/cgroup/memory.current")
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
    """The page cache of the (written-back) files under `directory` given up: it is charged to the container like everything else."""
    for root, _, files in os.walk(directory):
        for fn in files:
            try:
                fd = os.open(os.path.join(root, fn), os.O_RDONLY)
                try:
                    os.fsync(fd)
                    os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
                finally:
                    os.close(fd)
            except OSError:
                pass


def finish_background(timeout=900):
    """Wait for jobs a test left running beside the suite (the reference binary at hg38 scale: 60 GB) before a test that needs the memory."""
    import time

    t0 = time.time()
    for job in BACKGROUND.values():
        p = job.get("ref")
        while p is not None and p.poll() is None and time.time() - t0 < timeout:
            time.sleep(1.0)


@pytest.fixture(autouse=True)
def _big_files_do_not_pile_up(request):
    """Every index `vargeno index` writes carries 1.3 GB of bit-vector files whatever the genome; pytest keeps a session's tmp_path
    directories until the session ends, and some thirty tests build an index -- on a 79 GB root next to the 48 GB of hg38 index
    files.  What a test leaves in its tmp_path beyond 32 MB per file goes when the test is over."""
    yield
    if "tmp_path" in request.fixturenames:
        try:
            d = str(request.getfixturevalue("tmp_path"))
        except Exception:
            return
        for root, _, files in os.walk(d):
            for fn in files:
                fp = os.path.join(root, fn)
                try:
                    if not os.path.islink(fp) and os.path.getsize(fp) > (32 << 20):
                        os.remove(fp)
                except OSError:
                    pass


@pytest.fixture(scope="session", autouse=True)
def _memory_watchdog():
    import threading
    import time

    mx = _cg("/sys/fs/cgroup/memory.max")
    stop = threading.Event()
    if mx is None or mx > (2 << 40):
        yield
        return

    def watch():
        warned = False
        while not stop.wait(0.5):
            room = cgroup_room()
            if room is None:
                continue
            cur = mx - room                                          # (what cannot be reclaimed: anonymous memory + tmpfs)
            if cur > 0.85 * mx and not warned:
                warned = True
                sys.stderr.write("\n[memory watchdog] %.0f of %.0f GB in use: dropping the page cache of /tmp/vg_bench\n" % (cur / 1e9, mx / 1e9))
                drop_file_cache("/tmp/vg_bench")
            try:                                                     # (the root file system: 79 GB on the pool's boxes, and a full one costs the box too)
                st = os.statvfs("/tmp")
                disk_left = st.f_bavail * st.f_frsize
            except OSError:
                disk_left = None
            if cur > 0.95 * mx or (disk_left is not None and disk_left < 3e9):
                sys.stderr.write("\n[memory watchdog] %.0f of %.0f GB of memory in use, %s GB of /tmp left: ending the session before the container is killed\n" % (cur / 1e9, mx / 1e9, "%.1f" % (disk_left / 1e9) if disk_left is not None else "?"))
                sys.stderr.flush()
                for job in BACKGROUND.values():
                    p = job.get("ref")
                    if p is not None and p.poll() is None:
                        p.kill()
                import shutil
                shutil.rmtree("/dev/shm/vg_bench", ignore_errors=True)
                os._exit(86)
    t = threading.Thread(target=watch, daemon=True)
    t.start()
    yield
    stop.set()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_context_first():
    """On a GPU box create torch's HIP context before the library's first big allocations: torch initialised late,
    after several 35 GB index open/close cycles in the same process, has been seen to report "No HIP GPUs are
    available" (torch only does plumbing in these tests: device buffers for vg_reads_process_device)."""
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda:0")
    except Exception:
        pass
    yield


def _gunzip(src, dst):
    with gzip.open(src, "rb") as f, open(dst, "wb") as g:
        shutil.copyfileobj(f, g)


@pytest.fixture(scope="session")
def ftiny_dir(tmp_path_factory):
    """The committed F-tiny fixture as files: inputs + the index the REFERENCE wrote for them
    (bit-vector files re-created sparse from the committed set-bit lists)."""
    from vargeno_amd import index_io

    d = str(tmp_path_factory.mktemp("ftiny"))
    for name in ("ref.fa", "snps.vcf", "reads.fq", "idx.ref.dict", "idx.snp.dict"):
        _gunzip(os.path.join(GOLDEN, "ftiny.%s.gz" % name), os.path.join(d, name))
    shutil.copy(os.path.join(GOLDEN, "ftiny.idx.chrlens"), os.path.join(d, "idx.chrlens"))
    z = np.load(os.path.join(GOLDEN, "ftiny.bf.npz"))
    index_io.write_bf_sparse(os.path.join(d, "idx.ref.bf"), int(z["ref_bits"]), z["ref_set"])
    index_io.write_bf_sparse(os.path.join(d, "idx.snp.bf"), int(z["snp_bits"]), z["snp_set"])
    _gunzip(os.path.join(GOLDEN, "ftiny.out.vcf.gz"), os.path.join(d, "golden.out.vcf"))
    return d


@pytest.fixture(scope="session")
def ftiny_reads():
    from vargeno_amd import synth

    return synth.f_tiny()[2]


def read_sha256_list(name):
    out = {}
    with open(os.path.join(GOLDEN, name + ".sha256")) as f:
        for line in f:
            h, fn = line.split()
            out[fn] = h
    return out


def bench_dir(need_gb=0):
    """Directory for index files shared with bench.py (built once per box): $VG_BENCH_DIR, else /tmp/vg_bench -- unless it
    lacks the room a big index needs (hg38 + full dbSNP: ~110 GB of files) and /dev/shm (memory-backed) has it."""
    d = os.environ.get("VG_BENCH_DIR")
    if d:
        return d
    d = "/tmp/vg_bench"
    if need_gb:
        def free_gb(path):
            try:
                st = os.statvfs(path)
                return st.f_bavail * st.f_frsize / 1e9
            except OSError:
                return 0.0
        if free_gb("/tmp") < need_gb and free_gb("/dev/shm") >= need_gb:
            d = "/dev/shm/vg_bench"
    return d
