"""ctypes binding of libvargeno_hip.so (include/vargeno_hip.h).  No fallback of any kind: if the
library is missing or the machine has no HIP device, the calls raise."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# VARGENO_HIP_LIB: development aid -- load another build of the same library (A/B runs of kernel variants, profiles/*.sh)
LIB_PATH = os.environ.get("VARGENO_HIP_LIB") or os.path.join(HERE, "csrc", "libvargeno_hip.so")

VG_ERRORS = {-1: "VG_EINVAL", -2: "VG_EIO", -3: "VG_ENOMEM", -4: "VG_ENODEV", -5: "VG_ETOOBIG", -6: "VG_EBADREAD"}

STAT_FIELDS = ["reads", "reads_n", "reads_invalid", "passes", "passes_ok", "chunks", "gate_open",
               "refbf_pos", "snpbf_pos", "large_block", "ref_query", "snp_query", "ref_probe", "snp_probe",
               "scan_ref", "scan_snp", "scan_oob", "aux_ref", "aux_snp", "site_test", "ctx", "walks", "incr",
               "ingest_bytes", "overflow_reads", "overflow_deep", "alg_bytes"]

# every symbol include/vargeno_hip.h declares (tests/test_abi.py checks the header against this list and the .so)
SYMBOLS = ["vg_last_error", "vg_build_id", "vg_device_count", "vg_device_memory", "vg_share_budget", "vg_link_rate", "vg_host_alloc_pinned", "vg_host_free_pinned", "vg_index_open", "vg_index_open_ex", "vg_index_plan", "vg_index_open_report", "vg_index_create", "vg_index_close",
           "vg_index_device_bytes", "vg_index_views", "vg_reads_submit", "vg_reads_process_device", "vg_reads_process_device_gated", "vg_reads_submit_packed", "vg_reads_submit_packed_async", "vg_read_store_create", "vg_read_store_push", "vg_read_store_flush", "vg_read_store_reads", "vg_read_store_bytes_used", "vg_reads_submit_store", "vg_read_store_destroy", "vg_fastq_submit", "vg_fastq_stream_begin", "vg_fastq_stream_begin_packed", "vg_fastq_stream_push", "vg_fastq_stream_end",
           "vg_packer_create", "vg_packer_destroy", "vg_packer_begin", "vg_packer_reads_cap", "vg_packer_kmers_cap", "vg_packer_push", "vg_packer_end", "vg_sync", "vg_stats_get",
           "vg_set_stats", "vg_timing_get", "vg_num_sites", "vg_sites_fetch", "vg_counts_fetch", "vg_counts_reset",
           "vg_counts_device_ptr", "vg_counts_allreduce", "vg_counts_allreduce_devices"]


class VgStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in STAT_FIELDS]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in STAT_FIELDS}


class VgTiming(C.Structure):
    _fields_ = [("ms_total", C.c_float), ("ms_pack", C.c_float), ("ms_main", C.c_float), ("ms_tail", C.c_float), ("batches", C.c_uint32), ("ms_deep_lists", C.c_float)]


u64p, u32p, u8p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)


class VgIndexArrays(C.Structure):
    _fields_ = [("n_ref", C.c_uint64), ("ref_kmer", u64p), ("ref_pos", u32p), ("ref_amb", u8p),
                ("n_ref_aux", C.c_uint64), ("ref_aux", u32p),
                ("n_snp", C.c_uint64), ("snp_kmer", u64p), ("snp_pos", u32p),
                ("snp_info", u8p), ("snp_amb", u8p), ("snp_rf", u8p), ("snp_af", u8p),
                ("n_snp_aux", C.c_uint64), ("snp_aux_pos", u32p), ("snp_aux_info", u8p),
                ("ref_bf_bits", C.c_uint64), ("ref_bf_words", u64p),
                ("snp_bf_bits", C.c_uint64), ("snp_bf_words", u64p)]


class VgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (VG_ERRORS.get(code, "VG_E?"), code, msg))
        self.code = code


_lib = None


def lib():
    """Load the HIP library; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.vg_last_error.restype = C.c_char_p
        L.vg_build_id.restype = C.c_char_p
        L.vg_device_count.restype = C.c_int
        try:
            L.vg_device_memory.argtypes = [C.c_int]
            L.vg_device_memory.restype = C.c_uint64
            L.vg_share_budget.argtypes = [C.c_int, C.c_int]
            L.vg_share_budget.restype = C.c_uint64
            L.vg_link_rate.argtypes = [C.c_int]
            L.vg_link_rate.restype = C.c_double
        except AttributeError:
            if not os.environ.get("VARGENO_HIP_LIB"):
                raise
        L.vg_index_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
        L.vg_index_create.argtypes = [C.POINTER(VgIndexArrays), C.c_int, C.POINTER(vp)]
        L.vg_index_close.argtypes = [vp]
        L.vg_index_close.restype = None
        L.vg_index_device_bytes.argtypes = [vp]
        L.vg_index_device_bytes.restype = C.c_uint64
        L.vg_index_views.argtypes = [vp]
        L.vg_index_views.restype = C.c_uint32
        L.vg_reads_submit.argtypes = [vp, vp, vp, vp, C.c_uint64]
        L.vg_reads_process_device.argtypes = [vp, vp, vp, vp, C.c_uint64]
        L.vg_reads_process_device_gated.argtypes = [vp, vp, vp, vp, C.c_uint64]
        L.vg_fastq_submit.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.vg_fastq_stream_begin.argtypes = [vp]
        try:
            L.vg_fastq_stream_begin_packed.argtypes = [vp, C.c_int]
            L.vg_reads_submit_packed.argtypes = [vp, vp, vp, vp, C.c_uint64]
            L.vg_reads_submit_packed_async.argtypes = [vp, vp, vp, vp, C.c_uint64]
            L.vg_read_store_create.argtypes = [C.c_int, C.c_uint64, C.POINTER(vp)]
            L.vg_read_store_push.argtypes = [vp, vp, vp, vp, C.c_uint64]
            L.vg_read_store_flush.argtypes = [vp]
            L.vg_read_store_reads.argtypes = [vp]
            L.vg_read_store_reads.restype = C.c_uint64
            L.vg_read_store_bytes_used.argtypes = [vp]
            L.vg_read_store_bytes_used.restype = C.c_uint64
            L.vg_reads_submit_store.argtypes = [vp, vp]
            L.vg_read_store_destroy.argtypes = [vp]
            L.vg_read_store_destroy.restype = None
            L.vg_packer_create.argtypes = [C.c_int, C.POINTER(vp)]
            L.vg_packer_destroy.argtypes = [vp]
            L.vg_packer_destroy.restype = None
            L.vg_packer_begin.argtypes = [vp]
            L.vg_packer_reads_cap.argtypes = [C.c_uint64]
            L.vg_packer_reads_cap.restype = C.c_uint64
            L.vg_packer_kmers_cap.argtypes = [C.c_uint64]
            L.vg_packer_kmers_cap.restype = C.c_uint64
            L.vg_packer_push.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, vp, vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
            L.vg_packer_end.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
            L.vg_index_open_ex.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.POINTER(vp)]
            L.vg_index_plan.argtypes = [vp]
            L.vg_index_plan.restype = C.c_char_p
            L.vg_index_open_report.argtypes = [vp]
            L.vg_index_open_report.restype = C.c_char_p
        except AttributeError:
            if not os.environ.get("VARGENO_HIP_LIB"):                   # (an older build loaded for an A/B run may lack the round-4 entry points)
                raise
        L.vg_fastq_stream_push.argtypes = [vp, vp, C.c_uint64]
        L.vg_fastq_stream_end.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        L.vg_host_alloc_pinned.argtypes = [C.c_size_t]
        L.vg_host_alloc_pinned.restype = vp
        L.vg_host_free_pinned.argtypes = [vp]
        L.vg_host_free_pinned.restype = None
        L.vg_sync.argtypes = [vp]
        L.vg_stats_get.argtypes = [vp, C.POINTER(VgStats)]
        L.vg_set_stats.argtypes = [vp, C.c_int]
        L.vg_timing_get.argtypes = [vp, C.POINTER(VgTiming)]
        L.vg_num_sites.argtypes = [vp]
        L.vg_num_sites.restype = C.c_uint64
        L.vg_sites_fetch.argtypes = [vp, vp, vp, vp, vp, vp]
        L.vg_counts_fetch.argtypes = [vp, vp, vp]
        L.vg_counts_reset.argtypes = [vp]
        L.vg_counts_device_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint64)]
        L.vg_counts_allreduce.argtypes = [vp, vp]
        L.vg_counts_allreduce_devices.argtypes = [C.POINTER(vp), C.c_int]
        # The library is a build product that travels next to its sources (not in git): refuse a stale one.  Its build id is the
        # sha256 of the sources it was compiled from (csrc/Makefile); VARGENO_HIP_LIB (A/B variants) opts out.
        if not os.environ.get("VARGENO_HIP_LIB"):
            want = source_build_id()
            got = L.vg_build_id().decode()
            if want is not None and got != want:
                raise RuntimeError("stale %s: built from sources with id %s, the sources next to it have id %s -- rebuild with "
                                   "`python -c 'import __graft_entry__ as g; g.build()'`" % (LIB_PATH, got, want))
        _lib = L
    return _lib


# the files (in this order) whose sha256 csrc/Makefile compiles into the library as vg_build_id()
LIB_SOURCES = [os.path.join(HERE, "..", "include", "vargeno_hip.h")] + [os.path.join(HERE, "csrc", f) for f in ("vg_device.h", "vg_wave.h", "vargeno_hip.hip", "vg_sort.hip", "vg_arena.h", "vg_hostpack.h", "vg_hostpack.cpp", "vg_allreduce_plan.h", "vg_hostpack_impl.h", "vg_hostpack_impl.inc", "vg_hostpack_avx2.cpp")]


def source_build_id():
    """Build id the library would have if compiled from the sources as they are now; None when they are not there."""
    import hashlib

    h = hashlib.sha256()
    try:
        for f in LIB_SOURCES:
            with open(f, "rb") as fh:
                h.update(fh.read())
    except OSError:
        return None
    return h.hexdigest()[:16]


def check(rc):
    if rc != 0:
        raise VgError(rc, lib().vg_last_error().decode(errors="replace"))
