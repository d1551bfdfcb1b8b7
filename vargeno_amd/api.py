"""Python host mirror of the `vargeno geno` read path over the C-ABI (include/vargeno_hip.h).

The reference is one C++ function (`genotype()`, src/qv.cc:475); its stages map to:
    loader              qv.cc:519-695    -> GenoIndex.open / GenoIndex.create
    FASTQ loop body     qv.cc:760-1558   -> GenoIndex.submit / GenoIndex.process_device
    counters to caller  qv.cc:1573-1626  -> GenoIndex.counts() (+ all_reduce_counts across ranks)
PyTorch appears only as plumbing (device buffers, torch.distributed); the computation is the HIP
library.  Nothing here can run without it.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import VgIndexArrays, VgStats, VgTiming, check, lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class GenoIndex:
    """One device-resident index replica + its pile-up counters (one per GPU / rank)."""

    def __init__(self, handle, device):
        self._h = handle
        self.device = device
        self._stores = []                 # read stores whose batches are in flight: they must outlive the next vg_sync

    # ---- construction -------------------------------------------------------------------
    @classmethod
    def open(cls, prefix, device=0, max_device_bytes=None, sharers=1):
        """max_device_bytes: this replica's device-memory budget (vg_index_open_ex); None: vg_index_open (the whole device) --
        or, with sharers > 1 replicas on this device, an equal share of it (vg_share_budget).  vg_share_budget divides what is free
        NOW by the number of sharers: it is only the same number for every sharer when all of them ask BEFORE any of them opens --
        one process that opens its replicas itself (the command line does).  Sharers in different processes must agree on one number
        first and pass it as max_device_bytes (bench.py: every rank asks, the minimum over the ranks is everybody's budget)."""
        h = C.c_void_p()
        if max_device_bytes is None and sharers > 1:
            max_device_bytes = int(lib().vg_share_budget(device, int(sharers))) or None
        if max_device_bytes is None:
            check(lib().vg_index_open(os.fsencode(prefix), device, C.byref(h)))
        else:
            check(lib().vg_index_open_ex(os.fsencode(prefix), device, int(max_device_bytes), C.byref(h)))
        return cls(h, device)

    @property
    def plan(self):
        """What the device-memory budget bought (vg_index_plan): views kept, views left out and what each costs."""
        try:
            return lib().vg_index_plan(self._h).decode()
        except AttributeError:                                   # an older build loaded for an A/B run (VARGENO_HIP_LIB)
            return ""

    @property
    def open_report(self):
        """Where the handle's start-up time went, phase by phase, and the memory it ended up with (vg_index_open_report)."""
        try:
            return lib().vg_index_open_report(self._h).decode()
        except AttributeError:                                   # an older build loaded for an A/B run (VARGENO_HIP_LIB)
            return ""

    @classmethod
    def create(cls, arrays, device=0):
        """arrays: dict as returned by vargeno_amd.index_io.read_index."""
        keep = {}

        def c(k, dt):
            keep[k] = np.ascontiguousarray(arrays[k], dtype=dt)
            return keep[k]

        a = VgIndexArrays()
        rk = c("ref_kmer", np.uint64)
        a.n_ref = len(rk)
        a.ref_kmer = rk.ctypes.data_as(_lib.u64p)
        a.ref_pos = c("ref_pos", np.uint32).ctypes.data_as(_lib.u32p)
        a.ref_amb = c("ref_amb", np.uint8).ctypes.data_as(_lib.u8p)
        rx = c("ref_aux", np.uint32)
        a.n_ref_aux = rx.size // 10
        a.ref_aux = rx.ctypes.data_as(_lib.u32p)
        sk = c("snp_kmer", np.uint64)
        a.n_snp = len(sk)
        a.snp_kmer = sk.ctypes.data_as(_lib.u64p)
        a.snp_pos = c("snp_pos", np.uint32).ctypes.data_as(_lib.u32p)
        a.snp_info = c("snp_info", np.uint8).ctypes.data_as(_lib.u8p)
        a.snp_amb = c("snp_amb", np.uint8).ctypes.data_as(_lib.u8p)
        a.snp_rf = c("snp_rf", np.uint8).ctypes.data_as(_lib.u8p)
        a.snp_af = c("snp_af", np.uint8).ctypes.data_as(_lib.u8p)
        sxp = c("snp_aux_pos", np.uint32)
        a.n_snp_aux = sxp.size // 10
        a.snp_aux_pos = sxp.ctypes.data_as(_lib.u32p)
        a.snp_aux_info = c("snp_aux_info", np.uint8).ctypes.data_as(_lib.u8p)
        a.ref_bf_bits = int(arrays["ref_bf_bits"])
        a.ref_bf_words = c("ref_bf_words", np.uint64).ctypes.data_as(_lib.u64p)
        a.snp_bf_bits = int(arrays["snp_bf_bits"])
        a.snp_bf_words = c("snp_bf_words", np.uint64).ctypes.data_as(_lib.u64p)
        h = C.c_void_p()
        check(lib().vg_index_create(C.byref(a), device, C.byref(h)))
        return cls(h, device)

    def close(self):
        if self._h:
            lib().vg_index_close(self._h)
            self._h = None
        self._pin = None
        self._stores = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the read loop -------------------------------------------------------------------
    def submit(self, bases, quals, offsets):
        """Host batch: flat uint8 ASCII bases / quality chars, uint64 offsets[n+1]."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        check(lib().vg_reads_submit(self._h, _ptr(bases), _ptr(quals), _ptr(offsets), len(offsets) - 1))

    def submit_fastq(self, text):
        """Raw FASTQ bytes (numpy uint8 / bytes): framed on the device.  Returns (records, bytes consumed, start of
        the last complete record)."""
        text = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else np.ascontiguousarray(text, dtype=np.uint8)
        n, used, last = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(lib().vg_fastq_submit(self._h, _ptr(text), len(text), C.byref(n), C.byref(used), C.byref(last)))
        return int(n.value), int(used.value), int(last.value)

    def submit_packed(self, kmers, meta, chunk_offsets):
        """Host batch that is already framed and 2-bit packed (see HostPacker): uint64 chunk k-mers, uint64 meta words, uint64
        chunk_offsets[n+1]."""
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        meta = np.ascontiguousarray(meta, dtype=np.uint64)
        chunk_offsets = np.ascontiguousarray(chunk_offsets, dtype=np.uint64)
        check(lib().vg_reads_submit_packed(self._h, _ptr(kmers), _ptr(meta), _ptr(chunk_offsets), len(chunk_offsets) - 1))

    def submit_store(self, store):
        """Every batch of a ReadStore (same device) through the read loop, in push order; asynchronous (sync / counts wait)."""
        store.flush()
        check(lib().vg_reads_submit_store(self._h, store._h))
        self._stores.append(store)        # (released by sync / counts / stats / close: the batches read the store's memory until then)

    def fastq_stream(self, chunks, host_threads=None):
        """FASTQ text as a stream of byte chunks cut anywhere (numpy uint8 arrays / bytes; pinned host memory copies at link
        speed): records are framed across the cuts -- on the device (host_threads None or 0), or framed and 2-bit packed by
        that many host threads inside the library (-1: the library picks).  Returns (records, bytes consumed, start of the
        last framed record, refused) once everything pushed has been processed."""
        if host_threads is None:
            check(lib().vg_fastq_stream_begin(self._h))
        else:
            check(lib().vg_fastq_stream_begin_packed(self._h, int(host_threads)))
        for ch in chunks:
            a = np.frombuffer(ch, dtype=np.uint8) if isinstance(ch, (bytes, bytearray, memoryview)) else np.ascontiguousarray(ch, dtype=np.uint8)
            check(lib().vg_fastq_stream_push(self._h, _ptr(a), len(a)))
        n, used, last, refused = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int()
        check(lib().vg_fastq_stream_end(self._h, C.byref(n), C.byref(used), C.byref(last), C.byref(refused)))
        return int(n.value), int(used.value), int(last.value), bool(refused.value)

    def process_device(self, d_bases, d_quals, d_offsets, n_reads):
        """Device-resident batch: torch CUDA tensors (uint8, uint8, int64/uint64 offsets[n+1])."""
        check(lib().vg_reads_process_device(self._h, C.c_void_p(d_bases.data_ptr()), C.c_void_p(d_quals.data_ptr()),
                                            C.c_void_p(d_offsets.data_ptr()), int(n_reads)))

    def process_device_gated(self, d_bases, d_gate_words, d_offsets, n_reads):
        """Device-resident batch whose quality strings are reduced to one gate word per read (int32/uint32 tensor; bit c = quality
        character c < '8', see gate_words())."""
        check(lib().vg_reads_process_device_gated(self._h, C.c_void_p(d_bases.data_ptr()), C.c_void_p(d_gate_words.data_ptr()),
                                                  C.c_void_p(d_offsets.data_ptr()), int(n_reads)))

    def sync(self):
        check(lib().vg_sync(self._h))
        self._stores = []

    def set_stats(self, enable):
        check(lib().vg_set_stats(self._h, 1 if enable else 0))

    def stats(self):
        s = VgStats()
        check(lib().vg_stats_get(self._h, C.byref(s)))
        return s.as_dict()

    def timing(self):
        t = VgTiming()
        check(lib().vg_timing_get(self._h, C.byref(t)))
        return dict(ms_total=float(t.ms_total), ms_pack=float(t.ms_pack), ms_main=float(t.ms_main), ms_tail=float(t.ms_tail), ms_deep_lists=float(t.ms_deep_lists), batches=int(t.batches))

    # ---- results -------------------------------------------------------------------------
    @property
    def num_sites(self):
        return int(lib().vg_num_sites(self._h))

    @property
    def device_bytes(self):
        return int(lib().vg_index_device_bytes(self._h))

    VIEW_NAMES = {1: "sec", 2: "mx", 4: "dx", 8: "snp_probe", 16: "snp_jg32", 32: "hx", 64: "snp_sig", 128: "sec_is_bf", 256: "ssec"}

    @property
    def views(self):
        """Names of the optional re-laid-out views this handle holds (vg_index_views): they change speed, never results."""
        m = int(lib().vg_index_views(self._h))
        return [n for b, n in sorted(self.VIEW_NAMES.items()) if m & b]

    def sites(self):
        n = self.num_sites
        pos = np.empty(n, np.uint32)
        u8 = [np.empty(n, np.uint8) for _ in range(4)]
        check(lib().vg_sites_fetch(self._h, _ptr(pos), *[_ptr(x) for x in u8]))
        return dict(pos=pos, ref_base=u8[0], alt_base=u8[1], ref_freq=u8[2], alt_freq=u8[3])

    def counts(self, copy=True):
        """Clamped counters (ref, alt).  copy=False: views of a page-locked buffer that the next call overwrites (the device-to-host
        copy then runs at link speed instead of through the runtime's pageable staging: bench.py's timed fetch)."""
        n = self.num_sites
        if copy:
            r, a = np.empty(n, np.uint8), np.empty(n, np.uint8)
            check(lib().vg_counts_fetch(self._h, _ptr(r), _ptr(a)))
            return r, a
        if getattr(self, "_pin", None) is None or len(self._pin[0]) != 2 * n:
            self._pin = pinned_buffer(max(2 * n, 1))
        buf = self._pin[0]
        check(lib().vg_counts_fetch(self._h, _ptr(buf[:n]), _ptr(buf[n:])))
        return buf[:n], buf[n:2 * n]

    def reset(self):
        check(lib().vg_counts_reset(self._h))

    def counts_device(self):
        p, n = C.c_void_p(), C.c_uint64()
        check(lib().vg_counts_device_ptr(self._h, C.byref(p), C.byref(n)))
        return int(p.value or 0), int(n.value)

    def counts_tensor(self):
        """The device counter array as a torch int32 tensor aliasing the library's memory (a u32 sum
        and an i32 sum are the same bits), for torch.distributed.all_reduce over RCCL."""
        import torch

        ptr, n = self.counts_device()

        class _Alias:
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 2, "strides": None}

        return torch.as_tensor(_Alias(), device="cuda:%d" % self.device)


class HostPacker:
    """The host-side FASTQ framing + 2-bit packing of the library on its own (vg_packer_*: no device involved)."""

    def __init__(self, threads=1):
        self._h = C.c_void_p()
        check(lib().vg_packer_create(int(threads), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().vg_packer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def begin(self):
        check(lib().vg_packer_begin(self._h))

    def push(self, text):
        """-> (kmers uint64[chunks], meta uint64[n], chunk_offsets uint64[n+1], n_invalid) of the complete records framed."""
        a = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray, memoryview)) else np.ascontiguousarray(text, dtype=np.uint8)
        rc_, kc_ = int(lib().vg_packer_reads_cap(len(a))), int(lib().vg_packer_kmers_cap(len(a)))
        kmers, meta, offs = np.zeros(kc_, np.uint64), np.zeros(rc_, np.uint64), np.zeros(rc_ + 1, np.uint64)
        n, nc, bad = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(lib().vg_packer_push(self._h, _ptr(a), len(a), _ptr(kmers), kc_, _ptr(meta), _ptr(offs), rc_, C.byref(n), C.byref(nc), C.byref(bad)))
        return kmers[:nc.value].copy(), meta[:n.value].copy(), offs[:n.value + 1].copy(), int(bad.value)

    def end(self):
        """-> (records, bytes consumed, start of the last framed record, refused)"""
        n, used, last, refused = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int()
        check(lib().vg_packer_end(self._h, C.byref(n), C.byref(used), C.byref(last), C.byref(refused)))
        return int(n.value), int(used.value), int(last.value), bool(refused.value)


class ReadStore:
    """Packed batches parked in device memory before an index handle exists (vg_read_store_*): what the command line packs while
    vg_index_open runs.  `GenoIndex.submit_store(store)` runs them through an open handle on the same device."""

    def __init__(self, device=0, max_bytes=1 << 30):
        self._h = C.c_void_p()
        self._keep = None
        check(lib().vg_read_store_create(int(device), int(max_bytes), C.byref(self._h)))

    def push(self, kmers, meta, chunk_offsets):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        meta = np.ascontiguousarray(meta, dtype=np.uint64)
        chunk_offsets = np.ascontiguousarray(chunk_offsets, dtype=np.uint64)
        check(lib().vg_read_store_push(self._h, _ptr(kmers), _ptr(meta), _ptr(chunk_offsets), len(chunk_offsets) - 1))
        self._keep = (kmers, meta, chunk_offsets)          # untouched until the next push / flush returns

    def flush(self):
        check(lib().vg_read_store_flush(self._h))
        self._keep = None

    @property
    def reads(self):
        return int(lib().vg_read_store_reads(self._h))

    @property
    def bytes_used(self):
        return int(lib().vg_read_store_bytes_used(self._h))

    def close(self):
        if self._h:
            lib().vg_read_store_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pinned_buffer(nbytes):
    """Page-locked host memory: (numpy uint8 view, owner).  Keep `owner` alive as long as the view is used."""
    p = lib().vg_host_alloc_pinned(int(nbytes))
    if not p:
        raise MemoryError("vg_host_alloc_pinned(%d)" % nbytes)

    class _Owner:
        def __init__(self, ptr):
            self.ptr = ptr
            self.buf = (C.c_uint8 * int(nbytes)).from_address(ptr)

        def __del__(self):
            lib().vg_host_free_pinned(self.ptr)

    own = _Owner(p)
    return np.ctypeslib.as_array(own.buf), own


def all_reduce_devices(indexes):
    """One process, several devices (what `vargeno geno` does with VARGENO_GPUS=n): RCCL communicator over the handles'
    devices + one grouped all-reduce of their counters, inside the library."""
    arr = (C.c_void_p * len(indexes))(*[ix._h for ix in indexes])
    check(lib().vg_counts_allreduce_devices(arr, len(indexes)))


def all_reduce_sum_(tensor, group=None):
    """In-place sum over ranks of an integer tensor (device or host); no-op for a single process."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
    return tensor


def clamp_counts(summed):
    """The reference's 6-bit saturation (vartype.h:27) commutes with the sum over shards:
    min(63, sum_i c_i) == min(63, sum_i min(63, c_i)), so ranks may exchange exact or clamped sums."""
    return np.minimum(np.asarray(summed, dtype=np.int64), 63).astype(np.uint8)


def all_reduce_counts(index, group=None):
    """The path's one exchange step: sum the per-site counters over the ranks that each processed a
    shard of the reads.  RCCL (backend 'nccl') over xGMI on GPUs.

    The collective runs on a torch-owned staging tensor, the library's array is copied into it and back (two device
    copies of 8 bytes per site): every rank issues exactly the same collective sequence whatever its allocator thinks
    of memory it does not own, so a refusal on one rank cannot desynchronise the job."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    index.sync()
    mine = index.counts_tensor()                       # alias of the library's u32 array (as int32: same bits under a sum)
    # RCCL reduces device memory; any other backend (gloo in the CPU / shared-GPU plumbing tests) gets a host copy
    on_device = dist.get_backend(group) == "nccl"
    stage = torch.empty_like(mine) if on_device else torch.empty(mine.shape, dtype=mine.dtype)
    stage.copy_(mine)
    all_reduce_sum_(stage, group)
    mine.copy_(stage)
    torch.cuda.synchronize(index.device)


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items for `rank` of `world` (reads shard by index, no exchange)."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gate_words(d_quals, d_offsets):
    """Gate words of a device-resident batch (torch tensors: uint8 quality characters, int64 offsets[n+1]): bit c of word r is set
    iff quality character c of read r is below '8' (qv.cc:836, 943), for the read's chunk numbers c < len // 32."""
    import torch

    off = d_offsets.to(torch.int64)
    n_chunks = ((off[1:] - off[:-1]) >> 5).clamp(max=32)
    out = torch.zeros(len(off) - 1, dtype=torch.int64, device=d_quals.device)
    if d_quals.numel() == 0:
        return out.to(torch.int32)
    sq = d_quals.view(torch.int8)                       # the path compares a signed char with '8' (qv.cc:836): bytes >= 0x80 are below it
    for c in range(int(n_chunks.max().item()) if len(n_chunks) else 0):
        live = n_chunks > c
        idx = torch.where(live, off[:-1] + c, torch.zeros_like(off[:-1]))
        low = (sq[idx].to(torch.int16) < 56) & live
        out |= low.to(torch.int64) << c
    return out.to(torch.int32) if out.numel() == 0 else (out & 0xFFFFFFFF).to(torch.int64).view(torch.int32)[::2].contiguous()
