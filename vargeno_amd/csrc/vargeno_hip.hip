// vargeno_hip.hip -- HIP kernels (gfx950) and the C-ABI of include/vargeno_hip.h.
//
// Path: the per-read loop of `vargeno geno` (reference src/qv.cc:760-1558): 2-bit encode of the
// 32-base chunks, exact ref/SNP dictionary lookups, quality-gated Hamming-1 neighbour search bounded
// by two bit vectors, order-dependent position vote, pile-up counter updates.
//
// Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared (see Makefile).  No CPU fallback exists:
// every entry point fails with VG_ENODEV when no HIP device is usable.
#include "../../include/vargeno_hip.h"
#include "vg_device.h"
#include "vg_wave.h"
#include "vg_hostpack.h"
#include "vg_allreduce_plan.h"
#include "vg_arena.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <chrono>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

using namespace vg;

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, const char *a = "", const char *b = "")
{
	snprintf(g_err, sizeof g_err, fmt, a, b);
	return code;
}
#define HIP_TRY(expr)                                                                              \
	do {                                                                                           \
		hipError_t e_ = (expr);                                                                    \
		if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? VG_ENOMEM : VG_ENODEV, "%s: %s", #expr, hipGetErrorString(e_)); \
	} while (0)

// No C++ exception may cross the C boundary: host allocations (new[], std::vector) inside an entry point are guarded.
template <class F>
static int guarded(F &&body)
{
	try { return body(); }
	catch (const std::bad_alloc &) { return fail(VG_ENOMEM, "host allocation failed"); }
	catch (...) { return fail(VG_EINVAL, "unexpected C++ exception"); }
}

extern "C" const char *vg_last_error(void) { return g_err; }
#ifndef VG_LIB_BUILD_ID
#define VG_LIB_BUILD_ID "unknown"
#endif
extern "C" const char *vg_build_id(void) { return VG_LIB_BUILD_ID; }

// page-locked host memory for callers without HIP headers (the CLI reads FASTQ chunks straight into it: H2D then runs
// at link speed instead of through the runtime's pageable staging)
extern "C" void *vg_host_alloc_pinned(size_t bytes)
{
	void *p = nullptr;
	if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
	return p;
}
extern "C" void vg_host_free_pinned(void *p) { if (p) (void)hipHostFree(p); }
extern "C" int vg_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}
extern "C" uint64_t vg_device_memory(int device)
{
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess) { (void)hipGetLastError(); return 0; }
	return (uint64_t)prop.totalGlobalMem;
}
// host -> device rate of page-locked memory over this device's link, bytes per second (0 on failure): two timed copies of 64 MiB
extern "C" double vg_link_rate(int device)
{
	if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return 0.0; }
	const size_t n = 64ull << 20;
	void *h = nullptr, *d = nullptr;
	hipStream_t s = nullptr;
	double best = 0.0;
	if (hipHostMalloc(&h, n, hipHostMallocDefault) == hipSuccess && hipMalloc(&d, n) == hipSuccess && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess) {
		memset(h, 1, n);
		for (int it = 0; it < 3; it++) {
			const auto t0 = std::chrono::steady_clock::now();
			if (hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) break;
			const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			if (it && dt > 0 && (double)n / dt > best) best = (double)n / dt;        // (the first copy warms the path up)
		}
	}
	(void)hipGetLastError();
	if (s) (void)hipStreamDestroy(s);
	if (d) (void)hipFree(d);
	if (h) (void)hipHostFree(h);
	return best;
}
extern "C" uint64_t vg_share_budget(int device, int replicas)
{
	uint64_t total = vg_device_memory(device);
	const uint64_t reserve = 12ull << 30;
	if (total == 0 || replicas < 1) return 0;
	// what is free NOW, when that is less (another process on the device, the read stores the command line has just taken)
	size_t fr = 0, tot = 0;
	if (hipSetDevice(device) == hipSuccess && hipMemGetInfo(&fr, &tot) == hipSuccess) total = std::min<uint64_t>(total, (uint64_t)fr);
	(void)hipGetLastError();
	return (total > reserve ? total - reserve : total) / (uint64_t)replicas;
}

// ------------------------------------------------------------------------------------------------
// kernels: index construction
// ------------------------------------------------------------------------------------------------
constexpr int JG_SPAN = 16384;                       // bucket ids per workgroup
constexpr int JG_LDS = JG_SPAN + JG_SPAN / 32;       // padded: one spare word per 32 keeps the per-thread runs off one bank

__device__ inline uint64_t lower_bound_dev(const uint64_t *a, uint64_t n, uint64_t key)
{
	uint64_t lo = 0, hi = n;
	while (lo < hi) { uint64_t m = lo + ((hi - lo) >> 1); if (a[m] < key) lo = m + 1; else hi = m; }
	return lo;
}

// jump table: jg[h] = number of entries whose (kmer >> SHIFT) < h  (= index of the first entry with
// HI >= h, = n past the last used HI: exactly what src/qv.cc:539-584 / :622-678 build).  One
// workgroup owns JG_SPAN consecutive h: LDS histogram of its slice of the sorted array, LDS scan,
// coalesced write.  jg has n_buckets + 1 entries; the last one is the sentinel n.
__global__ __launch_bounds__(256) void vg_build_jumpgate(const uint64_t *__restrict__ kmer, uint64_t n, uint32_t *__restrict__ jg, uint64_t n_buckets, const int SHIFT)
{
	__shared__ uint32_t hist[JG_LDS];
	__shared__ uint64_t range[2];
	__shared__ uint32_t wave_tot[4];
	const uint64_t h0 = (uint64_t)blockIdx.x * JG_SPAN;
	const uint32_t t = threadIdx.x;
	if (t < 2) {
		const uint64_t h = h0 + (t ? JG_SPAN : 0);
		range[t] = (h >= n_buckets) ? n : lower_bound_dev(kmer, n, h << SHIFT);
	}
	for (uint32_t i = t; i < JG_LDS; i += 256) hist[i] = 0;
	__syncthreads();
	const uint64_t e0 = range[0], e1 = range[1];
	for (uint64_t e = e0 + t; e < e1; e += 256) {
		const uint32_t b = (uint32_t)((kmer[e] >> SHIFT) - h0);
		atomicAdd(&hist[b + (b >> 5)], 1u);
	}
	__syncthreads();
	// thread t owns buckets [64t, 64t+64): serial exclusive scan in place, then a block scan of the 256 run totals
	uint32_t run = 0;
	for (uint32_t j = 0; j < 64; j++) {
		const uint32_t b = t * 64 + j, at = b + (b >> 5);
		const uint32_t v = hist[at];
		hist[at] = run;
		run += v;
	}
	uint32_t incl = run;
	const uint32_t lane = t & 63, wv = t >> 6;
	for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(incl, o); if ((int)lane >= o) incl += y; }
	if (lane == 63) wave_tot[wv] = incl;
	__syncthreads();
	uint32_t base = incl - run;
	for (uint32_t w = 0; w < wv; w++) base += wave_tot[w];
	__syncthreads();
	for (uint32_t j = 0; j < 64; j++) { const uint32_t b = t * 64 + j, at = b + (b >> 5); hist[at] += base; }
	__syncthreads();
	for (uint32_t i = t; i < JG_SPAN; i += 256) {
		const uint64_t h = h0 + i;
		if (h < n_buckets) jg[h] = (uint32_t)(e0 + hist[i + (i >> 5)]);
	}
	if (blockIdx.x == gridDim.x - 1 && t == 0) jg[n_buckets] = (uint32_t)n;
}

size_t vg_dev_sort_pairs_temp_bytes(size_t n);                                                          // vg_sort.hip
int vg_dev_sort_pairs_u64_u32(uint64_t *keys_a, uint64_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, size_t n, hipStream_t stream, void *tmp, size_t tmp_bytes, bool *result_in_b);

__global__ void vg_make_sec_keys(const uint64_t *__restrict__ kmer, uint64_t n, uint64_t *__restrict__ key, uint32_t *__restrict__ val)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t k = kmer[i];
		key[i] = (k << 32) | (k >> 32);            // LO32 major, HI32 minor
		val[i] = (uint32_t)i;
	}
}

// 12-byte records of the LO32-ordered view (DevIndex::sec3) from the sorted keys (LO32 << 32 | HI32) and the entries they name
__global__ void vg_make_sec3(const uint64_t *__restrict__ skey, const uint32_t *__restrict__ sidx, const RefEnt *__restrict__ ref, uint64_t n, uint32_t *__restrict__ out)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t k = skey[i];
		const RefEnt e = ref[sidx[i]];
		out[3 * i] = (uint32_t)k;
		out[3 * i + 1] = e.pos;
		out[3 * i + 2] = ((uint32_t)(k >> 32) & 0x7FFFFFFFu) | (e.amb ? 0x80000000u : 0u);
	}
}
// ... and of the SNP dictionary's (DevIndex::ssec3): the low word keeps 26 bits of LO32 and carries SNP_INFO_POS (where in the k-mer the SNP
// sits: a neighbour whose mutated base is the SNP base itself is not a hit, qv.cc:1308-1352) and the ambiguity flag
__global__ void vg_make_ssec3(const uint64_t *__restrict__ skey, const uint32_t *__restrict__ sidx, const SnpEnt *__restrict__ snp, uint64_t n, uint32_t *__restrict__ out)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t k = skey[i];
		const SnpEnt e = snp[sidx[i]];
		out[3 * i] = (uint32_t)k;
		out[3 * i + 1] = e.pos;
		out[3 * i + 2] = ((uint32_t)(k >> 32) & 0x03FFFFFFu) | ((uint32_t)((e.key >> 43) & 0x1Fu) << 26) | (((e.key >> 48) & 0xFFu) ? 0x80000000u : 0u);
	}
}
// Is the reference bit vector exactly the set of LO32 values of the dictionary (qv.cc:955 tests bit hash32(LO32), and hash32 is a
// bijection)?  out[0]: dictionary LO32 values whose bit is NOT set; out[1]: distinct LO32 values; out[2]: bits set in the vector.
__global__ void vg_sec_bf_check(const uint64_t *__restrict__ skey, uint64_t n, const uint64_t *__restrict__ bf, unsigned long long *__restrict__ out)
{
	unsigned long long miss = 0, distinct = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t lo = (uint32_t)(skey[i] >> 32);
		if (i == 0 || lo != (uint32_t)(skey[i - 1] >> 32)) {
			distinct++;
			const uint32_t h = hash32(lo);
			if (!((bf[h >> 6] >> (h & 63u)) & 1ull)) miss++;
		}
	}
	if (miss) atomicAdd(&out[0], miss);
	if (distinct) atomicAdd(&out[1], distinct);
}
__global__ void vg_popcount_words(const uint64_t *__restrict__ w, uint64_t n, unsigned long long *__restrict__ out)
{
	unsigned long long c = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) c += (unsigned long long)__popcll(w[i]);
	for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
	if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
// The merged view is keyed by the CANONICAL form of a k-mer (the smaller of it and its reverse complement, mixed so that the
// buckets fill evenly; r03): one look-up of a read's chunk then answers both strands -- the entry of the k-mer itself
// (a hit of the current pass) and the entry of its reverse complement (a hit of the OTHER pass, for the mirrored chunk).  A read of
// the reverse strand, whose forward pass finds nothing, then runs its one useful pass on the look-ups it has just made.
// strand bit of entry i (1: the dictionary's k-mer is the reverse complement of its canonical form)
__global__ void vg_canon_keys(uint64_t *__restrict__ key, uint64_t n, uint32_t *__restrict__ strand_bits)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t k = key[i], r = revcomp64(k), ck = k < r ? k : r;
		if (k != ck) atomicOr(&strand_bits[i >> 5], 1u << (i & 31));
		key[i] = fmix64(ck);
	}
}
// merged exact-match view: after the stable sort, val = index into the concatenation [ref | snp].  Position and ambiguity come
// from the dictionaries' 16-byte entries (r05: the columns they were unpacked from are long gone by now -- the construction keeps
// as little alive as it can, vg_arena.h).  The entry's last word carries HI32 of its key, i.e. its bucket, until the direct
// table has been built from it (vg_make_direct; vg_inline_pairs then puts the word to its real use).
// dx_bits < 32 (DevIndex::dx_bits): bits 16-31 of the flags word carry field F, the key's high-word bits below the bucket, left-aligned.
__global__ void vg_make_mx_entries(const uint64_t *__restrict__ key, const uint32_t *__restrict__ val, uint64_t n, uint64_t n_ref,
                                   const RefEnt *__restrict__ ref, const SnpEnt *__restrict__ snp, uint4 *__restrict__ out, const uint32_t *__restrict__ strand_bits, const uint32_t dx_bits)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t v = val[i];
		const bool is_snp = v >= n_ref;
		uint32_t pos, amb;
		if (is_snp) { const SnpEnt e = snp[v - n_ref]; pos = e.pos; amb = (uint32_t)(e.key >> 48) & 0xFFu; }
		else { const RefEnt e = ref[v]; pos = e.pos; amb = e.amb; }
		const uint32_t strand = (strand_bits[v >> 5] >> (v & 31)) & 1u;                          // flag 8: the dictionary's k-mer is the reverse complement of its canonical form
		const uint32_t hi = (uint32_t)(key[i] >> 32), F = dx_bits < 32u ? (hi << dx_bits) & 0xFFFF0000u : 0u;
		out[i] = make_uint4((uint32_t)key[i], pos, (is_snp ? 1u : 0u) | ((amb & 1u) << 1) | (strand << 3) | F, hi);
	}
}
// An ambiguous k-mer (2-10 copies) points at an auxiliary row; when the row holds exactly two positions -- the usual case --
// both fit the entry itself (flag PAIR: y = first position, w = second), and stage A needs no row gather for it.
__device__ inline bool aux_pair(const uint32_t *__restrict__ aux, uint32_t row, uint32_t &p0, uint32_t &p1)
{
	if (row == POS_AMBIGUOUS) return false;
	const uint32_t *r = aux + (uint64_t)row * AUX_COLS;
	p0 = r[0]; p1 = r[1];
	return p1 != 0 && r[2] == 0;
}
// direct table: the first entry of every HI32 bucket of the merged view, inline.  flags: 1 non-empty, 2 SNP entry, 4 ambiguous,
// 8 PAIR (single-entry buckets only: a longer bucket needs w for the index of its entries), 16 TIE (the second entry has the
// first one's k-mer: a query that matches the first entry of a bucket without it needs no further entry), bits 8.. = entries.
// r05: built from the entries themselves -- each carries its bucket in its last word (vg_make_mx_entries), the thread of a
// bucket's first entry writes the record, the table was zeroed before -- so that no 16 GiB jump table has to exist next to the
// 64 GiB table while it is filled: that was the one moment at which construction held more than the finished index.
__global__ void vg_make_direct(const uint4 *__restrict__ mx, uint64_t n, uint4 *__restrict__ dx,
                               const uint32_t *__restrict__ ref_aux, const uint32_t *__restrict__ snp_aux_pos, uint32_t *__restrict__ too_big, const uint32_t dx_bits)
{
	// dx_bits < 32: a bucket = the top dx_bits bits of the key's high word; the record's flags word carries the first entry's field F
	// (copied from the entry, bits 16-31) and an 8-bit count; TIE compares the whole of what is left of the key, (F, lo32)
	const uint32_t sh = 32u - dx_bits;
	const uint64_t cmax = dx_bits < 32u ? 0xFFull : 0xFFFFFFull;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint4 e = mx[i];
		const uint32_t bkt = e.w >> sh;
		if (i && (mx[i - 1].w >> sh) == bkt) continue;             // not the first of its bucket
		uint64_t hi = i + 1;
		while (hi < n && hi - i <= cmax && (mx[hi].w >> sh) == bkt) hi++;
		uint32_t cnt = (uint32_t)(hi - i);
		if (cnt > cmax) { atomicOr(too_big, 1u); cnt = (uint32_t)cmax; }        // the count field is 24 (8) bits wide: the host keeps the jump-table form (drops the view)
		uint4 r = make_uint4(e.x, e.y, 1u | ((e.z & 1u) << 1) | (((e.z >> 1) & 1u) << 2) | (((e.z >> 3) & 1u) << 5) | (cnt << 8) | (e.z & 0xFFFF0000u), (uint32_t)i);   // (flag 32: strand)
		if (cnt > 1u && mx[i + 1].x == e.x && mx[i + 1].w == e.w) r.z |= 16u;            // TIE: the second entry carries the same k-mer (reference + SNP dictionary)
		uint32_t p0, p1;
		if (cnt == 1u && (e.z & 2u) && aux_pair((e.z & 1u) ? snp_aux_pos : ref_aux, e.y, p0, p1)) { r.y = p0; r.w = p1; r.z |= 8u; }
		dx[bkt] = r;
	}
}
// the jump table of the merged view from the same bucket words (only when the direct table could not be kept after all)
__global__ void vg_jumpgate_from_buckets(const uint4 *__restrict__ mx, uint64_t n, uint32_t *__restrict__ jg)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t h1 = i < n ? (uint64_t)mx[i].w : (1ull << 32), h0 = i ? (uint64_t)mx[i - 1].w + 1 : 0ull;     // buckets (previous entry's, this entry's] start here
		for (uint64_t h = h0; h <= h1; h++) jg[h] = (uint32_t)i;
	}
}
// the same for the entries of the merged view themselves (read for buckets of several entries); mx flags: 1 SNP, 2 ambiguous, 4 PAIR
__global__ void vg_inline_pairs(uint4 *__restrict__ mx, uint64_t n, const uint32_t *__restrict__ ref_aux, const uint32_t *__restrict__ snp_aux_pos)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		uint4 e = mx[i];
		uint32_t p0, p1;
		e.w = 0u;                                                   // (the bucket word has done its job)
		if ((e.z & 2u) && aux_pair((e.z & 1u) ? snp_aux_pos : ref_aux, e.y, p0, p1)) { e.y = p0; e.w = p1; e.z |= 4u; }
		mx[i] = e;
	}
}
// strided-probe view of the SNP dictionary (DevIndex::snp_probe)
__global__ void vg_make_snp_probe(const uint64_t *__restrict__ kmer, const uint32_t *__restrict__ jg, uint64_t n, uint64_t *__restrict__ out)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (uint64_t)gridDim.x * blockDim.x) {
		if (i == n) { out[i] = 0ull; continue; }                    // (the kernel reads the view two entries at a time)
		const uint64_t slo = jg[kmer[i] >> 40];
		const uint64_t t = slo + (i - slo) * SNP_STRIDE;
		out[i] = t < n ? (kmer[t] & LO40_MASK) : 0ull;
	}
}
// paired HI32 table (DevIndex::hx), r05: from the dictionaries' entries, each of which carries its HI32 bucket in its spare word --
// no jump table is built for it (r03-r04 built two of 16 GiB each, one after the other, next to the 64 GiB table).  The table is
// zeroed first; the thread of a bucket's first entry writes the bucket's part of the record: the reference side writes whole
// records, the SNP side completes them.  A filter is computed over at most 64 entries; a longer bucket gets the mask that lets
// everything through.  Counts saturate at 0xFFFF: the bucket's end is then the NEXT record's start, which the same thread
// writes too (an empty bucket has no thread of its own; a non-empty one writes the same value).
__global__ void vg_hx_fill_ref(const RefEnt *__restrict__ ref, uint64_t n, uint4 *__restrict__ hx)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t h = ref[i].pad;
		if (i && ref[i - 1].pad == h) continue;
		uint64_t hi = i + 1;
		while (hi < n && ref[hi].pad == h) hi++;
		const uint64_t cnt = hi - i;
		uint32_t f = 0;
		if (cnt == 1u) f = hx_fp16(ref[i].lo);
		else if (cnt > 64u) f = 0xFFFFu;
		else for (uint64_t e = i; e < hi; e++) f |= hx_bit(ref[e].lo);
		hx[h] = make_uint4((uint32_t)i, 0u, cnt < 0xFFFFu ? (uint32_t)cnt : 0xFFFFu, f);
		if (cnt >= 0xFFFFu) atomicMax(&hx[(uint64_t)h + 1].x, (uint32_t)hi);      // (atomic: the next bucket's own thread may write its record at the same moment -- with this very start)
	}
}
__global__ void vg_hx_fill_snp(const SnpEnt *__restrict__ snp, uint64_t n, uint4 *__restrict__ hx)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t h = snp[i].pad;
		if (i && snp[i - 1].pad == h) continue;
		uint64_t hi = i + 1;
		while (hi < n && snp[hi].pad == h) hi++;
		const uint64_t cnt = hi - i;
		uint32_t f = 0;
		if (cnt == 1u) f = hx_fp16((uint32_t)snp[i].key);
		else if (cnt > 64u) f = 0xFFFFu;
		else for (uint64_t e = i; e < hi; e++) f |= hx_bit((uint32_t)snp[e].key);
		// (the reference side ran before: its words of the record stand)
		uint32_t *r = reinterpret_cast<uint32_t *>(hx + h);
		r[1] = (uint32_t)i;
		atomicOr(&r[2], (cnt < 0xFFFFu ? (uint32_t)cnt : 0xFFFFu) << 16);
		atomicOr(&r[3], f << 16);
		if (cnt >= 0xFFFFu) atomicMax(&hx[(uint64_t)h + 1].y, (uint32_t)hi);
	}
}
__global__ void vg_hx_sentinel(uint4 *__restrict__ hx, uint32_t n_ref, uint32_t n_snp) { hx[1ull << 32] = make_uint4(n_ref, n_snp, 0u, 0u); }
// signature form of the strided-probe view (DevIndex::snp_sig); 16 zero signatures behind the last entry (items read eight at a time)
__global__ void vg_make_snp_sig(const uint64_t *__restrict__ kmer, const uint32_t *__restrict__ jg, uint64_t n, uint16_t *__restrict__ out)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n + 16; i += (uint64_t)gridDim.x * blockDim.x) {
		if (i >= n) { out[i] = 0; continue; }
		const uint64_t slo = jg[kmer[i] >> 40];
		const uint64_t t = slo + (i - slo) * SNP_STRIDE;
		out[i] = (uint16_t)sig16(t < n ? (kmer[t] & LO40_MASK) : 0ull);
	}
}
__global__ void vg_iota_u32(uint32_t *v, uint64_t n) { for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) v[i] = (uint32_t)i; }

// SoA as the dictionary file has it -> one 16-byte entry per k-mer (a hit then costs one line)
__global__ void vg_make_ref_entries(const uint64_t *__restrict__ kmer, const uint32_t *__restrict__ pos, const uint8_t *__restrict__ amb, uint64_t n, RefEnt *__restrict__ out)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
		out[i] = RefEnt{(uint32_t)kmer[i], pos[i], (uint32_t)amb[i], (uint32_t)(kmer[i] >> 32)};       // (the spare word: HI32, the entry's bucket -- see vg_hx_fill_ref)
}
__global__ void vg_make_snp_entries(const uint64_t *__restrict__ kmer, const uint32_t *__restrict__ pos, const uint8_t *__restrict__ info, const uint8_t *__restrict__ amb, uint64_t n, SnpEnt *__restrict__ out)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
		out[i] = SnpEnt{(kmer[i] & LO40_MASK) | ((uint64_t)info[i] << 40) | ((uint64_t)amb[i] << 48), pos[i], (uint32_t)(kmer[i] >> 32)};
}

// ------------------------------------------------------------------------------------------------
// kernels: the read loop
// ------------------------------------------------------------------------------------------------

// Streaming pre-pass: ASCII -> chunk k-mers + one flag word per read (gate bits: chunk c is gate-open iff
// qual[c] < '8', src/qv.cc:836, 943 -- the chunk NUMBER indexes the quality string).  Chunk c of read r
// lands at pk_kmer[(offsets[r] >> 5) + c]; slots of different reads cannot collide.
//
// r05: the text is packed POSITION by position, not read by read.  A wave takes a tile of 64 consecutive reads: their bases are
// one contiguous span of text, which the lanes fetch as aligned 16-byte pieces (coalesced, PACK_G per lane in flight), pack to
// 32 bits each (pack16: v_perm_b32 + v_dot4_u32_u8, ~8 instructions per 4 bases) and lay side by side in LDS -- the tile's text
// as a 2-bit stream.  A read's k-mers are then 64-bit windows of that stream at bit offset 2 x (its byte offset): five LDS
// words and four v_alignbit_b32 per pair of chunks.  The r04 kernel staged the ASCII text in LDS and let every lane pack its own
// read with 64-bit SWAR arithmetic: ~1 500 vector instructions per tile, 147 VGPRs, three waves per SIMD -- its 0.38 ms per
// 8 M reads were 0.31 ms of vector issue (125 000 tiles x 1 500 instructions over 1 024 SIMDs at one per four cycles), not the
// memory system.  Pieces that hold a byte other than ACGTacgt are flagged per piece (one ballot per round of 64 pieces); only
// a read whose chunks touch a flagged piece -- an N run, in practice -- classifies its own text (classify_bad_wide).
// Workgroups are four INDEPENDENT waves (a CU takes at most 16 workgroups; with ~48 VGPRs the kernel wants 32 waves per CU):
// no barrier, wave-scope fences only.  Only the first n quality characters of a read are ever looked at.
constexpr uint32_t PACK_T = 64;                 // reads per tile = lanes of the wave that packs it
constexpr uint32_t PACK_WPB = 4;                // waves per workgroup
constexpr uint32_t PACK_MAXLEN = 160;           // tiles whose reads average at most this many bases go through LDS; longer reads take the direct path
constexpr uint32_t PACK_PIECES = PACK_T * PACK_MAXLEN / 16 + 1;      // aligned 16-byte pieces under a tile (one more than the span needs)
constexpr uint32_t PACK_ROUNDS = (PACK_PIECES + 63) / 64;
#ifndef VG_PACK_G
#define VG_PACK_G 5                             // pieces per lane in flight together (150 bp reads: two groups per tile)
#endif
#ifndef VG_PACK_NT
#define VG_PACK_NT 1                            // `nt` on the loads of the base text, read once
#endif
__global__ __launch_bounds__(PACK_T * PACK_WPB) void vg_pack_kernel(const uint8_t *__restrict__ bases, const uint8_t *__restrict__ quals, const uint64_t *__restrict__ offsets,
                                                      uint64_t n_reads_arg, uint64_t *__restrict__ pk_kmer, uint64_t *__restrict__ pk_meta, uint32_t *__restrict__ invalid_reads,
                                                      const uint32_t *__restrict__ n_reads_dev, const uint32_t *__restrict__ gate, const uint32_t TR)
{
	// TR: reads per tile (<= 64; the host picks it from the batch's mean read length so that a tile's text fits the LDS stream:
	// 64 for reads of up to 160 bases, 40 for 250 bp ...; lanes beyond TR only help to fetch and pack the text)
	// gate != nullptr: the batch comes with one gate word per read (bit c = quality character c < '8') instead of quality strings --
	// the kernel then streams the bases only.  (With strings it fetches one line per read for the <= 4 characters a 150 bp read's
	// gate can see: at a 150-byte stride that is every line of the quality array, nearly as much traffic again as the bases.)
	__shared__ uint32_t sm_pk[PACK_WPB][PACK_PIECES + 7];                 // the tile's text, 16 bases per word
	__shared__ unsigned long long sm_bad[PACK_WPB][PACK_ROUNDS + 1];      // bit p: piece p holds a byte other than ACGTacgt
	const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), ln = threadIdx.x & 63u;
	const uint64_t n_reads = n_reads_dev ? (uint64_t)*n_reads_dev : n_reads_arg;     // a batch framed on the device knows its size there
	const uint64_t n_tiles = (n_reads + TR - 1) / TR, stride = (uint64_t)gridDim.x * PACK_WPB;
	uint64_t t = (uint64_t)blockIdx.x * PACK_WPB + wv;
	// a tile's own offsets (and gate words) are fetched while the tile before it is packed: one dependent wait per tile, not two
	uint64_t off = 0, off1 = 0;
	uint32_t gw = 0;
	auto fetch = [&](uint64_t tile, uint64_t &o, uint64_t &o1, uint32_t &g) {
		const uint64_t r = tile * TR + ln;
		o = o1 = 0; g = 0;
		if (ln < TR && r < n_reads) { o = offsets[r]; o1 = offsets[r + 1]; if (gate) g = gate[r]; }
	};
	if (t < n_tiles) fetch(t, off, off1, gw);
	for (; t < n_tiles; t += stride) {
		const uint64_t r0 = t * TR;
		const uint64_t r = ln < TR ? r0 + ln : n_reads;                                                    // (lanes beyond the tile have no read)
		const uint32_t last = (uint32_t)((r0 + TR < n_reads ? r0 + TR : n_reads) - r0) - 1u;               // last live lane of the tile
		const uint64_t base0 = __shfl(off, 0), span = __shfl(off1, (int)last) - base0;
		uint64_t noff = 0, noff1 = 0;
		uint32_t ngw = 0;
		if (t + stride < n_tiles) fetch(t + stride, noff, noff1, ngw);
		const uint32_t n = (uint32_t)((off1 - off) >> 5);
		uint32_t q4 = 0;
		if (!gate && n && r < n_reads) __builtin_memcpy(&q4, quals + off, 4);           // 4 <= n + 3 <= the read's own length: never past it
		// aligned 16-byte pieces: a piece that holds a byte of the text lies in that byte's page, so the first and the last piece
		// may reach past the text (a batch may start at any byte)
		const uint64_t a_text = (uint64_t)(bases + base0);
		const uint32_t shift = (uint32_t)(a_text & 15u);
		const uint8_t *abase = (const uint8_t *)(a_text - shift);
		const bool staged = (uint64_t)shift + span <= (uint64_t)PACK_T * PACK_MAXLEN;
		int cls = 0;
		if (staged) {
			const uint32_t n_pieces = (uint32_t)__builtin_amdgcn_readfirstlane((int)((shift + (uint32_t)span + 15u) >> 4));
			bool tile_bad = false;                                        // wave-uniform
			constexpr uint32_t G = VG_PACK_G;
			for (uint32_t j0 = 0; j0 * 64u < n_pieces; j0 += G) {
				uint4 v[G];
				#pragma unroll
				for (uint32_t q = 0; q < G; q++) {
					const uint32_t p = (j0 + q) * 64u + ln;
					v[q] = make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);
					if (p < n_pieces) v[q] = load_policy<VG_PACK_NT != 0, uint4, 16>(abase + 16ull * p);
				}
				#pragma unroll
				for (uint32_t q = 0; q < G; q++) {
					const uint32_t p = (j0 + q) * 64u + ln;
					uint32_t br = 0;
#ifdef VG_PACK_DRY
					const uint32_t w = v[q].x ^ v[q].y ^ v[q].z ^ v[q].w;      // timing experiment only (wrong results): the kernel's traffic without its arithmetic
#else
					const uint32_t w = pack16(v[q], br);
#endif
					if (p < n_pieces) sm_pk[wv][p] = w;
					const unsigned long long bm = __ballot((br & PACK_CASE_MASK) != 0u);
					tile_bad = tile_bad || bm != 0ull;
					if (ln == 0 && (j0 + q) * 64u < n_pieces) sm_bad[wv][j0 + q] = bm;
				}
			}
			VG_WAVE_SYNC();
			if (r < n_reads && n) {
				const uint32_t o = shift + (uint32_t)(off - base0);          // byte offset of the read in the tile's pieces
				const uint32_t w0 = o >> 4, s = 2u * (o & 15u);
				uint64_t *dst = pk_kmer + (off >> 5);
				auto win = [&](uint32_t hi, uint32_t lo) -> uint32_t { return __builtin_amdgcn_alignbit(hi, lo, s); };
				uint32_t c = 0;
				for (; c + 2 <= n; c += 2) {                               // a read's k-mers are contiguous: 16-byte stores
					uint32_t d[5];
					__builtin_memcpy(d, &sm_pk[wv][w0 + 2u * c], 20);
					const ulonglong2 kv = make_ulonglong2((uint64_t)win(d[1], d[0]) | ((uint64_t)win(d[2], d[1]) << 32), (uint64_t)win(d[3], d[2]) | ((uint64_t)win(d[4], d[3]) << 32));
					__builtin_memcpy(dst + c, &kv, 16);
				}
				if (c < n) {
					uint32_t d[3];
					__builtin_memcpy(d, &sm_pk[wv][w0 + 2u * c], 12);
					dst[c] = (uint64_t)win(d[1], d[0]) | ((uint64_t)win(d[2], d[1]) << 32);
				}
				if (tile_bad) {
					// does one of the pieces under this read's chunks hold an offending byte?  (the flags are per piece: the bases a read's
					// trim drops, and its neighbours' bases, may be what was seen -- the read's own text decides)
					bool any = false;
					for (uint32_t p = w0, left = ((o + 32u * n - 1u) >> 4) - w0 + 1u; left;) {          // (a 150 bp read lies under 9 pieces: one or two words)
						const uint32_t b = p & 63u, take = 64u - b < left ? 64u - b : left;
						unsigned long long m = sm_bad[wv][p >> 6] >> b;
						if (take < 64u) m &= (1ull << take) - 1ull;
						any = any || m != 0ull;
						p += take; left -= take;
					}
					if (any) cls = classify_bad_wide(bases + off, n);
				}
			}
		} else if (r < n_reads) {
			uint64_t bad = 0;
			for (uint32_t c = 0; c < n; c++) pk_kmer[(off >> 5) + c] = encode32(bases + off + 32 * c, bad);
			if (bad) cls = classify_bad(bases + off, n);
		}
		if (r < n_reads) {
			// quality gate bits (qv.cc:836): character c of the quality line, four characters per gather (the first four are in hand)
			uint64_t meta = 0;
			if (gate) meta = n >= 32 ? gw : (gw & ((1u << n) - 1u));
			else for (uint32_t c0 = 0; c0 < n && c0 < 32; c0 += 4) {
				if (c0) __builtin_memcpy(&q4, quals + off + c0, 4);       // c0 + 4 <= n + 3 <= the read's own length
				for (uint32_t j = 0; j < 4 && c0 + j < n && c0 + j < 32; j++) if ((int)(int8_t)(q4 >> (8 * j)) - '8' < 0) meta |= 1ull << (c0 + j);
			}
			if (cls) meta |= cls == 1 ? PK_SKIP_N : PK_INVALID;
			if (n > 32) meta |= gate ? PK_INVALID : PK_LONG;            // (a gate word has 32 bits; the reference's line buffer admits 31 chunks)
			if (meta & PK_INVALID) atomicAdd(invalid_reads, 1u);       // the reference aborts on such a read (util.c:103): the caller is told
			pk_meta[r] = meta;
		}
		VG_WAVE_SYNC();                                                // the tile's LDS words are read: the next tile may overwrite them
		off = noff; off1 = noff1; gw = ngw;
	}
}

// base-indexed counters of the wave kernel -> the ref / alt sums (and zero them for the next round)
__global__ void vg_fold_counters(uint32_t *__restrict__ cnt4, const uint8_t *__restrict__ site_ba, uint32_t *__restrict__ cnt, uint64_t n_sites)
{
	for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_sites; s += (uint64_t)gridDim.x * blockDim.x) {
		uint4 v = ((const uint4 *)cnt4)[s];
		if ((v.x | v.y | v.z | v.w) == 0) continue;
		const uint32_t q[4] = {v.x, v.y, v.z, v.w};
		const uint32_t ba = site_ba[s];
		cnt[2 * s] += q[ba & 3u];
		cnt[2 * s + 1] += q[(ba >> 2) & 3u];
		((uint4 *)cnt4)[s] = make_uint4(0u, 0u, 0u, 0u);
	}
}

// MAX_COV saturation (src/vartype.h:27, qv.cc:1411, 1419) of the exact sums, on the way to the host: 2 bytes per site cross the link instead of 8
__global__ void vg_clamp_counters(const uint32_t *__restrict__ cnt, uint64_t n_sites, uint8_t *__restrict__ ref_cnt, uint8_t *__restrict__ alt_cnt)
{
	for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_sites; s += (uint64_t)gridDim.x * blockDim.x) {
		const uint2 v = ((const uint2 *)cnt)[s];
		ref_cnt[s] = (uint8_t)(v.x < 63u ? v.x : 63u);
		alt_cnt[s] = (uint8_t)(v.y < 63u ? v.y : 63u);
	}
}

// One lane = one read: forward pass, then the reverse-complement retry (src/qv.cc:1504-1510).
// A read touches the counters only at the end of its last pass, so a lane that runs out of scratch
// simply drops the read onto the overflow list and the same kernel re-runs it with a deep scratch.
// packed: the batch came 2-bit packed (vg_reads_submit_packed, the host-packed FASTQ stream): chunk k-mers and flag words
// are read from pk_kmer / pk_meta (the layout the pack kernel writes) instead of being encoded from text.
template <bool STATS>
__global__ __launch_bounds__(256) void vg_lane_kernel(DevIndex d, Scratch s, const uint8_t *__restrict__ bases, const uint8_t *__restrict__ quals,
                                                      const uint64_t *__restrict__ offsets, uint64_t n_reads_arg, const uint32_t *__restrict__ read_ids,
                                                      const uint32_t *__restrict__ n_ids, uint32_t *overflow_list, uint32_t *overflow_count, unsigned long long *stats, uint32_t *invalid_reads,
                                                      const uint32_t *__restrict__ gate, const uint64_t *__restrict__ pk_kmer, const uint64_t *__restrict__ pk_meta, const bool packed)
{
	const uint64_t n_reads = n_ids ? (uint64_t)*n_ids : n_reads_arg;          // a list launch (or a device-framed batch) is sized on the device: no host round trip
	const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t stride = gridDim.x * blockDim.x;
	Lane<STATS> L(d, s, gtid);
	LaneStats<STATS> tot;
	if constexpr (STATS) for (int i = 0; i < S_COUNT; i++) tot.v[i] = 0;
	// entry r of the work goes to lane r / n_waves of wave r % n_waves: a list of a dozen deep reads lands on a dozen waves,
	// not on twelve lanes of one wave that would run their divergent walks one after the other
	const uint32_t n_waves = stride >> 6;
	for (uint64_t r = (uint64_t)(gtid & 63u) * n_waves + (gtid >> 6); r < n_reads; r += stride) {
		const uint64_t rid = read_ids ? read_ids[r] : r;
		const uint64_t off = offsets[rid];
		const uint32_t n = (uint32_t)((offsets[rid + 1] - off) >> 5);            // src/qv.cc:778-779: len = (read_len/32)*32
		const uint8_t *p = bases + off;
		const uint64_t pmeta = packed ? pk_meta[rid] : 0ull;
		// src/qv.cc:836, 943: the chunk NUMBER indexes the quality string (which is NOT reversed for the second pass, qv.cc:786-806);
		// a batch may bring the comparison's results as one word per read instead of the strings (a read has at most 31 chunks)
		const uint32_t gw = packed ? (uint32_t)pmeta : gate ? gate[rid] : 0u;
		auto gate_open = [&](uint32_t c) -> bool { return (gate || packed) ? ((gw >> (c & 31u)) & 1u) != 0u : (int)(int8_t)quals[off + c] - '8' < 0; };
		auto kmer_at = [&](uint32_t c) -> uint64_t { uint64_t b2 = 0; return packed ? pk_kmer[(off >> 5) + c] : encode32(p + 32 * c, b2); };
		if constexpr (STATS) { for (int i = 0; i < S_COUNT; i++) L.st.v[i] = 0; }
		L.st.add(S_READS, 1);
		L.st.add(S_INGEST, 9 * n);
		L.overflow = false;

		int cls = 0;
		if (packed) cls = (pmeta & PK_INVALID) ? 2 : (pmeta & PK_SKIP_N) ? 1 : 0;
		else {
			uint64_t bad = 0;
			for (uint32_t c = 0; c < n; c++) (void)encode32(p + 32 * c, bad);
			if (bad) cls = classify_bad(p, n);
		}
		if (cls == 1) L.st.add(S_READS_N, 1);
		if (cls == 0 && (gate || packed) && n > 32u) cls = 2;                       // (a gate word has 32 bits: the pack kernel counts such a read invalid too)
		if (cls == 2) { L.st.add(S_READS_INVALID, 1); if (invalid_reads) atomicAdd(invalid_reads, 1u); }
		if (cls == 0) {
			bool ok = false;
			L.reset_pass();
			for (uint32_t c = 0; c < n && !L.overflow; c++) L.do_chunk(kmer_at(c), c, gate_open(c));
			if (!L.overflow) ok = L.finish_pass();
			if (!L.overflow && !ok) {
				L.reset_pass();
				for (uint32_t c = 0; c < n && !L.overflow; c++) L.do_chunk(revcomp64(kmer_at(n - 1 - c)), c, gate_open(c));
				if (!L.overflow) (void)L.finish_pass();
			}
		}
		if (L.overflow) {
			const uint32_t at = atomicAdd(overflow_count, 1u);
			overflow_list[at] = (uint32_t)rid;
		} else if constexpr (STATS) {
			for (int i = 0; i < S_COUNT; i++) tot.v[i] += L.st.v[i];
		}
	}
	if constexpr (STATS) {
		for (int i = 0; i < S_COUNT; i++) if (tot.v[i]) atomicAdd(&stats[i], (unsigned long long)tot.v[i]);
	}
}



// The reads the deep tier leaves behind (listB: more vote keys or neighbour contexts than its LDS tables hold) are finished by the
// lane machine, one read per lane with its lists in HBM: 6-12 ms for the handful of 250 bp reads per 8 M that reach it, during
// which the batch's slot -- and with three slots the main stream -- waited (r05: 6.9 ms per step for 5.1 ms of kernels).  Counters
// are sums, so WHEN such a read is finished does not matter: this kernel copies the packed form of those reads (chunk k-mers + flag
// word) out of the batch's slot into a small store that belongs to the handle, the slot is free at once, and the lane machine runs
// over the store ONCE, when the caller next synchronises (finish_pending).  One 64-bit atomic reserves a read's place and its
// chunks' (reads << 32 | chunks: both cursors move together, so the offsets stay a prefix sum); what does not fit -- or has more
// chunks than a flag word has gate bits -- goes to the residual list and takes the per-batch lane launch as before.
struct LateStore {
	uint64_t *kmers = nullptr, *meta = nullptr, *offsets = nullptr;      // [cap_chunks + 2], [cap_reads], [cap_reads + 1] (offsets in bases: 32 x chunks before)
	unsigned long long *state = nullptr;                                 // [0] reads << 32 | chunks reserved so far, [1] reads that fit (a prefix)
	uint32_t *lost = nullptr, *lost_n = nullptr;                         // reads that outgrew the lane machine's deep scratch too
	uint32_t cap_reads = 0, cap_chunks = 0;
};
__global__ __launch_bounds__(256) void vg_late_collect(LateStore ls, const uint64_t *__restrict__ pk_kmer, const uint64_t *__restrict__ pk_meta, const uint64_t *__restrict__ offsets,
                                                       const uint32_t *__restrict__ list, const uint32_t *__restrict__ n_list, uint32_t *__restrict__ residual, uint32_t *__restrict__ n_residual)
{
	const uint32_t n = *n_list;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t rid = list[i];
		const uint64_t off = offsets[rid];
		const uint32_t nch = (uint32_t)((offsets[rid + 1] - off) >> 5);
		bool kept = false;
		if (nch <= 32u) {
			const unsigned long long old = atomicAdd(&ls.state[0], (1ull << 32) | (unsigned long long)nch);
			const uint64_t r = old >> 32, c0 = old & 0xFFFFFFFFull;
			if (r < ls.cap_reads && c0 + nch <= ls.cap_chunks) {
				for (uint32_t c = 0; c < nch; c++) ls.kmers[c0 + c] = pk_kmer[(off >> 5) + c];
				ls.meta[r] = pk_meta[rid];
				ls.offsets[r] = 32ull * c0;
				ls.offsets[r + 1] = 32ull * (c0 + nch);
				atomicMax(&ls.state[1], (unsigned long long)(r + 1));
				kept = true;
			}
		}
		if (!kept) residual[atomicAdd(n_residual, 1u)] = rid;
	}
}

// ------------------------------------------------------------------------------------------------
// kernels: FASTQ framing on the device (reference src/qv.cc:760-784: four fgets() per record)
//
// The text arrives as a STREAM of arbitrary byte chunks (vg_fastq_stream_push).  Everything a chunk's framing needs to know
// about its predecessor -- the bytes of the record the previous chunk ended in the middle of -- stays on the device, so the
// host never waits for a result: it only moves bytes.  A chunk's buffer keeps FQ_CARRY bytes free in front of the copied
// text; the carried bytes are put right before it and the chunk's text starts there.
// ------------------------------------------------------------------------------------------------
size_t vg_dev_scan_temp_bytes(size_t n_u32, size_t n_u64);                                             // vg_sort.hip
int vg_dev_exclusive_scan_u32(const uint32_t *in, uint32_t *out, size_t n, hipStream_t stream, void *tmp, size_t tmp_bytes);
int vg_dev_exclusive_scan_u64(const uint64_t *in, uint64_t *out, size_t n, hipStream_t stream, void *tmp, size_t tmp_bytes);

constexpr uint32_t FQ_TILE = 4096;                   // bytes per workgroup tile (256 lanes x 16 bytes)
constexpr uint32_t FQ_CARRY = 1u << 16;              // room in front of a chunk for the unfinished record of the chunk before (4 lines <= 4 KiB)

struct FqStream {                                    // one per handle, device resident
	unsigned long long consumed;                     // bytes of the stream framed into complete records so far
	unsigned long long records;                      // ... and their number
	unsigned long long last_record;                  // stream offset of the last of them
	uint32_t carry;                                  // bytes of the previous chunk's text that belong to its unfinished last record
	uint32_t poisoned;                               // a chunk could not be framed here: it and everything after it is left to the host
};
struct FqChunk {                                     // one per batch slot, device resident
	uint32_t start, len;                             // the chunk's text is buf[start, start + len)
	uint32_t n_reads;                                // complete records framed (0 when refused)
	uint32_t bad;                                    // this chunk needs the host's framing (a line beyond fgets' 1023 characters, ...)
	unsigned long long total;                        // bases of the batch
};

// carried bytes in front of the new text; start / length of the chunk's text
__global__ __launch_bounds__(256) void vg_fqs_prepare(const FqStream *__restrict__ st, FqChunk *__restrict__ ck, const uint8_t *__restrict__ prev_end, uint8_t *__restrict__ buf, uint32_t nbytes)
{
	const uint32_t c = prev_end ? st->carry : 0u;
	for (uint32_t i = threadIdx.x; i < c; i += 256) buf[FQ_CARRY - c + i] = prev_end[(int64_t)i - (int64_t)c];
	if (threadIdx.x == 0) { ck->start = FQ_CARRY - c; ck->len = c + nbytes; ck->n_reads = 0; ck->bad = st->poisoned; ck->total = 0; }
}

// newlines per 4 KiB tile of the buffer (bytes outside the chunk's text do not count)
__global__ __launch_bounds__(256) void vg_fq_count_newlines(const uint8_t *__restrict__ buf, const FqChunk *__restrict__ ck, uint32_t *__restrict__ tile_cnt)
{
	__shared__ uint32_t wsum[4];
	const uint32_t lo = ck->start, hi = lo + ck->len;
	const uint32_t base = blockIdx.x * FQ_TILE + threadIdx.x * 16;
	uint32_t c = 0;
	if (base + 16 > lo && base < hi) {
		uint4 v;
		__builtin_memcpy(&v, buf + base, 16);                     // the buffer is padded to a whole tile
		const uint8_t *b = (const uint8_t *)&v;
		for (uint32_t j = 0; j < 16; j++) if (base + j >= lo && base + j < hi && b[j] == '\n') c++;
	}
	for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
	if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
	__syncthreads();
	if (threadIdx.x == 0) tile_cnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// line_start[g + 1] = byte after the g-th newline; line_start[0] = start of the text
__global__ __launch_bounds__(256) void vg_fq_line_starts(const uint8_t *__restrict__ buf, FqChunk *__restrict__ ck, const uint32_t *__restrict__ tile_off, uint32_t *__restrict__ line_start, uint32_t cap_lines)
{
	__shared__ uint32_t wsum[4];
	const uint32_t lo = ck->start, hi = lo + ck->len;
	const uint32_t base = blockIdx.x * FQ_TILE + threadIdx.x * 16;
	uint32_t mask = 0;
	if (base + 16 > lo && base < hi) {
		uint4 v;
		__builtin_memcpy(&v, buf + base, 16);
		const uint8_t *b = (const uint8_t *)&v;
		for (uint32_t j = 0; j < 16; j++) if (base + j >= lo && base + j < hi && b[j] == '\n') mask |= 1u << j;
	}
	const uint32_t c = (uint32_t)__popc(mask);
	uint32_t incl = c;
	const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o); if ((int)lane >= o) incl += y; }
	if (lane == 63) wsum[wv] = incl;
	__syncthreads();
	uint32_t g = tile_off[blockIdx.x] + incl - c;
	for (uint32_t w = 0; w < wv; w++) g += wsum[w];
	if (blockIdx.x == 0 && threadIdx.x == 0) line_start[0] = lo;
	while (mask) {
		const uint32_t j = (uint32_t)__ffs((int)mask) - 1;
		mask &= mask - 1;
		++g;
		if (g < cap_lines) line_start[g] = base + j + 1; else ck->bad = 1u;     // lines shorter than 8 bytes on average: host framing
	}
}

// one lane per record: read length = strlen(read line) - 1 (qv.cc:778); slots past the last record get length 0 (the scan
// that follows runs over the whole capacity).  A line longer than fgets' 1023 characters, or a quality line without a
// character for every chunk (it would expose the reference's stale buffer contents), refuses the chunk.
__global__ void vg_fq_record_lengths(const uint32_t *__restrict__ line_start, const uint32_t *__restrict__ n_lines_p, FqChunk *__restrict__ ck, uint64_t *__restrict__ rlen, uint32_t cap_rec, uint32_t cap_lines)
{
	const uint32_t n_lines = *n_lines_p;
	const uint32_t n_rec = n_lines + 1 <= cap_lines ? n_lines / 4 : 0u;
	if (blockIdx.x == 0 && threadIdx.x == 0 && (n_lines + 1 > cap_lines || n_rec > cap_rec)) ck->bad = 1u;
	for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= cap_rec; r += (uint64_t)gridDim.x * blockDim.x) {
		uint64_t len = 0;
		if (r < n_rec && r < cap_rec) {
			bool bad = false;
			for (int l = 0; l < 4; l++) bad |= (line_start[4 * r + l + 1] - line_start[4 * r + l]) > 1023u;
			len = line_start[4 * r + 2] - line_start[4 * r + 1] - 1u;             // the line's length minus its newline
			bad |= (line_start[4 * r + 4] - line_start[4 * r + 3] - 1u) < (len >> 5);
			if (bad) ck->bad = 1u;
		}
		rlen[r] = len;
	}
}

// after the offsets scan: the chunk's verdict, and the stream's state for the next chunk
__global__ void vg_fqs_finish(FqStream *__restrict__ st, FqChunk *__restrict__ ck, const uint32_t *__restrict__ n_lines_p, const uint32_t *__restrict__ line_start,
                              const uint64_t *__restrict__ offsets, uint32_t cap_rec)
{
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	if (st->poisoned || ck->bad) { st->poisoned = 1u; ck->bad = 1u; ck->n_reads = 0; ck->total = 0; return; }
	const uint32_t n_rec = *n_lines_p / 4;
	const uint32_t end = n_rec ? line_start[4 * n_rec] : ck->start;               // first byte after the last complete record
	const uint32_t used = end - ck->start, left = ck->len - used;
	if (left > FQ_CARRY) { st->poisoned = 1u; ck->bad = 1u; return; }            // an unfinished record of more than 64 KiB
	if (n_rec) st->last_record = st->consumed + (line_start[4 * (n_rec - 1)] - ck->start);
	st->consumed += used;
	st->records += n_rec;
	st->carry = left;
	ck->n_reads = n_rec;
	ck->total = offsets[n_rec < cap_rec ? n_rec : cap_rec];
}

// gather bases and quality characters of every record into the flat batch layout; a quality line shorter than the
// read keeps what the reference's buffer would hold there: its newline, then NULs (qv.cc:763, 836)
// The quality line is looked at here and nowhere else: all the path ever asks of it is whether character c is below '8' for the
// read's chunk numbers c (qv.cc:836, 943), so the batch carries one GATE WORD per read (bit c) instead of the strings.
__global__ __launch_bounds__(256) void vg_fq_gather(const uint8_t *__restrict__ text, const uint32_t *__restrict__ line_start, const uint64_t *__restrict__ offsets,
                                                    const FqChunk *__restrict__ ck, uint8_t *__restrict__ bases, uint32_t *__restrict__ gate)
{
	const uint64_t n_rec = ck->n_reads;
	const uint32_t lane = threadIdx.x & 63;
	const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
	for (uint64_t r = wave; r < n_rec; r += n_waves) {                          // one wave per record: coalesced copies
		const uint64_t o = offsets[r], len = offsets[r + 1] - o;
		const uint32_t s1 = line_start[4 * r + 1], s3 = line_start[4 * r + 3];
		const uint32_t qlen = line_start[4 * r + 4] - s3;                       // quality line incl. its newline
		for (uint64_t j = lane; j < len; j += 64) bases[o + j] = text[s1 + j];
		// (a record is only framed here when its quality line has a character for every chunk, vg_fq_record_lengths; a lane past
		// the line reads what the reference's buffer would hold there: its newline, then NULs)
		const uint32_t n = (uint32_t)(len >> 5);
		bool open = false;
		if (lane < n && lane < 32u) { const uint8_t q = lane < qlen ? text[s3 + lane] : (uint8_t)0; open = (int)(int8_t)q - '8' < 0; }
		const uint64_t m = __ballot(open);
		if (lane == 0) gate[r] = (uint32_t)m;
	}
}

// ------------------------------------------------------------------------------------------------
// host side of the handle
// ------------------------------------------------------------------------------------------------
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// hipMalloc that waits for memory on its way back.  Device memory that a handle -- or a process that has just ended -- gave up is
// not there for the next allocation at once: the driver clears it first, about 88 GB in 2.4 s on this part, and until then the
// runtime's book (hipMemGetInfo) already counts it free while hipMalloc still says no (profiles/vram_release_lag_r06.txt: four
// small jobs one after the other, each with the whole device's widest tables, had 300 GB "in use" between them).  So a refusal is
// final only when the book agrees: while it says the bytes are free -- or the free figure is still growing -- the call is repeated,
// for at most $VG_ALLOC_WAIT_S seconds (default 20; 0 = one attempt).
static hipError_t vg_malloc_patient(void **p, size_t bytes)
{
	hipError_t e = hipMalloc(p, bytes);
	if (e == hipSuccess) return e;
	(void)hipGetLastError();
	const char *w = getenv("VG_ALLOC_WAIT_S");
	const double limit = w && *w ? atof(w) : 20.0;
	const double t0 = now_s();
	double grew = t0;
	size_t best = 0;
	while (now_s() - t0 < limit) {
		size_t fr = 0, tot = 0;
		if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); break; }
		if (fr > best + (256u << 20)) { best = fr; grew = now_s(); }
		if (fr < bytes && now_s() - grew > 4.0) break;           // somebody holds it: that is the answer
		usleep(100000);
		e = hipMalloc(p, bytes);
		if (e == hipSuccess) {
			if (getenv("VG_VERBOSE")) fprintf(stderr, "[vargeno_hip] hipMalloc(%.1f GB) waited %.1f s for memory the driver was still clearing\n", bytes / 1e9, now_s() - t0);
			return e;
		}
		(void)hipGetLastError();
	}
	return e;
}

// the driver calls behind vg_arena.h: one hipMalloc per handle
struct HipBlock {
	static void *alloc(uint64_t bytes)
	{
		void *p = nullptr;
		if (vg_malloc_patient(&p, bytes) != hipSuccess) return nullptr;
		return p;
	}
	static void free(void *p) { (void)hipFree(p); }
};
typedef DevArenaT<HipBlock> DevArena;

struct ScratchBuf {
	Scratch s{};
	size_t bytes = 0;
};

// Per-batch resources.  A handle keeps NSLOT batches in flight: the wave tier of batch k+1 runs on the
// main stream while the (rare, latency-bound) lane tiers of batch k finish on the tail stream.
constexpr int NSLOT = 3;            // (5 and 8 were measured: no gain at 8 M-read batches, 10-30 % slower at 1 M -- more pack kernels run ahead and get in the wave kernel's way)
struct Slot {
	uint32_t *listA = nullptr, *listB = nullptr, *listC = nullptr;  uint64_t list_cap = 0;   // spill lists: main -> deep tier (A), deep tier -> lane tier (B), lost (C)
	uint32_t *ctr = nullptr;              // [0] wave-tier overflow, [1] lane-tier overflow, [2] lost -- this batch
	uint32_t *h_ctr = nullptr;            // page-locked copy of ctr[0..3], made on the tail stream when the batch's tiers are done
	// what the generic lane tier needs should the deep tier leave reads behind: it is only launched then (harvest), never empty
	const uint8_t *lt_bases = nullptr, *lt_quals = nullptr; const uint64_t *lt_offsets = nullptr; const uint32_t *lt_gate = nullptr; bool lt_packed = false, lt_stats = false, lt_enqueued = false, lt_late = false;      // lt_late: the deep tier's leftovers went through vg_late_collect (the per-batch lane tier then only has the residual list: listA, ctr[6])
	uint64_t *pk_kmer = nullptr, *pk_meta = nullptr;  uint64_t pk_kmer_cap = 0, pk_meta_cap = 0;   // packed reads of this batch
	uint8_t *st_bases = nullptr, *st_quals = nullptr; uint64_t *st_offsets = nullptr;   // staging of vg_reads_submit / vg_fastq_submit
	uint32_t *st_gate = nullptr; uint64_t st_gate_cap = 0;                              // gate words of a batch framed on the device
	uint64_t *hp_kmers = nullptr, *hp_meta = nullptr, *hp_offsets = nullptr; uint64_t hp_kmers_cap = 0, hp_reads_cap = 0;   // page-locked HOST staging of a batch framed + packed on the host
	uint8_t *fq_text = nullptr; uint32_t *fq_lines = nullptr, *fq_tiles = nullptr; uint64_t fq_text_cap = 0, fq_lines_cap = 0, fq_tiles_cap = 0;   // FASTQ framing
	void *fq_tmp = nullptr; uint64_t fq_tmp_cap = 0;                 // scan scratch (grow-only: no allocation per chunk)
	FqChunk *fq_chunk = nullptr;                                     // this chunk's framing results, device resident
	uint64_t fq_text_len = 0;                                        // bytes of text copied into fq_text (after the FQ_CARRY gap)
	hipEvent_t e_in = nullptr;                                       // the batch's buffers are complete (when another stream produced them)
	hipEvent_t e_fq = nullptr; bool fq_tail_wanted = false;          // the NEXT chunk's prepare kernel reads this text's tail: recorded after it
	uint64_t stage_bytes = 0, stage_quals_bytes = 0, stage_reads = 0;      // capacities of st_bases, st_quals, st_offsets
	hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr, e4 = nullptr, e5 = nullptr;
	bool busy = false;
};

struct vg_index {
	int device = 0;
	hipStream_t stream = nullptr, tail = nullptr;   // pack + wave tier | deep wave tier of earlier batches, one after the other
	// ... and their lane tiers + the copies of their counters, on a stream of their own (r05).  On the one tail stream the deep wave
	// tier of batch k+1 waited for the lane tier of batch k: with 250 bp reads a batch's lane tier takes 5-12 ms (a handful of reads
	// with thousands of contexts, latency-bound) against 5 ms per step, and the main stream waited for its slots (6.9 ms per step).
	// (A stream per batch slot -- six streams in all -- was tried and cost the DEFAULT workload 6 % and chr22-scale 25 %
	// (profiles/ab_tail_streams_r05.txt): more streams than hardware queues, and the main stream shares one.)
	hipStream_t tail2 = nullptr;
	hipStream_t ingest = nullptr;                   // FASTQ framing + pack kernel of the next batch, under the current batch's wave kernel
	int pack_overlap = -1;                          // the pack kernel of batch k+1 on the ingest stream, under batch k's wave kernel: 1 always, 0 never (VG_PACK_OVERLAP), -1: for small batches.
	                                                // Off by default: nothing fits beside a full set of main-tier workgroups (4 x 128 VGPRs per SIMD), so the
	                                                // two kernels only take turns on the CUs -- same reads/s (hg38 scale: 4.21 vs 4.26 ms per 8 M reads), but the
	                                                // wave kernel's duration then includes the time it spent waiting for the pack kernel (4.18 vs 3.55 ms)
	bool ingest_stream = true;                      // VG_NO_INGEST_STREAM: FASTQ framing on the main stream too
	DevIndex d{};
	DevArena arena;                       // the index's device memory (vg_arena.h): permanent arrays and the temporaries of its construction
	std::vector<void *> owned;            // device allocations of the index made with hipMalloc (small handle-lifetime buffers; everything, when the arena could not be set up)
	std::map<void *, uint64_t> owned_bytes;
	uint64_t arena_misses = 0, arena_miss_bytes = 0;      // requests the arena had no room for (served by hipMalloc)
	uint64_t dev_bytes = 0;               // device memory of the index: hipMalloc'ed buffers (counted as they are made) + the arena's mapped chunks (counted when construction is over)
	uint64_t n_sites = 0;
	bool cnt4_dirty = false;                           // base-indexed counters hold increments not yet folded into d.cnt
	std::vector<uint32_t> site_pos;
	std::vector<uint8_t> site_ref, site_alt, site_rf, site_af;
	ScratchBuf mid, big;                  // lane-tier scratch: every lane x 64 contexts; a few lanes x 16384 contexts
	Slot slot[NSLOT];
	int next_slot = 0;
	bool lane_tier_seen = false;          // some batch of this handle left reads for the per-batch lane tier (the late store did not take them): it is enqueued with every batch from then on
	LateStore late;                       // reads the deep tier left behind, kept for ONE lane-machine run at the next synchronisation (vg_late_collect)
	uint32_t late_h[4] = {0, 0, 0, 0};    // host copy of the store's counters at the last finish_pending
	bool late_dirty = false, late_stats = false;     // batches have been enqueued since the store was last run / they were counting-build batches
	uint64_t late_reads_run = 0;          // reads the late runs have finished since open
	bool spill_known = false;
	uint32_t spill_hint = 0;              // the deep tier's list of the last harvested batches (reads): sizes the next batch's deep-tier grid
	uint64_t cum[4] = {0, 0, 0, 0};       // since reset: [0] wave-tier overflow, [1] lane-tier overflow, [2] lost, [3] reads with a character other than ACGTN
	uint8_t *d_clamped = nullptr;         // [2 * n_sites] staging of vg_counts_fetch: min(63, sum), ref counts then alt counts
	unsigned long long *d_stats = nullptr;
	bool stats_enabled = true;
	bool force_generic = false;           // VG_FORCE_GENERIC=1: skip the wave tier (tests compare the tiers)
	double t_pack = 0, t_main = 0, t_tail = 0, t_total = 0, t_w2 = 0; uint64_t t_batches = 0;   // harvested event times since the last vg_timing_get
	int cus = 256;
	int lane_grid_blocks = 0, wave_grid = 0;
	uint32_t work_chunk = 128;            // reads a main-tier wave pulls from the launch's work counter at a time (VG_WORK_CHUNK)
	uint32_t w2_chunk = 8, w2_wpc = 3;    // deep tier: reads a wave pulls at a time (VG_W2_CHUNK), workgroups per CU of its grid (VG_W2_WPC)
	FqStream *d_fq = nullptr;             // FASTQ stream state (vg_fastq_stream_*)
	bool fq_open = false; int fq_prev_slot = -1;
	uint64_t max_device_bytes = 0;        // the caller's budget for this replica (vg_index_open_ex; 0: the whole device)
	std::string plan_text;                // what the budget bought: views kept / left out (vg_index_plan)
	std::string aux_note;                 // ... and what the loader found in the auxiliary rows, if anything
	std::string open_report;              // where vg_index_open's time went, phase by phase
	vgp::Packer *packer = nullptr;        // host-side framing + packing (vg_fastq_stream_begin_packed)
	bool fq_packed = false;               // the open FASTQ stream is framed + packed on the host
	uint64_t host_invalid = 0;            // reads with a character other than ACGTN found by the host packer since the last reset
};

// The index handle a thread is constructing (vg_index_open / vg_index_create run on the caller's thread; the CLI opens its replicas
// on one thread each): where TempDev finds the arena and the stream.
static thread_local vg_index *g_building = nullptr;
static thread_local double g_alloc_s = 0;                  // seconds spent inside allocation calls since the last lap (VG_VERBOSE)

// a permanent array of the index: out of the arena's bottom half; hipMalloc when the arena is not there (or `plain`: buffers that
// other libraries get to see -- RCCL reduces the counters in place)
template <class T>
static int dev_alloc(vg_index *ix, T **p, uint64_t count, bool zero = false, bool plain = false)
{
	void *q = nullptr;
	const size_t bytes = (size_t)(count ? count : 1) * sizeof(T);
	const double t0 = now_s();
	if (!plain) { q = ix->arena.take(bytes, false); if (!q && ix->arena.ready()) { ix->arena_misses++; ix->arena_miss_bytes += bytes; } }
	if (!q) {
		hipError_t e = vg_malloc_patient(&q, bytes);
		if (e != hipSuccess) { return fail(VG_ENOMEM, "hipMalloc(%s bytes): %s", std::to_string(bytes).c_str(), hipGetErrorString(e)); }
		ix->owned.push_back(q);
		ix->owned_bytes[q] = bytes;
		ix->dev_bytes += bytes;
	}
	g_alloc_s += now_s() - t0;
	if (zero) { hipError_t e = hipMemsetAsync(q, 0, bytes, ix->stream); if (e != hipSuccess) return fail(VG_ENODEV, "hipMemset: %s", hipGetErrorString(e)); }
	*p = (T *)q;
	return VG_OK;
}
// ... and one that is not needed any more (a jump table that a wider table has replaced); the device must be done with it
static void dev_release(vg_index *ix, const void *cp)
{
	void *p = const_cast<void *>(cp);
	if (!p) return;
	(void)hipStreamSynchronize(ix->stream);
	if (ix->arena.give(p)) return;
	for (size_t z = 0; z < ix->owned.size(); z++) if (ix->owned[z] == p) { ix->owned.erase(ix->owned.begin() + (long)z); break; }
	auto it = ix->owned_bytes.find(p);
	if (it != ix->owned_bytes.end()) { ix->dev_bytes -= it->second; ix->owned_bytes.erase(it); }
	(void)hipFree(p);
}
template <class T>
static int dev_upload(vg_index *ix, T **p, const T *src, uint64_t count)
{
	int rc = dev_alloc(ix, p, count);
	if (rc) return rc;
	if (count) HIP_TRY(hipMemcpy(*p, src, (size_t)count * sizeof(T), hipMemcpyHostToDevice));
	return VG_OK;
}
// a temporary of the construction, given back when it goes out of scope (or released earlier): out of the arena's top half while a
// handle is being built on this thread, hipMalloc otherwise.  Giving it back waits for the stream first, as hipFree does.
template <class T>
struct TempDev {
	T *p = nullptr;
	vg_index *ix = nullptr;                        // the handle whose arena holds p (nullptr: hipMalloc)
	~TempDev() { release(); }
	int alloc(uint64_t count)
	{
		release();
		const size_t bytes = (size_t)(count ? count : 1) * sizeof(T);
		const double t0 = now_s();
		if (g_building) { p = (T *)g_building->arena.take(bytes, true); if (p) ix = g_building; else if (g_building->arena.ready()) { g_building->arena_misses++; g_building->arena_miss_bytes += bytes; } }
		if (!p) {
			hipError_t e = vg_malloc_patient((void **)&p, bytes);
			if (e != hipSuccess) { p = nullptr; return fail(VG_ENOMEM, "hipMalloc(staging): %s", hipGetErrorString(e)); }
		}
		g_alloc_s += now_s() - t0;
		return VG_OK;
	}
	int upload(const T *src, uint64_t count)
	{
		int rc = alloc(count);
		if (rc) return rc;
		if (count) HIP_TRY(hipMemcpy(p, src, (size_t)count * sizeof(T), hipMemcpyHostToDevice));
		return VG_OK;
	}
	void release()
	{
		if (!p) return;
		const double t0 = now_s();
		if (ix) { (void)hipStreamSynchronize(ix->stream); (void)ix->arena.give(p); }
		else (void)hipFree(p);
		g_alloc_s += now_s() - t0;
		p = nullptr; ix = nullptr;
	}
};

static int alloc_scratch(vg_index *ix, ScratchBuf &b, uint32_t nlanes, uint32_t cap, uint32_t kcap)
{
	b.s.nlanes = nlanes; b.s.cap = cap; b.s.kcap = kcap;
	int rc;
	if ((rc = dev_alloc(ix, &b.s.ctx_kmer, (uint64_t)cap * nlanes))) return rc;
	if ((rc = dev_alloc(ix, &b.s.ctx_kpos, (uint64_t)cap * nlanes))) return rc;
	if ((rc = dev_alloc(ix, &b.s.ctx_meta, (uint64_t)cap * nlanes))) return rc;
	if ((rc = dev_alloc(ix, &b.s.key_index, (uint64_t)kcap * nlanes))) return rc;
	if ((rc = dev_alloc(ix, &b.s.key_first, (uint64_t)kcap * nlanes))) return rc;
	if ((rc = dev_alloc(ix, &b.s.key_fm, (uint64_t)kcap * nlanes))) return rc;
	b.bytes = ((uint64_t)cap * 16 + (uint64_t)kcap * 12) * nlanes;
	return VG_OK;
}

extern "C" void vg_index_close(vg_index *ix)
{
	if (!ix) return;
	(void)hipSetDevice(ix->device);
	if (ix->stream) (void)hipStreamSynchronize(ix->stream);
	if (ix->tail) (void)hipStreamSynchronize(ix->tail);
	if (ix->tail2) (void)hipStreamSynchronize(ix->tail2);
	if (ix->ingest) (void)hipStreamSynchronize(ix->ingest);
	for (void *p : ix->owned) (void)hipFree(p);
	ix->arena.destroy();
	for (Slot &sl : ix->slot) {
		void *extra[] = {sl.listA, sl.listB, sl.listC, sl.st_bases, sl.st_quals, sl.st_gate, sl.st_offsets, sl.pk_kmer, sl.pk_meta, sl.fq_text, sl.fq_lines, sl.fq_tiles, sl.fq_tmp, sl.fq_chunk};
		for (void *p : extra) if (p) (void)hipFree(p);
		hipEvent_t evs[] = {sl.e0, sl.e1, sl.e2, sl.e3, sl.e4, sl.e5, sl.e_fq, sl.e_in};
		for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
		void *host[] = {sl.hp_kmers, sl.hp_meta, sl.hp_offsets, sl.h_ctr};
		for (void *p : host) if (p) (void)hipHostFree(p);
	}
	delete ix->packer;
	if (ix->stream) (void)hipStreamDestroy(ix->stream);
	if (ix->tail) (void)hipStreamDestroy(ix->tail);
	if (ix->tail2) (void)hipStreamDestroy(ix->tail2);
	if (ix->ingest) (void)hipStreamDestroy(ix->ingest);
	delete ix;
}

// Wall time of the phases of vg_index_open / vg_index_create (start-up is SURVEY.md §8f-4): kept in the handle (vg_index_open_report),
// and on stderr under VG_VERBOSE=1.  Every lap also says how much of it was spent inside allocation calls.
struct PhaseClock {
	vg_index *ix;
	const bool on = getenv("VG_VERBOSE") != nullptr;
	std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
	explicit PhaseClock(vg_index *h) : ix(h) { g_alloc_s = 0; }
	void lap(const char *what)
	{
		(void)hipDeviceSynchronize();
		const auto n = std::chrono::steady_clock::now();
		const double sec = std::chrono::duration<double>(n - t).count();
		char line[200];
		snprintf(line, sizeof line, "%s%s %.2f s (allocation calls %.2f s)", ix->open_report.empty() ? "" : "; ", what, sec, g_alloc_s);
		ix->open_report += line;
		if (on) fprintf(stderr, "[vargeno_hip] %-44s %.2f s  (allocation calls %.2f s)\n", what, sec, g_alloc_s);
		g_alloc_s = 0;
		t = n;
	}
};

// reader threads of the index loader
static unsigned host_threads()
{
	static const unsigned n = [] {
		unsigned h = std::thread::hardware_concurrency();
		if (const char *e = getenv("VG_HOST_THREADS")) h = (unsigned)std::max(1, atoi(e));
		return std::max(1u, std::min(h, 64u));
	}();
	return n;
}

// ------------------------------------------------------------------------------------------------
// index construction.  Everything from the dictionaries' columns onward happens on the device: the host only brings
// bytes (from the caller's arrays, or straight from the files -- read by several threads into pinned staging buffers and
// copied up while the next piece is being read; the packed records are unpacked by a kernel).
// ------------------------------------------------------------------------------------------------

// unpack the reference dictionary's 13-byte records {u64 k-mer, u32 pos, u8 ambig} (dictgen.c:63-154): a workgroup stages the
// 3 328 contiguous bytes of 256 records in LDS with 16-byte loads, then every lane takes its own record out of LDS
__global__ __launch_bounds__(256) void vg_unpack_ref(const uint8_t *__restrict__ raw, uint64_t n, uint64_t *__restrict__ kmer, uint32_t *__restrict__ pos, uint8_t *__restrict__ amb)
{
	__shared__ __attribute__((aligned(16))) uint8_t sm[256 * 13 + 48];
	for (uint64_t r0 = (uint64_t)blockIdx.x * 256; r0 < n; r0 += (uint64_t)gridDim.x * 256) {
		const uint64_t b0 = 13 * r0, a0 = b0 & ~15ull;              // the tile's first byte, and the 16-byte boundary below it (raw itself is aligned)
		const uint32_t shift = (uint32_t)(b0 - a0);
		const uint64_t nrec = n - r0 < 256 ? n - r0 : 256;
		const uint32_t nbytes = shift + 13u * (uint32_t)nrec;
		__syncthreads();                                            // previous tile fully consumed
		for (uint32_t i = threadIdx.x * 16; i < nbytes; i += 256 * 16) *reinterpret_cast<uint4 *>(sm + i) = *reinterpret_cast<const uint4 *>(raw + a0 + i);   // (the buffer has slack behind its last byte)
		__syncthreads();
		if (threadIdx.x < nrec) {
			const uint8_t *q = sm + shift + 13 * threadIdx.x;
			uint64_t k; uint32_t p2;
			__builtin_memcpy(&k, q, 8); __builtin_memcpy(&p2, q + 8, 4);
			const uint64_t i = r0 + threadIdx.x;
			kmer[i] = k; pos[i] = p2; amb[i] = q[12];
		}
	}
}
// ... the SNP dictionary's 16-byte records {u64 k-mer, u32 pos, u8 snp_info, ambig, ref_freq, alt_freq} (dictgen.c:156-275): one aligned 16-byte load each
__global__ void vg_unpack_snp(const uint8_t *__restrict__ raw, uint64_t n, uint64_t *__restrict__ kmer, uint32_t *__restrict__ pos,
                              uint8_t *__restrict__ info, uint8_t *__restrict__ amb, uint8_t *__restrict__ rf, uint8_t *__restrict__ af)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint4 v = *reinterpret_cast<const uint4 *>(raw + 16 * i);
		kmer[i] = ((uint64_t)v.y << 32) | v.x; pos[i] = v.z;
		info[i] = (uint8_t)v.w; amb[i] = (uint8_t)(v.w >> 8); rf[i] = (uint8_t)(v.w >> 16); af[i] = (uint8_t)(v.w >> 24);
	}
}
// ... auxiliary rows: the reference dictionary's are 10 x u32 already (copied bytewise: the file offset is not aligned); the SNP
// dictionary's are {u64 k-mer, 10 x {u32 pos, u8 snp_info, u8 ref_freq, u8 alt_freq}} = 78 bytes, of which the path reads pos and snp_info
__global__ void vg_unpack_ref_aux(const uint8_t *__restrict__ raw, uint64_t n_words, uint32_t *__restrict__ out)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint8_t *q = raw + 4 * i;
		out[i] = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
	}
}
__global__ void vg_unpack_snp_aux(const uint8_t *__restrict__ raw, uint64_t n_rows, uint32_t *__restrict__ pos, uint8_t *__restrict__ info)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows * AUX_COLS; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint8_t *q = raw + 78 * (i / AUX_COLS) + 8 + 7 * (i % AUX_COLS);
		pos[i] = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
		info[i] = q[4];
	}
}

// largest genome position any entry names (the reference sizes its pile-up table max(raw pos field) + 33, i.e. 2^32 + 32 entries as soon
// as one k-mer is POS_AMBIGUOUS; only real positions are ever indexed)
__global__ void vg_max_pos(const uint32_t *__restrict__ pos, const uint8_t *__restrict__ amb, uint64_t n, unsigned long long *out)
{
	unsigned long long m = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t p = pos[i];
		if ((!amb || amb[i] == 0) && p != POS_AMBIGUOUS && p > m) m = p;
	}
	for (int o = 32; o > 0; o >>= 1) { const unsigned long long y = __shfl_xor(m, o); m = y > m ? y : m; }
	if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
// What the kernels take for granted about a dictionary and a corrupt file could violate: k-mers strictly increasing (the jump
// tables and every bisection), and an entry flagged "several positions" naming a row the auxiliary table has.  bad[0] / bad[1]
// count the violations; the caller refuses the index (a wild row index would be a wild read on the device).
__global__ void vg_check_columns(const uint64_t *__restrict__ kmer, const uint32_t *__restrict__ pos, const uint8_t *__restrict__ amb, uint64_t n, uint64_t n_aux,
                                 unsigned long long *bad)
{
	unsigned long long unsorted = 0, wild = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		if (i + 1 < n && kmer[i] >= kmer[i + 1]) unsorted++;
		if (amb[i] != 0 && pos[i] != POS_AMBIGUOUS && pos[i] >= n_aux) wild++;
	}
	if (unsorted) atomicAdd(&bad[0], unsorted);
	if (wild) atomicAdd(&bad[1], wild);
}
// An auxiliary row lists the positions of a k-mer with 2-10 occurrences (dictgen.c:63-154); they are distinct unless the SNP list
// holds the very same record several times (then the k-mer "occurs" that often at one position).  The wave kernel's key table
// gives a chunk one vote per key, so it has to know (DevIndex::aux_dups): it counts rows with a repeated position.
__global__ void vg_check_aux_rows(const uint32_t *__restrict__ rows, uint64_t n_rows, unsigned long long *dups)
{
	unsigned long long mine = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += (uint64_t)gridDim.x * blockDim.x) {
		uint32_t v[AUX_COLS];
		for (int j = 0; j < AUX_COLS; j++) v[j] = rows[i * AUX_COLS + j];
		bool live = true, rep = false;
		for (int j = 0; j < AUX_COLS; j++) {
			live = live && v[j] != 0;
			if (!live) break;
			for (int k = 0; k < j; k++) rep = rep || v[k] == v[j];
		}
		if (rep) mine++;
	}
	if (mine) atomicAdd(dups, mine);
}
// Pile-up seeding in FILE ORDER, last writer wins (qv.cc:637-659): every position first learns the index of the LAST SNP-dictionary
// entry that seeds it ...
__global__ void vg_site_winner(const uint32_t *__restrict__ pos, const uint8_t *__restrict__ info, const uint8_t *__restrict__ amb, uint64_t n, uint32_t *__restrict__ winner)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t f = info[i];
		if ((f & 4u) == 0 && pos[i] != POS_AMBIGUOUS && amb[i] == 0) atomicMax(&winner[(uint64_t)pos[i] + (f >> 3)], (uint32_t)(i + 1));   // n < 2^32 - 1
	}
}
// ... then every position applies that one entry: one wave per 64-position block -- the ballot of "is a site" is the block's rank mask
__global__ __launch_bounds__(256) void vg_site_blocks(const uint32_t *__restrict__ winner, const uint64_t *__restrict__ kmer, const uint8_t *__restrict__ info,
                                                      uint64_t plen, uint8_t *__restrict__ pile, ulonglong2 *__restrict__ rank, uint64_t *__restrict__ blk_sites)
{
	const uint64_t nblk = plen / 64 + 1;
	const uint32_t lane = threadIdx.x & 63;
	for (uint64_t b = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; b < nblk; b += ((uint64_t)gridDim.x * blockDim.x) >> 6) {
		const uint64_t p = b * 64 + lane;
		uint32_t w = 0;
		if (p < plen) {
			const uint32_t win = winner[p];
			if (win) {
				const uint32_t f = info[win - 1];
				w = (f & 3u) | (((uint32_t)(kmer[win - 1] >> (2 * (f >> 3))) & 3u) << 2);
			}
		}
		const bool site = (w & 3u) != ((w >> 2) & 3u);
		if (p < plen) pile[p] = (uint8_t)(w | (site ? 16u : 0u));
		const uint64_t m = __ballot(site);
		if (lane == 0) { rank[b] = make_ulonglong2(m, 0ull); blk_sites[b] = (uint64_t)__popcll(m); }
	}
}
__global__ __launch_bounds__(256) void vg_site_tables(const uint32_t *__restrict__ winner, const uint8_t *__restrict__ pile, const uint8_t *__restrict__ rf, const uint8_t *__restrict__ af,
                                                      uint64_t plen, ulonglong2 *__restrict__ rank, const uint64_t *__restrict__ blk_before,
                                                      uint32_t *__restrict__ s_pos, uint8_t *__restrict__ s_ref, uint8_t *__restrict__ s_alt, uint8_t *__restrict__ s_rf, uint8_t *__restrict__ s_af, uint8_t *__restrict__ s_ba)
{
	const uint64_t nblk = plen / 64 + 1;
	const uint32_t lane = threadIdx.x & 63;
	for (uint64_t b = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; b < nblk; b += ((uint64_t)gridDim.x * blockDim.x) >> 6) {
		const uint64_t before = blk_before[b], m = rank[b].x;
		if (lane == 0) rank[b].y = before;
		if ((m >> lane) & 1ull) {
			const uint64_t p = b * 64 + lane, s = before + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
			const uint32_t w = pile[p], win = winner[p];
			s_pos[s] = (uint32_t)p; s_ref[s] = (uint8_t)(w & 3u); s_alt[s] = (uint8_t)((w >> 2) & 3u); s_ba[s] = (uint8_t)(w & 15u);
			s_rf[s] = rf[win - 1]; s_af[s] = af[win - 1];
		}
	}
}

// the dictionaries' columns on the device: temporaries of the construction (freed when it is done)
struct DevCols {
	uint64_t n_ref = 0, n_ref_aux = 0, n_snp = 0, n_snp_aux = 0;
	TempDev<uint64_t> ref_kmer, snp_kmer;
	TempDev<uint32_t> ref_pos, snp_pos;
	TempDev<uint8_t> ref_amb, snp_info, snp_amb, snp_rf, snp_af;
	uint32_t *ref_aux = nullptr, *snp_aux_pos = nullptr; uint8_t *snp_aux_info = nullptr;     // final arrays, owned by the index
	int alloc()
	{
		int rc;
		if ((rc = ref_kmer.alloc(n_ref)) || (rc = ref_pos.alloc(n_ref)) || (rc = ref_amb.alloc(n_ref))) return rc;
		if ((rc = snp_kmer.alloc(n_snp)) || (rc = snp_pos.alloc(n_snp)) || (rc = snp_info.alloc(n_snp)) || (rc = snp_amb.alloc(n_snp)) || (rc = snp_rf.alloc(n_snp)) || (rc = snp_af.alloc(n_snp))) return rc;
		return VG_OK;
	}
};

static int init_handle(vg_index *ix, int device)
{
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VG_ENODEV, "no HIP device available (this library has no CPU fallback)");
	if (device < 0 || device >= ndev) return fail(VG_EINVAL, "device index out of range");
	ix->device = device;
	HIP_TRY(hipSetDevice(device));
	HIP_TRY(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
	{
		// the spill tiers are a few hundred small workgroups that can only start when main-tier workgroups retire: with a
		// higher priority they are placed first at every such moment instead of queueing behind the next wave kernel
		int lo_p = 0, hi_p = 0;
		(void)hipDeviceGetStreamPriorityRange(&lo_p, &hi_p);
		if (getenv("VG_TAIL_PRIO") && atoi(getenv("VG_TAIL_PRIO")) == 0) hi_p = 0;
		HIP_TRY(hipStreamCreateWithPriority(&ix->tail, hipStreamNonBlocking, hi_p));
		HIP_TRY(hipStreamCreateWithPriority(&ix->tail2, hipStreamNonBlocking, hi_p));
	}
	HIP_TRY(hipStreamCreateWithFlags(&ix->ingest, hipStreamNonBlocking));
	if (const char *e = getenv("VG_PACK_OVERLAP")) ix->pack_overlap = atoi(e) != 0 ? 1 : 0;
	ix->ingest_stream = getenv("VG_NO_INGEST_STREAM") == nullptr;
	for (Slot &sl : ix->slot) HIP_TRY(hipHostMalloc((void **)&sl.h_ctr, 64, hipHostMallocDefault));
	for (Slot &sl : ix->slot) { HIP_TRY(hipEventCreate(&sl.e0)); HIP_TRY(hipEventCreate(&sl.e1)); HIP_TRY(hipEventCreate(&sl.e2)); HIP_TRY(hipEventCreate(&sl.e3)); HIP_TRY(hipEventCreate(&sl.e4)); HIP_TRY(hipEventCreate(&sl.e5)); HIP_TRY(hipEventCreateWithFlags(&sl.e_fq, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&sl.e_in, hipEventDisableTiming)); }
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device));
	ix->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	ix->lane_grid_blocks = ix->cus * 8;                          // 2048 lanes per CU = every wave slot
	int wpc = 16;                                                // waves per CU of the wave-tier grid: 128 VGPRs -> 4 waves per SIMD
	if (const char *e = getenv("VG_WAVES_PER_CU")) wpc = std::max(1, atoi(e));
	ix->wave_grid = ix->cus * wpc;
	if (const char *e = getenv("VG_FORCE_GENERIC")) ix->force_generic = atoi(e) != 0;
	if (const char *e = getenv("VG_WORK_CHUNK")) ix->work_chunk = (uint32_t)std::max(1, atoi(e));
	if (const char *e = getenv("VG_W2_CHUNK")) ix->w2_chunk = (uint32_t)std::max(1, atoi(e));
	if (const char *e = getenv("VG_W2_WPC")) ix->w2_wpc = (uint32_t)std::max(1, atoi(e));
	return VG_OK;
}

// Which optional views a replica gets is decided BEFORE anything is built, from the dictionaries' sizes and a byte budget alone
// (vg_index_open_ex; without one: the device's total memory less 12 GiB) -- never from what happens to be free at that moment, so
// the same files and the same budget always give the same views (r03 asked hipMemGetInfo while building: a caller that shared
// the device got the slower layout without being told).  Views are taken in a fixed order while the planned total stays within
// the budget; vg_index_plan() says what was kept, what was left out and what that costs.
struct ViewPlan {
	bool mx = false, dx = false, sec = false, sig = false, probe = false, hx = false, jg32 = false, ssec = false;
	uint64_t base = 0, total = 0, budget = 0;
	uint64_t arena = 0;                    // bytes of the handle's one block (vg_arena.h): the finished index less what lives outside it
	bool limited = false;                  // views were left out for the budget
	uint32_t dx_bits = 32, ref_jg_bits = 32;   // buckets of the direct table / of the reference dictionary's jump table (DevIndex)
	std::string text;
	bool same_views(const ViewPlan &o) const { return mx == o.mx && dx == o.dx && sec == o.sec && sig == o.sig && probe == o.probe && hx == o.hx && jg32 == o.jg32 && ssec == o.ssec && dx_bits == o.dx_bits && ref_jg_bits == o.ref_jg_bits; }
};
// Tables that scale with the index AND the budget (r06).  The reference's jump table has 2^32 entries whatever the genome
// (qv.cc:539-584), and so had this library's direct table: a chr22-scale index (1 GB of files) took 92 GB of HBM, 99 % of it empty
// buckets.  A table over the top b bits of the same word finds the same entries as long as what the bucket leaves undecided is
// compared (ref_bounds, DevIndex::dx_bits), so the width is now a planning decision.  The NATURAL width of a table is the power of
// two at or above its entry count (0.5-1 entry per bucket, like hg38's 0.75 in 2^32; from 2^31 entries on that is 2^32 itself).
// Measured at chr22 scale (profiles/ab_chr22_table_bits_r06.txt): sparser is FASTER -- 2^32 buckets 0.333 ms per 1 M reads, 2^28
// 0.389, 2^27 (natural) 0.401, 2^26 0.437 -- because a bucket with two or more entries costs a second line and an empty or
// single-entry one is settled by its record; a table that would fit the 256 MB Infinity Cache is not faster for it (the L2-miss
// path is the limit).  So the plan takes the WIDEST table the budget holds, 2^32 first, then natural + 2 down to
// natural - 4: a whole device still gives the small index its 2^32 buckets, a budget of 8 GB gives it 2^27 and the same results.
// VG_DX_BITS / VG_REF_JG_BITS force a width (tests, A/B runs).
static uint32_t table_bits_for(uint64_t entries, const char *env)
{
	if (const char *e = getenv(env)) { const int b = atoi(e); if (b >= 16 && b <= 32) return (uint32_t)b; }
	uint32_t b = 16;
	while (b < 32 && (1ull << b) < entries) b++;
	return b >= 31 ? 32u : b;
}
static ViewPlan plan_views(const DevCols &c, uint64_t maxp, uint64_t ref_bf_bits, uint64_t snp_bf_bits, uint64_t budget_arg, uint64_t device_total, int cus)
{
	ViewPlan p;
	const uint64_t GiB = 1ull << 30, n = c.n_ref, m = c.n_snp, J32 = ((1ull << 32) + 1) * 4;
	p.budget = budget_arg ? budget_arg : (device_total > 12 * GiB ? device_total - 12 * GiB : device_total);
	const uint64_t plen = maxp + 64, sites = m / 32 + 1;                       // (an SNP seeds at most one site; ~32 k-mers per SNP)
	// lane-tier scratch: the deep one (a few lanes x 16384 contexts) always; the wide shallow one (every lane x 64 contexts) only for
	// the lane machine as the WHOLE path (VG_FORCE_GENERIC=1: tests compare the tiers) -- it was 0.74 GB that no other run touched
	const bool lane_only = getenv("VG_FORCE_GENERIC") && atoi(getenv("VG_FORCE_GENERIC")) != 0;
	const uint64_t scratch = (lane_only ? (uint64_t)cus * 8 * 256 * (64 * 16 + 32 * 12) : 0ull) + 4096ull * (16384 * 16 + 2048 * 12);
	p.ref_jg_bits = table_bits_for(n, "VG_REF_JG_BITS");         // (the mandatory part holds the natural width; 2^32 entries are an upgrade, below)
	const uint64_t Jref = ((1ull << p.ref_jg_bits) + 1) * 4;
	// what every layout holds: both dictionaries in file order with their jump tables, auxiliary rows, bit vectors, pile-up
	// sites and counters, lane-tier scratch, and room for three batch slots of a few million reads
	p.base = Jref + 16 * n + 40 * c.n_ref_aux + ((1ull << 24) + 1) * 4 + 16 * m + 50 * c.n_snp_aux + std::min<uint64_t>(ref_bf_bits, 1ull << 32) / 8 + snp_bf_bits / 8
	       + plen + plen / 4 + sites * 32 + scratch + 2 * GiB;
	p.total = p.base;
	const bool can_mx = !getenv("VG_NO_MX") && n + m < (1ull << 32);
	uint32_t bits = 14;
	while (bits < 30 && (1ull << bits) < n) bits++;
	char line[512];
	std::string kept, dropped;
	auto take = [&](bool allowed, uint64_t bytes, const char *name, const char *cost, bool &flag) {
		if (!allowed) return;
		if (p.total + bytes <= p.budget) { flag = true; p.total += bytes; snprintf(line, sizeof line, "%s%s %.1f GB", kept.empty() ? "" : ", ", name, bytes / 1e9); kept += line; }
		else { snprintf(line, sizeof line, "%s%s (%.1f GB; %s)", dropped.empty() ? "" : "; ", name, bytes / 1e9, cost); dropped += line; }
	};
	const bool want_probe = !getenv("VG_NO_PROBE_VIEW");
	take(want_probe && !getenv("VG_NO_SIG_VIEW"), (m + 16) * 2, "signature view of the strided SNP scan", "the scan reads the dictionary entries themselves: 11 lines per 8 entries instead of 1", p.sig);
	take(want_probe && getenv("VG_NO_SIG_VIEW") != nullptr, (m + 1) * 8, "probe view of the strided SNP scan", "the scan reads the dictionary entries themselves", p.probe);
	take(!getenv("VG_NO_SEC"), 12 * n + 16 + ((1ull << bits) + 1) * 4, "LO32-ordered view", "the 48 high-half neighbour queries of a gate-open chunk are made one by one: ~2 x the stage-B time", p.sec);
	if (can_mx) {
		// merged view + direct table together, the table as wide as the index wants it; under a budget that does not hold that, with
		// half and a quarter of the buckets (two and four entries' worth of key per bucket: more look-ups read a second line) before
		// the table is given up for the jump-table form (2^32 entries: the form the look-up without a direct table is written for)
		const uint32_t nat = table_bits_for(n + m, "VG_DX_BITS");
		const bool forced = getenv("VG_DX_BITS") != nullptr;
		uint32_t cand[8]; int nc = 0;
		if (forced) cand[nc++] = nat;
		else { cand[nc++] = 32u; for (int k = 2; k >= -4; k--) { const int b = (int)nat + k; if (b >= 16 && b < 32) cand[nc++] = (uint32_t)b; } }
		if (!getenv("VG_NO_DIRECT")) for (int ci = 0; ci < nc && !p.dx; ci++) {
			const uint32_t b = cand[ci];
			const uint64_t bytes = 16 * (n + m) + (1ull << b) * 16;
			if (p.total + bytes <= p.budget) {
				p.mx = p.dx = true; p.dx_bits = b; p.total += bytes;
				snprintf(line, sizeof line, "%smerged exact-match view %.1f GB, direct table of 2^%u buckets %.1f GB%s", kept.empty() ? "" : ", ", 16 * (n + m) / 1e9, b, (double)((1ull << b) * 16) / 1e9,
				         b + 1 == nat ? " (HALF the buckets the index wants: the budget)" : b + 2 == nat ? " (A QUARTER of the buckets the index wants: the budget)" : b + 2 < nat ? " (AN EIGHTH OR LESS of the buckets the index wants: the budget)" : b < 32u && ci > 0 ? " (the widest the budget holds; 2^32 is faster)" : "");
				kept += line;
			}
		}
		if (!p.dx) {
			take(true, 16 * (n + m) + J32, "merged exact-match view", "two look-ups per chunk instead of one, no both-strands shortcut: ~+45 % kernel time (r03 A/B)", p.mx);
			if (!getenv("VG_NO_DIRECT")) { snprintf(line, sizeof line, "%sdirect table (%.1f GB at 2^%u buckets; a jump-table gather in front of every exact look-up: ~+15 %% kernel time, r02 A/B)", dropped.empty() ? "" : "; ", (double)((1ull << (nat >= 18u ? nat - 2u : 16u)) * 16) / 1e9, nat >= 18u ? nat - 2u : 16u); dropped += line; }
		}
	} else {
		const bool want32 = !getenv("VG_NO_SNP_JG32");
		take(want32 && !getenv("VG_NO_HX"), ((1ull << 32) + 1) * 16 - (p.ref_jg_bits == 32u ? J32 : 0), "paired HI32 table", "separate jump tables, no absence filters: +25 % kernel time (r03: 6.34 vs 5.03 ms)", p.hx);
		take(want32 && !p.hx, J32, "HI32 jump table of the SNP dictionary", "SNP look-ups bisect HI24 buckets of ~190 entries: 8 dependent probes", p.jg32);
	}
	{
		uint32_t sb = 14;
		while (sb < 30 && (1ull << sb) < m) sb++;
		take(!getenv("VG_NO_SSEC") && m > 0, 12 * m + 16 + ((1ull << sb) + 1) * 4, "LO32-ordered view of the SNP dictionary", "the high-half SNP neighbour queries of a chunk whose SNP bit-vector probe is positive are made one by one: up to 36 bisections of a HI24 bucket, in stage B1's rounds", p.ssec);
	}
	if (p.ref_jg_bits < 32u && !getenv("VG_REF_JG_BITS") && !p.hx && p.total + (J32 - Jref) <= p.budget) {
		// the reference's own table, one entry per HI32 value: a gate-open chunk's bucket bounds in one gather (the coarse table reads
		// the bucket's entries to tell HI32 values apart: +7 % kernel time at chr22 scale)
		p.total += J32 - Jref; p.base += J32 - Jref; p.ref_jg_bits = 32u;
	}
	snprintf(line, sizeof line, "budget %.1f GB (%s): %.1f GB planned = %.1f GB of dictionaries, tables (reference jump table: 2^%u entries), sites and scratch", p.budget / 1e9,
	         budget_arg ? "the caller's, vg_index_open_ex" : "the device's memory less 12 GiB", p.total / 1e9, p.base / 1e9, p.ref_jg_bits);
	p.text = line;
	if (!kept.empty()) p.text += " + " + kept;
	p.text += dropped.empty() ? "; nothing left out" : "; LEFT OUT for the budget: " + dropped;
	if (p.base > p.budget) p.text += "; THE BUDGET IS BELOW THE SMALLEST LAYOUT";
	// the block holds everything but the batch slots (2 GiB reserved above, allocated with the first batches), the counters that
	// RCCL reduces in place and the clamped copy the host fetches (10 bytes per site); 64 MiB for the alignment of ~40 arrays
	p.arena = p.total - 2 * GiB - std::min<uint64_t>(sites * 10, p.total / 2) + (64ull << 20);
	// with views left out the finished index is smaller than what construction has alive at its peak (the columns beside the
	// entries, the sorts' buffers): the block then holds the permanent arrays only (vg_arena.h, set_temp_floor)
	p.limited = !dropped.empty() || (p.dx && p.dx_bits < table_bits_for(n + m, "VG_DX_BITS")) || (!getenv("VG_REF_JG_BITS") && !p.hx && p.ref_jg_bits < 32u && table_bits_for(n, "VG_REF_JG_BITS") < 32u);
	return p;
}

// From the columns (device) + the two bit vectors (host words) to the resident index.
// Device memory comes out of the handle's arena (vg_arena.h): permanent arrays with dev_alloc, temporaries as TempDev; one
// stream (ix->stream) carries every kernel, and a temporary is only given back after the stream has drained.  The ORDER below
// keeps what is alive at any moment -- permanent arrays so far + the temporaries of the step -- within the size of the finished
// index, which is the size of the arena: every column goes as soon as its last reader is done (the dictionaries' 16-byte entries
// are built early and serve the later steps in the columns' place), the radix sorts ping-pong between two buffer pairs, and
// neither wide table needs a jump table beside it (the entries carry their buckets).
static int build_on_device(vg_index *ix, DevCols &c, const ViewPlan &plan, uint64_t ref_bf_bits, const uint64_t *ref_bf_words, uint64_t snp_bf_bits, const uint64_t *snp_bf_words, PhaseClock &pc)
{
	int rc;
	DevIndex &d = ix->d;
	d.n_ref = c.n_ref; d.n_snp = c.n_snp;
	d.dx_bits = 32u; d.ref_jg_bits = 32u;
	d.ref_aux = c.ref_aux; d.snp_aux_pos = c.snp_aux_pos; d.snp_aux_info = c.snp_aux_info;
	hipStream_t st = ix->stream;
	// ---- largest position any entry names, and what a file that `vargeno index` did not write could get wrong: one wait for both
	unsigned long long maxp = 0;
	{
		TempDev<unsigned long long> dchk;                  // [0] max position, [1] k-mers out of order, [2] wild row indices, [3] rows that repeat a position
		if ((rc = dchk.alloc(4))) return rc;
		HIP_TRY(hipMemsetAsync(dchk.p, 0, 32, st));
		if (c.n_ref) vg_max_pos<<<2048, 256, 0, st>>>(c.ref_pos.p, c.ref_amb.p, c.n_ref, dchk.p);
		if (c.n_ref_aux) vg_max_pos<<<1024, 256, 0, st>>>(c.ref_aux, nullptr, c.n_ref_aux * AUX_COLS, dchk.p);
		if (c.n_snp) vg_max_pos<<<2048, 256, 0, st>>>(c.snp_pos.p, c.snp_amb.p, c.n_snp, dchk.p);
		if (c.n_snp_aux) vg_max_pos<<<1024, 256, 0, st>>>(c.snp_aux_pos, nullptr, c.n_snp_aux * AUX_COLS, dchk.p);
		if (c.n_ref) vg_check_columns<<<2048, 256, 0, st>>>(c.ref_kmer.p, c.ref_pos.p, c.ref_amb.p, c.n_ref, c.n_ref_aux, dchk.p + 1);
		if (c.n_snp) vg_check_columns<<<2048, 256, 0, st>>>(c.snp_kmer.p, c.snp_pos.p, c.snp_amb.p, c.n_snp, c.n_snp_aux, dchk.p + 1);
		if (c.n_ref_aux) vg_check_aux_rows<<<1024, 256, 0, st>>>(c.ref_aux, c.n_ref_aux, dchk.p + 3);
		if (c.n_snp_aux) vg_check_aux_rows<<<1024, 256, 0, st>>>(c.snp_aux_pos, c.n_snp_aux, dchk.p + 3);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(st));
		unsigned long long chk[4] = {0, 0, 0, 0};
		HIP_TRY(hipMemcpy(chk, dchk.p, 32, hipMemcpyDeviceToHost));
		maxp = chk[0];
		d.aux_dups = chk[3] || getenv("VG_FORCE_AUX_DUPS") ? 1u : 0u;       // (the knob: tests drive the careful path on ordinary indexes)
		if (chk[3]) {
			char msg[200];
			snprintf(msg, sizeof msg, "%llu auxiliary rows repeat a position (the SNP list holds a record several times): rows are expanded column by column", chk[3]);
			ix->aux_note = msg;
		}
		if (chk[1] || chk[2]) {
			char msg[200];
			snprintf(msg, sizeof msg, "k-mers out of order in %llu places, %llu entries naming auxiliary rows the file does not have", chk[1], chk[2]);
			return fail(VG_EIO, "not an index `vargeno index` wrote: %s", msg);
		}
	}
	// ---- bit vectors: the reference addresses bit (hash % bits); hash32 is 32 bits wide, so only the first
	//      2^32 bits of the 9.6 Gbit reference vector can ever be read (src/generate_bf.h:112-128)
	{
		const uint64_t rbits = std::min<uint64_t>(ref_bf_bits, 1ull << 32);
		uint64_t *r = nullptr, *s2 = nullptr;
		if ((rc = dev_upload(ix, &r, ref_bf_words, (rbits + 63) / 64))) return rc;
		if ((rc = dev_upload(ix, &s2, snp_bf_words, (snp_bf_bits + 63) / 64))) return rc;
		d.ref_bf = r; d.ref_bf_bits = ref_bf_bits; d.snp_bf = s2; d.snp_bf_bits = snp_bf_bits;
	}
	{
		// the plan was made before anything was allocated, with the genome length <prefix>.chrlens gives (0 without it); what the
		// dictionaries really name decides the text -- and must still fit the budget
		size_t fr = 0, tot = 0;
		if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); tot = 0; }
		const ViewPlan real = plan_views(c, maxp, ref_bf_bits, snp_bf_bits, ix->max_device_bytes, (uint64_t)tot, ix->cus);
		ix->plan_text = real.same_views(plan) ? real.text : plan.text + "; (planned with the genome length of the .chrlens file: the dictionaries name positions beyond it)";
		if (!ix->aux_note.empty()) ix->plan_text += "; " + ix->aux_note;
		if (getenv("VG_VERBOSE")) fprintf(stderr, "[vargeno_hip] %s\n", ix->plan_text.c_str());
		if (real.base > real.budget) return fail(VG_ENOMEM, "the device-memory budget is below the smallest layout of this index: %s", real.text.c_str());
	}
	const bool want_mx = plan.mx;
	// sorted pairs end up in the FIRST buffer pair, the one allocated first and therefore higher in the arena: the pair that is
	// given back is the one next to the permanent arrays
	auto sort_into_a = [&](TempDev<uint64_t> &ka, TempDev<uint64_t> &kb, TempDev<uint32_t> &va, TempDev<uint32_t> &vb, uint64_t n) -> int {
		// (the sorts' own scratch is a few MB -- they ping-pong between the two buffer pairs, vg_sort.hip -- and lives only this long: a
		// temporary that outlives its step sits in a permanent array's way)
		TempDev<uint8_t> sort_tmp;
		const size_t sort_tmp_bytes = vg_dev_sort_pairs_temp_bytes((size_t)n);
		int rc2 = sort_tmp.alloc(sort_tmp_bytes);
		if (rc2) return rc2;
		bool in_b = false;
		const int se = vg_dev_sort_pairs_u64_u32(ka.p, kb.p, va.p, vb.p, n, st, sort_tmp.p, sort_tmp_bytes, &in_b);
		if (se != 0) return fail(VG_ENODEV, "device radix sort failed: %s", hipGetErrorString((hipError_t)se));
		if (in_b && n) {
			HIP_TRY(hipMemcpyAsync(ka.p, kb.p, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
			HIP_TRY(hipMemcpyAsync(va.p, vb.p, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
		}
		kb.release(); vb.release();
		return VG_OK;
	};
	pc.lap("checks, bit vectors");
	// ---- pile-up sites (src/qv.cc:602-603, 637-659): first, while little else is resident (their position-wide scratch is 4 bytes
	//      per genome position), and the SNP dictionary's frequency columns go right after
	{
		const uint64_t plen = maxp + 64, nblk = plen / 64 + 1;
		TempDev<uint32_t> winner; TempDev<uint64_t> blk; TempDev<uint8_t> tmp;
		if ((rc = winner.alloc(plen)) || (rc = blk.alloc(nblk + 1))) return rc;
		HIP_TRY(hipMemsetAsync(winner.p, 0, plen * 4, st));
		uint8_t *dp = nullptr; ulonglong2 *dr = nullptr;
		if ((rc = dev_alloc(ix, &dp, plen))) return rc;
		if ((rc = dev_alloc(ix, &dr, nblk))) return rc;
		if (c.n_snp) vg_site_winner<<<2048, 256, 0, st>>>(c.snp_pos.p, c.snp_info.p, c.snp_amb.p, c.n_snp, winner.p);
		vg_site_blocks<<<ix->cus * 16, 256, 0, st>>>(winner.p, c.snp_kmer.p, c.snp_info.p, plen, dp, dr, blk.p);
		HIP_TRY(hipMemsetAsync(blk.p + nblk, 0, 8, st));
		HIP_TRY(hipGetLastError());
		{
			const size_t need = vg_dev_scan_temp_bytes(1, nblk + 1);
			if ((rc = tmp.alloc(need))) return rc;
			const int se = vg_dev_exclusive_scan_u64(blk.p, blk.p, nblk + 1, st, tmp.p, need);
			if (se != 0) return fail(VG_ENODEV, "device scan failed: %s", hipGetErrorString((hipError_t)se));
		}
		HIP_TRY(hipStreamSynchronize(st));
		uint64_t nsites = 0;
		HIP_TRY(hipMemcpy(&nsites, blk.p + nblk, 8, hipMemcpyDeviceToHost));
		if (nsites >= (1ull << 31)) return fail(VG_ETOOBIG, "more than 2^31 SNP sites");
		ix->n_sites = nsites;
		TempDev<uint32_t> s_pos; TempDev<uint8_t> s_ref, s_alt, s_rf, s_af;
		uint8_t *dba = nullptr;
		if ((rc = s_pos.alloc(nsites)) || (rc = s_ref.alloc(nsites)) || (rc = s_alt.alloc(nsites)) || (rc = s_rf.alloc(nsites)) || (rc = s_af.alloc(nsites))) return rc;
		if ((rc = dev_alloc(ix, &dba, nsites + 1, true))) return rc;
		vg_site_tables<<<ix->cus * 16, 256, 0, st>>>(winner.p, dp, c.snp_rf.p, c.snp_af.p, plen, dr, blk.p, s_pos.p, s_ref.p, s_alt.p, s_rf.p, s_af.p, dba);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(st));
		ix->site_pos.resize(nsites); ix->site_ref.resize(nsites); ix->site_alt.resize(nsites); ix->site_rf.resize(nsites); ix->site_af.resize(nsites);
		if (nsites) {
			HIP_TRY(hipMemcpy(ix->site_pos.data(), s_pos.p, nsites * 4, hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(ix->site_ref.data(), s_ref.p, nsites, hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(ix->site_alt.data(), s_alt.p, nsites, hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(ix->site_rf.data(), s_rf.p, nsites, hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(ix->site_af.data(), s_af.p, nsites, hipMemcpyDeviceToHost));
		}
		uint32_t *dc = nullptr, *dc4 = nullptr;
		if ((rc = dev_alloc(ix, &dc, 2 * nsites + 2, true, true))) return rc;       // (plain hipMalloc: RCCL reduces these in place)
		if ((rc = dev_alloc(ix, &dc4, 4 * nsites + 4, true))) return rc;
		d.srank = dr; d.pile = dp; d.pile_len = plen; d.cnt = dc; d.site_ba = dba; d.cnt4 = dc4;
		c.snp_rf.release(); c.snp_af.release();
	}
	pc.lap("pile-up sites");
	// ---- SNP dictionary: jump table over HI24, 16-byte entries, the strided scan's view; then its columns go (the k-mers stay for
	//      the merged view's keys, or -- an index without one -- for the HI32 jump table, if that is what the plan holds)
	{
		uint32_t *jg = nullptr; SnpEnt *ent = nullptr;
		if ((rc = dev_alloc(ix, &jg, (1ull << 24) + 1))) return rc;
		if ((rc = dev_alloc(ix, &ent, c.n_snp))) return rc;
		vg_build_jumpgate<<<(unsigned)((1ull << 24) / JG_SPAN), 256, 0, st>>>(c.snp_kmer.p, c.n_snp, jg, 1ull << 24, 40);
		vg_make_snp_entries<<<2048, 256, 0, st>>>(c.snp_kmer.p, c.snp_pos.p, c.snp_info.p, c.snp_amb.p, c.n_snp, ent);
		HIP_TRY(hipGetLastError());
		d.snp_jg = jg; d.snp = ent;
		c.snp_pos.release(); c.snp_info.release(); c.snp_amb.release();
		// the strided scan's view of the SNP dictionary: signatures (2 bytes per entry) by default, the probed values themselves
		// (8 bytes per entry) under VG_NO_SIG_VIEW, neither under VG_NO_PROBE_VIEW
		if (plan.sig) {
			uint16_t *sv = nullptr;
			if ((rc = dev_alloc(ix, &sv, c.n_snp + 16))) return rc;
			vg_make_snp_sig<<<2048, 256, 0, st>>>(c.snp_kmer.p, jg, c.n_snp, sv);
			HIP_TRY(hipGetLastError());
			d.snp_sig = sv;
		} else if (plan.probe) {
			uint64_t *pv = nullptr;
			if ((rc = dev_alloc(ix, &pv, c.n_snp + 1))) return rc;
			vg_make_snp_probe<<<2048, 256, 0, st>>>(c.snp_kmer.p, jg, c.n_snp, pv);
			HIP_TRY(hipGetLastError());
			d.snp_probe = pv;
		}
		// an index too large for the merged view and without the paired table gets a HI32 jump table of the SNP dictionary (17 GB):
		// its HI24 buckets hold ~190 entries there, 8 dependent bisection probes per look-up
		if (plan.jg32) {
			uint32_t *j32 = nullptr;
			if ((rc = dev_alloc(ix, &j32, (1ull << 32) + 1))) return rc;
			vg_build_jumpgate<<<(unsigned)((1ull << 32) / JG_SPAN), 256, 0, st>>>(c.snp_kmer.p, c.n_snp, j32, 1ull << 32, 32);
			HIP_TRY(hipGetLastError());
			d.snp_jg32 = j32;
		}
		// LO32-ordered view of the SNP dictionary: device radix sort of the swapped k-mers + a jump table over LO32's top bits
		if (plan.ssec && c.n_snp) {
			uint32_t sb = 14;
			while (sb < 30 && (1ull << sb) < c.n_snp) sb++;
			TempDev<uint64_t> ka, kb; TempDev<uint32_t> va, vb;
			if ((rc = ka.alloc(c.n_snp)) || (rc = va.alloc(c.n_snp))) return rc;
			vg_make_sec_keys<<<2048, 256, 0, st>>>(c.snp_kmer.p, c.n_snp, ka.p, va.p);
			HIP_TRY(hipGetLastError());
			if ((rc = kb.alloc(c.n_snp)) || (rc = vb.alloc(c.n_snp))) return rc;
			if ((rc = sort_into_a(ka, kb, va, vb, c.n_snp))) return rc;
			uint32_t *sjg = nullptr, *s3 = nullptr;
			if ((rc = dev_alloc(ix, &sjg, (1ull << sb) + 1))) return rc;
			if ((rc = dev_alloc(ix, &s3, 3 * c.n_snp + 4))) return rc;
			vg_build_jumpgate<<<(unsigned)((1ull << sb) / JG_SPAN), 256, 0, st>>>(ka.p, c.n_snp, sjg, 1ull << sb, (int)(64 - sb));
			vg_make_ssec3<<<2048, 256, 0, st>>>(ka.p, va.p, ent, c.n_snp, s3);
			HIP_TRY(hipGetLastError());
			HIP_TRY(hipStreamSynchronize(st));
			d.ssec3 = s3; d.ssec_jg = sjg; d.ssec_bits = sb;
		}
		if (!want_mx) c.snp_kmer.release();
	}
	pc.lap("SNP dictionary, scan view, LO32-ordered view");
	// ---- reference dictionary: 16-byte entries, and -- unless the paired HI32 table will stand in for it -- the jump table over HI32
	{
		uint32_t *jg = nullptr; RefEnt *ent = nullptr;
		if (!plan.hx) {
			// (2^32 entries -- the reference's table, qv.cc:539-584 -- for an hg38-scale dictionary; a coarse table over the top bits of
			// HI32 for a small one: DevIndex::ref_jg_bits, ref_bounds)
			const uint32_t rb = plan.ref_jg_bits;
			if ((rc = dev_alloc(ix, &jg, (1ull << rb) + 1))) return rc;
			vg_build_jumpgate<<<(unsigned)(((1ull << rb) + JG_SPAN - 1) / JG_SPAN), 256, 0, st>>>(c.ref_kmer.p, c.n_ref, jg, 1ull << rb, (int)(64 - rb));
		}
		if ((rc = dev_alloc(ix, &ent, c.n_ref))) return rc;
		vg_make_ref_entries<<<2048, 256, 0, st>>>(c.ref_kmer.p, c.ref_pos.p, c.ref_amb.p, c.n_ref, ent);
		HIP_TRY(hipGetLastError());
		d.ref_jg = jg; d.ref = ent; d.ref_jg_bits = plan.hx ? 32u : plan.ref_jg_bits;
		c.ref_pos.release(); c.ref_amb.release();
		// secondary view ordered by (LO32, HI32): device radix sort of the swapped k-mers + a jump table over LO32's top bits
		if (plan.sec) {
			uint32_t bits = 14;
			while (bits < 30 && (1ull << bits) < c.n_ref) bits++;          // ~1-3 entries per bucket
			TempDev<uint64_t> ka, kb; TempDev<uint32_t> va, vb;
			if ((rc = ka.alloc(c.n_ref)) || (rc = va.alloc(c.n_ref))) return rc;
			vg_make_sec_keys<<<2048, 256, 0, st>>>(c.ref_kmer.p, c.n_ref, ka.p, va.p);
			HIP_TRY(hipGetLastError());
			if (!want_mx) c.ref_kmer.release();                    // (the merged view's keys are its last reader otherwise)
			if ((rc = kb.alloc(c.n_ref)) || (rc = vb.alloc(c.n_ref))) return rc;
			if ((rc = sort_into_a(ka, kb, va, vb, c.n_ref))) return rc;
			uint32_t *sjg = nullptr, *sec3 = nullptr;
			if ((rc = dev_alloc(ix, &sjg, (1ull << bits) + 1))) return rc;
			if ((rc = dev_alloc(ix, &sec3, 3 * c.n_ref + 4))) return rc;
			vg_build_jumpgate<<<(unsigned)((1ull << bits) / JG_SPAN), 256, 0, st>>>(ka.p, c.n_ref, sjg, 1ull << bits, (int)(64 - bits));
			vg_make_sec3<<<2048, 256, 0, st>>>(ka.p, va.p, ent, c.n_ref, sec3);
			// the reference bit vector against the dictionary: when they name the same LO32 values, the view answers qv.cc:955 too
			unsigned long long chk[3] = {1, 0, 0};
			if (ref_bf_bits >= (1ull << 32) && !getenv("VG_NO_BF_FROM_SEC")) {
				TempDev<unsigned long long> dchk;
				if ((rc = dchk.alloc(3))) return rc;
				HIP_TRY(hipMemsetAsync(dchk.p, 0, 24, st));
				vg_sec_bf_check<<<2048, 256, 0, st>>>(ka.p, c.n_ref, d.ref_bf, dchk.p);
				vg_popcount_words<<<2048, 256, 0, st>>>(d.ref_bf, (1ull << 32) / 64, dchk.p + 2);
				HIP_TRY(hipGetLastError());
				HIP_TRY(hipStreamSynchronize(st));
				HIP_TRY(hipMemcpy(chk, dchk.p, 24, hipMemcpyDeviceToHost));
			}
			HIP_TRY(hipGetLastError());
			d.sec3 = sec3; d.sec_jg = sjg; d.sec_bits = bits;
			d.sec_is_bf = (chk[0] == 0 && chk[1] == chk[2]) ? 1u : 0u;
		}
		if (!want_mx) c.ref_kmer.release();
	}
	pc.lap("reference dictionary + LO32-ordered view");
	// ---- paired HI32 table in place of the two HI32 jump tables (an index without merged view), from the entries' bucket words
	if (plan.hx) {
		uint4 *hx = nullptr;
		if ((rc = dev_alloc(ix, &hx, (1ull << 32) + 1))) return fail(VG_ENOMEM, "no room for the paired HI32 table although the plan had it -- is the device shared?  Pass a budget (vg_index_open_ex): %s", plan.text.c_str());
		HIP_TRY(hipMemsetAsync(hx, 0, ((1ull << 32) + 1) * 16, st));
		vg_hx_sentinel<<<1, 1, 0, st>>>(hx, (uint32_t)c.n_ref, (uint32_t)c.n_snp);
		if (c.n_ref) vg_hx_fill_ref<<<ix->cus * 32, 256, 0, st>>>(d.ref, c.n_ref, hx);
		if (c.n_snp) vg_hx_fill_snp<<<ix->cus * 32, 256, 0, st>>>(d.snp, c.n_snp, hx);
		if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail(VG_ENODEV, "building the paired HI32 table failed");
		d.hx = hx;
		pc.lap("paired HI32 table");
	}
	// ---- merged exact-match view (both dictionaries behind one HI32 table); its indices are 32 bits wide
	const uint64_t nm = c.n_ref + c.n_snp;
	if (want_mx) {
		TempDev<uint64_t> ka, kb; TempDev<uint32_t> va, vb;
		if ((rc = ka.alloc(nm))) return rc;
		if (c.n_ref) HIP_TRY(hipMemcpyAsync(ka.p, c.ref_kmer.p, (size_t)c.n_ref * 8, hipMemcpyDeviceToDevice, st));
		if (c.n_snp) HIP_TRY(hipMemcpyAsync(ka.p + c.n_ref, c.snp_kmer.p, (size_t)c.n_snp * 8, hipMemcpyDeviceToDevice, st));
		c.ref_kmer.release(); c.snp_kmer.release();            // 26 GB at hg38 scale: the sort's other buffers take their place
		TempDev<uint32_t> strand_bits;
		if ((rc = va.alloc(nm)) || (rc = strand_bits.alloc(nm / 32 + 2))) return rc;
		vg_iota_u32<<<2048, 256, 0, st>>>(va.p, nm);
		HIP_TRY(hipMemsetAsync(strand_bits.p, 0, (nm / 32 + 2) * 4, st));
		vg_canon_keys<<<2048, 256, 0, st>>>(ka.p, nm, strand_bits.p);
		HIP_TRY(hipGetLastError());
		if ((rc = kb.alloc(nm)) || (rc = vb.alloc(nm))) return rc;
		if ((rc = sort_into_a(ka, kb, va, vb, nm))) return rc;     // stable: ref before snp on equal k-mers
		uint32_t *mjg = nullptr; uint4 *mx = nullptr;
		if (!plan.dx) {
			if ((rc = dev_alloc(ix, &mjg, (1ull << 32) + 1))) return rc;
			vg_build_jumpgate<<<(unsigned)((1ull << 32) / JG_SPAN), 256, 0, st>>>(ka.p, nm, mjg, 1ull << 32, 32);
		}
		if ((rc = dev_alloc(ix, &mx, nm))) return rc;
		const uint32_t dxb = plan.dx ? plan.dx_bits : 32u;
		vg_make_mx_entries<<<2048, 256, 0, st>>>(ka.p, va.p, nm, c.n_ref, d.ref, d.snp, mx, strand_bits.p, dxb);
		HIP_TRY(hipGetLastError());
		ka.release(); va.release(); strand_bits.release();
		d.mx_jg = mjg; d.mx = mx;
		// direct table (64 GiB) in place of the merged jump table (16 GiB) when the plan has the room, from the entries' bucket
		// words; no bucket may exceed the 24-bit count field (it would be a >16 M-fold repeated 16-mer)
		if (plan.dx) {
			uint4 *dx = nullptr;
			TempDev<uint32_t> big;
			if ((rc = big.alloc(1))) return rc;
			HIP_TRY(hipMemsetAsync(big.p, 0, 4, st));
			uint32_t too_big = 0;
			if ((rc = dev_alloc(ix, &dx, 1ull << dxb))) return fail(VG_ENOMEM, "no room for the direct table although the plan had it -- is the device shared?  Pass a budget (vg_index_open_ex): %s", plan.text.c_str());
			HIP_TRY(hipMemsetAsync(dx, 0, (1ull << dxb) * 16, st));
			vg_make_direct<<<ix->cus * 32, 256, 0, st>>>(mx, nm, dx, d.ref_aux, d.snp_aux_pos, big.p, dxb);
			HIP_TRY(hipGetLastError());
			HIP_TRY(hipStreamSynchronize(st));
			HIP_TRY(hipMemcpy(&too_big, big.p, 4, hipMemcpyDeviceToHost));
			if (too_big && dxb < 32u) {
				// a bucket of more than 255 entries in a table of fewer than 2^32 buckets (the mixed keys spread evenly: not seen): the entries
				// carry that table's field F, so the merged view goes with it -- the look-up takes both dictionaries' own tables instead
				dev_release(ix, dx); dx = nullptr;
				dev_release(ix, mx); mx = nullptr;
				d.mx = nullptr; d.mx_jg = nullptr;
				ix->plan_text += "; merged view + direct table not kept: a bucket of more than 255 entries";
			} else if (too_big) {
				// jump-table form after all: the 64 GiB go back, the jump table is made from the same bucket words
				dev_release(ix, dx); dx = nullptr;
				if ((rc = dev_alloc(ix, &mjg, (1ull << 32) + 1))) return rc;
				vg_jumpgate_from_buckets<<<ix->cus * 32, 256, 0, st>>>(mx, nm, mjg);
				HIP_TRY(hipGetLastError());
				d.mx_jg = mjg;
				ix->plan_text += "; direct table not kept: a bucket of more than 2^24 - 1 entries";
			} else { d.dx = dx; d.dx_bits = dxb; }
		}
		// (after the table: it reads the row form and the bucket words.  Without the table the entries keep the row form -- the
		// jump-table look-up expands rows itself -- and their bucket words, which nothing reads)
		if (d.dx) vg_inline_pairs<<<2048, 256, 0, st>>>(mx, nm, d.ref_aux, d.snp_aux_pos);
		HIP_TRY(hipGetLastError());
		pc.lap("merged view, direct table");
	}
	// ---- scratch of the lane tier, overflow counters, stats
	// VG_SCRATCH_CAP / VG_SCRATCH_KCAP shrink the per-lane scratch so tests can drive every tier
	uint32_t cap = 64, kcap = 32;
	if (const char *e = getenv("VG_SCRATCH_CAP")) cap = (uint32_t)std::max(1, atoi(e));
	if (const char *e = getenv("VG_SCRATCH_KCAP")) kcap = (uint32_t)std::max(1, atoi(e));
	if (ix->force_generic && (rc = alloc_scratch(ix, ix->mid, (uint32_t)ix->lane_grid_blocks * 256u, cap, kcap))) return rc;
	if ((rc = alloc_scratch(ix, ix->big, 64u * 64u, 16384, 2048))) return rc;
	if (!getenv("VG_NO_LATE_STORE")) {
		// the late store (vg_late_collect): 65 536 reads / 2^20 chunks between two synchronisations (VG_LATE_READS: tests fill it up)
		LateStore &ls = ix->late;
		ls.cap_reads = 1u << 16; ls.cap_chunks = 1u << 20;
		if (const char *e = getenv("VG_LATE_READS")) { ls.cap_reads = (uint32_t)std::max(1, atoi(e)); ls.cap_chunks = std::min<uint32_t>(ls.cap_chunks, 32u * ls.cap_reads); }
		if ((rc = dev_alloc(ix, &ls.kmers, (uint64_t)ls.cap_chunks + 2, false, true)) || (rc = dev_alloc(ix, &ls.meta, ls.cap_reads, false, true)) || (rc = dev_alloc(ix, &ls.offsets, (uint64_t)ls.cap_reads + 1, false, true)) ||
		    (rc = dev_alloc(ix, &ls.lost, ls.cap_reads, false, true)) || (rc = dev_alloc(ix, &ls.lost_n, 1, true, true)) || (rc = dev_alloc(ix, &ls.state, 2, true, true))) return rc;
	}
	for (Slot &sl : ix->slot) if ((rc = dev_alloc(ix, &sl.ctr, 16, true, true))) return rc;       // [0..2] spill counts, [3] invalid reads, [4],[5] work counters of the two wave tiers, [6] reads left for the per-batch lane tier
	if ((rc = dev_alloc(ix, &ix->d_clamped, 2 * ix->n_sites + 2, false, true))) return rc;
	if ((rc = dev_alloc(ix, &ix->d_fq, 1, true, true))) return rc;
	if ((rc = dev_alloc(ix, &ix->d_stats, S_COUNT, true, true))) return rc;
	HIP_TRY(hipStreamSynchronize(st));
#ifdef VG_VIEW_COUNTERS
	{
		VcRange r[24]; int nr = 0;
		auto add = [&](const void *p, uint64_t bytes, int id) { if (p && nr < 24) { r[nr].lo = (unsigned long long)p; r[nr].hi = r[nr].lo + bytes; r[nr].id = id; r[nr].pad = 0; nr++; } };
		add(d.dx, (1ull << d.dx_bits) * 16, VC_DX); add(d.mx, nm * 16, VC_MX); add(d.ref_jg, ((1ull << d.ref_jg_bits) + 1) * 4, VC_REF_JG); add(d.ref, c.n_ref * 16, VC_REF);
		add(d.snp_jg, ((1ull << 24) + 1) * 4, VC_SNP_JG); add(d.snp, c.n_snp * 16, VC_SNP); add(d.sec_jg, ((1ull << d.sec_bits) + 1) * 4, VC_SEC_JG); add(d.sec3, c.n_ref * 12 + 16, VC_SEC3);
		add(d.snp_sig, (c.n_snp + 16) * 2, VC_SIG); add(d.ref_bf, 1ull << 29, VC_REF_BF); add(d.snp_bf, (d.snp_bf_bits + 7) / 8, VC_SNP_BF);
		add(d.ref_aux, c.n_ref_aux * 40, VC_AUX_ANY); add(d.snp_aux_pos, c.n_snp_aux * 40, VC_AUX_ANY); add(d.snp_aux_info, c.n_snp_aux * 10, VC_AUX_ANY);
		add(d.pile, d.pile_len, VC_PILE_ANY); add(d.srank, (d.pile_len / 64 + 1) * 16, VC_SRANK); add(d.cnt4, (4 * ix->n_sites + 4) * 4, VC_CNT4);
		HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(vg_vc_ranges), r, sizeof r));
		HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(vg_vc_nranges), &nr, sizeof nr));
	}
#endif
	return VG_OK;
}

// Before anything is allocated: the plan (which views the budget buys) and the handle's one block of device memory, sized for the
// finished index.  maxp_est: the largest genome position, as far as it is known up front (0: unknown -- the arrays that are one
// byte or so per position then find no room in the block and are hipMalloc'ed beside it).
static int plan_and_arena(vg_index *ix, const DevCols &c, uint64_t maxp_est, uint64_t ref_bf_bits, uint64_t snp_bf_bits, ViewPlan &plan)
{
	if (c.n_ref >= 0xFFFFFFFFull || c.n_snp >= 0xFFFFFFFFull) return fail(VG_ETOOBIG, "dictionary too large (limit: 2^32 32-mers)");
	if (ref_bf_bits == 0 || snp_bf_bits == 0) return fail(VG_EINVAL, "empty bit vector");
	size_t fr = 0, tot = 0;
	if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); tot = 0; }
	if (tot == 0 && ix->max_device_bytes == 0) return fail(VG_ENODEV, "hipMemGetInfo failed and no device-memory budget was given (vg_index_open_ex): the views cannot be planned");
	plan = plan_views(c, maxp_est, ref_bf_bits, snp_bf_bits, ix->max_device_bytes, (uint64_t)tot, ix->cus);
	if (plan.base > plan.budget) return fail(VG_ENOMEM, "the device-memory budget is below the smallest layout of this index: %s", plan.text.c_str());
	// VG_NO_ARENA=1: every buffer its own hipMalloc / hipFree, as through round 4 (A/B runs).  A block that cannot be had (somebody
	// else holds the memory) is not an error here: the individual allocations will say so if they fail too.
	if (!getenv("VG_NO_ARENA")) {
		const double t0 = now_s();
		bool use_block = true;
		if (plan.limited) {
			// A plan with views left out: the block holds the permanent arrays only and the construction's temporaries (the columns
			// beside their entries, a sort's two buffer pairs) are taken from the device BESIDE it and given back -- that keeps the
			// finished handle within its budget, but needs the room: block + temporaries at their peak.  When the plan is limited by
			// the DEVICE itself (the default budget on a part smaller than the full layout) that room does not exist, and the open
			// used to fail in its staging allocations instead of building the smaller layout it had planned (round 5's advisor): then
			// no block at all -- every buffer its own allocation, so that what is alive at any moment is only what construction needs
			// at that moment (slower: memory that has just been freed is cleared at allocation, vg_arena.h).
			const uint64_t nm = plan.mx ? c.n_ref + c.n_snp : c.n_ref;
			const uint64_t temps = 13 * c.n_ref + 16 * c.n_snp + 24 * nm + (1ull << 30);
			if ((uint64_t)fr < plan.arena + temps) {
				use_block = false;
				if (getenv("VG_VERBOSE")) fprintf(stderr, "[vargeno_hip] %.1f GB free, block %.1f GB + temporaries %.1f GB would not fit: no block, one allocation per buffer\n", fr / 1e9, plan.arena / 1e9, temps / 1e9);
			}
		}
		if (use_block) {
			(void)ix->arena.init(plan.arena);
			if (plan.limited) ix->arena.set_temp_floor(UINT64_MAX);
		}
		g_alloc_s += now_s() - t0;
	}
	return VG_OK;
}

// while a handle is under construction on this thread its temporaries come out of its arena
struct Building {
	explicit Building(vg_index *ix) { g_building = ix; }
	~Building() { g_building = nullptr; }
};
// construction is over (every temporary has been given back): the handle's size is final
static void finish_construction(vg_index *ix)
{
	uint64_t plain = 0;
	for (const auto &kv : ix->owned_bytes) plain += kv.second;
	ix->dev_bytes = plain + ix->arena.size();
	char line[400];
	if (ix->arena.ready())
		snprintf(line, sizeof line, "; memory: one block of %.1f GB (%.1f GB of it in use now, at most %.1f GB during construction; %llu requests = %.1f GB did not fit and went to hipMalloc) + %.1f GB of hipMalloc'ed buffers",
		         ix->arena.size() / 1e9, ix->arena.in_use() / 1e9, ix->arena.peak() / 1e9, (unsigned long long)ix->arena_misses, ix->arena_miss_bytes / 1e9, plain / 1e9);
	else snprintf(line, sizeof line, "; memory: no arena (hipMalloc / hipFree per buffer), %.1f GB", plain / 1e9);
	ix->open_report += line;
	if (getenv("VG_VERBOSE")) fprintf(stderr, "[vargeno_hip] %s\n", line + 2);
}

// $VG_MAX_DEVICE_BYTES: the budget of vg_index_open / vg_index_create (vg_index_open_ex takes it as an argument)
static uint64_t env_budget()
{
	const char *e = getenv("VG_MAX_DEVICE_BYTES");
	return e && *e ? strtoull(e, nullptr, 10) : 0ull;
}

static int create_impl(const vg_index_arrays *a, int device, vg_index *ix)
{
	int rc = init_handle(ix, device);
	if (rc) return rc;
	Building guard(ix);
	PhaseClock pc(ix);
	DevCols c;
	c.n_ref = a->n_ref; c.n_ref_aux = a->n_ref_aux; c.n_snp = a->n_snp; c.n_snp_aux = a->n_snp_aux;
	// the largest position the arrays name (the loader's own rule, vg_max_pos): it sizes the arrays that are per genome position
	uint64_t maxp_est = 0;
	{
		auto scan = [&](const uint32_t *pos, const uint8_t *amb, uint64_t n) { for (uint64_t i = 0; i < n; i++) if ((!amb || amb[i] == 0) && pos[i] != POS_AMBIGUOUS && pos[i] > maxp_est) maxp_est = pos[i]; };
		scan(a->ref_pos, a->ref_amb, a->n_ref); scan(a->ref_aux, nullptr, a->n_ref_aux * AUX_COLS);
		scan(a->snp_pos, a->snp_amb, a->n_snp); scan(a->snp_aux_pos, nullptr, a->n_snp_aux * AUX_COLS);
	}
	ViewPlan plan;
	if ((rc = plan_and_arena(ix, c, maxp_est, a->ref_bf_bits, a->snp_bf_bits, plan))) return rc;
	if ((rc = c.ref_kmer.upload(a->ref_kmer, a->n_ref)) || (rc = c.ref_pos.upload(a->ref_pos, a->n_ref)) || (rc = c.ref_amb.upload(a->ref_amb, a->n_ref))) return rc;
	if ((rc = c.snp_kmer.upload(a->snp_kmer, a->n_snp)) || (rc = c.snp_pos.upload(a->snp_pos, a->n_snp)) || (rc = c.snp_info.upload(a->snp_info, a->n_snp)) ||
	    (rc = c.snp_amb.upload(a->snp_amb, a->n_snp)) || (rc = c.snp_rf.upload(a->snp_rf, a->n_snp)) || (rc = c.snp_af.upload(a->snp_af, a->n_snp))) return rc;
	if ((rc = dev_upload(ix, &c.ref_aux, a->ref_aux, a->n_ref_aux * AUX_COLS))) return rc;
	if ((rc = dev_upload(ix, &c.snp_aux_pos, a->snp_aux_pos, a->n_snp_aux * AUX_COLS))) return rc;
	if ((rc = dev_upload(ix, &c.snp_aux_info, a->snp_aux_info, a->n_snp_aux * AUX_COLS))) return rc;
	pc.lap("block allocated, columns copied to the device");
	return build_on_device(ix, c, plan, a->ref_bf_bits, a->ref_bf_words, a->snp_bf_bits, a->snp_bf_words, pc);
}

extern "C" int vg_index_create(const vg_index_arrays *a, int device, vg_index **out)
{
	if (!a || !out) return fail(VG_EINVAL, "null argument");
	*out = nullptr;
	vg_index *ix = new (std::nothrow) vg_index();
	if (!ix) return fail(VG_ENOMEM, "host allocation failed");
	ix->max_device_bytes = env_budget();
	int rc = guarded([&] { return create_impl(a, device, ix); });
	if (rc) { vg_index_close(ix); return rc; }
	finish_construction(ix);
	*out = ix;
	return VG_OK;
}

// ---- index files (formats: SURVEY.md §8f-1; writers src/dictgen.c:63-275, sdsl int_vector.hpp:1563-1595)
// A byte range of a file -> device memory: reader threads pread() pieces into a ring of pinned buffers, each piece is copied up
// as soon as it is complete (the page cache / an NVMe array serve the threads concurrently; one fread of the 43 GB hg38
// dictionary is a single memcpy stream, and a copy from pageable memory is staged a second time by the runtime).
// the page-locked staging of the file reads: made once per vg_index_open (eight 64 MiB buffers take 0.12 s to pin and 0.08 s to
// release, tools/alloc_probe), shared by both dictionary files
struct FileRing {
	static constexpr uint64_t PIECE = 64ull << 20;
	static constexpr int NBUF = 8;
	uint8_t *buf[NBUF] = {};
	hipEvent_t copied[NBUF] = {};
	hipStream_t cs = nullptr;
	int init()
	{
		for (int i = 0; i < NBUF; i++) {
			if (hipHostMalloc((void **)&buf[i], PIECE, hipHostMallocDefault) != hipSuccess) return fail(VG_ENOMEM, "hipHostMalloc(staging) failed");
			if (hipEventCreateWithFlags(&copied[i], hipEventDisableTiming) != hipSuccess) return fail(VG_ENODEV, "hipEventCreate failed");
		}
		if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) return fail(VG_ENODEV, "hipStreamCreate failed");
		return VG_OK;
	}
	~FileRing()
	{
		for (auto p : buf) if (p) (void)hipHostFree(p);
		for (auto e : copied) if (e) (void)hipEventDestroy(e);
		if (cs) (void)hipStreamDestroy(cs);
	}
};
static int file_to_device(FileRing &R, int fd, uint64_t off, uint64_t bytes, uint8_t *dst, const std::string &path)
{
	if (bytes == 0) return VG_OK;
	constexpr uint64_t PIECE = FileRing::PIECE;
	constexpr int NBUF = FileRing::NBUF;
	const uint64_t n_pieces = (bytes + PIECE - 1) / PIECE;
	int rc = VG_OK;
	std::mutex mu; std::condition_variable cv;
	std::vector<uint8_t> state(n_pieces, 0);                    // 1 = read into its ring slot
	uint64_t issued = 0;                                        // pieces whose copy has been enqueued AND whose slot is free again once copied[slot] fires
	uint64_t freed = 0;                                         // pieces known to have left their ring slot
	std::atomic<uint64_t> next{0};
	bool io_error = false;
	const unsigned nt = std::min<unsigned>(host_threads(), 16u);
	std::vector<std::thread> readers;
	for (unsigned t = 0; t < nt; t++) readers.emplace_back([&] {
		for (;;) {
			const uint64_t p = next.fetch_add(1);
			if (p >= n_pieces) return;
			{ std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return p < freed + (uint64_t)NBUF || io_error; }); if (io_error) return; }
			const uint64_t o = p * PIECE, n = std::min(PIECE, bytes - o);
			uint64_t done = 0;
			while (done < n) {
				const ssize_t g = pread(fd, R.buf[p % NBUF] + done, (size_t)(n - done), (off_t)(off + o + done));
				if (g <= 0) break;
				done += (uint64_t)g;
			}
			std::lock_guard<std::mutex> g(mu);
			if (done < n) io_error = true;
			state[p] = 1;
			cv.notify_all();
		}
	});
	for (uint64_t p = 0; p < n_pieces; p++) {
		{ std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return state[p] != 0 || io_error; }); if (io_error) break; }
		const uint64_t o = p * PIECE, n = std::min(PIECE, bytes - o);
		if (hipMemcpyAsync(dst + o, R.buf[p % NBUF], n, hipMemcpyHostToDevice, R.cs) != hipSuccess || hipEventRecord(R.copied[p % NBUF], R.cs) != hipSuccess) {
			std::lock_guard<std::mutex> g(mu); io_error = true; rc = fail(VG_ENODEV, "host-to-device copy failed"); cv.notify_all(); break;
		}
		issued = p + 1;
		// the slot of piece p - NBUF + 1 .. is reusable once its copy has finished: wait for the oldest outstanding one when the ring is full
		if (issued - freed >= (uint64_t)NBUF - 1) {
			(void)hipEventSynchronize(R.copied[freed % NBUF]);
			std::lock_guard<std::mutex> g(mu); freed++; cv.notify_all();
		}
	}
	(void)hipStreamSynchronize(R.cs);
	{ std::lock_guard<std::mutex> g(mu); freed = n_pieces + NBUF; cv.notify_all(); }
	for (auto &t : readers) t.join();
	if (rc) return rc;
	if (io_error) return fail(VG_EIO, "short read on %s", path.c_str());
	return VG_OK;
}

static int read_bf(const std::string &path, uint64_t cap_bits, uint64_t &bits, std::vector<uint64_t> &words)
{
	FILE *f = fopen(path.c_str(), "rb");
	if (!f) return fail(VG_EIO, "cannot open %s", path.c_str());
	if (fread(&bits, 8, 1, f) != 1) { fclose(f); return fail(VG_EIO, "short read on %s", path.c_str()); }
	// sdsl int_vector<1> (int_vector.hpp:1563-1595): u64 bit count, then ceil(bits / 64) words.  A header that does not match the
	// file's size is a corrupt file: the kernels index the words with (hash % bits) >> 6
	{
		struct stat sb;
		const uint64_t words_all = bits / 64 + (bits % 64 != 0);
		if (fstat(fileno(f), &sb) != 0 || bits == 0 || words_all > ((uint64_t)sb.st_size - 8) / 8 || (uint64_t)sb.st_size != 8 + 8 * words_all) {
			fclose(f);
			return fail(VG_EIO, "%s: size does not match its header", path.c_str());
		}
	}
	const uint64_t capped = std::min(bits, cap_bits);
	const uint64_t nw = capped / 64 + (capped % 64 != 0);
	words.resize(nw);
	const size_t got = nw ? fread(words.data(), 8, nw, f) : 0;
	fclose(f);
	if (got != nw) return fail(VG_EIO, "short read on %s", path.c_str());
	return VG_OK;
}

struct Fd { int fd = -1; ~Fd() { if (fd >= 0) close(fd); } };

static int open_impl(const char *prefix, int device, vg_index *ix)
{
	const std::string pre(prefix);
	int rc;
	Fd rf, sf;
	rf.fd = open((pre + ".ref.dict").c_str(), O_RDONLY);
	if (rf.fd < 0) return fail(VG_EIO, "cannot open %s.ref.dict", prefix);
	sf.fd = open((pre + ".snp.dict").c_str(), O_RDONLY);
	if (sf.fd < 0) return fail(VG_EIO, "cannot open %s.snp.dict", prefix);
	struct stat rs, ss;
	if (fstat(rf.fd, &rs) != 0 || fstat(sf.fd, &ss) != 0) return fail(VG_EIO, "cannot stat the dictionary files of %s", prefix);
	const uint64_t rsize = (uint64_t)rs.st_size, ssize = (uint64_t)ss.st_size;
	if (rsize < 16 || ssize < 16) return fail(VG_EIO, "dictionary file too short: %s", prefix);
	uint64_t rh[2], sh[2];
	if (pread(rf.fd, rh, 16, 0) != 16 || pread(sf.fd, sh, 16, 0) != 16) return fail(VG_EIO, "short read on the dictionary files of %s", prefix);
	const uint64_t n_ref = rh[0], n_ref_aux = rh[1], n_snp = sh[0], n_snp_aux = sh[1];
	if (n_ref > (1ull << 32) || n_snp > (1ull << 32)) return fail(VG_ETOOBIG, "dictionary too large (limit: 2^32 32-mers)");
	// every count is bounded by the file it came from before it is multiplied (a corrupt header must not wrap the size check)
	if (n_ref > rsize / 13 || n_ref_aux > rsize / 40 || rsize != 16 + 13 * n_ref + 40 * n_ref_aux) return fail(VG_EIO, "%s.ref.dict: size does not match its header", prefix);
	if (n_snp > ssize / 16 || n_snp_aux > ssize / 78 || ssize != 16 + 16 * n_snp + 78 * n_snp_aux) return fail(VG_EIO, "%s.snp.dict: size does not match its header", prefix);
	uint64_t rbits = 0, sbits = 0;
	std::vector<uint64_t> rw, sw;
	const double t_bf0 = now_s();
	if ((rc = read_bf(pre + ".ref.bf", 1ull << 32, rbits, rw))) return rc;
	if ((rc = read_bf(pre + ".snp.bf", ~0ull, sbits, sw))) return rc;
	const double t_bf1 = now_s();
	if ((rc = init_handle(ix, device))) return rc;
	Building guard(ix);
	PhaseClock pc(ix);
	{
		char line[160];
		snprintf(line, sizeof line, "bit-vector files read %.2f s; device, streams, events %.2f s", t_bf1 - t_bf0, now_s() - t_bf1);
		ix->open_report = line;
		if (pc.on) fprintf(stderr, "[vargeno_hip] %s\n", line);
	}
	DevCols c;
	c.n_ref = n_ref; c.n_ref_aux = n_ref_aux; c.n_snp = n_snp; c.n_snp_aux = n_snp_aux;
	// the genome's length, from <prefix>.chrlens when `vargeno index` left one ("name length" per line): positions are 1-based over
	// the concatenated sequences, so their sum bounds every position the dictionaries name
	uint64_t maxp_est = 0;
	if (FILE *f = fopen((pre + ".chrlens").c_str(), "r")) {
		char name[256]; unsigned long long len = 0;
		while (fscanf(f, "%255s %llu", name, &len) == 2) maxp_est += len;
		fclose(f);
		if (maxp_est > (1ull << 32)) maxp_est = 0;
	}
	ViewPlan plan;
	if ((rc = plan_and_arena(ix, c, maxp_est, rbits, sbits, plan))) return rc;
	if ((rc = c.alloc())) return rc;
	if ((rc = dev_alloc(ix, &c.ref_aux, n_ref_aux * AUX_COLS))) return rc;
	if ((rc = dev_alloc(ix, &c.snp_aux_pos, n_snp_aux * AUX_COLS))) return rc;
	if ((rc = dev_alloc(ix, &c.snp_aux_info, n_snp_aux * AUX_COLS))) return rc;
	{
	FileRing ring;
	if ((rc = ring.init())) return rc;
	{
		// the files' bytes, as they are, next to the columns they unpack into (the larger one first; freed right after)
		TempDev<uint8_t> raw;
		if ((rc = raw.alloc(rsize - 16 + 64))) return rc;
		if ((rc = file_to_device(ring, rf.fd, 16, rsize - 16, raw.p, pre + ".ref.dict"))) return rc;
		vg_unpack_ref<<<4096, 256, 0, ix->stream>>>(raw.p, n_ref, c.ref_kmer.p, c.ref_pos.p, c.ref_amb.p);
		if (n_ref_aux) vg_unpack_ref_aux<<<1024, 256, 0, ix->stream>>>(raw.p + 13 * n_ref, n_ref_aux * AUX_COLS, c.ref_aux);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(ix->stream));
	}
	{
		TempDev<uint8_t> raw;
		if ((rc = raw.alloc(ssize - 16 + 64))) return rc;
		if ((rc = file_to_device(ring, sf.fd, 16, ssize - 16, raw.p, pre + ".snp.dict"))) return rc;
		vg_unpack_snp<<<4096, 256, 0, ix->stream>>>(raw.p, n_snp, c.snp_kmer.p, c.snp_pos.p, c.snp_info.p, c.snp_amb.p, c.snp_rf.p, c.snp_af.p);
		if (n_snp_aux) vg_unpack_snp_aux<<<1024, 256, 0, ix->stream>>>(raw.p + 16 * n_snp, n_snp_aux, c.snp_aux_pos, c.snp_aux_info);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(ix->stream));
	}
	}
	pc.lap("block allocated, dictionary files read, copied up, unpacked");
	return build_on_device(ix, c, plan, rbits, rw.data(), sbits, sw.data(), pc);
}

extern "C" int vg_index_open_ex(const char *prefix, int device, uint64_t max_device_bytes, vg_index **out)
{
	if (!prefix || !out) return fail(VG_EINVAL, "null argument");
	*out = nullptr;
	vg_index *ix = new (std::nothrow) vg_index();
	if (!ix) return fail(VG_ENOMEM, "host allocation failed");
	ix->max_device_bytes = max_device_bytes;
	const double t0 = now_s();
	int rc = guarded([&] { return open_impl(prefix, device, ix); });
	if (rc) { vg_index_close(ix); return rc; }
	finish_construction(ix);
	char line[80];
	snprintf(line, sizeof line, "; vg_index_open %.2f s in all", now_s() - t0);
	ix->open_report += line;
	*out = ix;
	return VG_OK;
}

extern "C" const char *vg_index_plan(const vg_index *ix) { return ix ? ix->plan_text.c_str() : ""; }
extern "C" const char *vg_index_open_report(const vg_index *ix) { return ix ? ix->open_report.c_str() : ""; }

extern "C" int vg_index_open(const char *prefix, int device, vg_index **out)
{
	return vg_index_open_ex(prefix, device, env_budget(), out);
}
extern "C" uint64_t vg_index_device_bytes(const vg_index *ix) { return ix ? ix->dev_bytes : 0; }
extern "C" uint64_t vg_num_sites(const vg_index *ix) { return ix ? ix->n_sites : 0; }
extern "C" uint32_t vg_index_views(const vg_index *ix)
{
	if (!ix) return 0;
	const DevIndex &d = ix->d;
	return (d.sec3 ? VG_VIEW_SEC : 0u) | (d.sec_is_bf ? VG_VIEW_SEC_IS_BF : 0u) | (d.mx ? VG_VIEW_MX : 0u) | (d.dx ? VG_VIEW_DX : 0u) | (d.snp_probe ? VG_VIEW_SNP_PROBE : 0u) | (d.snp_jg32 ? VG_VIEW_SNP_JG32 : 0u) | (d.hx ? VG_VIEW_HX : 0u) | (d.snp_sig ? VG_VIEW_SNP_SIG : 0u) | (d.ssec3 ? VG_VIEW_SSEC : 0u);
}

// ------------------------------------------------------------------------------------------------
// read batches
// ------------------------------------------------------------------------------------------------
static int harvest(vg_index *ix, Slot &sl)
{
	if (!sl.busy) return VG_OK;
	HIP_TRY(hipEventSynchronize(sl.e3));
	// reads left for the per-batch lane tier: what the late store did not take (ctr[6]), or -- VG_FORCE_GENERIC, VG_NO_LATE_STORE -- all of listB
	const uint32_t resid = sl.lt_late ? sl.h_ctr[6] : sl.h_ctr[1];
	uint32_t *const r_list = sl.lt_late ? sl.listA : sl.listB, *const r_cnt = sl.lt_late ? &sl.ctr[6] : &sl.ctr[1];
	if (resid) ix->lane_tier_seen = true;
	if (resid && !sl.lt_enqueued) {
		// the lane machine with its lists in HBM finishes them now
		if (sl.lt_stats) vg_lane_kernel<true><<<ix->big.s.nlanes / 64, 64, 0, ix->tail2>>>(ix->d, ix->big.s, sl.lt_bases, sl.lt_quals, sl.lt_offsets, 0, r_list, r_cnt, sl.listC, &sl.ctr[2], ix->d_stats, nullptr, sl.lt_gate, sl.pk_kmer, sl.pk_meta, sl.lt_packed);
		else vg_lane_kernel<false><<<ix->big.s.nlanes / 64, 64, 0, ix->tail2>>>(ix->d, ix->big.s, sl.lt_bases, sl.lt_quals, sl.lt_offsets, 0, r_list, r_cnt, sl.listC, &sl.ctr[2], ix->d_stats, nullptr, sl.lt_gate, sl.pk_kmer, sl.pk_meta, sl.lt_packed);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipMemcpyAsync(sl.h_ctr, sl.ctr, 32, hipMemcpyDeviceToHost, ix->tail2));
		HIP_TRY(hipStreamSynchronize(ix->tail2));
	}
	for (int i = 0; i < 4; i++) ix->cum[i] += sl.h_ctr[i];
	// the deep tier's grid follows the lists it has been getting (enqueue_batch): the larger of this batch's and half the hint before
	ix->spill_hint = std::max<uint32_t>(sl.h_ctr[0], ix->spill_hint / 2);
	ix->spill_known = true;
	float a = 0, b = 0, c = 0, t = 0;
	HIP_TRY(hipEventElapsedTime(&a, sl.e0, sl.e1)); HIP_TRY(hipEventElapsedTime(&b, sl.e5, sl.e2));
	float w2 = 0;
	HIP_TRY(hipEventElapsedTime(&c, sl.e2, sl.e3)); HIP_TRY(hipEventElapsedTime(&t, sl.e0, sl.e3)); HIP_TRY(hipEventElapsedTime(&w2, sl.e2, sl.e4));
	ix->t_pack += a; ix->t_main += b; ix->t_tail += c; ix->t_total += t; ix->t_w2 += w2; ix->t_batches++;
	sl.busy = false;
	return VG_OK;
}

// The late store's one lane-machine run (see vg_late_collect): every stream of the handle is idle when this is called.
static int run_late_store(vg_index *ix)
{
	if (!ix->late_dirty || !ix->late.state) return VG_OK;
	unsigned long long st[2] = {0, 0};
	HIP_TRY(hipMemcpy(st, ix->late.state, 16, hipMemcpyDeviceToHost));
	ix->late_dirty = false;
	const uint64_t n = st[1];
	if (st[0] == 0) return VG_OK;
	if (n) {
		if (ix->late_stats) vg_lane_kernel<true><<<ix->big.s.nlanes / 64, 64, 0, ix->tail>>>(ix->d, ix->big.s, nullptr, nullptr, ix->late.offsets, n, nullptr, nullptr, ix->late.lost, ix->late.lost_n, ix->d_stats, nullptr, nullptr, ix->late.kmers, ix->late.meta, true);
		else vg_lane_kernel<false><<<ix->big.s.nlanes / 64, 64, 0, ix->tail>>>(ix->d, ix->big.s, nullptr, nullptr, ix->late.offsets, n, nullptr, nullptr, ix->late.lost, ix->late.lost_n, ix->d_stats, nullptr, nullptr, ix->late.kmers, ix->late.meta, true);
		HIP_TRY(hipGetLastError());
	}
	uint32_t lost = 0;
	HIP_TRY(hipMemcpyAsync(&lost, ix->late.lost_n, 4, hipMemcpyDeviceToHost, ix->tail));
	HIP_TRY(hipMemsetAsync(ix->late.state, 0, 16, ix->tail));
	HIP_TRY(hipMemsetAsync(ix->late.lost_n, 0, 4, ix->tail));
	HIP_TRY(hipStreamSynchronize(ix->tail));
	ix->cum[2] += lost;
	ix->late_reads_run += n;
	return VG_OK;
}

// Drain both streams and turn "a read outgrew even the deep scratch" into an error code.
static int finish_pending(vg_index *ix)
{
	HIP_TRY(hipSetDevice(ix->device));
	HIP_TRY(hipStreamSynchronize(ix->stream));
	HIP_TRY(hipStreamSynchronize(ix->tail));
	HIP_TRY(hipStreamSynchronize(ix->tail2));
	// (the FASTQ stream's own work -- vg_fastq_stream_begin's reset of the stream state included -- is otherwise only ordered
	// before the batches it produced: an empty stream has none)
	if (ix->ingest) HIP_TRY(hipStreamSynchronize(ix->ingest));
	for (Slot &sl : ix->slot) { int rc = harvest(ix, sl); if (rc) return rc; }
	{ int rc = run_late_store(ix); if (rc) return rc; }
	if (ix->cnt4_dirty && ix->n_sites) {
		vg_fold_counters<<<(unsigned)std::min<uint64_t>((ix->n_sites + 255) / 256, 4096), 256, 0, ix->stream>>>(ix->d.cnt4, ix->d.site_ba, ix->d.cnt, ix->n_sites);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(ix->stream));
	}
	ix->cnt4_dirty = false;
	if (ix->cum[2]) return fail(VG_ENOMEM, "a read produced more hit contexts than the deep scratch holds (the reference overruns MAX_HITS=2000 long before, qv.cc:709)");
	return VG_OK;
}

// One batch = pack -> wave tier on the main stream, then lane tier (mid scratch) -> lane tier (deep scratch)
// on the tail stream; the list launches size themselves from device counters, so nothing waits for the host.
// n_reads: the batch's size, or (d_n_reads given) an upper bound of the size the device holds at d_n_reads
// packed: the batch is already packed (sl.pk_kmer / sl.pk_meta hold it, d_offsets = 32 x chunks before each read): no pack kernel
template <bool STATS>
static int enqueue_batch(vg_index *ix, Slot &sl, const uint8_t *d_bases, const uint8_t *d_quals, const uint32_t *d_gate, const uint64_t *d_offsets, uint64_t n_reads, hipStream_t produced_on, const uint32_t *d_n_reads, const bool packed, const uint64_t total_bases)
{
	uint32_t *ctr = sl.ctr;
	const unsigned g1 = (unsigned)std::min<uint64_t>((n_reads + 255) / 256, (uint64_t)ix->lane_grid_blocks);
	if (!ix->force_generic) {
		// main stream: pack, then the wave tier.  (Packing batch k+1 on a third stream under batch k's wave kernel was measured
		// with 8 M-read batches and lost 12 %: two co-scheduled kernels split the CUs.  With batches of a million reads the
		// launch gaps between the dependent kernels of one stream weigh more than that: chr22-scale, 1 M reads per step,
		// 0.384 -> 0.352 ms per step (profiles/ab_chr22_pack_overlap_r05.txt) -- so small batches do overlap.)
		hipStream_t ps = ix->stream;
		if (ix->ingest && (ix->pack_overlap == 1 || (ix->pack_overlap < 0 && n_reads <= (2u << 20)))) ps = ix->ingest;
		if (produced_on && produced_on != ps) {                     // a batch gathered by the FASTQ framing on the ingest stream
			HIP_TRY(hipEventRecord(sl.e_in, produced_on));
			HIP_TRY(hipStreamWaitEvent(ps, sl.e_in, 0));
		}
		HIP_TRY(hipMemsetAsync(ctr, 0, 64, ps));
		HIP_TRY(hipEventRecord(sl.e0, ps));
		static const int pack_bpc = getenv("VG_PACK_BPC") ? std::max(1, atoi(getenv("VG_PACK_BPC"))) : 16;          // workgroups (of PACK_WPB tiles at a time) per CU
		// reads per tile of the pack kernel: as many as the mean read length lets fit its LDS stream (64 up to 160 bases)
		const uint64_t mean_len = n_reads ? (total_bases + n_reads - 1) / n_reads : 1;
		const uint32_t pack_tr = (uint32_t)std::max<uint64_t>(4, std::min<uint64_t>(PACK_T, (uint64_t)PACK_T * PACK_MAXLEN / std::max<uint64_t>(mean_len, 1)));
		const unsigned pgrid = (unsigned)std::min<uint64_t>((n_reads + (uint64_t)pack_tr * PACK_WPB - 1) / ((uint64_t)pack_tr * PACK_WPB), (uint64_t)ix->cus * pack_bpc);
		const bool fused = VG_FUSE_PACK && !packed && !getenv("VG_NO_FUSE");          // the main tier encodes the reads itself (experiment)
		const FuseIn fin{fused ? d_bases : nullptr, d_quals, d_gate, &ctr[3]}, nofuse{nullptr, nullptr, nullptr, nullptr};
		if (!packed && !fused) vg_pack_kernel<<<pgrid, PACK_T * PACK_WPB, 0, ps>>>(d_bases, d_quals, d_offsets, n_reads, sl.pk_kmer, sl.pk_meta, &ctr[3], d_n_reads, d_gate, pack_tr);
		HIP_TRY(hipEventRecord(sl.e1, ps));
		if (ps != ix->stream) HIP_TRY(hipStreamWaitEvent(ix->stream, sl.e1, 0));
		// The previous batch's deep-list tier (tail stream) runs under this batch's pack kernel and, for what is left of it,
		// under the head of this batch's wave kernel: its few single-wave workgroups drain while the main tier pulls its work
		// dynamically, which costs less than holding the wave kernel back for them (0.71 -> 0.66 ms per 1 M-read step).
		HIP_TRY(hipEventRecord(sl.e5, ix->stream));               // the wave kernel's own start
		ix->cnt4_dirty = true;
		const unsigned wgrid = (unsigned)std::min<uint64_t>((n_reads + 64 * W1_WPB - 1) / (64 * W1_WPB), (uint64_t)ix->wave_grid / W1_WPB);
		const bool big = !STATS && ix->d.mx == nullptr;                // an index without the merged view: the kernel built for it
		const bool sdx = !STATS && !big && ix->d.dx != nullptr && ix->d.dx_bits < 32u;      // a direct table of fewer than 2^32 buckets: the instantiation that compares (F, lo32)
		if (big) vg_wave_kernel_big<W1_ECAP, W1_NCAP, W1_WPB><<<wgrid, 64 * W1_WPB, 0, ix->stream>>>(ix->d, sl.pk_kmer, sl.pk_meta, d_offsets, n_reads, nullptr, d_n_reads, sl.listA, &ctr[0], &ctr[4], ix->work_chunk, ix->d_stats, fin);
		else if (sdx) vg_wave_kernel<false, W1_ECAP, W1_NCAP, W1_WPB, true><<<wgrid, 64 * W1_WPB, 0, ix->stream>>>(ix->d, sl.pk_kmer, sl.pk_meta, d_offsets, n_reads, nullptr, d_n_reads, sl.listA, &ctr[0], &ctr[4], ix->work_chunk, ix->d_stats, fin);
		else vg_wave_kernel<STATS, W1_ECAP, W1_NCAP, W1_WPB><<<wgrid, 64 * W1_WPB, 0, ix->stream>>>(ix->d, sl.pk_kmer, sl.pk_meta, d_offsets, n_reads, nullptr, d_n_reads, sl.listA, &ctr[0], &ctr[4], ix->work_chunk, ix->d_stats, fin);
		HIP_TRY(hipEventRecord(sl.e2, ix->stream));
		// tail stream, the deep tier: the same kernel with deeper tables over the spill list (single-wave workgroups of 42 KB of LDS).
		// ONE deep tier (r04; r03 had a 40 + 16 tier in front of it): it starts the moment the main tier's workgroups retire, while
		// the CUs are free.  A tier enqueued behind another one found the NEXT batch's main-tier kernel on every CU and was only placed
		// when that kernel's workgroups retired, 2.2 ms later -- the tail stream was busy for a whole step, the handle's batch slots
		// waited for it and the main stream idled 0.2 ms per step (profiles/timeline_hg38_r04_three_tiers.txt).  With vote keys instead
		// of context lists the deep tier has 0.005-0.3 % of the reads to do, not 10 %.
		HIP_TRY(hipStreamWaitEvent(ix->tail, sl.e2, 0));
		// The deep tier's grid: single-wave workgroups of 42 KB of LDS that can only be PLACED where a main-tier workgroup has retired --
		// by then the next batch's kernels want the same CUs.  A full grid (3 per CU) for a list of 20 reads took 0.30 ms beside a
		// 0.32 ms main kernel at chr22 scale (profiles/rocprof_summary_r05_chr22.txt).  The list's size is only known on the device, but
		// lists of consecutive batches of one workload are alike: the grid follows the last harvested batches' lists (two reads per
		// wave's pull, and a floor), the full grid until a batch has been harvested.  A list longer than expected is still finished --
		// the waves pull their work from a counter and their pulls grow with the list -- just by fewer waves.
		unsigned w2grid = (unsigned)std::min<uint64_t>((n_reads + 63) / 64, (uint64_t)ix->cus * ix->w2_wpc);
		if (ix->spill_known && !getenv("VG_W2_FULL_GRID")) w2grid = std::min<unsigned>(w2grid, std::max<unsigned>(32u, (2u * ix->spill_hint + ix->w2_chunk - 1) / ix->w2_chunk + 16u));
		if (big) vg_wave_kernel_big<W3_ECAP, W3_NCAP, 1><<<w2grid, 64, 0, ix->tail>>>(ix->d, sl.pk_kmer, sl.pk_meta, d_offsets, 0, sl.listA, &ctr[0], sl.listB, &ctr[1], &ctr[5], ix->w2_chunk, ix->d_stats, nofuse);
		else if (sdx) vg_wave_kernel<false, W3_ECAP, W3_NCAP, 1, true><<<w2grid, 64, 0, ix->tail>>>(ix->d, sl.pk_kmer, sl.pk_meta, d_offsets, 0, sl.listA, &ctr[0], sl.listB, &ctr[1], &ctr[5], ix->w2_chunk, ix->d_stats, nofuse);
		else vg_wave_kernel<STATS, W3_ECAP, W3_NCAP, 1><<<w2grid, 64, 0, ix->tail>>>(ix->d, sl.pk_kmer, sl.pk_meta, d_offsets, 0, sl.listA, &ctr[0], sl.listB, &ctr[1], &ctr[5], ix->w2_chunk, ix->d_stats, nofuse);
		// what the deep tier leaves behind goes to the handle's late store (the lane machine runs over it once, at the next
		// synchronisation); the slot only keeps what the store cannot take (listA is free again: the deep tier has consumed it)
		sl.lt_late = ix->late.state != nullptr;
		if (sl.lt_late) {
			vg_late_collect<<<4, 256, 0, ix->tail>>>(ix->late, sl.pk_kmer, sl.pk_meta, d_offsets, sl.listB, &ctr[1], sl.listA, &ctr[6]);
			ix->late_dirty = true; ix->late_stats = STATS;
		}
		HIP_TRY(hipEventRecord(sl.e4, ix->tail));
	} else {
		sl.lt_late = false;
		if (produced_on && produced_on != ix->stream) {             // a batch gathered by the FASTQ framing on the ingest stream
			HIP_TRY(hipEventRecord(sl.e_in, produced_on));
			HIP_TRY(hipStreamWaitEvent(ix->stream, sl.e_in, 0));
		}
		HIP_TRY(hipMemsetAsync(ctr, 0, 64, ix->stream));
		HIP_TRY(hipEventRecord(sl.e0, ix->stream));
		HIP_TRY(hipEventRecord(sl.e1, ix->stream));
		HIP_TRY(hipEventRecord(sl.e5, ix->stream));
		vg_lane_kernel<STATS><<<g1, 256, 0, ix->stream>>>(ix->d, ix->mid.s, d_bases, d_quals, d_offsets, n_reads, nullptr, d_n_reads, sl.listB, &ctr[1], ix->d_stats, packed ? nullptr : &ctr[3], d_gate, sl.pk_kmer, sl.pk_meta, packed);
		HIP_TRY(hipEventRecord(sl.e2, ix->stream));
		HIP_TRY(hipStreamWaitEvent(ix->tail, sl.e2, 0));
		HIP_TRY(hipEventRecord(sl.e4, ix->tail));
	}
	// ... and the generic lane machine with the deep HBM scratch for whatever is left -- LATER, and only if anything is (r05): the
	// batch's counters come to the host behind its tiers, and harvest() launches the lane tier when they say that the deep tier left
	// reads behind (0-5 reads per 8 M at hg38 scale, mostly none).  Through r04 the kernel was enqueued unconditionally: 64 workgroups
	// that exit at once, but can only be PLACED when workgroups of the next batch's main tier retire -- 1.7-3.7 ms during which the
	// batch's slot stayed busy, and 2-4 ms at the end of every job.
	// (a workload whose batches DO leave reads for the lane tier -- long reads, tiny list capacities in the tests -- gets the kernel
	// enqueued behind the deep tier as before, from the first batch that showed it on: a launch at harvest time would stall the host
	// and the tail stream once per batch)
	sl.lt_bases = d_bases; sl.lt_quals = d_quals; sl.lt_offsets = d_offsets; sl.lt_gate = d_gate; sl.lt_packed = packed; sl.lt_stats = STATS;
	sl.lt_enqueued = ix->lane_tier_seen;
	// (the lane tiers, once a workload needs them, and the counter copies behind them go to `tail2`; a workload that never does keeps
	// to the one tail stream -- chr22-scale steps are 3 % slower with the fourth stream in use: profiles/ab_tail_streams_r05.txt)
	hipStream_t lt = ix->tail;
	if (sl.lt_enqueued) { lt = ix->tail2; HIP_TRY(hipStreamWaitEvent(lt, sl.e4, 0)); }
	if (sl.lt_enqueued) vg_lane_kernel<STATS><<<ix->big.s.nlanes / 64, 64, 0, lt>>>(ix->d, ix->big.s, d_bases, d_quals, d_offsets, 0, sl.lt_late ? sl.listA : sl.listB, sl.lt_late ? &ctr[6] : &ctr[1], sl.listC, &ctr[2], ix->d_stats, nullptr, d_gate, sl.pk_kmer, sl.pk_meta, packed);
	HIP_TRY(hipMemcpyAsync(sl.h_ctr, ctr, 32, hipMemcpyDeviceToHost, lt));
	HIP_TRY(hipEventRecord(sl.e3, lt));
	HIP_TRY(hipGetLastError());
	sl.busy = true;
	return VG_OK;
}

static int acquire_slot(vg_index *ix, Slot **out)
{
	Slot &sl = ix->slot[ix->next_slot];
	ix->next_slot = (ix->next_slot + 1) % NSLOT;
	int rc = harvest(ix, sl);                             // blocks only when NSLOT batches are already in flight
	if (rc) return rc;
	*out = &sl;
	return VG_OK;
}

// produced_on: the stream whose earlier work fills the batch buffers (nullptr: they are complete already).
// d_n_reads: the batch was framed on the device and only the device knows its size; n_reads and total_bases are then upper bounds.
static int launch_batch(vg_index *ix, Slot &sl, const uint8_t *d_bases, const uint8_t *d_quals, const uint64_t *d_offsets, uint64_t n_reads, hipStream_t produced_on = nullptr,
                        const uint32_t *d_n_reads = nullptr, uint64_t total_bases = 0, const uint32_t *d_gate = nullptr)
{
	if (n_reads >= (1ull << 32) - (1ull << 24)) return fail(VG_EINVAL, "more than 2^32 - 2^24 reads in one batch");
	// packed-read buffers are sized from the batch's total length (8 bytes from the device; the handle's streams are
	// non-blocking, so this copy does not wait for kernels in flight)
	uint64_t total = total_bases;
	if (!d_n_reads) HIP_TRY(hipMemcpy(&total, d_offsets + n_reads, 8, hipMemcpyDeviceToHost));
	if (total >= (1ull << 37)) return fail(VG_ETOOBIG, "a batch of 2^37 bases or more (the kernels address a batch's 32-base slots with 32 bits)");
	if (total && (!d_bases || (!d_quals && !d_gate))) return fail(VG_EINVAL, "null argument: a batch with bases needs the base text and its quality strings or gate words");
	const uint64_t need_k = (total >> 5) + 2, need_m = n_reads + 1;
	// the slot is idle (acquire_slot harvested it), so its buffers may be replaced
	if (need_k > sl.pk_kmer_cap) { if (sl.pk_kmer) (void)hipFree(sl.pk_kmer); sl.pk_kmer = nullptr; sl.pk_kmer_cap = 0; HIP_TRY(hipMalloc((void **)&sl.pk_kmer, need_k * 8)); sl.pk_kmer_cap = need_k; }
	if (need_m > sl.pk_meta_cap) { if (sl.pk_meta) (void)hipFree(sl.pk_meta); sl.pk_meta = nullptr; sl.pk_meta_cap = 0; HIP_TRY(hipMalloc((void **)&sl.pk_meta, need_m * 8)); sl.pk_meta_cap = need_m; }
	if (n_reads > sl.list_cap) {
		uint32_t **lists[] = {&sl.listA, &sl.listB, &sl.listC};
		sl.list_cap = 0;                                        // a failed allocation below must not leave the old size behind
		for (uint32_t **l : lists) { if (*l) (void)hipFree(*l); *l = nullptr; }
		for (uint32_t **l : lists) HIP_TRY(hipMalloc((void **)l, (size_t)n_reads * 4));
		sl.list_cap = n_reads;
	}
	const uint64_t tb = d_n_reads ? 0ull : total;                // (a batch framed on the device: sizes are upper bounds, the mean length is unknown here)
	return ix->stats_enabled ? enqueue_batch<true>(ix, sl, d_bases, d_quals, d_gate, d_offsets, n_reads, produced_on, d_n_reads, false, tb)
	                         : enqueue_batch<false>(ix, sl, d_bases, d_quals, d_gate, d_offsets, n_reads, produced_on, d_n_reads, false, tb);
}

extern "C" int vg_reads_process_device(vg_index *ix, const uint8_t *d_bases, const uint8_t *d_quals, const uint64_t *d_offsets, uint64_t n_reads)
{
	if (!ix || (!d_offsets && n_reads)) return fail(VG_EINVAL, "null argument");
	if (n_reads == 0) return VG_OK;
	HIP_TRY(hipSetDevice(ix->device));
	Slot *sl = nullptr;
	int rc = acquire_slot(ix, &sl);
	if (rc) return rc;
	return launch_batch(ix, *sl, d_bases, d_quals, d_offsets, n_reads);
}

extern "C" int vg_reads_process_device_gated(vg_index *ix, const uint8_t *d_bases, const uint32_t *d_gate_words, const uint64_t *d_offsets, uint64_t n_reads)
{
	if (!ix || ((!d_offsets || !d_gate_words) && n_reads)) return fail(VG_EINVAL, "null argument");
	if (n_reads == 0) return VG_OK;
	HIP_TRY(hipSetDevice(ix->device));
	Slot *sl = nullptr;
	int rc = acquire_slot(ix, &sl);
	if (rc) return rc;
	return launch_batch(ix, *sl, d_bases, nullptr, d_offsets, n_reads, nullptr, nullptr, 0, d_gate_words);
}

// A batch that is already 2-bit packed, in HOST memory ("pre-packed 2-bit + the <= 31 quality chars the gate can
// see", the latter reduced to the comparison's result): the slot's packed-read buffers are filled by copies instead of by the
// pack kernel.  copy_on: the stream the (asynchronous, page-locked source) copies go to, or nullptr for blocking copies.
static int launch_packed(vg_index *ix, Slot &sl, const uint64_t *kmers, const uint64_t *meta, const uint64_t *offsets, uint64_t n_reads, uint64_t n_chunks, hipStream_t copy_on)
{
	if (n_reads >= (1ull << 32) - (1ull << 24)) return fail(VG_EINVAL, "more than 2^32 - 2^24 reads in one batch");
	if (n_chunks >= (1ull << 32)) return fail(VG_ETOOBIG, "a batch of 2^32 chunks or more");
	const uint64_t need_k = n_chunks + 2, need_m = n_reads + 1;
	if (need_k > sl.pk_kmer_cap) { if (sl.pk_kmer) (void)hipFree(sl.pk_kmer); sl.pk_kmer = nullptr; sl.pk_kmer_cap = 0; HIP_TRY(hipMalloc((void **)&sl.pk_kmer, need_k * 8)); sl.pk_kmer_cap = need_k; }
	if (need_m > sl.pk_meta_cap) { if (sl.pk_meta) (void)hipFree(sl.pk_meta); sl.pk_meta = nullptr; sl.pk_meta_cap = 0; HIP_TRY(hipMalloc((void **)&sl.pk_meta, need_m * 8)); sl.pk_meta_cap = need_m; }
	if (n_reads + 1 > sl.stage_reads) {
		if (sl.st_offsets) (void)hipFree(sl.st_offsets);
		sl.st_offsets = nullptr; sl.stage_reads = 0;
		HIP_TRY(hipMalloc((void **)&sl.st_offsets, (n_reads + 1) * 8));
		sl.stage_reads = n_reads + 1;
	}
	if (n_reads > sl.list_cap) {
		uint32_t **lists[] = {&sl.listA, &sl.listB, &sl.listC};
		sl.list_cap = 0;
		for (uint32_t **l : lists) { if (*l) (void)hipFree(*l); *l = nullptr; }
		for (uint32_t **l : lists) HIP_TRY(hipMalloc((void **)l, (size_t)n_reads * 4));
		sl.list_cap = n_reads;
	}
	if (copy_on) {
		// (hipMemcpyDefault: the source is page-locked host memory, or -- a read store's batch -- device memory)
		if (n_chunks) HIP_TRY(hipMemcpyAsync(sl.pk_kmer, kmers, n_chunks * 8, hipMemcpyDefault, copy_on));
		HIP_TRY(hipMemcpyAsync(sl.pk_meta, meta, n_reads * 8, hipMemcpyDefault, copy_on));
		HIP_TRY(hipMemcpyAsync(sl.st_offsets, offsets, (n_reads + 1) * 8, hipMemcpyDefault, copy_on));
	} else {
		if (n_chunks) HIP_TRY(hipMemcpy(sl.pk_kmer, kmers, n_chunks * 8, hipMemcpyHostToDevice));
		HIP_TRY(hipMemcpy(sl.pk_meta, meta, n_reads * 8, hipMemcpyHostToDevice));
		HIP_TRY(hipMemcpy(sl.st_offsets, offsets, (n_reads + 1) * 8, hipMemcpyHostToDevice));
	}
	return ix->stats_enabled ? enqueue_batch<true>(ix, sl, nullptr, nullptr, nullptr, sl.st_offsets, n_reads, copy_on, nullptr, true, 0)
	                         : enqueue_batch<false>(ix, sl, nullptr, nullptr, nullptr, sl.st_offsets, n_reads, copy_on, nullptr, true, 0);
}

// pinned: the caller's arrays are page-locked and stay untouched until vg_sync -- the copies are asynchronous (ingest stream) and
// the call returns as soon as the batch is enqueued (vg_reads_submit_packed_async: a caller that packed a whole file ahead of the
// index hands over hundreds of batches; with blocking copies each cost ~10 ms of the host's time)
static int submit_packed_impl(vg_index *ix, const uint64_t *kmers, const uint64_t *meta, const uint64_t *chunk_offsets, uint64_t n_reads, bool pinned)
{
	HIP_TRY(hipSetDevice(ix->device));
	if (chunk_offsets[0] != 0) return fail(VG_EINVAL, "chunk_offsets[0] must be 0");
	uint64_t invalid = 0, bad = 0;
	for (uint64_t i = 0; i < n_reads; i++) {                     // (branch-free: one pass over three arrays, the compiler vectorises it)
		const uint64_t d = chunk_offsets[i + 1] - chunk_offsets[i];
		bad |= (chunk_offsets[i + 1] < chunk_offsets[i] ? 1ull : 0ull) | (d > 31 ? 2ull : 0ull) | ((meta[i] & 0x3FFFFFFF00000000ull) ? 4ull : 0ull);
		invalid += meta[i] >> 63;
	}
	if (bad & 1ull) return fail(VG_EINVAL, "chunk offsets not monotone");
	if (bad & 2ull) return fail(VG_EBADREAD, "a packed read of more than 31 chunks (a FASTQ line the reference can read holds at most 1022 bases, qv.cc:700)");
	if (bad & 4ull) return fail(VG_EINVAL, "a packed read's flag word has reserved bits set (bits 0-31: gate bits, 62: N inside the read, 63: another character; nothing else)");
	const uint64_t n_chunks = chunk_offsets[n_reads];
	if (n_chunks && !kmers) return fail(VG_EINVAL, "null argument");
	Slot *sl = nullptr;
	int rc = acquire_slot(ix, &sl);
	if (rc) return rc;
	// the flat-batch offsets of the trimmed reads (32 x chunks before each), in the slot's page-locked staging
	if (n_reads + 1 > sl->hp_reads_cap) {
		if (sl->hp_meta) (void)hipHostFree(sl->hp_meta);
		if (sl->hp_offsets) (void)hipHostFree(sl->hp_offsets);
		sl->hp_meta = sl->hp_offsets = nullptr; sl->hp_reads_cap = 0;
		const uint64_t cap = (n_reads + 1) * 5 / 4;
		if (hipHostMalloc((void **)&sl->hp_meta, cap * 8, hipHostMallocDefault) != hipSuccess || hipHostMalloc((void **)&sl->hp_offsets, cap * 8, hipHostMallocDefault) != hipSuccess)
			return fail(VG_ENOMEM, "hipHostMalloc(packed staging) failed");
		sl->hp_reads_cap = cap;
	}
	for (uint64_t i = 0; i <= n_reads; i++) sl->hp_offsets[i] = 32 * chunk_offsets[i];
	hipStream_t is = ix->ingest_stream ? ix->ingest : ix->stream;
	rc = launch_packed(ix, *sl, kmers, meta, sl->hp_offsets, n_reads, n_chunks, is);
	if (rc == VG_OK && !pinned) HIP_TRY(hipStreamSynchronize(is));          // the caller's arrays are free again
	if (rc == VG_OK) ix->host_invalid += invalid;
	return rc;
}
extern "C" int vg_reads_submit_packed(vg_index *ix, const uint64_t *kmers, const uint64_t *meta, const uint64_t *chunk_offsets, uint64_t n_reads)
{
	if (!ix || !chunk_offsets || (n_reads && !meta)) return fail(VG_EINVAL, "null argument");
	if (n_reads == 0) return VG_OK;
	return guarded([&]() -> int { return submit_packed_impl(ix, kmers, meta, chunk_offsets, n_reads, false); });
}
extern "C" int vg_reads_submit_packed_async(vg_index *ix, const uint64_t *kmers, const uint64_t *meta, const uint64_t *chunk_offsets, uint64_t n_reads)
{
	if (!ix || !chunk_offsets || (n_reads && !meta)) return fail(VG_EINVAL, "null argument");
	if (n_reads == 0) return VG_OK;
	return guarded([&]() -> int { return submit_packed_impl(ix, kmers, meta, chunk_offsets, n_reads, true); });
}

// ---- a read store: packed batches parked in device memory before (or beside) an index handle -------------------------------
// The command line packs the FASTQ file while vg_index_open runs.  Keeping what it packs in page-locked HOST memory until the
// handle exists cost 0.15 s per GB to lock and 0.1 s per GB to give back when the process ends (11 GB for 200 M reads: more
// than the whole read loop), and the copies up only started after the open.  The device has ~45 GB to spare beside an hg38
// index and the link is idle two thirds of the open's time: the batches go up as they are packed, out of two small staging
// buffers, and the open handle takes them from device memory.
struct StoredBatch { const uint64_t *kmers, *meta, *offsets; uint64_t n_reads, n_chunks; };
struct vg_read_store {
	int device = 0;
	uint8_t *block = nullptr;
	uint64_t bytes = 0, used = 0, reads = 0, invalid = 0;
	hipStream_t stream = nullptr;
	std::vector<StoredBatch> batches;
};
__global__ void vg_chunk_to_flat_offsets(uint64_t *__restrict__ o, const uint64_t n)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) o[i] <<= 5;
}
extern "C" int vg_read_store_create(int device, uint64_t max_bytes, vg_read_store **out)
{
	if (!out || max_bytes == 0) return fail(VG_EINVAL, "null argument / a store of no bytes");
	*out = nullptr;
	return guarded([&]() -> int {
		HIP_TRY(hipSetDevice(device));
		std::unique_ptr<vg_read_store> rs(new vg_read_store);
		rs->device = device;
		max_bytes = (max_bytes + 4095) & ~4095ull;
		hipError_t e = vg_malloc_patient((void **)&rs->block, max_bytes);
		if (e != hipSuccess) { char t[64]; snprintf(t, sizeof t, "%.1f GB", max_bytes / 1e9); return fail(VG_ENOMEM, "hipMalloc(read store, %s): %s", t, hipGetErrorString(e)); }
		rs->bytes = max_bytes;
		if (hipStreamCreateWithFlags(&rs->stream, hipStreamNonBlocking) != hipSuccess) { (void)hipFree(rs->block); return fail(VG_ENODEV, "hipStreamCreate(read store) failed"); }
		*out = rs.release();
		return VG_OK;
	});
}
extern "C" void vg_read_store_destroy(vg_read_store *rs)
{
	if (!rs) return;
	(void)hipSetDevice(rs->device);
	if (rs->stream) { (void)hipStreamSynchronize(rs->stream); (void)hipStreamDestroy(rs->stream); }
	if (rs->block) (void)hipFree(rs->block);
	delete rs;
}
extern "C" uint64_t vg_read_store_reads(const vg_read_store *rs) { return rs ? rs->reads : 0; }
extern "C" uint64_t vg_read_store_bytes_used(const vg_read_store *rs) { return rs ? rs->used : 0; }
extern "C" int vg_read_store_flush(vg_read_store *rs)
{
	if (!rs) return fail(VG_EINVAL, "null argument");
	HIP_TRY(hipSetDevice(rs->device));
	HIP_TRY(hipStreamSynchronize(rs->stream));
	return VG_OK;
}
extern "C" int vg_read_store_push(vg_read_store *rs, const uint64_t *kmers, const uint64_t *meta, const uint64_t *chunk_offsets, uint64_t n_reads)
{
	if (!rs || !chunk_offsets || (n_reads && !meta)) return fail(VG_EINVAL, "null argument");
	return guarded([&]() -> int {
		HIP_TRY(hipSetDevice(rs->device));
		HIP_TRY(hipStreamSynchronize(rs->stream));               // the arrays of the push before this one are free again
		if (n_reads == 0) return VG_OK;
		if (chunk_offsets[0] != 0) return fail(VG_EINVAL, "chunk_offsets[0] must be 0");
		uint64_t invalid = 0, bad = 0;
		for (uint64_t i = 0; i < n_reads; i++) {
			const uint64_t d = chunk_offsets[i + 1] - chunk_offsets[i];
			bad |= (chunk_offsets[i + 1] < chunk_offsets[i] ? 1ull : 0ull) | (d > 31 ? 2ull : 0ull) | ((meta[i] & 0x3FFFFFFF00000000ull) ? 4ull : 0ull);
			invalid += meta[i] >> 63;
		}
		if (bad & 1ull) return fail(VG_EINVAL, "chunk offsets not monotone");
		if (bad & 2ull) return fail(VG_EBADREAD, "a packed read of more than 31 chunks (a FASTQ line the reference can read holds at most 1022 bases, qv.cc:700)");
		if (bad & 4ull) return fail(VG_EINVAL, "a packed read's flag word has reserved bits set (bits 0-31: gate bits, 62: N inside the read, 63: another character; nothing else)");
		const uint64_t n_chunks = chunk_offsets[n_reads];
		if (n_chunks && !kmers) return fail(VG_EINVAL, "null argument");
		if (n_reads >= (1ull << 32) - (1ull << 24) || n_chunks >= (1ull << 32)) return fail(VG_ETOOBIG, "a batch of 2^32 chunks or more");
		auto up = [](uint64_t b) { return (b + 255) & ~255ull; };
		const uint64_t need = up((n_chunks + 2) * 8) + up(n_reads * 8) + up((n_reads + 1) * 8);
		if (rs->used + need > rs->bytes) { char t[96]; snprintf(t, sizeof t, "%.2f of %.2f GB used, this batch needs %.3f GB", rs->used / 1e9, rs->bytes / 1e9, need / 1e9); return fail(VG_ENOMEM, "the read store is full (%s)", t); }
		StoredBatch b;
		uint8_t *at = rs->block + rs->used;
		b.kmers = (uint64_t *)at; at += up((n_chunks + 2) * 8);
		b.meta = (uint64_t *)at; at += up(n_reads * 8);
		b.offsets = (uint64_t *)at;
		b.n_reads = n_reads; b.n_chunks = n_chunks;
		if (n_chunks) HIP_TRY(hipMemcpyAsync((void *)b.kmers, kmers, n_chunks * 8, hipMemcpyHostToDevice, rs->stream));
		HIP_TRY(hipMemcpyAsync((void *)b.meta, meta, n_reads * 8, hipMemcpyHostToDevice, rs->stream));
		HIP_TRY(hipMemcpyAsync((void *)b.offsets, chunk_offsets, (n_reads + 1) * 8, hipMemcpyHostToDevice, rs->stream));
		vg_chunk_to_flat_offsets<<<(unsigned)((n_reads + 1 + 255) / 256), 256, 0, rs->stream>>>((uint64_t *)b.offsets, n_reads + 1);
		HIP_TRY(hipGetLastError());
		rs->used += need; rs->reads += n_reads; rs->invalid += invalid;
		rs->batches.push_back(b);
		return VG_OK;
	});
}
extern "C" int vg_reads_submit_store(vg_index *ix, vg_read_store *rs)
{
	if (!ix || !rs) return fail(VG_EINVAL, "null argument");
	if (ix->device != rs->device) return fail(VG_EINVAL, "the read store and the index handle live on different devices");
	return guarded([&]() -> int {
		HIP_TRY(hipSetDevice(ix->device));
		HIP_TRY(hipStreamSynchronize(rs->stream));               // everything pushed is in device memory
		hipStream_t is = ix->ingest_stream ? ix->ingest : ix->stream;
		for (const StoredBatch &b : rs->batches) {
			Slot *sl = nullptr;
			int rc = acquire_slot(ix, &sl);
			if (rc) return rc;
			rc = launch_packed(ix, *sl, b.kmers, b.meta, b.offsets, b.n_reads, b.n_chunks, is);
			if (rc) return rc;
		}
		ix->host_invalid += rs->invalid;
		return VG_OK;
	});
}

static int submit_impl(vg_index *ix, const uint8_t *bases, const uint8_t *quals, const uint64_t *offsets, uint64_t n_reads);
extern "C" int vg_reads_submit(vg_index *ix, const uint8_t *bases, const uint8_t *quals, const uint64_t *offsets, uint64_t n_reads)
{
	if (!ix || !offsets) return fail(VG_EINVAL, "null argument");
	if (n_reads == 0) return VG_OK;
	return guarded([&] { return submit_impl(ix, bases, quals, offsets, n_reads); });
}
static int submit_impl(vg_index *ix, const uint8_t *bases, const uint8_t *quals, const uint64_t *offsets, uint64_t n_reads)
{
	HIP_TRY(hipSetDevice(ix->device));
	const uint64_t base0 = offsets[0];
	const uint64_t total = offsets[n_reads] - base0;
	for (uint64_t i = 0; i < n_reads; i++) {
		if (offsets[i + 1] < offsets[i]) return fail(VG_EINVAL, "offsets not monotone");
		if (offsets[i + 1] - offsets[i] > 1022) return fail(VG_EBADREAD, "read longer than 1022 bases (reference BUF_SIZE 1024, qv.cc:700)");
	}
	Slot *slp = nullptr;
	int rc = acquire_slot(ix, &slp);
	if (rc) return rc;
	Slot &sl = *slp;
	if (total + 64 > sl.stage_bytes) {
		if (sl.st_bases) (void)hipFree(sl.st_bases);
		sl.st_bases = nullptr; sl.stage_bytes = 0;
		HIP_TRY(hipMalloc((void **)&sl.st_bases, total + 64));
		sl.stage_bytes = total + 64;
	}
	if (total + 64 > sl.stage_quals_bytes) {
		if (sl.st_quals) (void)hipFree(sl.st_quals);
		sl.st_quals = nullptr; sl.stage_quals_bytes = 0;
		HIP_TRY(hipMalloc((void **)&sl.st_quals, total + 64));
		sl.stage_quals_bytes = total + 64;
	}
	if (n_reads + 1 > sl.stage_reads) {
		if (sl.st_offsets) (void)hipFree(sl.st_offsets);
		sl.st_offsets = nullptr; sl.stage_reads = 0;
		HIP_TRY(hipMalloc((void **)&sl.st_offsets, (n_reads + 1) * 8));
		sl.stage_reads = n_reads + 1;
	}
	std::vector<uint64_t> rel;
	const uint64_t *off = offsets;
	if (base0) { rel.resize(n_reads + 1); for (uint64_t i = 0; i <= n_reads; i++) rel[i] = offsets[i] - base0; off = rel.data(); }
	// plain (blocking) copies: when they return the caller's buffers are free again; kernels of earlier batches keep running
	HIP_TRY(hipMemcpy(sl.st_bases, bases + base0, total, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(sl.st_quals, quals + base0, total, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(sl.st_offsets, off, (n_reads + 1) * 8, hipMemcpyHostToDevice));
	return launch_batch(ix, sl, sl.st_bases, sl.st_quals, sl.st_offsets, n_reads);
}

// ---- FASTQ text in, framed and processed on the device (replaces the four fgets() + strlen of qv.cc:760-784) ---------------
template <class T>
static int grow_dev(T **p, uint64_t &cap, uint64_t need)
{
	if (need <= cap) return VG_OK;
	if (*p) (void)hipFree(*p);
	*p = nullptr; cap = 0;
	hipError_t e = hipMalloc((void **)p, (size_t)need * sizeof(T));
	if (e != hipSuccess) return fail(VG_ENOMEM, "hipMalloc(FASTQ staging): %s", hipGetErrorString(e));
	cap = need;
	return VG_OK;
}

// CPUs this process may actually use: the hardware threads, capped by a cgroup CPU quota (cpu.max: "<quota> <period>" -- the GPU
// boxes of this pool give a container 16 CPUs' worth of time on a 256-thread host; threads beyond the quota only take time from
// each other, and a spinning one takes it from the working ones)
static unsigned usable_cpus()
{
	unsigned h = std::thread::hardware_concurrency();
	if (h == 0) h = 1;
	for (const char *path : {"/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"}) {
		FILE *f = fopen(path, "r");
		if (!f) continue;
		long long q = -1, per = 100000;
		char a[64] = "";
		if (fscanf(f, "%63s %lld", a, &per) >= 1 && strcmp(a, "max") != 0) q = atoll(a);
		fclose(f);
		if (q > 0 && per > 0) { const unsigned c = (unsigned)((q + per - 1) / per); if (c && c < h) h = c; }
		break;
	}
	return h;
}
static int host_pack_threads_default()
{
	if (const char *e = getenv("VG_PACK_THREADS")) return std::max(0, atoi(e));
	const unsigned c = usable_cpus();
	// Host packing pays when the process has CPUs to spare BESIDE whoever reads the file and feeds the device: from 32 usable CPUs on,
	// half of them (at most 96).  Below that the device-side framing is the dependable route: on this pool's boxes (a 16-CPU quota)
	// 13 packing threads reach 2.3 x 10^8 reads/s when nothing else runs and 0.9 x 10^8 when two more processes do -- a process that
	// asks for more CPU time than its quota is stopped for the rest of the 100 ms period -- against a steady 1.65 x 10^8 for the
	// device framing.
	return c >= 32 ? (int)std::min(c / 2, 96u) : 0;
}

extern "C" int vg_fastq_stream_begin_packed(vg_index *ix, int host_threads)
{
	if (!ix) return fail(VG_EINVAL, "null argument");
	if (host_threads < 0) host_threads = host_pack_threads_default();
	if (host_threads == 0) return vg_fastq_stream_begin(ix);
	if (host_threads > 256) host_threads = 256;               // (threads beyond the CPUs there are only spin)
	return guarded([&]() -> int {
		HIP_TRY(hipSetDevice(ix->device));
		if (!ix->packer || ix->packer->threads() != host_threads) { delete ix->packer; ix->packer = nullptr; ix->packer = new vgp::Packer(host_threads); }
		ix->packer->begin();
		ix->fq_open = true; ix->fq_packed = true; ix->fq_prev_slot = -1;
		return VG_OK;
	});
}

// one chunk of a host-packed stream: frame + pack into the slot's page-locked staging (host threads; the caller's buffer is free
// when this returns), then asynchronous copies of the packed form on the ingest stream and the read loop behind them
static int push_packed(vg_index *ix, const uint8_t *text, uint64_t nbytes)
{
	HIP_TRY(hipSetDevice(ix->device));
	if (ix->packer->poisoned()) return VG_OK;                 // refused earlier: the rest of the stream is the host reader's
	Slot *slp = nullptr;
	int rc = acquire_slot(ix, &slp);                            // (its previous batch has finished: the staging is free)
	if (rc) return rc;
	Slot &sl = *slp;
	const uint64_t need_r = vgp::Packer::reads_cap(nbytes) + 1, need_k = vgp::Packer::kmers_cap(nbytes);
	if (need_k > sl.hp_kmers_cap) {
		if (sl.hp_kmers) (void)hipHostFree(sl.hp_kmers);
		sl.hp_kmers = nullptr; sl.hp_kmers_cap = 0;
		if (hipHostMalloc((void **)&sl.hp_kmers, need_k * 8, hipHostMallocDefault) != hipSuccess) return fail(VG_ENOMEM, "hipHostMalloc(packed staging) failed");
		sl.hp_kmers_cap = need_k;
	}
	if (need_r > sl.hp_reads_cap) {
		if (sl.hp_meta) (void)hipHostFree(sl.hp_meta);
		if (sl.hp_offsets) (void)hipHostFree(sl.hp_offsets);
		sl.hp_meta = sl.hp_offsets = nullptr; sl.hp_reads_cap = 0;
		if (hipHostMalloc((void **)&sl.hp_meta, need_r * 8, hipHostMallocDefault) != hipSuccess || hipHostMalloc((void **)&sl.hp_offsets, need_r * 8, hipHostMallocDefault) != hipSuccess)
			return fail(VG_ENOMEM, "hipHostMalloc(packed staging) failed");
		sl.hp_reads_cap = need_r;
	}
	vgp::Staging st;
	st.kmers = sl.hp_kmers; st.kmers_cap = sl.hp_kmers_cap; st.meta = sl.hp_meta; st.offsets = sl.hp_offsets; st.reads_cap = sl.hp_reads_cap;
	const vgp::ChunkResult r = ix->packer->push(text, nbytes, st);
	if (r.n_reads == 0) return VG_OK;                            // (a refused block poisons the stream; the records framed before it are still this batch)
	hipStream_t is = ix->ingest_stream ? ix->ingest : ix->stream;
	rc = launch_packed(ix, sl, sl.hp_kmers, sl.hp_meta, sl.hp_offsets, r.n_reads, r.n_chunks, is);
	if (rc == VG_OK) ix->host_invalid += r.n_invalid;
	return rc;
}

extern "C" int vg_fastq_stream_begin(vg_index *ix)
{
	if (!ix) return fail(VG_EINVAL, "null argument");
	ix->fq_packed = false;
	HIP_TRY(hipSetDevice(ix->device));
	hipStream_t is = ix->ingest_stream ? ix->ingest : ix->stream;
	HIP_TRY(hipMemsetAsync(ix->d_fq, 0, sizeof(FqStream), is));
	ix->fq_open = true; ix->fq_prev_slot = -1;
	return VG_OK;
}

// One chunk of the stream: a blocking host-to-device copy (the caller's buffer is free when the call returns), then framing
// and the read loop are only ENQUEUED -- record counts stay on the device until vg_fastq_stream_end.
extern "C" int vg_fastq_stream_push(vg_index *ix, const uint8_t *text, uint64_t nbytes)
{
	if (!ix || (!text && nbytes)) return fail(VG_EINVAL, "null argument");
	if (!ix->fq_open) return fail(VG_EINVAL, "vg_fastq_stream_push without vg_fastq_stream_begin");
	if (nbytes == 0) return VG_OK;
	if (nbytes >= (1ull << 31)) return fail(VG_EINVAL, "FASTQ chunk of 2 GiB or more");
	if (ix->fq_packed) return guarded([&] { return push_packed(ix, text, nbytes); });
	HIP_TRY(hipSetDevice(ix->device));
	const int slot_no = ix->next_slot;
	Slot *slp = nullptr;
	int rc = acquire_slot(ix, &slp);
	if (rc) return rc;
	Slot &sl = *slp;
	hipStream_t is = ix->ingest_stream ? ix->ingest : ix->stream;
	// the chunk after this slot's last one copied the tail of its text on the ingest stream: that must have happened before
	// the text is overwritten (the slot's own batch being finished does not imply it)
	if (sl.fq_tail_wanted) { HIP_TRY(hipEventSynchronize(sl.e_fq)); sl.fq_tail_wanted = false; }
	// capacities follow from the chunk's size alone: lines average at least 8 bytes (or the chunk is refused), a record has four
	const uint64_t span = (uint64_t)FQ_CARRY + nbytes;
	const uint64_t n_tiles = (span + FQ_TILE - 1) / FQ_TILE;
	const uint64_t cap_lines = span / 8 + 16, cap_rec = cap_lines / 4 + 1;
	if ((rc = grow_dev(&sl.fq_text, sl.fq_text_cap, n_tiles * FQ_TILE + 64))) return rc;
	if ((rc = grow_dev(&sl.fq_tiles, sl.fq_tiles_cap, n_tiles + 2))) return rc;
	if ((rc = grow_dev(&sl.fq_lines, sl.fq_lines_cap, cap_lines + 2))) return rc;
	if (!sl.fq_chunk) { uint64_t one = 0; if ((rc = grow_dev(&sl.fq_chunk, one, 1))) return rc; }
	if (cap_rec + 2 > sl.stage_reads) {
		if (sl.st_offsets) (void)hipFree(sl.st_offsets);
		sl.st_offsets = nullptr; sl.stage_reads = 0;
		HIP_TRY(hipMalloc((void **)&sl.st_offsets, (cap_rec + 2) * 8));
		sl.stage_reads = cap_rec + 2;
	}
	if (span + 64 > sl.stage_bytes) {                               // the bases of a chunk are shorter than its text
		if (sl.st_bases) (void)hipFree(sl.st_bases);
		sl.st_bases = nullptr; sl.stage_bytes = 0;
		HIP_TRY(hipMalloc((void **)&sl.st_bases, span + 64));
		sl.stage_bytes = span + 64;
	}
	if ((rc = grow_dev(&sl.st_gate, sl.st_gate_cap, cap_rec + 2))) return rc;
	{
		uint8_t *tmp = (uint8_t *)sl.fq_tmp;
		const uint64_t need = vg_dev_scan_temp_bytes(n_tiles + 1, cap_rec + 1);
		if ((rc = grow_dev(&tmp, sl.fq_tmp_cap, need))) { sl.fq_tmp = tmp; return rc; }
		sl.fq_tmp = tmp;
	}
	HIP_TRY(hipMemcpy(sl.fq_text + FQ_CARRY, text, nbytes, hipMemcpyHostToDevice));
	sl.fq_text_len = nbytes;
	const uint8_t *prev_end = nullptr;
	if (ix->fq_prev_slot >= 0) { const Slot &pv = ix->slot[ix->fq_prev_slot]; prev_end = pv.fq_text + FQ_CARRY + pv.fq_text_len; }
	vg_fqs_prepare<<<1, 256, 0, is>>>(ix->d_fq, sl.fq_chunk, prev_end, sl.fq_text, (uint32_t)nbytes);
	if (ix->fq_prev_slot >= 0) { Slot &pv = ix->slot[ix->fq_prev_slot]; HIP_TRY(hipEventRecord(pv.e_fq, is)); pv.fq_tail_wanted = true; }
	vg_fq_count_newlines<<<(unsigned)n_tiles, 256, 0, is>>>(sl.fq_text, sl.fq_chunk, sl.fq_tiles);
	HIP_TRY(hipMemsetAsync(sl.fq_tiles + n_tiles, 0, 4, is));
	HIP_TRY(hipGetLastError());
	int se = vg_dev_exclusive_scan_u32(sl.fq_tiles, sl.fq_tiles, n_tiles + 1, is, sl.fq_tmp, sl.fq_tmp_cap);
	if (se != 0) return fail(VG_ENODEV, "device scan failed: %s", hipGetErrorString((hipError_t)se));
	const uint32_t *d_n_lines = sl.fq_tiles + n_tiles;
	vg_fq_line_starts<<<(unsigned)n_tiles, 256, 0, is>>>(sl.fq_text, sl.fq_chunk, sl.fq_tiles, sl.fq_lines, (uint32_t)cap_lines);
	vg_fq_record_lengths<<<1024, 256, 0, is>>>(sl.fq_lines, d_n_lines, sl.fq_chunk, sl.st_offsets, (uint32_t)cap_rec, (uint32_t)cap_lines);
	HIP_TRY(hipGetLastError());
	se = vg_dev_exclusive_scan_u64(sl.st_offsets, sl.st_offsets, cap_rec + 1, is, sl.fq_tmp, sl.fq_tmp_cap);
	if (se != 0) return fail(VG_ENODEV, "device scan failed: %s", hipGetErrorString((hipError_t)se));
	vg_fqs_finish<<<1, 1, 0, is>>>(ix->d_fq, sl.fq_chunk, d_n_lines, sl.fq_lines, sl.st_offsets, (uint32_t)cap_rec);
	vg_fq_gather<<<(unsigned)std::min<uint64_t>((cap_rec + 3) / 4, (uint64_t)ix->cus * 32), 256, 0, is>>>(sl.fq_text, sl.fq_lines, sl.st_offsets, sl.fq_chunk, sl.st_bases, sl.st_gate);
	HIP_TRY(hipGetLastError());
	ix->fq_prev_slot = slot_no;
	return launch_batch(ix, sl, sl.st_bases, nullptr, sl.st_offsets, cap_rec, is, &sl.fq_chunk->n_reads, span, sl.st_gate);
}

static int fq_collect(vg_index *ix, bool drain, uint64_t *n_records, uint64_t *consumed, uint64_t *last_record_start, int *refused)
{
	HIP_TRY(hipSetDevice(ix->device));
	if (drain) { int rc = finish_pending(ix); if (rc) return rc; }
	else HIP_TRY(hipStreamSynchronize(ix->ingest_stream ? ix->ingest : ix->stream));       // framing only: the read loop runs on
	FqStream h;
	HIP_TRY(hipMemcpy(&h, ix->d_fq, sizeof h, hipMemcpyDeviceToHost));
	if (n_records) *n_records = h.records;
	if (consumed) *consumed = h.consumed;
	if (last_record_start) *last_record_start = h.last_record;
	if (refused) *refused = h.poisoned ? 1 : 0;
	return VG_OK;
}

extern "C" int vg_fastq_stream_end(vg_index *ix, uint64_t *n_records, uint64_t *consumed, uint64_t *last_record_start, int *refused)
{
	if (!ix) return fail(VG_EINVAL, "null argument");
	if (!ix->fq_open) return fail(VG_EINVAL, "vg_fastq_stream_end without vg_fastq_stream_begin");
	ix->fq_open = false;
	if (ix->fq_packed) {
		ix->fq_packed = false;
		int rc = finish_pending(ix);
		if (rc) return rc;
		if (n_records) *n_records = ix->packer->records();
		if (consumed) *consumed = ix->packer->consumed();
		if (last_record_start) *last_record_start = ix->packer->last_record_start();
		if (refused) *refused = ix->packer->poisoned() ? 1 : 0;
		return VG_OK;
	}
	return fq_collect(ix, true, n_records, consumed, last_record_start, refused);
}

// One self-contained chunk (a stream of one push): the older, synchronous form of the above -- the caller learns what was
// framed before it sends the next chunk (and resubmits the unconsumed tail itself).  Waits for the framing, not for the read loop.
extern "C" int vg_fastq_submit(vg_index *ix, const uint8_t *text, uint64_t nbytes, uint64_t *n_records, uint64_t *consumed, uint64_t *last_record_start)
{
	if (!ix || (!text && nbytes) || !n_records || !consumed) return fail(VG_EINVAL, "null argument");
	*n_records = 0; *consumed = 0;
	if (last_record_start) *last_record_start = 0;
	if (nbytes == 0) return VG_OK;
	if (ix->fq_open) return fail(VG_EINVAL, "vg_fastq_submit inside an open FASTQ stream");
	int rc = vg_fastq_stream_begin(ix);
	if (rc) return rc;
	rc = vg_fastq_stream_push(ix, text, nbytes);
	ix->fq_open = false;
	if (rc) return rc;
	int refused = 0;
	rc = fq_collect(ix, false, n_records, consumed, last_record_start, &refused);
	if (rc) return rc;
	if (refused) {
		*n_records = 0; *consumed = 0;
		return fail(VG_EBADREAD, "a FASTQ line longer than 1023 characters (reference BUF_SIZE 1024, qv.cc:700), a quality line shorter than the read's chunk count, or lines of fewer than 8 bytes on average: frame this chunk on the host");
	}
	return VG_OK;
}

// ---- host-side framing + packing alone: no device is touched (a caller that packs on its own threads and hands the batches to
// vg_reads_submit_packed; the CPU test-suite checks the framing rules through these)
struct vg_packer { vgp::Packer p; explicit vg_packer(int t) : p(t) {} };
extern "C" int vg_packer_create(int host_threads, vg_packer **out)
{
	if (!out) return fail(VG_EINVAL, "null argument");
	*out = nullptr;
	return guarded([&]() -> int {
		int t = host_threads;
		if (t <= 0) { t = host_pack_threads_default(); if (t <= 0) t = 1; }
		if (t > 256) t = 256;
		*out = new vg_packer(t);
		(*out)->p.begin();
		return VG_OK;
	});
}
extern "C" void vg_packer_destroy(vg_packer *pk) { delete pk; }
extern "C" int vg_packer_begin(vg_packer *pk)
{
	if (!pk) return fail(VG_EINVAL, "null argument");
	pk->p.begin();
	return VG_OK;
}
extern "C" uint64_t vg_packer_reads_cap(uint64_t nbytes) { return vgp::Packer::reads_cap(nbytes) + 1; }
extern "C" uint64_t vg_packer_kmers_cap(uint64_t nbytes) { return vgp::Packer::kmers_cap(nbytes); }
extern "C" int vg_packer_push(vg_packer *pk, const uint8_t *text, uint64_t nbytes, uint64_t *kmers, uint64_t kmers_cap, uint64_t *meta, uint64_t *chunk_offsets, uint64_t reads_cap,
                              uint64_t *n_reads, uint64_t *n_chunks, uint64_t *n_invalid)
{
	if (!pk || (!text && nbytes) || !kmers || !meta || !chunk_offsets || !n_reads || !n_chunks) return fail(VG_EINVAL, "null argument");
	return guarded([&]() -> int {
		vgp::Staging st;
		st.kmers = kmers; st.kmers_cap = kmers_cap; st.meta = meta; st.offsets = chunk_offsets; st.reads_cap = reads_cap;
		const vgp::ChunkResult r = pk->p.push(text, nbytes, st);
		*n_reads = r.n_reads; *n_chunks = r.n_chunks;
		if (n_invalid) *n_invalid = r.n_invalid;
		for (uint64_t i = 0; i <= r.n_reads && r.n_reads; i++) chunk_offsets[i] >>= 5;       // the packer writes flat-batch offsets (32 x chunks)
		return VG_OK;
	});
}
extern "C" int vg_packer_end(vg_packer *pk, uint64_t *n_records, uint64_t *consumed, uint64_t *last_record_start, int *refused)
{
	if (!pk) return fail(VG_EINVAL, "null argument");
	if (n_records) *n_records = pk->p.records();
	if (consumed) *consumed = pk->p.consumed();
	if (last_record_start) *last_record_start = pk->p.last_record_start();
	if (refused) *refused = pk->p.poisoned() ? 1 : 0;
	return VG_OK;
}

extern "C" int vg_sync(vg_index *ix)
{
	if (!ix) return fail(VG_EINVAL, "null argument");
	return finish_pending(ix);
}

extern "C" int vg_set_stats(vg_index *ix, int enable)
{
	if (!ix) return fail(VG_EINVAL, "null argument");
	if (ix->stats_enabled != (enable != 0)) { int rc = finish_pending(ix); if (rc) return rc; }
	ix->stats_enabled = enable != 0;
	return VG_OK;
}

extern "C" int vg_stats_get(vg_index *ix, vg_stats *out)
{
	if (!ix || !out) return fail(VG_EINVAL, "null argument");
	int rc = vg_sync(ix);
	if (rc) return rc;
	unsigned long long h[S_COUNT];
	HIP_TRY(hipMemcpy(h, ix->d_stats, sizeof h, hipMemcpyDeviceToHost));
	memset(out, 0, sizeof *out);
	out->reads = h[S_READS]; out->reads_n = h[S_READS_N]; out->reads_invalid = h[S_READS_INVALID];
	out->passes = h[S_PASSES]; out->passes_ok = h[S_PASSES_OK]; out->chunks = h[S_CHUNKS]; out->gate_open = h[S_GATE_OPEN];
	out->refbf_pos = h[S_REFBF_POS]; out->snpbf_pos = h[S_SNPBF_POS]; out->large_block = h[S_LARGE_BLOCK];
	out->ref_query = h[S_REF_QUERY]; out->snp_query = h[S_SNP_QUERY]; out->ref_probe = h[S_REF_PROBE]; out->snp_probe = h[S_SNP_PROBE];
	out->scan_ref = h[S_SCAN_REF]; out->scan_snp = h[S_SCAN_SNP]; out->scan_oob = h[S_SCAN_OOB];
	out->aux_ref = h[S_AUX_REF]; out->aux_snp = h[S_AUX_SNP]; out->site_test = h[S_SITE_TEST]; out->ctx = h[S_CTX];
	out->walks = h[S_WALKS]; out->incr = h[S_INCR]; out->ingest_bytes = h[S_INGEST];
	out->overflow_reads = ix->cum[0]; out->overflow_deep = ix->cum[1]; out->reads_invalid = ix->cum[3] + ix->host_invalid;
	const uint64_t scans = out->gate_open - out->large_block;
	out->alg_bytes = out->ingest_bytes + 8 * (out->ref_query + out->snp_query) + 9 * out->ref_probe + 11 * out->snp_probe
	               + 16 * out->gate_open + 16 * scans + 9 * out->scan_ref + 11 * out->scan_snp
	               + 40 * out->aux_ref + 50 * out->aux_snp + 4 * out->site_test + 128 * out->walks + 4 * out->incr;
	return VG_OK;
}

extern "C" int vg_timing_get(vg_index *ix, vg_timing *out)
{
	if (!ix || !out) return fail(VG_EINVAL, "null argument");
	memset(out, 0, sizeof *out);
	int rc = finish_pending(ix);
#ifdef VG_VIEW_COUNTERS
	{
		// the census since the last vg_timing_get (all launches of all tiers): loads and 128-byte lines asked for, per view
		static const char *names[VC_N] = {"other", "reads (k-mers, offsets, flag words)", "dx (direct table)", "mx (merged view)", "ref_jg", "ref entries", "snp_jg", "snp entries", "sec_jg", "sec3 (LO32-ordered view)",
		                                  "snp_sig", "ref_bf", "snp_bf", "aux rows, stage A", "aux rows, stage B (= any - A)", "-", "-", "srank (walk)", "cnt4 (atomics)", "aux rows, any stage", "pile (site bytes, stage B)"};
		unsigned long long ln[VC_N], ld[VC_N], zero[VC_N] = {0};
		if (hipMemcpyFromSymbol(ln, HIP_SYMBOL(vg_vc_lines), sizeof ln) == hipSuccess && hipMemcpyFromSymbol(ld, HIP_SYMBOL(vg_vc_loads), sizeof ld) == hipSuccess) {
			ln[VC_AUX_B] = ln[VC_AUX_ANY] - ln[VC_AUX_A]; ld[VC_AUX_B] = ld[VC_AUX_ANY] - ld[VC_AUX_A];
			unsigned long long tot = 0;
			for (int i = 0; i < VC_N; i++) if (i != VC_AUX_ANY) tot += ln[i];
			fprintf(stderr, "[view census] %llu batches; lines asked for (all lanes, before L1 / L2 merge anything): %llu\n", (unsigned long long)ix->t_batches, tot);
			for (int i = 0; i < VC_N; i++) if (ld[i] && i != VC_AUX_ANY) fprintf(stderr, "[view census]   %-38s loads %12llu  lines %12llu  (%.1f %%)\n", names[i], ld[i], ln[i], 100.0 * (double)ln[i] / (double)(tot ? tot : 1));
		}
		(void)hipMemcpyToSymbol(HIP_SYMBOL(vg_vc_lines), zero, sizeof zero);
		(void)hipMemcpyToSymbol(HIP_SYMBOL(vg_vc_loads), zero, sizeof zero);
	}
#endif
#ifdef VG_STAGE_CLOCKS
	{
		unsigned long long h[8];
		if (hipMemcpyFromSymbol(h, HIP_SYMBOL(vg_dbg_ovf), sizeof h) == hipSuccess)
			fprintf(stderr, "[dbg] list overflows (pass-level events) tier1: exact %llu neighbour %llu keys %llu | tier2: exact %llu neighbour %llu keys %llu\n", h[0], h[1], h[2], h[4], h[5], h[6]);
	}
#endif
	if (rc) return rc;
	if (!ix->t_batches) return fail(VG_EINVAL, "no batch has been processed since the last vg_timing_get");
	const double n = (double)ix->t_batches;
	out->ms_pack = (float)(ix->t_pack / n); out->ms_main = (float)(ix->t_main / n);
	out->ms_tail = (float)(ix->t_tail / n); out->ms_total = (float)(ix->t_total / n); out->ms_deep_lists = (float)(ix->t_w2 / n);
	out->batches = (uint32_t)ix->t_batches;
	ix->t_pack = ix->t_main = ix->t_tail = ix->t_total = ix->t_w2 = 0; ix->t_batches = 0;
	return VG_OK;
}

extern "C" int vg_sites_fetch(vg_index *ix, uint32_t *pos, uint8_t *ref_base, uint8_t *alt_base, uint8_t *ref_freq, uint8_t *alt_freq)
{
	if (!ix) return fail(VG_EINVAL, "null argument");
	const size_t n = ix->n_sites;
	if (pos) memcpy(pos, ix->site_pos.data(), n * 4);
	if (ref_base) memcpy(ref_base, ix->site_ref.data(), n);
	if (alt_base) memcpy(alt_base, ix->site_alt.data(), n);
	if (ref_freq) memcpy(ref_freq, ix->site_rf.data(), n);
	if (alt_freq) memcpy(alt_freq, ix->site_af.data(), n);
	return VG_OK;
}

extern "C" int vg_counts_fetch(vg_index *ix, uint8_t *ref_cnt, uint8_t *alt_cnt)
{
	if (!ix || !ref_cnt || !alt_cnt) return fail(VG_EINVAL, "null argument");
	int rc = vg_sync(ix);
	if (rc) return rc;
	if (ix->n_sites == 0) return VG_OK;
	vg_clamp_counters<<<(unsigned)std::min<uint64_t>((ix->n_sites + 255) / 256, 4096), 256, 0, ix->stream>>>(ix->d.cnt, ix->n_sites, ix->d_clamped, ix->d_clamped + ix->n_sites);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(ref_cnt, ix->d_clamped, ix->n_sites, hipMemcpyDeviceToHost, ix->stream));
	HIP_TRY(hipMemcpyAsync(alt_cnt, ix->d_clamped + ix->n_sites, ix->n_sites, hipMemcpyDeviceToHost, ix->stream));
	HIP_TRY(hipStreamSynchronize(ix->stream));
	return VG_OK;
}

extern "C" int vg_counts_reset(vg_index *ix)
{
	if (!ix) return fail(VG_EINVAL, "null argument");
	{ int rc = finish_pending(ix); if (rc) return rc; }
	HIP_TRY(hipMemsetAsync(ix->d.cnt, 0, (2 * ix->n_sites + 2) * 4, ix->stream));
	HIP_TRY(hipMemsetAsync(ix->d_stats, 0, S_COUNT * sizeof(unsigned long long), ix->stream));
	for (auto &c : ix->cum) c = 0;
	ix->host_invalid = 0;
	HIP_TRY(hipStreamSynchronize(ix->stream));
	return VG_OK;
}

extern "C" int vg_counts_device_ptr(vg_index *ix, void **d_counts, uint64_t *n_u32)
{
	if (!ix || !d_counts || !n_u32) return fail(VG_EINVAL, "null argument");
	{ int rc = finish_pending(ix); if (rc) return rc; }         // the sums are only meaningful once the batches in flight have landed
	*d_counts = ix->d.cnt; *n_u32 = 2 * ix->n_sites;
	return VG_OK;
}

// RCCL is resolved lazily (dlopen) so that the library loads, and the rest of the ABI works, on hosts without it; types,
// enumerators and prototypes come from its own header.
#include <rccl/rccl.h>
namespace {
struct Rccl {
	decltype(&ncclAllReduce) all_reduce = nullptr;
	decltype(&ncclCommInitAll) comm_init_all = nullptr;
	decltype(&ncclCommDestroy) comm_destroy = nullptr;
	decltype(&ncclGroupStart) group_start = nullptr;
	decltype(&ncclGroupEnd) group_end = nullptr;
	decltype(&ncclGetErrorString) error_string = nullptr;
	bool ok = false;
};
const Rccl &rccl()
{
	static const Rccl r = [] {
		Rccl x;
		void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
		if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
		if (!h) return x;
		x.all_reduce = (decltype(x.all_reduce))dlsym(h, "ncclAllReduce");
		x.comm_init_all = (decltype(x.comm_init_all))dlsym(h, "ncclCommInitAll");
		x.comm_destroy = (decltype(x.comm_destroy))dlsym(h, "ncclCommDestroy");
		x.group_start = (decltype(x.group_start))dlsym(h, "ncclGroupStart");
		x.group_end = (decltype(x.group_end))dlsym(h, "ncclGroupEnd");
		x.error_string = (decltype(x.error_string))dlsym(h, "ncclGetErrorString");
		x.ok = x.all_reduce && x.comm_init_all && x.comm_destroy && x.group_start && x.group_end && x.error_string;
		return x;
	}();
	return r;
}
}  // namespace

extern "C" int vg_counts_allreduce(vg_index *ix, void *nccl_comm)
{
	if (!ix || !nccl_comm) return fail(VG_EINVAL, "null argument");
	const Rccl &R = rccl();
	if (!R.ok) return fail(VG_ENODEV, "cannot load librccl (or it lacks the collective entry points)");
	{ int rc = finish_pending(ix); if (rc) return rc; }
	if (ix->n_sites == 0) return VG_OK;
	const ncclResult_t rc = R.all_reduce(ix->d.cnt, ix->d.cnt, (size_t)(2 * ix->n_sites), ncclUint32, ncclSum, (ncclComm_t)nccl_comm, ix->stream);
	if (rc != ncclSuccess) return fail(VG_ENODEV, "ncclAllReduce: %s", R.error_string(rc));
	HIP_TRY(hipStreamSynchronize(ix->stream));
	return VG_OK;
}

__global__ void vg_add_counters(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint64_t n)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

// One process driving n replicas (the CLI with VARGENO_GPUS=n): communicator over the replicas' devices, one grouped in-place
// all-reduce of the counters, communicator torn down.  n = 1 is the identity and still goes through RCCL.  Replicas that share
// a device (small indexes; the one-GPU test boxes) are summed on that device first -- RCCL wants one rank per device --, the
// first of them takes part in the all-reduce, and the others get its result.
extern "C" int vg_counts_allreduce_devices(vg_index **handles, int n)
{
	if (!handles || n <= 0) return fail(VG_EINVAL, "null argument");
	for (int i = 0; i < n; i++) {
		if (!handles[i]) return fail(VG_EINVAL, "null handle");
		if (handles[i]->n_sites != handles[0]->n_sites) return fail(VG_EINVAL, "the handles do not hold replicas of one index");
		for (int j = 0; j < i; j++) if (handles[j] == handles[i]) return fail(VG_EINVAL, "the same handle twice");
	}
	const Rccl &R = rccl();
	if (!R.ok) return fail(VG_ENODEV, "cannot load librccl (or it lacks the collective entry points)");
	for (int i = 0; i < n; i++) { int rc = finish_pending(handles[i]); if (rc) return rc; }
	if (handles[0]->n_sites == 0) return VG_OK;
	return guarded([&]() -> int {
		// the order of operations lives in vg_allreduce_plan.h (so that it can be run against a mock); this is its HIP + RCCL backend
		struct Backend {
			vg_index **h; const Rccl &R; uint64_t words; std::vector<ncclComm_t> comms; ncclResult_t last = ncclSuccess;
			int device_of(int i) { return h[i]->device; }
			int add_into(int dst, int src)
			{
				if (hipSetDevice(h[dst]->device) != hipSuccess) return VG_ENODEV;
				vg_add_counters<<<(unsigned)std::min<uint64_t>((words + 255) / 256, 4096), 256, 0, h[dst]->stream>>>(h[dst]->d.cnt, h[src]->d.cnt, words);
				return hipGetLastError() == hipSuccess ? 0 : VG_ENODEV;
			}
			int comm_init(const int *devs, int nr) { comms.assign((size_t)nr, nullptr); last = R.comm_init_all(comms.data(), nr, devs); return last == ncclSuccess ? 0 : VG_ENODEV; }
			int group_start() { last = R.group_start(); return last == ncclSuccess ? 0 : VG_ENODEV; }
			int group_end() { const ncclResult_t r = R.group_end(); if (r != ncclSuccess) last = r; return r == ncclSuccess ? 0 : VG_ENODEV; }
			int all_reduce(int i, int rank)
			{
				if (hipSetDevice(h[i]->device) != hipSuccess) { last = ncclUnhandledCudaError; return VG_ENODEV; }
				last = R.all_reduce(h[i]->d.cnt, h[i]->d.cnt, (size_t)words, ncclUint32, ncclSum, comms[(size_t)rank], h[i]->stream);
				return last == ncclSuccess ? 0 : VG_ENODEV;
			}
			int copy_from(int dst, int src)
			{
				return hipSetDevice(h[src]->device) == hipSuccess && hipMemcpyAsync(h[dst]->d.cnt, h[src]->d.cnt, words * 4, hipMemcpyDeviceToDevice, h[src]->stream) == hipSuccess ? 0 : VG_ENODEV;
			}
			int sync(int i) { return hipSetDevice(h[i]->device) == hipSuccess && hipStreamSynchronize(h[i]->stream) == hipSuccess ? 0 : VG_ENODEV; }
			void comm_destroy() { for (ncclComm_t c : comms) if (c) (void)R.comm_destroy(c); comms.clear(); }
		} B{handles, R, 2 * handles[0]->n_sites, {}};
		const char *where = "";
		const int rc = run_allreduce(B, n, &where);
		if (rc) return fail(rc, "RCCL exchange of the site counters failed at: %s (%s)", where, B.last != ncclSuccess ? R.error_string(B.last) : "HIP error");
		return VG_OK;
	});
}
