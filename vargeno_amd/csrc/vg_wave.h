// vg_wave.h -- the wave-cooperative read-loop kernel (gfx950; workgroups of four independent 64-lane wavefronts).
//
// Why not one read per lane end to end: 8 % of the 32-base chunks are "gate-open" (src/qv.cc:943) and
// need ~100 more dictionary queries each, so in a lane-per-read kernel one lane of nearly every wave
// walks a 100-300 deep chain of dependent gathers while 63 lanes wait -- the first kernel of this
// repo ran at 14 G gathers/s, 29 % of the chip's measured random-gather ceiling, for that reason.
// Here a wave keeps 64 (read, pass) jobs in flight and runs each pass in three stages:
//   A  lane-parallel   exact ref/SNP look-ups of every chunk            (src/qv.cc:840-937)
//   B  wave-parallel   every gate-open (owner, chunk) pair of the wave gets a table row (B0, one pair per lane);
//                      the pairs' Hamming-1 neighbour queries / strided bucket-scan probes are laid end to end and
//                      dealt to the lanes 64 at a time (B1); accepted hits are compacted with a ballot + segmented
//                      prefix sum into the owner's list in canonical order   (src/qv.cc:943-1365)
//   C  lane-parallel   the order-dependent vote is replayed per read from the two short lists, then
//                      the supporting contexts walk the pile-up            (src/qv.cc:132-178, 1375-1502)
// What the kernel is bound by is the number of times a wave WAITS for memory per pass, so every stage
// issues all its independent gathers before it consumes any: four chunks' direct-table records, bucket remainders in
// pairs of chunks, the five table words of a pair, both dictionary queries of a neighbour k-mer in lock step, the
// rank blocks under a whole read.  Wave-uniform values are forced into scalar registers: the kernel sits exactly at
// the 128-VGPR step of 4 waves per SIMD.
// Lists and vote keys live in LDS ([slot][lane], conflict-free).  Neighbour contexts
// whose implied read position is not the position of any exact hit of the same pass can neither
// vote (qv.cc:134-139) nor support the winner, so stage B drops them -- the lists stay tiny.
// A job that outgrows its lists touches no counter and is handed, whole, to the next tier: the same
// kernel with deeper lists, then the generic lane machine of vg_device.h (exactness is never traded).
#pragma once
#include "vg_device.h"

#ifndef VG_WPE
#define VG_WPE 4          // waves per SIMD the main tier is compiled for (128 VGPRs)
#endif
#ifndef VG_SCAN_W
#define VG_SCAN_W 2        // further entries of a multi-entry bucket fetched together in stage A
#endif

// -DVG_STAGE_CLOCKS: development aid, never in the shipped build -- a sample of waves prints the core-clock cycles they spent per stage
#ifdef VG_STAGE_CLOCKS
#define VG_CLK(i) do { const long long t_ = clock64(); clk[i] += t_ - tlast; tlast = t_; } while (0)
#define VG_CLKW(i) do { __builtin_amdgcn_s_waitcnt(0); VG_CLK(i); } while (0)
#else
#define VG_CLK(i) do { } while (0)
#define VG_CLKW(i) do { } while (0)
#endif

#ifdef VG_STAGE_CLOCKS
__device__ unsigned long long vg_dbg_ovf[8];     // [tier*4 + reason]: 0 key table full, 1 neighbour list / two scan candidates in one item, 2 a chunk voted twice for a key
#define VG_OVF(r) atomicAdd(&vg_dbg_ovf[(WPB > 1 ? 0 : 4) + (r)], 1ull)
#else
#define VG_OVF(r) do { } while (0)
#endif

namespace vg {

// Capacities per job-pass (LDS, [slot][lane]): vote keys and neighbour contexts.  The main tier keeps 16 waves per CU
// resident; the deeper tiers take the reads that spill from it (more distinct positions than it has key slots, many
// neighbour contexts) -- still wave-parallel, so a heavy read costs a few dozen dependent gathers instead of the
// thousands the sequential lane machine needs.
#ifndef VG_W1_ECAP
#define VG_W1_ECAP 14     // vote keys (distinct implied read positions of a pass's exact contexts) per lane in the main tier
#endif
#ifndef VG_W1_NCAP
#define VG_W1_NCAP 6      // neighbour contexts per lane in the main tier (with VG_W1_ECAP: the most that still leaves 4 workgroups per CU)
#endif
#ifndef VG_W1_WPB
#define VG_W1_WPB 4
#endif
constexpr int W1_ECAP = VG_W1_ECAP, W1_NCAP = VG_W1_NCAP, W1_WPB = VG_W1_WPB;   // W1_WPB: waves per workgroup of the main tier
// Deep tier: 48 keys + 40 neighbour contexts per lane, single-wave workgroups of 42 000 bytes of LDS (a workgroup of it fits where
// ONE main-tier workgroup has retired: 38 928 + the 4 300 the four of them leave free per CU); then the generic lane machine with
// its lists in HBM.  (r03 had a 40 + 16 tier in between, six workgroups per CU, for the 10 % of a repeat-rich genome's reads that
// outgrew 14-context lists; with vote keys 0.3 % do, and a tier enqueued behind another one costs more than it brings -- see
// enqueue_batch.)
constexpr int W3_ECAP = 48, W3_NCAP = 40;
constexpr uint32_t NOHIT = 0xFFFFFFFFu;   // "no entry": (uint32_t)-1, which is also what a failed query's -1 truncates to
#ifndef VG_SEC_W
#define VG_SEC_W 8
#endif
constexpr int SEC_W = VG_SEC_W;   // entries of an LO32-view bucket fetched in one go (buckets average 1-3 entries; hg38: 2.7)
#ifndef VG_SEC_LONG
#define VG_SEC_LONG 256
#endif
constexpr uint32_t SEC_LONG = VG_SEC_LONG;   // a longer bucket, up to this many entries, is dealt to the lanes record by record in stage B1
constexpr int PCAP = 32;         // rows of the stage-B pair table (a wave with more gate-open chunks takes several windows)
constexpr int HCAP = 4;          // high-half reference hits per pair kept from the LO32-ordered view (more: the 48 queries are issued)

// packed reads: chunk k-mers at [offsets[r] >> 5 ...), one flag word per read
constexpr uint64_t PK_SKIP_N = 1ull << 62;     // an N inside the trimmed read: skipped (qv.cc:815-828)
constexpr uint64_t PK_INVALID = 1ull << 63;    // another character: the reference aborts (util.c:103)
constexpr uint64_t PK_LONG = 1ull << 61;       // more than 32 chunks: generic tier

// context meta word: chunk (5) | mod (5) << 5 | neighbour flag << 10 | new base (2) << 11
__device__ inline uint32_t mk_meta(uint32_t chunk, uint32_t mod, bool neigh, uint32_t nbase) { return chunk | (mod << 5) | ((neigh ? 1u : 0u) << 10) | (nbase << 11); }

// -DVG_FUSE_PACK (experiment, r04): the main tier encodes the reads itself as it takes them (no pack kernel in front of it):
// a lane loads its own read's bases, packs them (SWAR, like the pack kernel), writes k-mers and flag word where the later
// stages, the deeper tiers and the lane machine read them.  bases == nullptr: the batch is packed already.
struct FuseIn {
	const uint8_t *bases, *quals;
	const uint32_t *gate;
	uint32_t *invalid_reads;
};
#ifndef VG_FUSE_PACK
#define VG_FUSE_PACK 0
#endif

template <bool WIDE> struct XKey { typedef uint32_t type; };
template <> struct XKey<true> { typedef uint64_t type; };
template <bool NARROW> struct KMask { typedef uint32_t type; };
template <> struct KMask<true> { typedef uint16_t type; };

template <bool STATS>
__device__ inline uint32_t wave_sum(uint32_t v)
{
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	return v;
}

// read_ids == nullptr: the wave owns a contiguous range of the batch's n_reads_arg reads;
// otherwise the jobs are read_ids[0 .. *n_ids) (the spill list of the previous tier, sized on the device).
// Waves of one workgroup never talk to each other: WPB > 1 only exists because a CU takes at most 16
// workgroups, so single-wave groups cap residency at 16 waves per CU (measured: a 17th wave per CU queues).
// Cross-lane hand-offs through LDS stay inside a wave, where DS operations issue and complete in order;
// the wavefront-scope fence pair keeps the compiler from moving LDS accesses across the hand-off.
#define VG_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// WORK_CHUNK: reads a wave pulls from the launch's work counter at a time (large for the main tier, a handful for the
// spill tier, whose few hundred heavy reads must spread over all its waves)
// NOMX: the instantiation for an index without the merged view / direct table (2^32 or more k-mers in the two dictionaries
// together, or VG_NO_MX).  A kernel of its own (vg_wave_kernel_big) rather than a branch: the two exact-look-up stages would
// otherwise share one register allocation, and the kernel has no register to spare.
// SDX: the instantiation for a direct table of fewer than 2^32 buckets (DevIndex::dx_bits < 32: small indexes, and hg38-scale ones
// under a budget): the look-up compares (F, lo32) -- the key's bits below the bucket -- where the 2^32-bucket form compares lo32.
// A kernel of its own for the same reason as NOMX: the headline instantiation keeps its registers and its instruction stream.
template <bool STATS, int W_ECAP, int W_NCAP, int WPB, bool NOMX, bool SDX = false>
__device__ __forceinline__ void vg_wave_body(const DevIndex &d, const uint64_t *__restrict__ pk_kmer, const uint64_t *__restrict__ pk_meta,
                                             const uint64_t *__restrict__ offsets, uint64_t n_reads_arg,
                                             const uint32_t *__restrict__ read_ids, const uint32_t *__restrict__ n_ids,
                                             uint32_t *overflow_list, uint32_t *overflow_count, uint32_t *work_next, const uint32_t WORK_CHUNK_ARG, unsigned long long *stats,
                                             const FuseIn fuse)
{
	// Exact contexts are kept BY VOTE KEY, not one by one (r04).  A vote key (qv.cc:132-178) is the IMPLIED READ POSITION of a
	// context (k-mer position - 32 chunk) and is always opened by an exact context -- neighbour contexts never open one,
	// :134-139.  All the rest of the pass asks of the exact contexts is, per key, WHICH CHUNKS had an exact context with it:
	//   * the vote's outcome is a function of per-key totals -- by induction over improved_index_table_add, `best` is always a key
	//     of maximal frequency among the keys that have seen two different k-mer positions (= votes from two different chunks),
	//     and `ambiguous` says whether another such key has the same frequency -- so the order of the votes does not matter, only
	//     each key's count (exact contexts + the neighbour contexts of chunks not before the key's first exact chunk) and whether
	//     two chunks contributed (tests/test_vote_aggregate.py replays the reference's own state machine against this rule);
	//   * the walk needs every supporting context's k-mer position: key + 32 chunk, for the chunks of the winning key's mask;
	//   * stage B's filter asks whether a neighbour's implied position is one of the keys.
	// So a lane holds K_idx[slot] = key, K_mask[slot] = chunks: a read inside a c-copy repeat (4 chunks x c positions: 4c contexts,
	// which outgrew the 14-context lists of r03 from c = 4 on, twice per read) is c slots.  Dictionary positions of one chunk are
	// distinct (`vargeno index` skips ALT = REF records, dictgen.c:745), so a chunk cannot vote twice for a key with exact contexts;
	// should a hand-made index do it anyway the read is passed on, down to the lane machine, which keeps contexts one by one.
	// The main tier's masks are 16 bits wide (reads of more than 16 chunks -- 543 bases -- start in the next tier).
	using kmask_t = typename KMask<(WPB > 1)>::type;
	constexpr uint32_t KMASK_CHUNKS = WPB > 1 ? 16u : 32u;
	__shared__ uint32_t K_idx[W_ECAP][64 * WPB], N_kpos[W_NCAP][64 * WPB];
	__shared__ kmask_t K_mask[W_ECAP][64 * WPB];
	__shared__ uint16_t N_meta[W_NCAP][64 * WPB];
	// stage B pair table: one row per gate-open (owner, chunk) pair of the wave, PCAP rows at a time
	__shared__ uint32_t P_klo[PCAP][WPB], P_khi[PCAP][WPB], P_lo[PCAP][WPB], P_hi[PCAP][WPB], P_slo[PCAP][WPB], P_shi[PCAP][WPB];
	__shared__ uint32_t P_meta[PCAP][WPB], P_cnt[PCAP][WPB], P_off[PCAP][WPB], P_hu[PCAP][WPB], P_hidx[HCAP][PCAP][WPB];
	__shared__ uint8_t P_ecnt[PCAP][WPB], P_hamb[PCAP][WPB];
	__shared__ uint16_t N_cnt[64 * WPB];
	__shared__ uint8_t N_ovf[64 * WPB];
	__shared__ unsigned long long Q_state;              // work pool of the workgroup: hi32 = end, lo32 = cursor (may run past the end)
	__shared__ uint32_t Q_lock, Q_done;
	constexpr int NSH = 11;                               // event counters that helper lanes bump on behalf of an owner
	constexpr int SH_IDS[NSH] = {S_REF_QUERY, S_SNP_QUERY, S_REF_PROBE, S_SNP_PROBE, S_SCAN_REF, S_SCAN_SNP, S_SCAN_OOB, S_AUX_REF, S_AUX_SNP, S_SITE_TEST, S_CTX};
	__shared__ uint32_t S_own[STATS ? NSH : 1][STATS ? 64 * WPB : 1];
	const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform
	// The secondary (LO32-ordered) view answers the 48 high-half reference queries with one bucket read.  The counting
	// build keeps the 48 individual queries because the event counters price each of them.
	const bool use_sec = !STATS && d.sec3 != nullptr;
	const bool bf_from_sec = use_sec && d.sec_is_bf != 0u;          // the reference bit vector is the LO32 set of the dictionary (verified by the loader)
#if defined(VG_AB_NO_SSEC_CODE)                                     // (A/B builds only: the SNP view's code compiled out)
	constexpr bool use_ssec = false;
#else
	const bool use_ssec = !STATS && d.ssec3 != nullptr;
#endif              // ... and the SNP dictionary's counterpart answers the (up to 36 + 12) high-half SNP queries
	const bool use_mx = !STATS && !NOMX && d.mx != nullptr;
	const bool use_sig = !STATS && d.snp_sig != nullptr;
	const bool use_probe = !STATS && !use_sig && d.snp_probe != nullptr;
	// 32 entries of the SNP bucket (their signatures: 64 contiguous bytes) per stage-B item (r05; r03-r04: eight), and four records of a
	// long LO32 bucket, in the main tier only: an item with two candidates sends its read to the next tier, and the deep-list tier
	// must be able to finish such a read itself (the lane tier behind it takes milliseconds per read).  Stage B1 is bound by the
	// instructions it issues per ROUND of 64 items, whatever the items hold (profiles/stageb1_rounds_r04.txt): at hg38 scale a SNP
	// bucket of ~19 entries is one item instead of three, a 100-record LO32 bucket of a repeat family 25 items instead of 100.
#ifndef VG_SIG_W
#define VG_SIG_W 4                                                                    // 16-byte loads of signatures per item: 4 = 32 signatures (1, 2: experiments)
#endif
#ifndef VG_LW_LOG
#define VG_LW_LOG 2                                                                   // log2 of the LO32-bucket records per item
#endif
	constexpr uint32_t SIG_W = VG_SIG_W, SIG_LOG = SIG_W == 4 ? 5u : SIG_W == 2 ? 4u : 3u;
	const uint32_t sw_log = WPB > 1 && use_sig ? SIG_LOG : 0u;                        // log2 of the entries per item (wave-uniform)
	const uint32_t sw_m1 = (1u << sw_log) - 1u;
	constexpr uint32_t LW_LOG = WPB > 1 ? VG_LW_LOG : 0u, LW = 1u << LW_LOG;          // records of a long LO32 bucket per item
	const uint32_t col0 = wv << 6;                       // first column of this wave (scalar)
	// lane in the wave / this lane's LDS column: recomputed where they are used (two ALU operations) instead of held in a register
	// from the first line of the kernel to its last -- the main tier sits exactly at its register budget
#define lane (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)))
#define col (col0 + lane)
	const uint64_t n_reads = n_ids ? (uint64_t)*n_ids : n_reads_arg;     // (n_ids without read_ids: a batch framed on the device, which alone knows its size)
	// A list-driven launch (the spill tier) does not know its size on the host: its chunk grows with the list, from the
	// WORK_CHUNK_ARG reads that spread a few hundred heavy reads over all waves up to 16 when there are tens of thousands
	uint32_t WORK_CHUNK = WORK_CHUNK_ARG;
	if constexpr (WPB == 1) if (read_ids) {                 // (the spill tier is the single-wave instantiation)
		const uint64_t fair = n_reads / (2ull * gridDim.x * WPB);
		if (fair > WORK_CHUNK) WORK_CHUNK = fair < 16 ? (uint32_t)fair : 16u;
	}
	// work distribution: a workgroup pulls POOL consecutive reads at a time from one device counter (zeroed per launch)
	// into an LDS pool that its waves drain a refill at a time (see the refill stage), so the last waves to finish differ
	// by part of a pool instead of by the variance of a static 1/n_waves share
	const uint32_t POOL = WORK_CHUNK * (WPB > 1 ? WPB / 2 : 1);
	// the first pool of every workgroup is its own slice of the first gridDim.x * POOL reads -- no counter traffic while all
	// workgroups start at once; the launch counter hands out what lies beyond (work_base)
	const uint32_t work_base = gridDim.x * POOL;          // (< 2^19; a launch holds < 2^32 - 2^24 reads, so the sums below fit 32 bits)
	if (threadIdx.x == 0) {
		const uint64_t p0 = (uint64_t)blockIdx.x * POOL, p1 = p0 + POOL < n_reads ? p0 + POOL : n_reads;
		Q_state = p0 < n_reads ? ((unsigned long long)p1 << 32) | p0 : 0ull;
		Q_lock = 0u; Q_done = 0u;
	}
	__syncthreads();
	uint32_t cursor = 0, end = 0;                        // wave-uniform (kept in scalar registers); a launch holds < 2^32 reads
	bool drained = false;

	bool active = false;
#ifdef VG_STAGE_CLOCKS
	long long clk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
	uint32_t iters = 0, dbg_pairs = 0, dbg_items = 0, dbg_rounds = 0, dbg_dq = 0, dbg_rows = 0, dbg_large = 0, dbg_secbad = 0;
#endif
	uint32_t rid = 0, n = 0, gates = 0, pass = 0;
	uint32_t slot0 = 0;                                  // first k-mer slot of the read (a batch holds < 2^37 bases: checked on the host)
	LaneStats<STATS> tot, cur;
	tot.clear(); cur.clear();

	auto chunk_kmer = [&](uint32_t c) -> uint64_t {
		VG_VC_AS(VC_READS, pk_kmer + ((uint64_t)slot0 + (pass ? n - 1 - c : c)), 8);
		const uint64_t kf = pk_kmer[(uint64_t)slot0 + (pass ? n - 1 - c : c)];
		return pass ? revcomp64(kf) : kf;
	};

	// chunks c and c + 1 of the current strand sit side by side in pk_kmer whichever the strand: one 16-byte gather
	auto chunk_kmer2 = [&](uint32_t c, uint64_t &k0, uint64_t &k1) {
		ulonglong2 v;
		VG_VC_AS(VC_READS, pk_kmer + ((uint64_t)slot0 + (pass ? n - 2 - c : c)), 16);
		__builtin_memcpy(&v, pk_kmer + ((uint64_t)slot0 + (pass ? n - 2 - c : c)), 16);
		k0 = pass ? revcomp64(v.y) : v.x;
		k1 = pass ? revcomp64(v.x) : v.y;
	};

	for (;;) {
		// ------------------------------------------------------------------ refill free lanes
		// every free lane gets a read as long as the launch has any: a chunk that runs out mid-refill is followed by the next
		for (;;) {
			const uint64_t freem = __ballot(!active);
			if (!freem) break;
			if (cursor == end) {
				if (drained) break;
				// The workgroup's waves share a pool of POOL consecutive reads taken from the launch's counter; a wave takes from
				// the pool exactly as many reads as it has free lanes (LDS atomic), so within a workgroup a slow wave ends up
				// with fewer reads, and the launch's counter is touched once per POOL reads.
				uint32_t c0 = 0, c1 = 0;
				if (lane == 0) {
					const uint32_t want = (uint32_t)__popcll(freem);
					for (;;) {
						const unsigned long long old = atomicAdd(&Q_state, (unsigned long long)want);
						if ((uint32_t)old < (uint32_t)(old >> 32)) { c0 = (uint32_t)old; c1 = (uint32_t)(old >> 32) - c0 < want ? (uint32_t)(old >> 32) : c0 + want; break; }
						// the pool is empty: one wave refills it, the others wait for that (bounded: after a while a wave helps itself)
						bool refilled = false;
						for (uint32_t spin = 0; spin < 4096u && !refilled; spin++) {
							if (atomicAdd(&Q_done, 0u)) break;
							if (atomicCAS(&Q_lock, 0u, 1u) == 0u) {
								const unsigned long long now = atomicAdd(&Q_state, 0ull);
								if ((uint32_t)now >= (uint32_t)(now >> 32) && !atomicAdd(&Q_done, 0u)) {
									const uint32_t g0 = work_base + atomicAdd(work_next, POOL);
									if ((uint64_t)g0 >= n_reads) atomicExch(&Q_done, 1u);
									else { const uint32_t g1 = (uint64_t)g0 + POOL < n_reads ? g0 + POOL : (uint32_t)n_reads; atomicExch(&Q_state, ((unsigned long long)g1 << 32) | g0); }
								}
								atomicExch(&Q_lock, 0u);
								refilled = true;
							} else {
								__builtin_amdgcn_s_sleep(4);
								const unsigned long long now = atomicAdd(&Q_state, 0ull);
								refilled = (uint32_t)now < (uint32_t)(now >> 32);
							}
						}
						if (atomicAdd(&Q_done, 0u)) { c0 = c1 = 0xFFFFFFFFu; break; }
						if (!refilled) {                                      // (never seen) the refilling wave is taking its time: a private chunk
							const uint32_t g0 = work_base + atomicAdd(work_next, want);
							if ((uint64_t)g0 >= n_reads) { c0 = c1 = 0xFFFFFFFFu; } else { c0 = g0; c1 = (uint64_t)g0 + want < n_reads ? g0 + want : (uint32_t)n_reads; }
							break;
						}
					}
				}
				c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0);
				c1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c1);
				if (c0 == 0xFFFFFFFFu) { drained = true; break; }
				cursor = c0; end = c1;
			}
			const uint32_t avail = end - cursor;
			const uint32_t nfree = (uint32_t)__popcll(freem);
			const uint32_t take = avail < nfree ? avail : nfree;
			if (!active) {
				const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(freem >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)freem, 0u));   // free lanes below this one
				if (rank < take) {
					rid = read_ids ? read_ids[cursor + rank] : (uint32_t)(cursor + rank);
					VG_VC_AS(VC_READS, offsets + rid, 16); VG_VC_AS(VC_READS, pk_meta + rid, 8);
					const uint64_t off = offsets[rid];
					n = (uint32_t)((offsets[rid + 1] - off) >> 5);
					slot0 = (uint32_t)(off >> 5);
					uint64_t meta;
					if (VG_FUSE_PACK && WPB > 1 && fuse.bases) {
						// encode the read here: chunk k-mers out of the lane's own 32 n bases (16-byte loads at the read's own alignment, two
						// chunks' worth in flight: four spill registers), the flag word as the pack kernel makes it
						uint64_t bad = 0;
						meta = 0;
						uint64_t *dst = const_cast<uint64_t *>(pk_kmer) + slot0;
						const uint8_t *src = fuse.bases + off;
						for (uint32_t c0 = 0; c0 < n; c0 += 2) {
							const bool two = c0 + 1 < n;
							uint4 t[4];
							#pragma unroll
							for (uint32_t q = 0; q < 4; q++) if (q < 2 || two) t[q] = load_policy<false, uint4, 1>(src + 32 * c0 + 16 * q);
							#pragma unroll
							for (uint32_t z = 0; z < 2; z++) if (z == 0 || two) {
								const uint4 a = t[2 * z], b = t[2 * z + 1];
								const uint64_t k = (uint64_t)pack8(((uint64_t)a.y << 32) | a.x, bad) | ((uint64_t)pack8(((uint64_t)a.w << 32) | a.z, bad) << 16) |
								                   ((uint64_t)pack8(((uint64_t)b.y << 32) | b.x, bad) << 32) | ((uint64_t)pack8(((uint64_t)b.w << 32) | b.z, bad) << 48);
								dst[c0 + z] = k;
							}
						}
						if (fuse.gate) meta = n >= 32 ? fuse.gate[rid] : (fuse.gate[rid] & ((1u << n) - 1u));
						else for (uint32_t c0 = 0; c0 < n && c0 < 32; c0 += 4) {
							uint32_t q4;
							__builtin_memcpy(&q4, fuse.quals + off + c0, 4);       // c0 + 4 <= n + 3 <= the read's own length
							for (uint32_t j = 0; j < 4 && c0 + j < n && c0 + j < 32; j++) if ((int)(int8_t)(q4 >> (8 * j)) - '8' < 0) meta |= 1ull << (c0 + j);
						}
						if (bad) meta |= classify_bad(src, n) == 1 ? PK_SKIP_N : PK_INVALID;
						if (n > 32) meta |= fuse.gate ? PK_INVALID : PK_LONG;
						if (meta & PK_INVALID) atomicAdd(fuse.invalid_reads, 1u);
						const_cast<uint64_t *>(pk_meta)[rid] = meta;
					} else meta = pk_meta[rid];
					gates = (uint32_t)meta;
					pass = 0;
					cur.clear();
					cur.add(S_READS, 1);
					cur.add(S_INGEST, 9 * n);
					if (meta & (PK_SKIP_N | PK_INVALID)) {
						cur.add((meta & PK_INVALID) ? S_READS_INVALID : S_READS_N, 1);
						if constexpr (STATS) for (int i = 0; i < S_COUNT; i++) tot.v[i] += cur.v[i];
					} else if ((meta & PK_LONG) || n > KMASK_CHUNKS) {        // more chunks than this tier's key masks have bits
						overflow_list[atomicAdd(overflow_count, 1u)] = rid;
					} else {
						active = true;
					}
				}
			}
			cursor += take;
		}
		if (!__any(active)) { if (drained) break; continue; }
		VG_CLK(0);
#ifdef VG_STAGE_CLOCKS
		iters++;
#endif

		// ------------------------------------------------------------------ stage A: exact look-ups
		uint32_t kcnt = 0, ncnt = 0;                             // vote keys of this pass (slots [0, kcnt)), neighbour contexts
		uint32_t npend = 0;                                      // auxiliary rows this lane has queued for the wave to expand together (below)
		bool ovf = false;
		if (active) {
			// an exact context of chunk c at k-mer position p (qv.cc:850-937): find its vote key among slots [lo, hi) of the lane's
			// table, four candidates per LDS round trip starting at `hint` -- the auxiliary rows of consecutive chunks of a read inside
			// a repeat list the same copies in the same order, so the slot after the last match is usually the one.  Returns the slot,
			// or W_ECAP when the key is new (the caller inserts it: the forward list grows up from 0, the reverse-strand block down
			// from the top).  A chunk that has voted for the key already: dup = true (see above).
			uint32_t hint = 0;
			auto key_find = [&](uint32_t q, uint32_t c, uint32_t lo, uint32_t hi, bool &dup) -> uint32_t {
				uint32_t e = (uint32_t)W_ECAP;
				const uint32_t cnt = hi - lo;
				if (cnt) {
					if (hint < lo || hint >= hi) hint = lo;
					for (uint32_t done = 0; done < cnt && e == (uint32_t)W_ECAP; done += 4) {
						uint32_t v[4], at[4];
						#pragma unroll
						for (uint32_t z = 0; z < 4; z++) { uint32_t i = hint + done + z; if (i >= hi) i -= cnt; if (i >= hi) i = lo; at[z] = i; v[z] = K_idx[i][col]; }
						#pragma unroll
						for (uint32_t z = 0; z < 4; z++) if (e == (uint32_t)W_ECAP && done + z < cnt && v[z] == q) e = at[z];
					}
				}
				if (e != (uint32_t)W_ECAP) {
					const uint32_t m = K_mask[e][col];
					if ((m >> c) & 1u) dup = true; else K_mask[e][col] = (kmask_t)(m | (1u << c));
					hint = e + 1u;
				}
				return e;
			};
			auto push_exact = [&](uint32_t p, uint32_t c) {
				cur.add(S_CTX, 1);
				if (ovf) return;
				bool dup = false;
				const uint32_t q = p - 32u * c;
				if (key_find(q, c, 0u, kcnt, dup) == (uint32_t)W_ECAP) {
					if (kcnt < (uint32_t)W_ECAP) { K_idx[kcnt][col] = q; K_mask[kcnt][col] = (kmask_t)(1u << c); kcnt++; hint = kcnt; } else { VG_OVF(0); ovf = true; }
				} else if (dup) { VG_OVF(2); ovf = true; }
			};
			auto push_row = [&](const uint32_t *row, uint32_t c) {               // a row ends at its first 0
				// The whole row in one wait.  Then EVERY column against EVERY key of the table, four keys per LDS round trip (r04, second
				// half): the rows of a read's chunks list the copies of a repeat family that share THAT chunk's k-mer -- a different subset
				// of the family from chunk to chunk, as soon as the copies have diverged anywhere -- so nothing positional survives (the
				// first key-table build expected column j at slot s + j and searched the table once per column that was not there: 3.4
				// searches per row, 12 000 cycles per row of a wave in which two or three lanes had one; a third of the kernel's time on
				// the repeat-rich genome, profiles/rows_census_r04.txt).  Columns that matched set their chunk's bit; the others are new
				// keys and are appended without a search, at kcnt + their rank among the new ones.  A row's positions are distinct in
				// every index `vargeno index` writes from distinct records; an index whose rows repeat a position (the same SNP record
				// three times or more in the list: DevIndex::aux_dups, found at load time) takes the careful way, column by column, where
				// a chunk that votes twice for a key sends the read down to the lane machine.
#ifdef VG_DBG_NO_ROWS
				return;                                                 // timing experiment only (wrong results): what do the rows cost?
#endif
				uint32_t rw[AUX_COLS];
				VG_VC_AS(VC_AUX_A, row, 40);
				load_row10(row, rw);
				uint32_t todo = 0, newk = 0;                            // columns left to the careful path / columns whose keys are new
				if (!ovf) {
					uint32_t r = 0;
					#pragma unroll
					for (int j = 0; j < AUX_COLS; j++) if (r == (uint32_t)j && rw[j] != 0) r = (uint32_t)j + 1u;      // live columns
					cur.add(S_CTX, r);
					const uint32_t bit = 1u << c, back = 32u * c, live = (1u << r) - 1u;
					if (d.aux_dups) todo = live;
					else {
						uint32_t found = 0;
						bool dup = false;
						#pragma nounroll
						for (uint32_t e0 = 0; e0 < kcnt && found != live; e0 += 4) {
							uint32_t kv[4], km[4];
							#pragma unroll
							for (uint32_t t = 0; t < 4; t++) { const uint32_t e = e0 + t < (uint32_t)W_ECAP ? e0 + t : (uint32_t)W_ECAP - 1u; kv[t] = K_idx[e][col]; km[t] = K_mask[e][col]; }
							#pragma unroll
							for (uint32_t t = 0; t < 4; t++) if (e0 + t < kcnt) {
								uint32_t h = 0;
								#pragma unroll
								for (int j = 0; j < AUX_COLS; j++) h |= (rw[j] - back == kv[t] ? 1u : 0u) << j;
								h &= live;
								if (h) {
									if (km[t] & bit) dup = true;                        // the chunk has voted for this key already
									else K_mask[e0 + t][col] = (kmask_t)(km[t] | bit);
									found |= h;
								}
							}
						}
						if (dup) { VG_OVF(2); ovf = true; }
						else newk = live & ~found;
						hint = 0;
					}
				}
				if (newk && !ovf) {
					const uint32_t nn = (uint32_t)__popc(newk);
					if (kcnt + nn > (uint32_t)W_ECAP) { VG_OVF(0); ovf = true; }
					else {
						#pragma unroll
						for (int j = 0; j < AUX_COLS; j++) if ((newk >> j) & 1u) {
							const uint32_t e = kcnt + (uint32_t)__popc(newk & ((1u << j) - 1u));
							K_idx[e][col] = rw[j] - 32u * c; K_mask[e][col] = (kmask_t)(1u << c);
						}
						kcnt += nn;
					}
				}
				#pragma nounroll
				while (todo && !ovf) {
					const uint32_t j = (uint32_t)__ffs((int)todo) - 1u;
					todo &= todo - 1u;
					const uint32_t v = j == 0 ? rw[0] : j == 1 ? rw[1] : j == 2 ? rw[2] : j == 3 ? rw[3] : j == 4 ? rw[4] : j == 5 ? rw[5] : j == 6 ? rw[6] : j == 7 ? rw[7] : j == 8 ? rw[8] : rw[9];
					cur.add(S_CTX, (uint32_t)-1);                            // (push_exact counts it again)
					push_exact(v, c);
				}
			};
			auto emit_exact = [&](uint32_t c, bool rhit, uint32_t rpos, uint32_t ramb, bool shit, uint32_t spos, uint32_t samb) {
				if (rhit && rpos != POS_AMBIGUOUS) {
					if (ramb == 0) push_exact(rpos, c);
					else { cur.add(S_AUX_REF, 1); push_row(d.ref_aux + (uint64_t)rpos * AUX_COLS, c); }
				}
				if (shit && spos != POS_AMBIGUOUS) {
					if (samb == 0) push_exact(spos, c);
					else { cur.add(S_AUX_SNP, 1); push_row(d.snp_aux_pos + (uint64_t)spos * AUX_COLS, c); }
				}
			};
			if constexpr (NOMX) {
				// An index without the merged view: both dictionaries of four chunks at a time, bisected in lock step
				for (uint32_t c = 0; c < n; c += 4) {
					const uint32_t m = n - c < 4u ? n - c : 4u;
					uint64_t kq[4] = {0, 0, 0, 0};
					const bool want[4] = {true, m >= 2, m >= 3, m >= 4};
					if (m >= 2) chunk_kmer2(c, kq[0], kq[1]); else kq[0] = chunk_kmer(c);
					if (m >= 4) chunk_kmer2(c + 2, kq[2], kq[3]); else if (m == 3) kq[2] = chunk_kmer(c + 2);
					uint32_t rhit[4], shit[4], rpos[4], spos[4];
					exact_multi_nomx<4>(d, kq, want, rhit, rpos, shit, spos);
					#pragma unroll
					for (uint32_t z = 0; z < 4; z++) if (want[z]) {
						cur.add(S_CHUNKS, 1);
						emit_exact(c + z, rhit[z] != 0u, rpos[z], rhit[z] >> 1, shit[z] != 0u, spos[z], shit[z] >> 1);
					}
				}
			} else if (use_mx) {
				// The timed build reads the merged view: one jump-table gather + one bucket line answers both dictionaries.
				// Two chunks are in flight at a time (their gathers are issued back to back before either is consumed).
				uint32_t k_strand = 0; bool k_pal = false;                       // strand bit / palindrome flag of the k-mer being looked up (below)
				auto scan_bucket = [&](uint64_t k, uint32_t lo, uint32_t hi, uint4 first, bool &rhit, uint32_t &rpos, uint32_t &ramb, bool &shit, uint32_t &spos, uint32_t &samb) {
					const uint32_t key = (uint32_t)k;
					uint32_t ea = lo;
					if (hi - lo > 4) {                                            // rare big bucket (low-complexity HI32): bisect to the first candidate
						uint32_t eb = hi;
						while (ea < eb) { const uint32_t m = ea + ((eb - ea) >> 1); if (d.mx[m].x < key) ea = m + 1; else eb = m; }
					}
					for (uint32_t e = ea; e < hi; e++) {                          // buckets of the merged view mostly hold 0-2 entries
						const uint4 v = (e == lo) ? first : d.mx[e];
						if (v.x < key) continue;
						if (v.x > key) break;
						if (!(k_pal || ((v.z >> 3) & 1u) == k_strand)) continue;      // the entry of the reverse complement: not a hit of this pass
						if (v.z & 1u) { shit = true; spos = v.y; samb = (v.z >> 1) & 1u; }
						else { rhit = true; rpos = v.y; ramb = (v.z >> 1) & 1u; }
					}
				};
				if (d.dx) {
					// The merged view and its direct table are keyed by the CANONICAL form of a k-mer (the smaller of it and its reverse
					// complement, mixed so that the buckets fill evenly; r03), so the bucket of a chunk holds the entry of the chunk's
					// k-mer -- a hit of THIS pass -- and the entry of its reverse complement -- a hit of the other pass, for the mirrored
					// chunk.  In pass 0 those are collected too (the ones with a single position), in a block that grows down from the
					// top of the exact-context list; if the forward pass finds nothing at all -- every read of the reverse strand --
					// the block IS the exact-context list of pass 1 and the lane goes on as pass 1.
					// The block of reverse-strand contexts is [rc_top, W_ECAP); RC_BAD: it cannot be used (not pass 0; a hit with several
					// positions; no room)
					constexpr uint32_t RC_BAD = 0xFFFFu;
					uint32_t rc_top = pass != 0u ? RC_BAD : (uint32_t)W_ECAP;
					for (uint32_t c = 0; c < n; c += 4) {
						const uint32_t m = n - c < 4u ? n - c : 4u;
						uint64_t kq[4] = {0, 0, 0, 0};
						if (m >= 2) chunk_kmer2(c, kq[0], kq[1]); else kq[0] = chunk_kmer(c);
						if (m >= 4) chunk_kmer2(c + 2, kq[2], kq[3]); else if (m == 3) kq[2] = chunk_kmer(c + 2);
						// k-mer -> mixed canonical key; strand bits: bit z = the k-mer is the reverse complement of its canonical form, bit 4 + z = palindrome
						uint32_t sbits = 0;
						#pragma unroll
						for (uint32_t z = 0; z < 4; z++) {
							const uint64_t kk = kq[z], rr = revcomp64(kk), ck = kk < rr ? kk : rr;
							sbits |= (kk != ck ? 1u : 0u) << z | (kk == rr ? 1u : 0u) << (4u + z);
							kq[z] = fmix64(ck);
						}
						VG_CLKW(9);
						// what an entry is compared with: lo32 of the key -- and, under SDX, field F above it (the high word's bits below the
						// bucket, left-aligned; see DevIndex::dx_bits).  ekey() of a record / entry is the same 32 or 64 bits of ITS key.
						// (the 2^32-bucket instantiation compares 32-bit words, as through round 5: 64-bit keys there cost it four registers and
						// 3-12 % of its time -- profiles/ab_hg38_r06_bisect.txt)
						using xkey_t = typename XKey<SDX>::type;
						const uint32_t dxb = SDX ? d.dx_bits : 32u;
						auto ekey = [](const uint4 &e) -> xkey_t { if constexpr (SDX) return ((uint64_t)(e.z & 0xFFFF0000u) << 32) | e.x; else return e.x; };
						auto ecnt = [](const uint4 &b) -> uint32_t { return SDX ? (b.z >> 8) & 0xFFu : b.z >> 8; };
						uint4 bq[4];
						xkey_t kx[4];
						#pragma unroll
						for (uint32_t z = 0; z < 4; z++) {
							bq[z] = make_uint4(0, 0, 0, 0);
							if (z < m) bq[z] = gather<uint4>(d.dx + (SDX ? kq[z] >> (64u - dxb) : kq[z] >> 32));
							if constexpr (SDX) kx[z] = ((uint64_t)(((uint32_t)(kq[z] >> 32) << dxb) & 0xFFFF0000u) << 32) | (uint32_t)kq[z];      // (F, lo32): what is left to compare
							else kx[z] = (uint32_t)kq[z];
						}
						VG_CLKW(10);
						// `more`: the bucket has further entries that may hold the key (entries are sorted by lo: nothing below the first; a
						// match on the first entry is final unless the table says its successor has the same k-mer -- flag TIE)
						bool more[4];
						#pragma unroll
						for (uint32_t z = 0; z < 4; z++) more[z] = z < m && (bq[z].z & 1u) && ecnt(bq[z]) > 1u && (ekey(bq[z]) < kx[z] || (ekey(bq[z]) == kx[z] && (bq[z].z & 16u)));
						// Two chunks at a time: the rest of a small bucket -- up to VG_SCAN_W more entries -- arrives together (one wait;
						// anything deeper is rare and goes one by one); then each chunk's exact contexts are appended (qv.cc:850-937),
						// reference hit first, then SNP hit.  An ambiguous k-mer with exactly two positions carries both in its entry
						// (flag PAIR, set by vg_inline_pairs), so only k-mers with 3-10 copies still read their auxiliary row.
						#pragma unroll
						for (uint32_t z0 = 0; z0 < 4; z0 += 2) {
							constexpr uint32_t SW = VG_SCAN_W;
							uint4 sv[2][SW];
							#pragma unroll
							for (uint32_t y = 0; y < 2; y++) {
								const uint32_t cnt = ecnt(bq[z0 + y]), lo = bq[z0 + y].w;
								#pragma unroll
								for (uint32_t x = 0; x < SW; x++) { sv[y][x] = make_uint4(0xFFFFFFFFu, 0, 0xFFFF0000u, 0); if (more[z0 + y] && x + 1u < cnt) sv[y][x] = gather<uint4>(d.mx + (lo + 1u + x)); }
							}
							// An auxiliary row (a k-mer with 3-10 copies) is not expanded here but QUEUED -- the neighbour lists' LDS slots are idle
							// during stage A: row index and chunk per entry -- and the wave expands everybody's rows together after the look-ups
							// (below).  A pass with a row has exact hits of its own: the reverse-strand block will not be used.  An index whose
							// rows repeat positions (DevIndex::aux_dups), and a lane whose queue is full, expand the row here, the careful way.
							auto queue_row = [&](const uint32_t *aux, uint32_t row, uint32_t cc, uint32_t is_snp) {
								if (d.aux_dups || npend == (uint32_t)W_NCAP) { push_row(aux + (uint64_t)row * AUX_COLS, cc); return; }
								N_kpos[npend][col] = row; N_meta[npend][col] = (uint16_t)(cc | (is_snp << 5));
								npend++;
								rc_top = RC_BAD;
							};
							#pragma unroll
							for (uint32_t y = 0; y < 2; y++) {
								const uint32_t z = z0 + y;
								if (z >= m) continue;
								cur.add(S_CHUNKS, 1);
								const uint4 b = bq[z];
								const xkey_t key = kx[z];                               // (lo32, or (F, lo32) under SDX)
								// hit state: position (or row index), second position of a PAIR, flags 1 hit, 2 ambiguous, 4 PAIR -- for this pass's
								// strand, and (x...) for the other strand
								uint32_t rp = 0, rp2 = 0, rf = 0, sp = 0, sp2 = 0, sf = 0;
								const uint32_t ks = (sbits >> z) & 1u;
								const bool pal = (sbits >> (4u + z)) & 1u;
								// an entry with this chunk's key (es: its strand bit).  One of the other strand goes straight to the block that grows
								// down from the top of the list, as an exact context of pass 1's chunk n - 1 - (c + z) -- unless it has several
								// positions (flag 2, PAIR or not: left to the plain way, which keeps this path small) or there is no room
								auto note = [&](bool is_snp, uint32_t es, uint32_t p1, uint32_t p2, uint32_t f) {
									if (pal || es == ks) { if (is_snp) { sp = p1; sp2 = p2; sf = f; } else { rp = p1; rp2 = p2; rf = f; } }
									if ((pal || es != ks) && rc_top != RC_BAD) {
										if (f & 2u) rc_top = RC_BAD;
										else if (p1 != POS_AMBIGUOUS) {
											const uint32_t c1 = n - 1u - (c + z), q1 = p1 - 32u * c1;
											bool dup = false;
											if (key_find(q1, c1, rc_top, (uint32_t)W_ECAP, dup) == (uint32_t)W_ECAP) {
												if (rc_top <= kcnt) rc_top = RC_BAD;
												else { rc_top--; K_idx[rc_top][col] = q1; K_mask[rc_top][col] = (kmask_t)(1u << c1); }
											} else if (dup) rc_top = RC_BAD;
										}
									}
								};
								if ((b.z & 1u) && ekey(b) == key)                      // the inline first entry (dx flags: 2 SNP, 4 ambiguous, 8 PAIR, 32 strand)
									note((b.z & 2u) != 0u, (b.z >> 5) & 1u, b.y, b.w, 1u | (((b.z >> 2) & 1u) << 1) | (((b.z >> 3) & 1u) << 2));
								if (more[z]) {
									const uint32_t cnt = ecnt(b), lo = b.w, hi = lo + cnt;
									auto take = [&](const uint4 v) {                     // mx flags: 1 SNP, 2 ambiguous, 4 PAIR, 8 strand
										note((v.z & 1u) != 0u, (v.z >> 3) & 1u, v.y, v.w, 1u | (v.z & 2u) | (v.z & 4u));
									};
									#pragma unroll
									for (uint32_t x = 0; x < SW; x++) if (x + 1u < cnt && ekey(sv[y][x]) == key) take(sv[y][x]);
									if (cnt > SW + 1u && ekey(sv[y][SW - 1]) <= key) {  // the bucket goes on and may still hold the key
										uint32_t e = lo + SW + 1u;
										if (cnt > 8u) { uint32_t eb = hi; while (e < eb) { const uint32_t mm = e + ((eb - e) >> 1); if (ekey(d.mx[mm]) < key) e = mm + 1; else eb = mm; } }
										for (; e < hi; e++) {
											const uint4 v = d.mx[e];
											if (ekey(v) < key) continue;
											if (ekey(v) > key) break;
											take(v);
										}
									}
								}
								const bool r_ok = (rf & 1u) && ((rf & 4u) || rp != POS_AMBIGUOUS), s_ok = (sf & 1u) && ((sf & 4u) || sp != POS_AMBIGUOUS);
								const bool r_ax = r_ok && (rf & 6u) == 2u, s_ax = s_ok && (sf & 6u) == 2u;   // ambiguous and not a PAIR: read the row
								if (r_ok) {
									if (rf & 2u) cur.add(S_AUX_REF, 1);
									if (r_ax) queue_row(d.ref_aux, rp, c + z, 0u);
									else { push_exact(rp, c + z); if (rf & 4u) push_exact(rp2, c + z); }
								}
								if (s_ok) {
									if (sf & 2u) cur.add(S_AUX_SNP, 1);
									if (s_ax) queue_row(d.snp_aux_pos, sp, c + z, 1u);
									else { push_exact(sp, c + z); if (sf & 4u) push_exact(sp2, c + z); }
								}
							}
						}
					}
					// A forward pass without a single exact hit can neither vote nor be walked (the stage-B skip below): the lane goes on as
					// pass 1 with the keys collected above -- none at all means that pass 1 has nothing either, and the read is done
					// (the order of the keys does not matter, see the key table's description)
					if (pass == 0u && kcnt == 0u && rc_top != RC_BAD && !ovf) {
						const uint32_t nrc = (uint32_t)W_ECAP - rc_top;
						for (uint32_t i = 0; i < nrc; i++) { K_idx[i][col] = K_idx[rc_top + i][col]; K_mask[i][col] = K_mask[rc_top + i][col]; }
						kcnt = nrc;
						pass = 1u;
					}
				} else
				for (uint32_t c = 0; c < n; c += 2) {
					const bool two = c + 1 < n;
					uint64_t k0 = chunk_kmer(c), k1 = two ? chunk_kmer(c + 1) : 0;
					uint32_t st0, st1; bool pl0, pl1;                              // the merged view is keyed by mixed canonical k-mers
					{ const uint64_t r = revcomp64(k0), ck = k0 < r ? k0 : r; st0 = k0 != ck; pl0 = k0 == r; k0 = fmix64(ck); }
					{ const uint64_t r = revcomp64(k1), ck = k1 < r ? k1 : r; st1 = k1 != ck; pl1 = k1 == r; k1 = fmix64(ck); }
					uint32_t lo0, hi0, lo1 = 0, hi1 = 0;
					jg_pair(d.mx_jg, k0 >> 32, lo0, hi0);
					if (two) jg_pair(d.mx_jg, k1 >> 32, lo1, hi1);
					uint4 f0 = make_uint4(0, 0, 0, 0), f1 = f0;
					if (lo0 < hi0) f0 = d.mx[lo0];
					if (lo1 < hi1) f1 = d.mx[lo1];
					uint32_t rpos = 0, ramb = 0, spos = 0, samb = 0;
					bool rhit = false, shit = false;
					cur.add(S_CHUNKS, 1);
					k_strand = st0; k_pal = pl0;
					scan_bucket(k0, lo0, hi0, f0, rhit, rpos, ramb, shit, spos, samb);
					emit_exact(c, rhit, rpos, ramb, shit, spos, samb);
					if (two) {
						rhit = shit = false;
						cur.add(S_CHUNKS, 1);
						k_strand = st1; k_pal = pl1;
						scan_bucket(k1, lo1, hi1, f1, rhit, rpos, ramb, shit, spos, samb);
						emit_exact(c + 1, rhit, rpos, ramb, shit, spos, samb);
					}
				}
			} else {
				// the counting build does the reference's two separate walks, whose probes the event counters price
				for (uint32_t c = 0; c < n; c++) {
					const uint64_t k = chunk_kmer(c);
					cur.add(S_CHUNKS, 1);
					uint32_t lo, hi;
					RefEnt e; SnpEnt se;
					const bool rhit = ref_query(d, cur, k, lo, hi, e) >= 0;
					const bool shit = snp_query(d, cur, k, lo, hi, se) >= 0;
					emit_exact(c, rhit, rhit ? e.pos : 0u, rhit ? e.amb : 0u, shit, shit ? se.pos : 0u, shit ? (uint32_t)((se.key >> 48) & 0xFFu) : 0u);
				}
			}
		}
		VG_WAVE_SYNC();
		// ---- the queued auxiliary rows, dealt to the lanes of the WAVE (r05).  Through r04 a lane expanded its own rows inside the look-up
		// loop: gather the row (one wait), match it against the key table, next chunk -- four or five dependent gathers per pass for the
		// two or three lanes of 64 that had rows, with the other lanes waiting: a fifth of the kernel on a repeat-rich genome at hg38
		// scale (profiles/stage_clocks_r05.txt).  Now ALL queued rows of the wave are fetched together -- row i of the wave by lane i,
		// whoever owns it: one wait for all of them --, and only the matching is done in the owners' order (round r: every owner's r-th
		// row against its key table, by the lane that holds the row; LDS and arithmetic, no memory wait), so that a later row finds
		// the keys an earlier one appended.
		VG_CLK(11);
		if constexpr (!STATS && !NOMX) {
			const uint32_t mine = (active && !ovf) ? npend : 0u;
			uint32_t incl = mine;
			for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o); if ((int)lane >= o) incl += y; }
			const uint32_t base = incl - mine, total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
			uint32_t maxr = mine;
			for (int o = 32; o > 0; o >>= 1) { const uint32_t y = __shfl_xor(maxr, o); maxr = y > maxr ? y : maxr; }
			maxr = (uint32_t)__builtin_amdgcn_readfirstlane((int)maxr);
			for (uint32_t it0 = 0; it0 < total; it0 += 64) {
				const uint32_t item = it0 + lane;
				const bool has = item < total;
				uint32_t ow = 0;                                       // the owner: the last lane whose first item is not beyond this one
				for (uint32_t step = 32; step > 0; step >>= 1) { const uint32_t cand = ow + step, bb = __shfl(base, (int)(cand & 63u)); if (cand < 64u && bb <= item) ow = cand; }
				const uint32_t rr_mine = item - __shfl(base, (int)ow), ocol = col0 + ow;
				uint32_t rw[AUX_COLS], cc = 0;
				#pragma unroll
				for (int j = 0; j < AUX_COLS; j++) rw[j] = 0u;
				if (has) {
					const uint32_t row = N_kpos[rr_mine][ocol], mt = N_meta[rr_mine][ocol];
					cc = mt & 31u;
					VG_VC_AS(VC_AUX_A, ((mt >> 5) & 1u ? d.snp_aux_pos : d.ref_aux) + (uint64_t)row * AUX_COLS, 40);
					load_row10(((mt >> 5) & 1u ? d.snp_aux_pos : d.ref_aux) + (uint64_t)row * AUX_COLS, rw);
				}
				uint32_t nlive = 0;
				#pragma unroll
				for (int j = 0; j < AUX_COLS; j++) if (nlive == (uint32_t)j && rw[j] != 0u) nlive = (uint32_t)j + 1u;      // a row ends at its first 0
				const uint32_t bit = 1u << cc, back = 32u * cc, live = (1u << nlive) - 1u;
				for (uint32_t rr = 0; rr < maxr; rr++) {
					const uint32_t kc = __shfl(kcnt, (int)ow);          // the owner's keys as they stand before this round
					const uint32_t ow_ovf = __shfl((uint32_t)(ovf ? 1u : 0u), (int)ow);
					uint32_t add = 0, bad = 0;
					if (has && rr_mine == rr && !ow_ovf) {
						uint32_t found = 0;
						#pragma nounroll
						for (uint32_t e0 = 0; e0 < kc && found != live; e0 += 4) {
							uint32_t kv[4], km[4];
							#pragma unroll
							for (uint32_t t = 0; t < 4; t++) { const uint32_t e = e0 + t < (uint32_t)W_ECAP ? e0 + t : (uint32_t)W_ECAP - 1u; kv[t] = K_idx[e][ocol]; km[t] = K_mask[e][ocol]; }
							#pragma unroll
							for (uint32_t t = 0; t < 4; t++) if (e0 + t < kc) {
								uint32_t h = 0;
								#pragma unroll
								for (int j = 0; j < AUX_COLS; j++) h |= (rw[j] - back == kv[t] ? 1u : 0u) << j;
								h &= live;
								if (h) {
									if (km[t] & bit) bad = 2u;                      // the chunk has voted for this key already: next tier
									else K_mask[e0 + t][ocol] = (kmask_t)(km[t] | bit);
									found |= h;
								}
							}
						}
						const uint32_t newk = live & ~found;
						add = (uint32_t)__popc(newk);
						if (!bad && kc + add > (uint32_t)W_ECAP) bad = 1u;
						if (!bad) {
							#pragma unroll
							for (int j = 0; j < AUX_COLS; j++) if ((newk >> j) & 1u) {
								const uint32_t e = kc + (uint32_t)__popc(newk & ((1u << j) - 1u));
								K_idx[e][ocol] = rw[j] - back; K_mask[e][ocol] = (kmask_t)bit;
							}
						}
					}
					// the owner hears from the lane that held its rr-th row (if that row is in this batch of 64)
					const uint32_t src = base + rr - it0;
					const uint32_t add_o = __shfl(add, (int)(src & 63u)), bad_o = __shfl(bad, (int)(src & 63u));
					if (rr < mine && base + rr >= it0 && src < 64u) {
						if (bad_o) { if (bad_o == 2u) VG_OVF(2); else VG_OVF(0); ovf = true; }
						else kcnt += add_o;
					}
					VG_WAVE_SYNC();
				}
			}
		}
		VG_CLK(1);

		// ------------------------------------------------------------------ stage B: all gate-open chunks of the wave, flattened
		// (owner, chunk) pairs in (lane, chunk) order -> one table row each (LDS), their work items laid end to end and dealt
		// to the lanes 64 at a time, so a round is full whatever the size of each chunk's neighbourhood.
		{
			// A pass without a single exact hit has no vote key, and neighbour contexts cannot open one (qv.cc:134-139): nothing its
			// gate-open chunks could find would vote or be walked.  That is the whole forward pass of every reverse-strand read -- a
			// third of all passes -- so the timed build skips stage B for it (the counting build runs it: its events are priced).
#ifdef VG_NO_HOPELESS_SKIP
			const bool hopeless = false;
#else
			const bool hopeless = !STATS && kcnt == 0;
#endif
			const uint32_t pend = (active && !ovf && !hopeless) ? (n >= 32 ? gates : (gates & ((1u << n) - 1u))) : 0u;
			const uint32_t my_np = (uint32_t)__popc(pend);
			uint32_t pincl = my_np;
			for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(pincl, o); if ((int)lane >= o) pincl += y; }
			const uint32_t my_base = pincl - my_np, P = (uint32_t)__builtin_amdgcn_readlane((int)pincl, 63);    // wave-uniform: scalar
			N_cnt[col] = 0; N_ovf[col] = 0;
			if constexpr (STATS) for (int i = 0; i < NSH; i++) S_own[i][col] = 0;
			for (uint32_t w0 = 0; w0 < P; w0 += PCAP) {
				// ---- B0: owners name their pairs of this window (LDS), then lane p fills row p -- one pair per lane whatever
				// the owner, so a read with several gate-open chunks does not make its wave walk them one after the other
				{
					uint32_t q = my_base, bits = pend;
					while (bits) {
						const uint32_t c = (uint32_t)__ffs((int)bits) - 1;
						bits &= bits - 1;
						if (q >= w0 && q < w0 + PCAP) { P_meta[q - w0][wv] = lane | (c << 6); P_ecnt[q - w0][wv] = (uint8_t)kcnt; }
						q++;
					}
				}
				VG_WAVE_SYNC();
				{
					const uint32_t np0 = P - w0 < (uint32_t)PCAP ? P - w0 : (uint32_t)PCAP;
					const uint32_t p = lane < (uint32_t)PCAP ? lane : 0u;
					const bool mine = lane < np0;
					const uint32_t m0 = mine ? P_meta[p][wv] : 0u;
					const uint32_t own = m0 & 63u, c = (m0 >> 6) & 31u;
					// the owner's read: where its k-mers start, how many, which strand
					const uint32_t o_n = __shfl(n, own), o_pass = __shfl(pass, own);
					const uint64_t o_slot0 = __shfl(slot0, own);
					if (mine) {
						VG_VC_AS(VC_READS, pk_kmer + (o_slot0 + (o_pass ? o_n - 1 - c : c)), 8);
						const uint64_t kf = pk_kmer[o_slot0 + (o_pass ? o_n - 1 - c : c)];
						const uint64_t k = o_pass ? revcomp64(kf) : kf;
						const uint32_t klo = (uint32_t)k, khi = (uint32_t)(k >> 32);
						uint32_t lo, hi, slo, shi, b0 = 0, b1 = 0, fl = 0;
						ref_bounds(d, k >> 32, lo, hi);                          // check_block_size, qv.cc:242-264
						jg_pair(d.snp_jg, k >> 40, slo, shi);
						if (use_sec) jg_pair(d.sec_jg, klo >> (32 - d.sec_bits), b0, b1);
						const uint64_t rp = (uint64_t)hash32(klo) % d.ref_bf_bits;   // qv.cc:946-956
						const uint64_t sp = hash40(k & LO40_MASK) % d.snp_bf_bits;
						if (!bf_from_sec) { if ((gather_bf<uint64_t>(d.ref_bf + (rp >> 6)) >> (rp & 63)) & 1u) fl |= 1u; }
						if ((gather_bf<uint64_t>(d.snp_bf + (sp >> 6)) >> (sp & 63)) & 1u) fl |= 2u;
						const bool large = hi - lo >= BLOCK_THRESHOLD;
						// high-half SNP queries are live for a contiguous range of slots u = 3 * (pair - 16) + sel  (qv.cc:1303-1306)
						uint32_t s_lo = 0, s_hi = 0;
						if (fl & 2u) { s_lo = large ? 0u : 12u; s_hi = 48u; } else if (large) { s_lo = 0u; s_hi = 12u; }
						// high-half ref hits from the LO32-ordered view: every dictionary k-mer with the same first 16 bases
						// whose last 16 differ in exactly one base, kept sorted by slot
						uint32_t nh = 0, hu = 0, hamb = 0;                         // hu: slot of hit z in byte z (unsorted); hamb: bit z = hit z is ambiguous
						bool sec_ok = use_sec, longsec = false;
						uint32_t sec_b0 = 0;
						auto sec_entry = [&](uint32_t ehi, uint32_t epos, uint32_t eamb) {   // one view entry with this LO32: HI32, position (or row), ambiguity
							const int dd = onebase((uint64_t)(ehi ^ khi));
							if (dd < 0) return;
							if (nh == (uint32_t)HCAP) { sec_ok = false; return; }
							const uint32_t nbb = (ehi >> (2 * dd)) & 3u, base = (khi >> (2 * dd)) & 3u;
							P_hidx[nh][p][wv] = epos;
							hamb |= eamb << nh;
							hu |= ((uint32_t)dd * 3u + nbb - (nbb > base ? 1u : 0u)) << (8 * nh);
							nh++;
						};
						if (use_sec && (bf_from_sec || (fl & 1u)) && b1 > b0) {
							if (b1 - b0 <= (uint32_t)SEC_W) {
								// the usual bucket: its (at most SEC_W) 12-byte records in one go, no search.  With a verified bit vector the
								// records also answer "is the bit of this LO32 set" (qv.cc:955): it is iff one of them carries it.
								uint3 rec[SEC_W];
								VG_VC(d.sec3 + 3ull * b0, 12u * (b1 - b0));
								#pragma unroll
								for (uint32_t z = 0; z < (uint32_t)SEC_W; z++) { const uint32_t e = b0 + z < b1 ? b0 + z : b1 - 1; rec[z] = gather12(d.sec3 + 3ull * e); }
								#pragma unroll
								for (uint32_t z = 0; z < (uint32_t)SEC_W; z++) if (b0 + z < b1 && ((rec[z].z ^ klo) & 0x7FFFFFFFu) == 0u) { fl |= 1u; sec_entry(rec[z].x, rec[z].y, rec[z].z >> 31); }
							} else if (b1 - b0 <= SEC_LONG) {
								// A longer bucket -- a popular first half: the diverged copies of a repeat family -- is dealt to the lanes of stage B1
								// RECORD BY RECORD (r04): adjacent lanes read adjacent 12-byte records, one wait, no search, and a record that is a
								// neighbour carries its position.  r03 turned such a chunk into 48 dictionary queries of three dependent gathers
								// each: 7 % of the gate-open chunks of the repeat-rich genome made half of its stage-B items that way.
								longsec = true;
								sec_b0 = b0;
							} else {
								// beyond that the 48 queries stay (walking the run here -- bisection + dependent loads -- made one lane hold up its
								// wave: 4 % of the kernel at hg38 scale, r02)
								sec_ok = false;
								if (bf_from_sec) { if ((gather_bf<uint64_t>(d.ref_bf + (rp >> 6)) >> (rp & 63)) & 1u) fl |= 1u; }   // (and its bit is read after all)
							}
						}
						// high-half SNP hits from the SNP dictionary's LO32-ordered view (r06): the live SNP slots [s_lo, s_hi) -- 36 of them when the SNP
						// bit vector's probe is positive, 12 more under a large block -- used to be one stage-B1 item each, a jump-table gather and a
						// ~5-deep bisection of a HI24 bucket apiece, nearly all of them misses (the probe is a one-hash filter over the wrong k-mers:
						// SURVEY.md B2).  Every SNP k-mer with this chunk's first 16 bases lies in one short bucket of the view: a record whose last
						// 16 bases differ from the chunk's in exactly one base, at a live slot, whose SNP does not sit on the mutated base
						// (qv.cc:1308-1352; ambiguous entries check that per column of their row, later), IS the hit -- it joins the pair's hit
						// list (the reference hits first), and the pair has no SNP slots left to query.  Rare (1.5 % of the gate-open chunks):
						// its two dependent waits are only paid by waves that hold such a pair.
						bool ssec_ok = false;
						uint32_t hsnp = 0;                                         // bit z: hit z comes from the SNP dictionary
						if (use_ssec && s_hi > s_lo && !longsec) {
							uint32_t c0, c1;
							jg_pair(d.ssec_jg, klo >> (32 - d.ssec_bits), c0, c1);
							if (c1 - c0 <= (uint32_t)SEC_W) {
								ssec_ok = true;
								const uint32_t nh_ref = nh, hu_ref = hu, hamb_ref = hamb;
								if (c1 > c0) {
									uint3 rec[SEC_W];
									VG_VC(d.ssec3 + 3ull * c0, 12u * (c1 - c0));
									#pragma unroll
									for (uint32_t z = 0; z < (uint32_t)SEC_W; z++) { const uint32_t e = c0 + z < c1 ? c0 + z : c1 - 1; rec[z] = gather12(d.ssec3 + 3ull * e); }
									#pragma unroll
									for (uint32_t z = 0; z < (uint32_t)SEC_W; z++) if (c0 + z < c1 && ((rec[z].z ^ klo) & 0x03FFFFFFu) == 0u && ssec_ok) {
										const int dd = onebase((uint64_t)(rec[z].x ^ khi));
										if (dd < 0) continue;
										const uint32_t nbb = (rec[z].x >> (2 * dd)) & 3u, base = (khi >> (2 * dd)) & 3u;
										const uint32_t u = (uint32_t)dd * 3u + nbb - (nbb > base ? 1u : 0u), amb = rec[z].z >> 31;
										if (u < s_lo || u >= s_hi) continue;                   // not a slot the reference queries for this chunk
										if (!amb && ((rec[z].z >> 26) & 31u) == 16u + (uint32_t)dd) continue;      // the mutated base is the SNP base itself
										if (nh == (uint32_t)HCAP) { ssec_ok = false; continue; }
										P_hidx[nh][p][wv] = rec[z].y;
										hamb |= amb << nh; hsnp |= 1u << nh;
										hu |= u << (8 * nh);
										nh++;
									}
								}
								if (!ssec_ok) { nh = nh_ref; hu = hu_ref; hamb = hamb_ref; hsnp = 0; }       // more hits than the list holds: the slots are queried after all
							}
						}
						const bool snp_settled = s_hi == s_lo || ssec_ok;
						uint32_t mode = 0, u_lo = 0, nhigh = 0;                  // mode 0: slots [u_lo, u_lo + nhigh); mode 1: the nh hits
						if (!longsec) {
							// the (at most four) hits in SLOT order, each byte = slot << 2 | hit number (0xFF: none): sorted here, once per pair, so that
							// the items of stage B1 -- one round of 64 of them executes every path any of its lanes takes -- only pick a byte
							uint32_t b0_ = nh > 0u ? ((hu & 0xFFu) << 2) | 0u : 0xFFu, b1_ = nh > 1u ? (((hu >> 8) & 0xFFu) << 2) | 1u : 0xFFu;
							uint32_t b2_ = nh > 2u ? (((hu >> 16) & 0xFFu) << 2) | 2u : 0xFFu, b3_ = nh > 3u ? (((hu >> 24) & 0xFFu) << 2) | 3u : 0xFFu;
							auto cx = [](uint32_t &x, uint32_t &y) { const uint32_t lo_ = x < y ? x : y, hi_ = x < y ? y : x; x = lo_; y = hi_; };
							cx(b0_, b1_); cx(b2_, b3_); cx(b0_, b2_); cx(b1_, b3_); cx(b1_, b2_);
							hu = b0_ | (b1_ << 8) | (b2_ << 16) | (b3_ << 24);
						}
						if (longsec) { u_lo = s_lo; nhigh = (((b1 - b0) + LW - 1u) >> LW_LOG) + (s_hi - s_lo); hu = b1 - b0; }      // the bucket's records (LW per item), then the live SNP slots
						else if (snp_settled && (!(fl & 1u) || sec_ok)) { mode = 1; nhigh = nh; }      // both dictionaries' high-half neighbours are in the hit list
						else if (!(fl & 1u)) { u_lo = s_lo; nhigh = s_hi - s_lo; }
						else { u_lo = 0; nhigh = 48; }
						// (mode 0 never sees an SNP hit in the list: it is only consulted under sec_ok, and sec_ok without mode 1 means the SNP view did not settle its side)
						// items of the strided scans: one per reference-bucket entry; SNP-bucket entries eight per item (their signatures lie
						// side by side in the signature view): the SNP scan is most of stage B's items at hg38 scale, and every round of 64
						// items pays the full chain of dependent waits of the few items in it that do have something to look up
						const uint32_t Lsn = shi - slo;
						const uint32_t L = large ? 48u : (hi - lo) + ((Lsn + sw_m1) >> sw_log);
						P_klo[p][wv] = klo; P_khi[p][wv] = khi; P_lo[p][wv] = lo; P_hi[p][wv] = hi; P_slo[p][wv] = slo; P_shi[p][wv] = shi;
						P_meta[p][wv] = own | (c << 6) | (fl << 11) | ((large ? 1u : 0u) << 13) | (mode << 14) | ((sec_ok ? 1u : 0u) << 15) | (nh << 16) | (u_lo << 19) | ((nhigh < 63u ? nhigh : 63u) << 25) | ((longsec ? 1u : 0u) << 31);
						P_cnt[p][wv] = L + nhigh;
						if (longsec) P_hidx[0][p][wv] = sec_b0;
						P_hu[p][wv] = hu;
						P_hamb[p][wv] = (uint8_t)(hamb | (hsnp << 4));          // bits 0-3: hit z is ambiguous, bits 4-7: hit z is an SNP-dictionary hit
					}
				}
				if constexpr (STATS) {
					VG_WAVE_SYNC();
					uint32_t q = my_base, bits = pend;                       // the owner books the events of its own pairs
					while (bits) {
						bits &= bits - 1;
						if (q >= w0 && q < w0 + PCAP) {
							const uint32_t meta = P_meta[q - w0][wv];
							cur.add(S_GATE_OPEN, 1);
							cur.add(S_REFBF_POS, (meta >> 11) & 1u);
							cur.add(S_SNPBF_POS, (meta >> 12) & 1u);
							cur.add(S_LARGE_BLOCK, (meta >> 13) & 1u);
						}
						q++;
					}
				}
				VG_WAVE_SYNC();
				VG_CLK(2);
				const uint32_t np = P - w0 < (uint32_t)PCAP ? P - w0 : (uint32_t)PCAP;
				uint32_t T;
				{
					const uint32_t cntp = lane < np ? P_cnt[lane < (uint32_t)PCAP ? lane : 0][wv] : 0u;
					uint32_t ci = cntp;
					for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(ci, o); if ((int)lane >= o) ci += y; }
					if (lane < np) P_off[lane][wv] = ci - cntp;
					T = (uint32_t)__builtin_amdgcn_readlane((int)ci, 63);
				}
				VG_WAVE_SYNC();
				// ---- B1: rounds of 64 items.  (r02-r04 fetched an item's strided-scan probe one round ahead; with 32 signatures or four
				// records per item the loads are issued in the round itself, all of an item's together: a round is bound by the instructions
				// it issues, and a row search per round instead of two is worth more than the hidden wait.)
#ifdef VG_STAGE_CLOCKS
				dbg_pairs += np; dbg_items += T;
				{ uint32_t lg = 0, sb = 0; if (lane < np) { const uint32_t mt = P_meta[lane][wv]; lg = (mt >> 13) & 1u; sb = ((mt >> 11) & 1u) && !((mt >> 15) & 1u) ? 1u : 0u; } dbg_large += wave_sum<false>(lg); dbg_secbad += wave_sum<false>(sb); }
#endif
				for (uint32_t t0 = 0; t0 < T; t0 += 64) {
					const uint32_t g = t0 + lane;
					const bool valid = g < T;
#ifdef VG_STAGE_CLOCKS
					dbg_rounds++;
#endif
					// An item of LO32-bucket records can hold several neighbours (the diverged copies of a repeat family are one another's
					// neighbours): the first goes through the round's acceptance and compaction with everybody else's items, the others wait
					// in `hitmask` and get passes of their own (`extra`), in which only their lanes have anything -- rare, so the round's code
					// is not duplicated for them, it is run again.  (Sending such a read to the next tier instead -- the first build of the
					// four-records items did -- doubled the deep tier's work on the repeat-rich genome and cost its main tier 30 %.)
					uint32_t hitmask = 0, xrec = 0, xNI = 0;
					for (uint32_t extra = 0u;; extra = 1u) {
					uint32_t own = 64, c = 0, mod = 0, nbase = 0, o_ecnt = 0;
					uint32_t ri = NOHIT, si = NOHIT;                          // entry indices (dictionaries hold < 2^32 - 1 entries) or NOHIT
					uint32_t rdirect = 0;                                     // bit 0: ri holds the entry's POSITION field instead (bit 1: its ambig_flag)
					uint32_t sdirect = 0;                                     // the same for si (a hit of the SNP dictionary's LO32-ordered view)
					LaneStats<STATS> hs;
					hs.clear();
					if (extra) {
						// (every lane keeps its item's owner: the compaction below takes runs of lanes with one owner for one segment)
						uint32_t p = 0, meta = 64u;
						if (valid) {
							for (uint32_t step = PCAP / 2; step > 0; step >>= 1) if (p + step < np && P_off[p + step][wv] <= g) p += step;
							meta = P_meta[p][wv];
							own = meta & 63u; c = (meta >> 6) & 31u; o_ecnt = P_ecnt[p][wv];
						}
						if (hitmask) {
							const uint32_t khi = P_khi[p][wv];
							const uint32_t j = (uint32_t)__ffs((int)hitmask) - 1u;
							hitmask &= hitmask - 1u;
							VG_VC(d.sec3 + 3ull * ((uint64_t)xrec + (uint64_t)j * xNI), 12);
							const uint3 rec = gather12(d.sec3 + 3ull * ((uint64_t)xrec + (uint64_t)j * xNI));
							const int dd = onebase((uint64_t)(rec.x ^ khi));      // (a neighbour: the first pass has seen it)
							ri = rec.y; rdirect = 1u | ((rec.z >> 31) << 1); mod = 16u + (uint32_t)dd; nbase = (rec.x >> (2 * dd)) & 3u;
						}
					} else if (valid) {
						uint32_t p = 0;                                                  // last row with P_off <= g
						for (uint32_t step = PCAP / 2; step > 0; step >>= 1) if (p + step < np && P_off[p + step][wv] <= g) p += step;
						const uint32_t t = g - P_off[p][wv];
						const uint32_t meta = P_meta[p][wv];
						const uint32_t klo = P_klo[p][wv], khi = P_khi[p][wv], lo = P_lo[p][wv], hi = P_hi[p][wv], slo = P_slo[p][wv], shi = P_shi[p][wv];
						const uint64_t k = ((uint64_t)khi << 32) | klo;
						own = meta & 63u; c = (meta >> 6) & 31u;
						const uint32_t fl = (meta >> 11) & 3u, mode = (meta >> 14) & 1u, nh = (meta >> 16) & 7u, u_lo = (meta >> 19) & 63u;
						const bool large = (meta >> 13) & 1u, sec_ok = (meta >> 15) & 1u;
						const uint32_t Lsn = shi - slo;
						const uint32_t Lr = large ? 48u : hi - lo, L = large ? 48u : (hi - lo) + ((Lsn + sw_m1) >> sw_log);
						const uint32_t rsb = (fl & 1u) ? 64u : 32u, ssb = (fl & 2u) ? 64u : 40u;
						o_ecnt = P_ecnt[p][wv];
						bool q_r = false, q_s = false;                          // dictionary queries of the neighbour k-mer qk, issued together below
						uint64_t qk = 0;
						if (t < L) {
							if (large) {                                         // qv.cc:962-1109
								const uint32_t pair = t / 3, sel = t % 3, base = (uint32_t)(k >> (2 * pair)) & 3u;
								nbase = sel + (sel >= base ? 1u : 0u);
								mod = pair;
								qk = (k & ~(3ull << (2 * pair))) | ((uint64_t)nbase << (2 * pair));
								q_r = q_s = true;
							} else {
								// iterate_ref_dict, qv.cc:316-376 / iterate_snp_dict, qv.cc:413-464   (B1): entry lo + 9u (slo + 11u) is
								// tested, entry lo + u (slo + u) recorded.  Both dictionaries hold 16-byte entries: one gather site.
								const bool isr = t < Lr;
								if (!isr && use_sig) {
									// signatures of up to 32 consecutive entries of the SNP bucket (main tier; the deep tier: one): which of them can the
									// scan keep at all?  Per 32-bit word two signatures; y = one bit per base in which a signature differs from the k-mer's.
									const uint32_t u0 = (t - Lr) << sw_log, live = Lsn - u0, ks = sig16(k & LO40_MASK);      // live: entries of the bucket from u0 on (>= 1)
									uint32_t cand = 0;
									if constexpr (WPB > 1) {
										const uint32_t kk2 = ks | (ks << 16);
										uint4 sv[SIG_W];
										#pragma unroll
										for (uint32_t q = 0; q < SIG_W; q++) { sv[q] = make_uint4(0u, 0u, 0u, 0u); if (8u * q < live) sv[q] = gather<uint4, 1>(d.snp_sig + ((uint64_t)slo + u0 + 8u * q)); }
										auto two = [&](uint32_t w) -> uint32_t {
											const uint32_t x = w ^ kk2, y = (x | (x >> 1)) & 0x55555555u;
											return (__popc(y & 0xFFFFu) == 1 ? 1u : 0u) | (__popc(y >> 16) == 1 ? 2u : 0u);
										};
										#pragma unroll
										for (uint32_t q = 0; q < SIG_W; q++) cand |= (two(sv[q].x) | (two(sv[q].y) << 2) | (two(sv[q].z) << 4) | (two(sv[q].w) << 6)) << (8u * q);
										cand &= live >= 8u * SIG_W ? (uint32_t)((1ull << (8u * SIG_W)) - 1ull) : ((1u << live) - 1u);
									} else {
										const uint32_t x = (uint32_t)d.snp_sig[(uint64_t)slo + u0] ^ ks, y = (x | (x >> 1)) & 0x5555u;
										cand = __popc(y) == 1 ? 1u : 0u;
									}
									if (cand & (cand - 1u)) N_ovf[col0 + own] = 1;        // two candidates in one item: the read goes to the next tier
									else if (cand) {
										// the candidate's probed value itself: entry slo + 11 u of the dictionary (zero beyond its end)
										const uint32_t u = u0 + (uint32_t)__ffs((int)cand) - 1u;
										const uint64_t tt = (uint64_t)slo + (uint64_t)u * SNP_STRIDE;
										const uint64_t full = tt < d.n_snp ? gather<uint64_t>(&d.snp[tt].key) & LO40_MASK : 0ull;
										const int dd = onebase((k & LO40_MASK) ^ full);
										if (dd >= 0) { si = slo + u; mod = (uint32_t)dd; nbase = (uint32_t)(full >> (2 * dd)) & 3u; }
									}
								} else {
								const uint32_t u = isr ? t : t - Lr;
								const uint64_t tt = isr ? (uint64_t)lo + (uint64_t)u * REF_STRIDE : (uint64_t)slo + (uint64_t)u * SNP_STRIDE;
								const bool inr = tt < (isr ? d.n_ref : d.n_snp);
								uint4 v = make_uint4(0u, 0u, 0u, 0u);                      // zeros when the probe falls off the array
								if (!isr && use_probe) { const uint2 q = gather<uint2, 8>(d.snp_probe + ((uint64_t)slo + u)); v.x = q.x; v.y = q.y; }
								else if (inr) v = gather<uint4>(isr ? (const void *)(d.ref + tt) : (const void *)(d.snp + tt));
								hs.add(S_SCAN_REF, isr ? 1u : 0u);
								hs.add(S_SCAN_SNP, isr ? 0u : 1u);
								hs.add(S_SCAN_OOB, inr ? 0u : 1u);
								const uint64_t tlo = isr ? (uint64_t)v.x : ((((uint64_t)v.y << 32) | v.x) & LO40_MASK);
								const int dd = onebase(isr ? (uint64_t)(klo ^ v.x) : ((k & LO40_MASK) ^ tlo));
								if (dd >= 0) {
									if (isr) ri = lo + u; else si = slo + u;
									mod = (uint32_t)dd; nbase = (uint32_t)(tlo >> (2 * dd)) & 3u;
								}
								}
							}
						} else {                                                 // qv.cc:1213-1365
							uint32_t h = t - L;
							uint32_t u;
							bool have_ri = false, slot_item = true;
							const bool longsec = (meta >> 31) != 0u;
							if (longsec) {
								const uint32_t S = P_hu[p][wv], NI = (S + LW - 1u) >> LW_LOG;
								if (h < NI) {
									// records h, h + NI, h + 2 NI, h + 3 NI of the chunk's LO32 bucket (adjacent lanes read adjacent records, four gathers
									// in one wait): a dictionary k-mer with the chunk's first half whose last half differs in exactly one base is a
									// high-half neighbour (qv.cc:1213-1296), and the record carries its position.
									slot_item = false;
									const uint64_t rb = P_hidx[0][p][wv];
									uint3 rec[LW];
									#pragma unroll
									for (uint32_t j = 0; j < LW; j++) { const uint32_t r_ = h + j * NI; rec[j] = make_uint3(0u, 0u, ~klo); if (r_ < S) { VG_VC(d.sec3 + 3ull * (rb + r_), 12); rec[j] = gather12(d.sec3 + 3ull * (rb + r_)); } }
									#pragma unroll
									for (uint32_t j = LW; j-- > 0u;) if (((rec[j].z ^ klo) & 0x7FFFFFFFu) == 0u && (bf_from_sec || (fl & 1u))) {      // (downwards: the lowest hit is the one kept for this pass)
										const int dd = onebase((uint64_t)(rec[j].x ^ khi));
										if (dd >= 0) { hitmask |= 1u << j; ri = rec[j].y; rdirect = 1u | ((rec[j].z >> 31) << 1); mod = 16u + (uint32_t)dd; nbase = (rec[j].x >> (2 * dd)) & 3u; }
									}
									hitmask &= hitmask - 1u;                              // the others: passes of their own, below
									xrec = (uint32_t)rb + h; xNI = NI;
								} else h -= NI;
							}
							if (slot_item) {
							if (mode == 1) {                                     // the h-th hit in slot order
								const uint32_t hb = (P_hu[p][wv] >> (8u * (h & 3u))) & 0xFFu, zsel = hb & 3u;      // (sorted by slot in stage B0)
								const uint32_t ha = (uint32_t)P_hamb[p][wv];
								u = hb >> 2;
								// (a hit of either LO32-ordered view comes with its position: no entry to fetch)
								if ((ha >> (4u + zsel)) & 1u) { si = P_hidx[zsel][p][wv]; sdirect = 1u | ((ha >> zsel) & 1u) << 1; }
								else { ri = P_hidx[zsel][p][wv]; rdirect = 1u | ((ha >> zsel) & 1u) << 1; }
								have_ri = true;                                  // (both dictionaries' sides of this pair are settled by the list)
							} else u = u_lo + h;
							const uint32_t pair = 16u + u / 3, sel = u % 3, base = (uint32_t)(k >> (2 * pair)) & 3u;
							nbase = sel + (sel >= base ? 1u : 0u);
							mod = pair;
							qk = (k & ~(3ull << (2 * pair))) | ((uint64_t)nbase << (2 * pair));
							if (!have_ri && !longsec && 2 * pair < rsb) {
								if (sec_ok) { const uint32_t hu = P_hu[p][wv]; for (uint32_t z = 0; z < nh; z++) { const uint32_t hb = (hu >> (8 * z)) & 0xFFu, zz = hb & 3u; if ((hb >> 2) == u) { ri = P_hidx[zz][p][wv]; rdirect = 1u | (((uint32_t)P_hamb[p][wv] >> zz) & 1u) << 1; } } }
								else q_r = true;
							}
							q_s = mode == 0u && (large || 2 * pair >= 40u) && 2 * pair < ssb;
							}
						}
#ifdef VG_STAGE_CLOCKS
						dbg_dq += (q_r || q_s) ? 1u : 0u;
#endif
						if (q_r || q_s) dual_query(d, hs, qk, q_r, q_s, ri, si);
					}
					// is `position` one of the owner's vote keys (the implied read position of one of its exact hits)?
					auto in_keys = [&](uint32_t position) -> bool {                 // four keys per LDS round trip
						bool f = false;
						#pragma nounroll
						for (uint32_t e0 = 0; e0 < o_ecnt; e0 += 4) {
							uint32_t kv[4];
							#pragma unroll
							for (uint32_t t = 0; t < 4; t++) kv[t] = K_idx[e0 + t < (uint32_t)W_ECAP ? e0 + t : (uint32_t)W_ECAP - 1u][col0 + own];
							#pragma unroll
							for (uint32_t t = 0; t < 4; t++) f |= e0 + t < o_ecnt && kv[t] == position;
						}
						return f;
					};
					// acceptance (site / SNP-base tests) + key filter -> bit j: ref candidate j kept, bit 10+j: snp candidate j kept.
					// Loads are grouped so the wave waits once per group, not once per load: both entries; then the site byte or
					// four row columns at a time with their four site bytes.
					uint32_t keepm = 0, rpos = 0, spos = 0;                   // non-ambiguous kept positions travel to the write phase
					bool r_aux = false, s_aux = false;
					{
						RefEnt re; SnpEnt se;
						re.pos = POS_AMBIGUOUS; re.amb = 0; se.pos = POS_AMBIGUOUS; se.key = 0;
						if (rdirect) { re.pos = ri; re.amb = rdirect >> 1; }         // ri IS the position (or the row index) here
						else if (ri != NOHIT) re = gather<RefEnt>(d.ref + ri);
						if (sdirect) { se.pos = si; se.key = (uint64_t)(sdirect >> 1) << 48; }   // likewise (stage B0 has compared SNP_INFO_POS with the mutated base already: 0 here, never a high-half base)
						else if (si != NOHIT) se = gather<SnpEnt>(d.snp + si);
						const bool r_ok = re.pos != POS_AMBIGUOUS, s_ok = se.pos != POS_AMBIGUOUS;
						r_aux = r_ok && re.amb != 0; s_aux = s_ok && ((se.key >> 48) & 0xFFu) != 0;
						rpos = re.pos; spos = se.pos;
						if (r_ok && !r_aux) {
							// (the timed build asks the key table first -- LDS -- and reads the site byte only of a position that is a key)
							if constexpr (!STATS) { if (in_keys(rpos - 32u * c) && !site_loose(d, hs, rpos + mod)) keepm |= 1u; }
							else if (!site_loose(d, hs, rpos + mod)) { hs.add(S_CTX, 1); if (in_keys(rpos - 32u * c)) keepm |= 1u; }
						}
						if (r_aux) {
							// the whole row in one wait, then the site bytes of all its positions in a second one (r03 went through the row four
							// columns at a time: up to six dependent waits for a neighbour k-mer with nine copies)
							hs.add(S_AUX_REF, 1);
							uint32_t v[AUX_COLS], sb[AUX_COLS];
							load_row10(d.ref_aux + (uint64_t)rpos * AUX_COLS, v);
							if constexpr (!STATS) {
								// The timed build asks the key table first (every column against every key, four keys per LDS round trip) and
								// reads the site byte only of the columns that are keys: ten one-byte gathers, each a line of its own, were most
								// of what a neighbour k-mer inside a repeat family cost -- for columns the key filter then dropped.
								uint32_t lv = 0;
								#pragma unroll
								for (int j = 0; j < AUX_COLS; j++) if (lv == (uint32_t)j && v[j] != 0) lv = (uint32_t)j + 1u;
								uint32_t kmm = 0;
								#pragma nounroll
								for (uint32_t e0 = 0; e0 < o_ecnt; e0 += 4) {
									uint32_t kv[4];
									#pragma unroll
									for (uint32_t t = 0; t < 4; t++) kv[t] = K_idx[e0 + t < (uint32_t)W_ECAP ? e0 + t : (uint32_t)W_ECAP - 1u][col0 + own];
									#pragma unroll
									for (uint32_t t = 0; t < 4; t++) if (e0 + t < o_ecnt) {
										#pragma unroll
										for (int j = 0; j < AUX_COLS; j++) kmm |= (v[j] - 32u * c == kv[t] ? 1u : 0u) << j;
									}
								}
								kmm &= (1u << lv) - 1u;
								#pragma unroll
								for (int j = 0; j < AUX_COLS; j++) { const uint64_t a = (uint64_t)v[j] + mod; sb[j] = 0; if (((kmm >> j) & 1u) && a < d.pile_len) { VG_VC(d.pile + a, 1); sb[j] = d.pile[a]; } }
								#pragma unroll
								for (int j = 0; j < AUX_COLS; j++) if (((kmm >> j) & 1u) && !(sb[j] & 15u)) keepm |= 1u << j;
							} else {
							#pragma unroll
							for (int j = 0; j < AUX_COLS; j++) { const uint64_t a = (uint64_t)v[j] + mod; const bool in = v[j] && a < d.pile_len; sb[j] = d.pile[in ? a : 0]; if (!in) sb[j] = 0; }
							bool live = true;
							uint32_t cand = 0;
							#pragma unroll
							for (int j = 0; j < AUX_COLS; j++) {
								live = live && v[j] != 0;
								if (!live) continue;
								hs.add(S_SITE_TEST, 1);
								if (sb[j] & 15u) continue;
								hs.add(S_CTX, 1);
								cand |= 1u << j;
							}
							while (cand) {                                           // accepted columns are few: one key-filter loop
								const uint32_t j = (uint32_t)__ffs((int)cand) - 1;
								cand &= cand - 1;
								const uint32_t pp = j == 0 ? v[0] : j == 1 ? v[1] : j == 2 ? v[2] : j == 3 ? v[3] : j == 4 ? v[4] : j == 5 ? v[5] : j == 6 ? v[6] : j == 7 ? v[7] : j == 8 ? v[8] : v[9];
								if (in_keys(pp - 32u * c)) keepm |= 1u << j;
							}
							}
						}
						if (s_ok && !s_aux) {
							if ((uint32_t)((se.key >> 43) & 0x1Fu) != mod) { hs.add(S_CTX, 1); if (in_keys(spos - 32u * c)) keepm |= 1u << 10; }
						}
						if (s_aux) {
							const uint32_t *prow = d.snp_aux_pos + (uint64_t)spos * AUX_COLS;
							const uint8_t *irow = d.snp_aux_info + (uint64_t)spos * AUX_COLS;
							hs.add(S_AUX_SNP, 1);
							for (int j0 = 0; j0 < AUX_COLS; j0 += 4) {
								uint32_t v[4], inf[4];
								load_row4(prow, j0, v);
								VG_VC(irow + j0, 4);
								#pragma unroll
								for (int j = 0; j < 4; j++) inf[j] = irow[j0 + j < AUX_COLS ? j0 + j : 0];
								bool live = true;
								uint32_t cand = 0;
								#pragma unroll
								for (int j = 0; j < 4; j++) {
									live = live && v[j] != 0;
									if (!live) continue;
									if ((inf[j] >> 3) == mod) continue;
									hs.add(S_CTX, 1);
									cand |= 1u << j;
								}
								while (cand) {
									const uint32_t j = (uint32_t)__ffs((int)cand) - 1;
									cand &= cand - 1;
									const uint32_t pp = j == 0 ? v[0] : j == 1 ? v[1] : j == 2 ? v[2] : v[3];
									if (in_keys(pp - 32u * c)) keepm |= 1u << (10 + j0 + j);
								}
								if (!live) break;
							}
						}
					}
					if constexpr (STATS) {
						if (valid && !extra) for (int i = 0; i < NSH; i++) { const uint32_t v = hs.v[SH_IDS[i]]; if (v) atomicAdd(&S_own[i][col0 + own], v); }
					}
					const bool item = valid;
					if (__any(keepm != 0)) {
						// compaction per owner (a segment of consecutive lanes), canonical order = lane order; ref contexts of an
						// item before its SNP contexts
						const uint32_t keep = (uint32_t)__popc(keepm);
						uint32_t incl = keep;
						for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o); if ((int)lane >= o) incl += y; }
						const uint32_t prev = __shfl_up(own, 1);
						const uint64_t smask = __ballot(lane == 0 || own != prev);
						const uint64_t upto = ~0ull >> (63u - lane);              // lanes 0 .. this one
						const int ss = 63 - __clzll((long long)(smask & upto));
						const uint64_t above = smask & ~upto;
						const int se_l = above ? (__ffsll((long long)above) - 2) : 63;
						const uint32_t excl_ss = __shfl(incl - keep, ss);
						const uint32_t seg_total = __shfl(incl, se_l) - excl_ss;
						const uint32_t curc = item ? (uint32_t)N_cnt[col0 + own] : 0u;
						const bool fits = curc + seg_total <= (uint32_t)W_NCAP;
						if (item && keep) {
							if (!fits) N_ovf[col0 + own] = 1;
							else {
								uint32_t at = curc + (incl - keep) - excl_ss;
								const uint16_t mt = (uint16_t)mk_meta(c, mod, true, nbase);
								if (keepm & 0x3FFu) {
									if (!r_aux) { N_kpos[at][col0 + own] = rpos; N_meta[at][col0 + own] = mt; at++; }
									else {                                       // kept contexts out of a row are rare: re-read those columns
										const uint32_t *row = d.ref_aux + (uint64_t)rpos * AUX_COLS;
										VG_VC(row, 40);
										for (int j = 0; j < AUX_COLS; j++) if (keepm & (1u << j)) { N_kpos[at][col0 + own] = row[j]; N_meta[at][col0 + own] = mt; at++; }
									}
								}
								if (keepm >> 10) {
									if (!s_aux) { N_kpos[at][col0 + own] = spos; N_meta[at][col0 + own] = mt; at++; }
									else {
										const uint32_t *prow = d.snp_aux_pos + (uint64_t)spos * AUX_COLS;
										VG_VC(prow, 40);
										for (int j = 0; j < AUX_COLS; j++) if (keepm & (1u << (10 + j))) { N_kpos[at][col0 + own] = prow[j]; N_meta[at][col0 + own] = mt; at++; }
									}
								}
							}
						}
						if (item && fits && seg_total && (int)lane == se_l) N_cnt[col0 + own] = (uint16_t)(curc + seg_total);
					}
					VG_WAVE_SYNC();
					if (!__any(hitmask != 0u)) break;
					}
				}
				VG_CLK(3);
			}
			VG_WAVE_SYNC();
			ncnt = N_cnt[col];
			if (N_ovf[col]) { if (!ovf) VG_OVF(1); ovf = true; }
			if constexpr (STATS) for (int i = 0; i < NSH; i++) cur.v[SH_IDS[i]] += S_own[i][col];
		}
		VG_WAVE_SYNC();

		// ------------------------------------------------------------------ stage C: replay the vote, walk the pile-up
		if (active) {
			bool processed = false;
			if (!ovf) {
				// improved_index_table_add, qv.cc:132-178, from per-key totals (see the key table's description): a key's frequency is the
				// number of its exact contexts (one per chunk of its mask) plus its neighbour contexts from chunks not before its first
				// exact chunk (an earlier one found no key to vote for, :134-139); it takes part once two different chunks have voted
				// for it (:163-165); the pass is processed iff ONE key has the highest frequency (:1375).  Frequencies stay far below
				// the reference's uint8_t wrap (at most one exact context per chunk and W_NCAP neighbour contexts).
				int best = -1; bool amb = false;
				uint32_t bestf = 0, target = 0, bmask = 0;
				for (uint32_t e = 0; e < kcnt; e++) {
					const uint32_t idx = K_idx[e][col], m = K_mask[e][col];
					uint32_t f = (uint32_t)__popc(m), chunks = m;
					if (ncnt) {
						const uint32_t first = (uint32_t)__ffs((int)m) - 1u;
						for (uint32_t i = 0; i < ncnt; i++) {
							const uint32_t c = N_meta[i][col] & 31u;
							if (c >= first && N_kpos[i][col] - 32u * c == idx) { f++; chunks |= 1u << c; }
						}
					}
					if (!(chunks & (chunks - 1u))) continue;
					if (f > bestf) { best = (int)e; bestf = f; amb = false; target = idx; bmask = m; }
					else if (f == bestf) amb = true;
				}
				VG_CLK(4);
				if (!ovf) {
					cur.add(S_PASSES, 1);
					processed = best >= 0 && !amb;                               // qv.cc:1375 (a key that takes part has at least two votes)
					if (processed) {
						cur.add(S_PASSES_OK, 1);
						// Timed build, reads of up to four chunks (150 bp): all supporting contexts lie in [target, target + 32 n), which
						// two or three rank blocks cover.  Those blocks (site bits + rank) and the read's k-mers are all the walk needs
						// when the counters are indexed by the base the read shows (DevIndex::cnt4): ~2.5 block gathers + one atomic per
						// site per context, where the byte-per-position walk costs two window gathers per context + a rank gather + an
						// atomic per counted base.  Same sums: a site's ref and alt differ, and fetch reads cnt4[ref] and cnt4[alt].
						bool fast = false;
						if constexpr (!STATS) fast = n <= 4u && (uint64_t)target + 32u * n <= d.pile_len;
						if (fast) {
							const uint32_t blk0 = target >> 6, blk_last = (target + 32u * n - 1u) >> 6;
							// named scalars, not arrays: an array the compiler cannot keep in registers ends up in scratch memory
							const ulonglong2 zz = make_ulonglong2(0ull, 0ull);
							VG_VC_AS(VC_READS, pk_kmer + (uint64_t)slot0, 8u * n);
							const ulonglong2 r0 = gather_walk<ulonglong2>(d.srank + blk0);
							const ulonglong2 r1 = blk0 + 1u <= blk_last ? gather_walk<ulonglong2>(d.srank + (blk0 + 1u)) : zz;
							const ulonglong2 r2 = blk0 + 2u <= blk_last ? gather_walk<ulonglong2>(d.srank + (blk0 + 2u)) : zz;
							uint64_t f0, f1, f2 = 0, f3 = 0;                     // the read's k-mers in file order
							{
								ulonglong2 v;
								__builtin_memcpy(&v, pk_kmer + (uint64_t)slot0, 16); f0 = v.x; f1 = v.y;   // n >= 2: one chunk cannot win a vote
								if (n == 4u) { __builtin_memcpy(&v, pk_kmer + ((uint64_t)slot0 + 2), 16); f2 = v.x; f3 = v.y; }
								else if (n == 3u) f2 = pk_kmer[(uint64_t)slot0 + 2];
							}
							VG_CLKW(6);
							// one loop over both lists and no lambda: a select over variables captured by reference becomes a select of
							// addresses, which parks the whole closure in scratch memory
							// the supporting contexts: the winning key's exact contexts (one per chunk of its mask, at target + 32 chunk), then
							// the neighbour contexts with its implied position (whether or not they voted, qv.cc:1386-1502)
							uint32_t em = bmask, ni = 0;
							for (;;) {
								uint32_t c, mod = NOMOD;
								if (em) { c = (uint32_t)__ffs((int)em) - 1u; em &= em - 1u; }
								else {
									if (ni >= ncnt) break;
									const uint32_t mt = N_meta[ni][col], p = N_kpos[ni][col];
									ni++;
									c = mt & 31u; mod = (mt >> 5) & 31u;
									if (p - 32u * c != target) continue;
								}
								const uint32_t fi = pass ? n - 1u - c : c;
								uint64_t kk = fi == 0 ? f0 : fi == 1 ? f1 : fi == 2 ? f2 : f3;
								if (pass) kk = revcomp64(kk);
								const uint32_t o = (target & 63u) + 32u * c;         // bit offset of the window in the blocks
								const uint32_t z = o >> 6, sh = o & 63u;
								const uint64_t w0 = z == 0 ? r0.x : z == 1 ? r1.x : r2.x;
								const uint64_t w1 = z == 0 ? r1.x : r2.x;           // only read when the window crosses into it (then z <= 1)
								uint32_t sites = (uint32_t)(w0 >> sh);
								if (sh > 32u) sites |= (uint32_t)(w1 << (64u - sh));
								if (mod < 32u) sites &= ~(1u << mod);
								while (sites) {
									const uint32_t b = (uint32_t)__ffs((int)sites) - 1;
									sites &= sites - 1;
									const uint32_t ob = o + b, zb = ob >> 6;
									const uint64_t mx_ = zb == 0 ? r0.x : zb == 1 ? r1.x : r2.x, my_ = zb == 0 ? r0.y : zb == 1 ? r1.y : r2.y;
									const uint32_t sid = (uint32_t)my_ + (uint32_t)__popcll(mx_ & ((1ull << (ob & 63u)) - 1ull));
									VG_VC(&d.cnt4[4ull * sid + ((uint32_t)(kk >> (2 * b)) & 3u)], 4);
									atomicAdd(&d.cnt4[4ull * sid + ((uint32_t)(kk >> (2 * b)) & 3u)], 1u);
								}
							}
							VG_CLK(7);
							VG_CLKW(8);
						} else if (!STATS && n <= 8u && (uint64_t)target + 32u * n <= d.pile_len) {
							// Reads of five to eight chunks (250 bp: seven) the same way (r06): their supporting contexts lie in [target, target + 256),
							// which five rank blocks cover -- one or two lines for the whole read, where the byte-per-position walk below costs a pile
							// line and a rank line PER CONTEXT (5.6 contexts per 250 bp read: 11 of its ~25 lines; profiles/traffic_r05_len250.json had
							// the kernel at 1.38 x its algorithmic bytes for that).  A block of its own, so that the four-chunk walk keeps its registers.
							const uint32_t blk0 = target >> 6, blk_last = (target + 32u * n - 1u) >> 6;
							const ulonglong2 zz = make_ulonglong2(0ull, 0ull);
							VG_VC_AS(VC_READS, pk_kmer + (uint64_t)slot0, 8u * n);
							const ulonglong2 r0 = gather_walk<ulonglong2>(d.srank + blk0);
							const ulonglong2 r1 = blk0 + 1u <= blk_last ? gather_walk<ulonglong2>(d.srank + (blk0 + 1u)) : zz;
							const ulonglong2 r2 = blk0 + 2u <= blk_last ? gather_walk<ulonglong2>(d.srank + (blk0 + 2u)) : zz;
							const ulonglong2 r3 = blk0 + 3u <= blk_last ? gather_walk<ulonglong2>(d.srank + (blk0 + 3u)) : zz;
							const ulonglong2 r4 = blk0 + 4u <= blk_last ? gather_walk<ulonglong2>(d.srank + (blk0 + 4u)) : zz;
							uint64_t f0, f1, f2, f3, f4, f5 = 0, f6 = 0, f7 = 0;   // the read's k-mers in file order (n >= 5)
							{
								ulonglong2 v;
								__builtin_memcpy(&v, pk_kmer + (uint64_t)slot0, 16); f0 = v.x; f1 = v.y;
								__builtin_memcpy(&v, pk_kmer + ((uint64_t)slot0 + 2), 16); f2 = v.x; f3 = v.y;
								if (n >= 6u) { __builtin_memcpy(&v, pk_kmer + ((uint64_t)slot0 + 4), 16); f4 = v.x; f5 = v.y; } else f4 = pk_kmer[(uint64_t)slot0 + 4];
								if (n == 8u) { __builtin_memcpy(&v, pk_kmer + ((uint64_t)slot0 + 6), 16); f6 = v.x; f7 = v.y; } else if (n == 7u) f6 = pk_kmer[(uint64_t)slot0 + 6];
							}
							uint32_t em = bmask, ni = 0;
							for (;;) {
								uint32_t c, mod = NOMOD;
								if (em) { c = (uint32_t)__ffs((int)em) - 1u; em &= em - 1u; }
								else {
									if (ni >= ncnt) break;
									const uint32_t mt = N_meta[ni][col], p = N_kpos[ni][col];
									ni++;
									c = mt & 31u; mod = (mt >> 5) & 31u;
									if (p - 32u * c != target) continue;
								}
								const uint32_t fi = pass ? n - 1u - c : c;
								uint64_t kk = fi == 0 ? f0 : fi == 1 ? f1 : fi == 2 ? f2 : fi == 3 ? f3 : fi == 4 ? f4 : fi == 5 ? f5 : fi == 6 ? f6 : f7;
								if (pass) kk = revcomp64(kk);
								const uint32_t o = (target & 63u) + 32u * c;         // bit offset of the window in the blocks (< 288: blocks 0 .. 4)
								const uint32_t z = o >> 6, sh = o & 63u;
								const uint64_t w0 = z == 0 ? r0.x : z == 1 ? r1.x : z == 2 ? r2.x : z == 3 ? r3.x : r4.x;
								const uint64_t w1 = z == 0 ? r1.x : z == 1 ? r2.x : z == 2 ? r3.x : r4.x;       // only read when the window crosses into it (then z <= 3)
								uint32_t sites = (uint32_t)(w0 >> sh);
								if (sh > 32u) sites |= (uint32_t)(w1 << (64u - sh));
								if (mod < 32u) sites &= ~(1u << mod);
								while (sites) {
									const uint32_t b = (uint32_t)__ffs((int)sites) - 1;
									sites &= sites - 1;
									const uint32_t ob = o + b, zb = ob >> 6;
									const uint64_t mx_ = zb == 0 ? r0.x : zb == 1 ? r1.x : zb == 2 ? r2.x : zb == 3 ? r3.x : r4.x;
									const uint64_t my_ = zb == 0 ? r0.y : zb == 1 ? r1.y : zb == 2 ? r2.y : zb == 3 ? r3.y : r4.y;
									const uint32_t sid = (uint32_t)my_ + (uint32_t)__popcll(mx_ & ((1ull << (ob & 63u)) - 1ull));
									VG_VC(&d.cnt4[4ull * sid + ((uint32_t)(kk >> (2 * b)) & 3u)], 4);
									atomicAdd(&d.cnt4[4ull * sid + ((uint32_t)(kk >> (2 * b)) & 3u)], 1u);
								}
							}
						} else {
						// Reads of more than eight chunks, and the counting build: the byte-per-position walk of qv.cc:1386-1436 -- per
					// supporting context its k-mer and its 32-byte pile window, then one rank-block gather + atomic per counted base
					// (the base a neighbour context was found with lies at the one position the walk leaves out, :1397)
					auto count = [&](uint32_t p, uint32_t which) { cur.add(S_INCR, 1); bump_site(d, p, which); };
					uint32_t em = bmask, ni = 0;
					for (;;) {
						uint32_t c, wp, mod = NOMOD;
						if (em) { c = (uint32_t)__ffs((int)em) - 1u; em &= em - 1u; wp = target + 32u * c; }
						else {
							if (ni >= ncnt) break;
							const uint32_t mt = N_meta[ni][col];
							wp = N_kpos[ni][col];
							ni++;
							c = mt & 31u; mod = (mt >> 5) & 31u;
							if (wp - 32u * c != target) continue;
						}
						cur.add(S_WALKS, 1);
						if ((uint64_t)wp + 32 > d.pile_len) continue;
						const uint64_t kk = chunk_kmer(c);
						uint4 pw[2];
						load_pile_window(d, wp, pw);
						walk_matches(pw, kk, wp, mod, count);
					}
					}
					}
				}
			}
			if (ovf) {
				overflow_list[atomicAdd(overflow_count, 1u)] = rid;              // counters untouched: the generic tier redoes the read
				active = false;
			} else if (processed || pass == 1) {
				if constexpr (STATS) for (int i = 0; i < S_COUNT; i++) tot.v[i] += cur.v[i];
				active = false;
			} else {
				pass = 1;                                                        // reverse-complement retry, qv.cc:1504-1510
			}
		}
		VG_WAVE_SYNC();
		VG_CLK(5);
	}
#ifdef VG_STAGE_CLOCKS
	{ const uint32_t dq = wave_sum<false>(dbg_dq); if (!STATS && lane == 0 && (blockIdx.x % 97u) == 0 && wv == 0) printf("DBG blk %u iters %u pairs %u items %u rounds %u dualq %u large %u secbad %u\n", blockIdx.x, iters, dbg_pairs, dbg_items, dbg_rounds, dq, dbg_large, dbg_secbad); }
	if (!STATS && lane == 0 && (blockIdx.x % 97u) == 0 && wv == 0)
		printf("CLK blk %u iters %u refill %lld A %lld B0 %lld B1 %lld vote %lld walk %lld wload %lld wloop %lld watom %lld Akmer %lld Adx %lld Ascan %lld ecap %d\n", blockIdx.x, iters, clk[0], clk[1], clk[2], clk[3], clk[4], clk[5], clk[6], clk[7], clk[8], clk[9], clk[10], clk[11], W_ECAP);
#endif
	if constexpr (STATS) {
		for (int i = 0; i < S_COUNT; i++) if (tot.v[i]) atomicAdd(&stats[i], (unsigned long long)tot.v[i]);
	}
}

#undef lane
#undef col

template <bool STATS, int W_ECAP, int W_NCAP, int WPB, bool SDX = false>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(WPB > 1 ? VG_WPE : 1))) void vg_wave_kernel(DevIndex d, const uint64_t *__restrict__ pk_kmer, const uint64_t *__restrict__ pk_meta,
                                                     const uint64_t *__restrict__ offsets, uint64_t n_reads_arg,
                                                     const uint32_t *__restrict__ read_ids, const uint32_t *__restrict__ n_ids,
                                                     uint32_t *overflow_list, uint32_t *overflow_count, uint32_t *work_next, const uint32_t WORK_CHUNK_ARG, unsigned long long *stats, const FuseIn fuse)
{
	vg_wave_body<STATS, W_ECAP, W_NCAP, WPB, false, SDX>(d, pk_kmer, pk_meta, offsets, n_reads_arg, read_ids, n_ids, overflow_list, overflow_count, work_next, WORK_CHUNK_ARG, stats, fuse);
}

// the timed build for an index without the merged view (see vg_wave_body)
template <int W_ECAP, int W_NCAP, int WPB>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(WPB > 1 ? VG_WPE : 1))) void vg_wave_kernel_big(DevIndex d, const uint64_t *__restrict__ pk_kmer, const uint64_t *__restrict__ pk_meta,
                                                     const uint64_t *__restrict__ offsets, uint64_t n_reads_arg,
                                                     const uint32_t *__restrict__ read_ids, const uint32_t *__restrict__ n_ids,
                                                     uint32_t *overflow_list, uint32_t *overflow_count, uint32_t *work_next, const uint32_t WORK_CHUNK_ARG, unsigned long long *stats, const FuseIn fuse)
{
	vg_wave_body<false, W_ECAP, W_NCAP, WPB, true>(d, pk_kmer, pk_meta, offsets, n_reads_arg, read_ids, n_ids, overflow_list, overflow_count, work_next, WORK_CHUNK_ARG, stats, fuse);
}

}  // namespace vg
