// vg_device.h -- device-side data layout and the per-read state machine of the `vargeno geno`
// hot path (reference: src/qv.cc:760-1558 of medvedevgroup/vargeno), written for gfx950.
//
// Everything here is integer / bit work bounded by random gathers; there is no MFMA-shaped
// computation anywhere on this path.  Measured on MI355X (tools/gather_probe, tools/line_probe): the chip
// sustains ~49 G random gathers/s from a table far larger than its caches, and every one of them moves a
// whole 128-byte line from HBM (49 G x 128 B = 6.3 TB/s: that rate IS the HBM roofline of this access
// shape), so the layout below is chosen to touch as few distinct lines per query as possible -- with as
// few load instructions as possible: a second load into a line that is already on its way is not free.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vg {

constexpr uint32_t POS_AMBIGUOUS = 0xFFFFFFFFu;   // src/vartype.h:38
constexpr uint32_t NOMOD = 10086u;                 // src/qv.cc:711
constexpr int AUX_COLS = 10;                       // src/vartype.h:93
constexpr uint32_t BLOCK_THRESHOLD = 100;          // src/vartype.h:103
constexpr int REF_STRIDE = 9;                      // sizeof(struct kmer_entry), src/vartype.h:64-72      (bug B1)
constexpr int SNP_STRIDE = 11;                     // sizeof(struct snp_kmer_entry), src/vartype.h:74-79  (bug B1)
constexpr uint64_t LO40_MASK = 0xFFFFFFFFFFull;

// One reference-dictionary entry, 16 bytes so that a hit costs one line (the reference packs 9).
struct __attribute__((aligned(16))) RefEnt { uint32_t lo, pos, amb, pad; };
// One SNP-dictionary entry: key = LO40 | snp_info << 40 | ambig_flag << 48.
struct __attribute__((aligned(16))) SnpEnt { uint64_t key; uint32_t pos, pad; };

// HBM layout of one index replica.  B1's index arithmetic is evaluated against these arrays in
// FILE ORDER, which is all the reference's behaviour depends on.
struct DevIndex {
	// reference dictionary (src/qv.cc:519-590)
	const uint32_t *ref_jg;        // [2^ref_jg_bits + 1] jump table over the top ref_jg_bits bits of HI32; the last entry = n_ref (sentinel replaces the 0xFFFFFFFF special case)
	// ref_jg_bits = 32: the reference's table, one entry per HI32 value (16 GiB whatever the genome, qv.cc:539-584).  A dictionary of
	// n k-mers fills 2^32 buckets to n / 2^32 -- 1 % for a 40 Mbp genome -- so a small index gets a COARSE table (r06: ~2 n buckets, the
	// top bits of HI32): a coarse bucket holds the HI32 bucket asked for and maybe a neighbour or two, and the entries say which is
	// which -- RefEnt::pad carries HI32 of the entry's k-mer (ref_bounds below).  Same [lo, hi) as the 2^32-entry table gives.
	uint32_t ref_jg_bits;
	const RefEnt   *ref;           // [n_ref]
	const uint32_t *ref_aux;       // [n_ref_aux][10]
	uint32_t aux_dups;             // some auxiliary row (either dictionary) lists a position twice: the wave kernel expands rows column by column
	uint64_t n_ref;
	// secondary view of the reference dictionary ordered by (LO32, HI32): every k-mer that shares a chunk's first 16 bases is
	// adjacent, so the 48 "last 16 bases differ in one base" neighbour queries of qv.cc:1213-1296 become one bucket read.
	// One 12-byte record per k-mer {HI32, pos, LO32 & 0x7FFFFFFF | ambig_flag << 31} (r03: what a hit needs -- position or row
	// index, ambiguity -- comes with the record, so a hit costs no gather of the dictionary entry; the top bit of LO32 is implied
	// by the bucket), sec_jg over the top sec_bits (>= 14) of LO32.
	// sec_is_bf: the loader has verified that the reference bit vector is exactly the set of LO32 values of this dictionary
	// (hash32 is a bijection; it holds for every index `vargeno index` writes from an upper-case FASTA): "is the bit set" is then
	// "does the bucket hold an entry with this LO32", and the probe of the 512 MiB vector is not made.
	const uint32_t *sec3;          // [n_ref][3]
	const uint32_t *sec_jg;        // [2^sec_bits + 1]
	uint32_t sec_bits;
	uint32_t sec_is_bf;
	// merged exact-match view: the reference and SNP dictionaries sorted together (reference entry first on ties) by the
	// CANONICAL form of their k-mers -- key = fmix64(min(k, revcomp k)), flag 8 of an entry: its k-mer is the reverse complement
	// of that form -- behind ONE jump table over HI32 of the key, so the two exact look-ups of a chunk (qv.cc:840-841) cost one
	// jump-table gather + at most one bucket line instead of two of each.  mx entry: {lo32, pos, flags, pos2} with
	// flags bit 0 = SNP-dictionary entry, bit 1 = ambig_flag, bit 2 = PAIR (set once the direct table exists: the k-mer's
	// auxiliary row holds exactly two positions and they are pos, pos2 -- no row gather in stage A).
	const uint32_t *mx_jg;         // [2^32 + 1]   (dropped when the direct table below could be allocated)
	const uint4 *mx;               // [n_ref + n_snp]
	// direct table over HI32: one 16-byte record per bucket = the bucket's FIRST merged entry inline {lo32, pos, flags,
	// index of that entry in mx}, flags bit 0 = bucket non-empty, bit 1 = SNP entry, bit 2 = ambig_flag, bit 5 = strand, bit 4 = TIE (the second
	// entry has the same k-mer), bit 3 = PAIR (single-
	// entry buckets only: then the last word is the second position), bits 8.. = entries in the bucket.  A bucket with one entry -- the common case -- is settled, hit or miss, by ONE gather.  64 GiB.
	const uint4 *dx;
	// dx_bits = 32: the table above.  dx_bits < 32 (r06: an index of n + m k-mers gets ~2 (n + m) buckets instead of 2^32 -- chr22-scale
	// 2 GB instead of 64 GiB --, and an hg38-scale index under a budget can take 2^31 or 2^30): buckets = the TOP dx_bits bits of the
	// mixed key; the bits of the key's high word that the bucket does not fix (32 - dx_bits <= 16 of them) travel left-aligned in
	// bits 16-31 of the flags word of every dx record and mx entry (field F), entries inside a bucket are ordered by (F, lo32) --
	// i.e. by the key --, and a dx record's count is 8 bits wide (bits 8-15; a bucket of more than 255 entries makes the loader drop
	// the merged view).  The kernel instantiation for this form compares (F, lo32) where the 2^32 form compares lo32.
	uint32_t dx_bits;
	// SNP dictionary (src/qv.cc:606-695)
	const uint32_t *snp_jg;        // [2^24 + 1]
	const SnpEnt   *snp;           // [n_snp]
	const uint32_t *snp_aux_pos;   // [n_snp_aux][10]
	const uint8_t  *snp_aux_info;  // [n_snp_aux][10]
	uint64_t n_snp;
	// strided-probe view of the SNP dictionary: snp_probe[i] = LO40 of entry slo + 11 (i - slo), slo = start of i's HI24 bucket
	// (0 where that index lies beyond the array) -- exactly the values iterate_snp_dict's scan (bug B1, qv.cc:447-455) tests for a
	// bucket, laid side by side: at hg38 scale a bucket holds ~19 entries, i.e. 19 lines 176 bytes apart become 3 adjacent ones.
	const uint64_t *snp_probe;     // [n_snp] (timed build only; nullptr: the probes read `snp` itself)
	// signature form of the same view (the default): snp_sig[i] = the 40 bits of snp_probe[i] folded to 16 (sig16 below).  Two 40-bit
	// values that differ in exactly one base have signatures that differ in exactly one base (the fold is linear and keeps a base's
	// two bits together), so the signatures say which entries of a bucket the strided scan can possibly keep -- 24 of 65 536 patterns
	// pass by chance -- and only those have their full value fetched (entry slo + 11 (i - slo) of `snp` itself).  2 bytes per entry
	// where the probe view holds 8: the ~190-entry buckets of hg38 + full dbSNP are 3 lines instead of 12, and one 16-byte load
	// covers eight entries of the scan.
	const uint16_t *snp_sig;       // [n_snp + 16]
	// LO32-ordered view of the SNP dictionary (r06), the counterpart of sec3 above: every SNP k-mer that shares a chunk's first 16 bases
	// is adjacent, so the high-half SNP neighbour queries of a gate-open chunk (qv.cc:1303-1352: up to 36 of them when the SNP bit
	// vector's probe is positive, each a jump-table gather + a ~5-deep bisection of a HI24 bucket, nearly all of them misses) are one
	// short bucket read in stage B0.  One 12-byte record per k-mer: {HI32, pos (or auxiliary row), LO32 & 0x03FFFFFF |
	// SNP_INFO_POS << 26 | ambig_flag << 31}; ssec_jg over the top ssec_bits (>= 14) bits of LO32, which with the record's 26 low bits
	// pin all 32.  profiles/view_census_*_r06.txt: those queries asked for 1.4 of a default read's 19.8 lines and 3.7 of a
	// repeat-rich read's 25.2, in dependent chains.
	const uint32_t *ssec3;         // [n_snp][3]
	const uint32_t *ssec_jg;       // [2^ssec_bits + 1]
	uint32_t ssec_bits;
	// jump table of the SNP dictionary over HI32 (2^32 + 1 words, like ref_jg): built only for an index too large for the merged view
	// (2^32 or more k-mers in the two dictionaries together: hg38 + full dbSNP), where a HI24 bucket holds ~190 entries and the
	// reference's bsearch would be 8 dependent probes; a HI32 bucket holds one or two.  Same entries found: the dictionary is
	// sorted by the whole k-mer, so a HI32 bucket is a contiguous piece of its HI24 bucket.
	const uint32_t *snp_jg32;
	// paired HI32 table (an index too large for the merged view): ONE 16-byte record per HI32 value in place of the two jump tables
	// ref_jg / snp_jg32 -- {first reference entry, first SNP entry, reference count | SNP count << 16, reference filter | SNP
	// filter << 16}; record 2^32 is the sentinel {n_ref, n_snp, 0, 0}.  Counts saturate at 0xFFFF (then the next record's start is
	// the bucket's end).  A filter says which low halves a bucket can hold, so that a look-up of a k-mer the dictionary does not
	// have -- every look-up of a reverse-strand read's forward pass, the SNP side of nearly every look-up, nearly every neighbour
	// query -- ends at this record instead of fetching a bucket entry to find that out: 16-bit fingerprint of the only entry
	// (count 1), 16-bit presence mask over a 4-bit hash of the entries (count > 1).  ref_jg == snp_jg32 == nullptr when hx is built.
	const uint4 *hx;
	// bit vectors (src/generate_bf.h:112-142)
	const uint64_t *ref_bf; uint64_t ref_bf_bits;
	const uint64_t *snp_bf; uint64_t snp_bf_bits;
	// one byte per genome position: bits 0-1 ref, 2-3 alt as seeded from the SNP dictionary (src/vartype.h:81-90,
	// qv.cc:637-659), bit 4 = "is a SNP site" (ref != alt).  A 32-base pile-up walk reads 32 contiguous bytes (the
	// reference's packed_pileup_entry table costs 128).  The site id behind a position comes from a rank block:
	// per 64 positions {bit mask of sites, number of sites before the block}; counters live in `cnt`.
	const uint8_t *pile; uint64_t pile_len;
	const ulonglong2 *srank;       // [pile_len / 64 + 1]  .x = site bits of the block, .y = sites before it
	uint32_t *cnt;                 // [2 * n_sites] exact sums: [2s] ref, [2s+1] alt
	// The wave kernel's walks count by the base the read shows -- [4s + base] -- which takes the site's ref/alt bases (and
	// with them the 32-byte pile window) out of the walk: the rank block alone says where the sites are.  Folded into
	// `cnt` with site_ba (ref | alt << 2 per site) before anything reads the sums (vg_fold_counters).
	uint32_t *cnt4;                // [4 * n_sites]
	const uint8_t *site_ba;        // [n_sites]
};

enum StatId {
	S_READS, S_READS_N, S_READS_INVALID, S_PASSES, S_PASSES_OK, S_CHUNKS, S_GATE_OPEN, S_REFBF_POS, S_SNPBF_POS,
	S_LARGE_BLOCK, S_REF_QUERY, S_SNP_QUERY, S_REF_PROBE, S_SNP_PROBE, S_SCAN_REF, S_SCAN_SNP, S_SCAN_OOB,
	S_AUX_REF, S_AUX_SNP, S_SITE_TEST, S_CTX, S_WALKS, S_INCR, S_INGEST, S_COUNT
};

template <bool STATS> struct LaneStats;
template <> struct LaneStats<false> {
	static constexpr bool counting = false;
	__device__ inline void add(int, uint32_t) {}
	__device__ inline void clear() {}
};
template <> struct LaneStats<true> {
	static constexpr bool counting = true;
	uint32_t v[S_COUNT];
	__device__ inline void add(int id, uint32_t x) { v[id] += x; }
	__device__ inline void clear() { for (int i = 0; i < S_COUNT; i++) v[i] = 0; }
};

__device__ inline uint32_t hash32(uint32_t x) { x = ((x >> 16) ^ x) * 0x45d9f3bu; x = ((x >> 16) ^ x) * 0x45d9f3bu; return (x >> 16) ^ x; }
__device__ inline uint64_t hash40(uint64_t x) { x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull; return x ^ (x >> 31); }

__device__ inline uint32_t ceil_log2_p1(uint32_t b) { return 32u - (uint32_t)__clz((int)b); }   // ceil(log2(b+1)) for b >= 1

// 40 bits -> 16, linear over xor, bases stay whole (bit pairs map to bit pairs)
__device__ __host__ inline uint32_t sig16(uint64_t lo40) { return (uint32_t)((lo40 ^ (lo40 >> 16) ^ (lo40 >> 32)) & 0xFFFFull); }
// is x (16 bits) non-zero and confined to one base?
__device__ inline bool onebase16(uint32_t x) { return x != 0u && (x & ~(3u << ((uint32_t)(__ffs((int)x) - 1) & ~1u))) == 0u; }

// one_hamming_distance_32/64 (src/qv.cc:267-312): x != 0 and confined to one base -> base index, else -1
__device__ inline int onebase(uint64_t x)
{
	if (x == 0) return -1;
	int d = (int)(__ffsll((long long)x) - 1) >> 1;
	return (x & ~(3ull << (2 * d))) ? -1 : d;
}

// reverse complement of a 32-mer in 2-bit space (what src/qv.cc:786-806 does on characters)
__device__ inline uint64_t revcomp64(uint64_t k)
{
	k = ((k >> 2) & 0x3333333333333333ull) | ((k & 0x3333333333333333ull) << 2);
	k = ((k >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((k & 0x0F0F0F0F0F0F0F0Full) << 4);
	k = __builtin_bswap64(k);
	return ~k;
}

// bijective 64-bit mix (murmur3's finaliser): spreads canonical k-mers evenly over the HI32 buckets of the merged view
__device__ __host__ inline uint64_t fmix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

// 4 ASCII bases (little-endian in v) -> 8 bits, base 0 in bits 0-1 (encode_kmer, src/util.c:89-111: A0 C1 G2 T3,
// case-insensitive), in four instructions (r05; the SWAR form it replaces took ~40 and made the pack kernel VALU-bound).
// Bits 1-2 of an ASCII base tell the four letters apart (A 00, C 01, T 10, G 11), so `v & 0x06` per byte is a byte selector
// for v_perm_b32 into two 4-entry tables held in a register pair: the letter that selector stands for, and its 2-bit code.
// The codes' dot product with (1, 4, 16, 64) is the packed byte (v_dot4_u32_u8).  `badraw` collects letter ^ byte; a byte is
// one of ACGTacgt iff that is zero outside the case bit 0x20 (for each value of bits 1-2 exactly one upper-case letter fits).
__device__ __forceinline__ uint32_t pack4(uint32_t v, uint32_t &badraw)
{
	const uint32_t sel = v & 0x06060606u;                                             // A 0, C 2, T 4, G 6: bytes 0, 2 of the second table word, bytes 0, 2 of the first
	badraw |= __builtin_amdgcn_perm(0x00470054u, 0x00430041u, sel) ^ v;
	return __builtin_amdgcn_udot4(__builtin_amdgcn_perm(0x00020003u, 0x00010000u, sel), 0x40100401u, 0u, false);
}
constexpr uint32_t PACK_CASE_MASK = 0xDFDFDFDFu;         // badraw & this != 0: some byte was not one of ACGTacgt
// 16 ASCII bases -> 32 bits
__device__ __forceinline__ uint32_t pack16(uint4 v, uint32_t &badraw)
{
	return pack4(v.x, badraw) | (pack4(v.y, badraw) << 8) | (pack4(v.z, badraw) << 16) | (pack4(v.w, badraw) << 24);
}
// 8 ASCII bases -> 16 bits; `bad` gets a non-zero value if any byte is not one of ACGTacgt.
__device__ inline uint32_t pack8(uint64_t v, uint64_t &bad)
{
	uint32_t br = 0;
	const uint32_t r = pack4((uint32_t)v, br) | (pack4((uint32_t)(v >> 32), br) << 8);
	bad |= (uint64_t)(br & PACK_CASE_MASK);
	return r;
}

__device__ inline uint64_t load8(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

__device__ inline uint64_t encode32(const uint8_t *p, uint64_t &bad)
{
	return (uint64_t)pack8(load8(p), bad) | ((uint64_t)pack8(load8(p + 8), bad) << 16) |
	       ((uint64_t)pack8(load8(p + 16), bad) << 32) | ((uint64_t)pack8(load8(p + 24), bad) << 48);
}

// Exact classification when some byte is not ACGT: the reference encodes chunk 0..n-1, each from base
// 31 down to 0, and the FIRST offending character decides: N/n -> skip the read (src/qv.cc:815-828),
// anything else -> assert(0) (src/util.c:103).  1 = N, 2 = invalid.
__device__ inline int classify_bad(const uint8_t *p, uint32_t n)
{
	for (uint32_t c = 0; c < n; c++)
		for (int j = 31; j >= 0; j--) {
			const uint8_t ch = p[32 * c + j] & 0xDF;
			if (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T') continue;
			return ch == 'N' ? 1 : 2;
		}
	return 0;
}

// The same classification with 16-byte loads and bit masks instead of a byte walk (the pack kernel's path for the few reads
// whose pieces of text hold an offending byte): per chunk, bit i of `off` = byte i is not one of ACGTacgt, bit i of `isn` = it
// is N / n; the reference's scan meets the HIGHEST offending byte of the first offending chunk first.
__device__ inline int classify_bad_wide(const uint8_t *p, uint32_t n)
{
	for (uint32_t c = 0; c < n; c++) {
		uint4 q[2];
		__builtin_memcpy(&q[0], p + 32 * c, 16);
		__builtin_memcpy(&q[1], p + 32 * c + 16, 16);
		const uint32_t w[8] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w};
		uint32_t off = 0, isn = 0;
		#pragma unroll
		for (int i = 0; i < 8; i++) {
			const uint32_t u = w[i] & PACK_CASE_MASK;
			const uint32_t d = __builtin_amdgcn_perm(0x00470054u, 0x00430041u, w[i] & 0x06060606u) ^ u, z = u ^ 0x4E4E4E4Eu;
			const uint32_t dn = (((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u;              // 0x80 per non-zero byte of d
			const uint32_t zn = ~(((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u;             // 0x80 per zero byte of z
			off |= __builtin_amdgcn_udot4(dn >> 7, 0x08040201u, 0u, false) << (4 * i);
			isn |= __builtin_amdgcn_udot4(zn >> 7, 0x08040201u, 0u, false) << (4 * i);
		}
		if (off) return ((isn >> (31 - __clz((int)off))) & 1u) ? 1 : 2;
	}
	return 0;
}

// Loads with the `nt` bit of gfx950's memory instructions ("nothing will touch this line again soon").  A probe of bare random
// gathers from a table far larger than L2 (tools/gather_policy_probe, profiles/gather_policy_probe_r02.jsonl) runs at 55.0 G/s
// with it and 50.9 G/s without, whatever the width (sc0 / sc1 change nothing; an L2-resident table: 250 G/s) -- but the read
// loop is not that probe.  Measured on the hg38-scale workload (profiles/ab_hg38_r02_a12.txt, _a13.txt; run-to-run spread
// +-1.5 %): the bit on the dictionary / view gathers changes nothing, on the rank blocks under a read it costs ~3 % (their
// neighbours are wanted a moment later), on the two bit-vector probes of a gate-open chunk it is worth 1 % at most.  Hence
// three switches, by what is loaded.  ALIGN: what the address is known to be aligned to.
#ifndef VG_NT_TAB
#define VG_NT_TAB 0
#endif
#ifndef VG_NT_BF
#define VG_NT_BF 1
#endif
#ifndef VG_NT_WALK
#define VG_NT_WALK 0
#endif

// -DVG_VIEW_COUNTERS: development aid, never in the shipped build -- a census of the 128-byte lines the read loop ASKS for, per view of
// the index (profiles/view_traffic_r06.txt: which view do the extra lines of a repeat-rich genome belong to?).  Every load that goes
// through gather<> / gather_bf / gather_walk / load_row10 / load_row4 is classified by its ADDRESS against the handle's arrays (a table
// of ranges the loader fills in: vg_vc_ranges); the few plain dereferences and the row / site-byte loads whose stage matters are counted
// at their sites under an explicit id.  "Lines" = the 128-byte lines one load's bytes span, summed over lanes: what is requested of the
// memory system before L1 / L2 merge anything -- to be read beside TCC_MISS of the same launch.
#ifdef VG_VIEW_COUNTERS
enum VcId { VC_OTHER, VC_READS, VC_DX, VC_MX, VC_REF_JG, VC_REF, VC_SNP_JG, VC_SNP, VC_SEC_JG, VC_SEC3, VC_SIG, VC_REF_BF, VC_SNP_BF, VC_AUX_A, VC_AUX_B, VC_PILE_B, VC_PILE_C, VC_SRANK, VC_CNT4, VC_AUX_ANY, VC_PILE_ANY, VC_N };
struct VcRange { unsigned long long lo, hi; int id; int pad; };
__device__ VcRange vg_vc_ranges[24];
__device__ int vg_vc_nranges;
__device__ unsigned long long vg_vc_lines[VC_N], vg_vc_loads[VC_N];
__device__ inline void vc_count_as(int id, const void *p, uint32_t bytes)
{
	const unsigned long long a = (unsigned long long)p;
	atomicAdd(&vg_vc_loads[id], 1ull);
	atomicAdd(&vg_vc_lines[id], (a + bytes - 1) / 128 - a / 128 + 1);
}
__device__ inline void vc_count(const void *p, uint32_t bytes)
{
	const unsigned long long a = (unsigned long long)p;
	int id = VC_OTHER;
	for (int i = 0; i < vg_vc_nranges; i++) if (a >= vg_vc_ranges[i].lo && a < vg_vc_ranges[i].hi) { id = vg_vc_ranges[i].id; break; }
	vc_count_as(id, p, bytes);
}
#define VG_VC(p, bytes) vc_count((p), (bytes))
#define VG_VC_AS(id, p, bytes) vc_count_as((id), (p), (bytes))
#else
#define VG_VC(p, bytes) do { } while (0)
#define VG_VC_AS(id, p, bytes) do { } while (0)
#endif
typedef uint32_t nt_v4a16 __attribute__((ext_vector_type(4), aligned(16)));
typedef uint32_t nt_v4a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef uint32_t nt_v2a8 __attribute__((ext_vector_type(2), aligned(8)));
typedef uint32_t nt_v2a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t nt_v4a1 __attribute__((ext_vector_type(4), aligned(1)));
template <bool NT, typename T, int ALIGN>
__device__ __forceinline__ T load_policy(const void *p)
{
	static_assert(sizeof(T) == 4 || sizeof(T) == 8 || sizeof(T) == 16, "4, 8 or 16 bytes");
	static_assert(ALIGN == 1 || ALIGN == 4 || ALIGN == 8 || ALIGN == 16, "alignment 1, 4, 8 or 16");
	T r;
	if constexpr (!NT) __builtin_memcpy(&r, __builtin_assume_aligned(p, ALIGN), sizeof(T));
	else if constexpr (sizeof(T) == 16 && ALIGN >= 16) { const nt_v4a16 v = __builtin_nontemporal_load((const nt_v4a16 *)p); __builtin_memcpy(&r, &v, 16); }
	else if constexpr (sizeof(T) == 16 && ALIGN == 8) { const nt_v4a8 v = __builtin_nontemporal_load((const nt_v4a8 *)p); __builtin_memcpy(&r, &v, 16); }
	else if constexpr (sizeof(T) == 16) { static_assert(ALIGN == 1, "16 bytes: aligned to 16, 8 or not at all"); const nt_v4a1 v = __builtin_nontemporal_load((const nt_v4a1 *)p); __builtin_memcpy(&r, &v, 16); }
	else if constexpr (sizeof(T) == 8 && ALIGN >= 8) { const nt_v2a8 v = __builtin_nontemporal_load((const nt_v2a8 *)p); __builtin_memcpy(&r, &v, 8); }
	else if constexpr (sizeof(T) == 8) { static_assert(ALIGN == 4, "8 bytes: aligned to 8 or 4"); const nt_v2a4 v = __builtin_nontemporal_load((const nt_v2a4 *)p); __builtin_memcpy(&r, &v, 8); }
	else { const uint32_t v = __builtin_nontemporal_load((const uint32_t *)p); __builtin_memcpy(&r, &v, 4); }
	return r;
}
template <typename T, int ALIGN = alignof(T)> __device__ __forceinline__ T gather(const void *p) { VG_VC(p, sizeof(T)); return load_policy<VG_NT_TAB != 0, T, ALIGN>(p); }        // dictionaries, views, jump tables
template <typename T> __device__ __forceinline__ T gather_bf(const T *p) { VG_VC(p, sizeof(T)); return load_policy<VG_NT_BF != 0, T, alignof(T)>(p); }                           // bit vectors
template <typename T> __device__ __forceinline__ T gather_walk(const T *p) { VG_VC(p, sizeof(T)); return load_policy<VG_NT_WALK != 0, T, alignof(T)>(p); }                       // rank blocks

// filters of the paired HI32 table (DevIndex::hx), over the low 32 bits of a k-mer
__device__ __host__ inline uint32_t hx_fp16(uint32_t lo) { return (lo ^ (lo >> 16)) & 0xFFFFu; }
__device__ __host__ inline uint32_t hx_bit(uint32_t lo) { return 1u << ((lo * 0x9E3779B1u) >> 28); }
// can a bucket with `cnt` entries and filter word `f` hold low half `lo`?  (never a false "no")
__device__ inline bool hx_may_hold(uint32_t cnt, uint32_t f, uint32_t lo)
{
	return cnt > 1u ? (f & hx_bit(lo)) != 0u : (cnt == 1u && f == hx_fp16(lo));
}

// one 12-byte record (4-byte aligned)
__device__ __forceinline__ uint3 gather12(const uint32_t *p) { uint3 r; __builtin_memcpy(&r, __builtin_assume_aligned(p, 4), 12); return r; }

// bucket bounds: one 8-byte gather (two adjacent jump-table words)
__device__ inline void jg_pair(const uint32_t *jg, uint64_t h, uint32_t &lo, uint32_t &hi)
{
	const uint64_t v = gather<uint64_t, 4>(jg + h);
	lo = (uint32_t)v; hi = (uint32_t)(v >> 32);
}

// bounds of the HI32 bucket of the reference dictionary, from whichever table the index has
__device__ inline void ref_bounds(const DevIndex &d, uint64_t h, uint32_t &lo, uint32_t &hi)
{
	if (d.hx) {
		const uint4 r = gather<uint4>(d.hx + h);
		const uint32_t c = r.z & 0xFFFFu;
		lo = r.x;
		hi = c == 0xFFFFu ? gather<uint32_t>(&d.hx[h + 1].x) : lo + c;
	}
#if defined(VG_AB_NO_COARSE)                                       // (A/B builds only: the reference jump table's coarse form compiled out)
	else jg_pair(d.ref_jg, h, lo, hi);
#else
	else if (d.ref_jg_bits >= 32u) jg_pair(d.ref_jg, h, lo, hi);
	else {
		// coarse table: the bucket of h's top bits, then the run of entries whose k-mer has exactly this HI32 (RefEnt::pad) inside it --
		// the entries are sorted by the whole k-mer, so by HI32 first.  Mostly zero to two entries: walked; a long bucket: bisected.
		uint32_t a, e;
		jg_pair(d.ref_jg, h >> (32u - d.ref_jg_bits), a, e);
		const uint32_t hh = (uint32_t)h;
		if (e - a > 8u) {
			uint32_t x = a, y = e;
			while (x < y) { const uint32_t m = x + ((y - x) >> 1); if (gather<uint32_t>(&d.ref[m].pad) < hh) x = m + 1; else y = m; }
			lo = x; y = e;
			while (x < y) { const uint32_t m = x + ((y - x) >> 1); if (gather<uint32_t>(&d.ref[m].pad) <= hh) x = m + 1; else y = m; }
			hi = x;
		} else {
			// (all of the bucket's entries asked for before any is looked at: one wait, not one per entry)
			uint32_t pv[8], below = 0, same = 0;
			#pragma unroll
			for (uint32_t i = 0; i < 8u; i++) { pv[i] = 0u; if (a + i < e) pv[i] = gather<uint32_t>(&d.ref[a + i].pad); }
			#pragma unroll
			for (uint32_t i = 0; i < 8u; i++) if (a + i < e) { below += pv[i] < hh ? 1u : 0u; same += pv[i] == hh ? 1u : 0u; }
			lo = a + below; hi = lo + same;
		}
	}
#endif
}

// columns [j0, j0 + 4) of an auxiliary-table row (AUX_COLS = 10 positions, rows 8-byte aligned) as two independent 8-byte
// gathers; columns past the row read as 0.  Rows end at their first 0, and most hold two or three positions, so one call
// -- one wait -- usually settles a row that a column-by-column walk would wait on three or four times.
__device__ inline void load_row4(const uint32_t *row, int j0, uint32_t (&v)[4])
{
	uint2 a, b = make_uint2(0u, 0u);
	VG_VC(row + j0, j0 + 2 < AUX_COLS ? 16 : 8);
	__builtin_memcpy(&a, row + j0, 8);
	if (j0 + 2 < AUX_COLS) __builtin_memcpy(&b, row + j0 + 2, 8);
	v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}

// a whole row (AUX_COLS = 10 positions, 8-byte aligned) as three independent gathers: one wait
__device__ inline void load_row10(const uint32_t *row, uint32_t (&v)[AUX_COLS])
{
	VG_VC(row, 40);
	const uint4 a = load_policy<VG_NT_TAB != 0, uint4, 8>(row), b = load_policy<VG_NT_TAB != 0, uint4, 8>(row + 4);
	const uint2 c = load_policy<VG_NT_TAB != 0, uint2, 8>(row + 8);
	v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; v[8] = c.x; v[9] = c.y;
}

// query_ref_dict, src/qv.cc:206-240.  Returns the entry index or -1; lo/hi = bucket of HI32(k); `ent` = the entry
// found (every probe fetches the whole 16-byte entry, so a hit needs no second gather).  Keys are unique within
// a bucket, so an equality search returns what bsearch() returns.
template <class ST>
__device__ inline int64_t ref_query(const DevIndex &d, ST &st, uint64_t k, uint32_t &lo, uint32_t &hi, RefEnt &ent)
{
	ref_bounds(d, k >> 32, lo, hi);
	st.add(S_REF_QUERY, 1);
	if (lo == hi) return -1;                          // also covers lo == n_ref (then hi == n_ref)
	st.add(S_REF_PROBE, ceil_log2_p1(hi - lo));
	const uint32_t key = (uint32_t)k;
	uint32_t a = lo, b = hi;
	while (a < b) {
		const uint32_t m = a + ((b - a) >> 1);
		ent = d.ref[m];
		if (ent.lo == key) return (int64_t)m;
		if (ent.lo < key) a = m + 1; else b = m;
	}
	return -1;
}
template <class ST>
__device__ inline int64_t ref_query(const DevIndex &d, ST &st, uint64_t k, uint32_t &lo, uint32_t &hi)
{
	RefEnt e;
	return ref_query(d, st, k, lo, hi, e);
}
// query_snp_dict, src/qv.cc:385-411
template <class ST>
__device__ inline int64_t snp_query(const DevIndex &d, ST &st, uint64_t k, uint32_t &lo, uint32_t &hi, SnpEnt &ent)
{
	jg_pair(d.snp_jg, k >> 40, lo, hi);
	st.add(S_SNP_QUERY, 1);
	if (lo == hi) return -1;
	st.add(S_SNP_PROBE, ceil_log2_p1(hi - lo));
	const uint64_t key = k & LO40_MASK;
	uint32_t a = lo, b = hi;
	while (a < b) {
		const uint32_t m = a + ((b - a) >> 1);
		ent = d.snp[m];
		const uint64_t ek = ent.key & LO40_MASK;
		if (ek == key) return (int64_t)m;
		if (ek < key) a = m + 1; else b = m;
	}
	return -1;
}
template <class ST>
__device__ inline int64_t snp_query(const DevIndex &d, ST &st, uint64_t k, uint32_t &lo, uint32_t &hi)
{
	SnpEnt e;
	return snp_query(d, st, k, lo, hi, e);
}

// both HI32 buckets of k-mer k from the paired table, already narrowed by its filters: an empty range = "cannot be there"
__device__ inline void hx_bounds(const DevIndex &d, uint64_t k, bool want_r, bool want_s, uint32_t &ra, uint32_t &rb, uint32_t &sa, uint32_t &sb)
{
	const uint64_t h = k >> 32;
	const uint4 r = gather<uint4>(d.hx + h);
	const uint32_t rc = r.z & 0xFFFFu, sc = r.z >> 16, lo = (uint32_t)k;
	ra = rb = sa = sb = 0;
	if (want_r && hx_may_hold(rc, r.w & 0xFFFFu, lo)) { ra = r.x; rb = rc == 0xFFFFu ? gather<uint32_t>(&d.hx[h + 1].x) : r.x + rc; }
	if (want_s && hx_may_hold(sc, r.w >> 16, lo)) { sa = r.y; sb = sc == 0xFFFFu ? gather<uint32_t>(&d.hx[h + 1].y) : r.y + sc; }
}

// Both dictionary queries of one neighbour k-mer in lock step: the two jump-table gathers go out together and so do the two
// probes of every bisection step, so the pair costs 1 + max(depth) waits instead of 2 + the sum.  Same results and the same
// event counts as ref_query + snp_query.  ri / si are left untouched on a miss.
template <class ST>
__device__ inline void dual_query(const DevIndex &d, ST &st, uint64_t k, bool want_r, bool want_s, uint32_t &ri, uint32_t &si)
{
	uint32_t ra = 0, rb = 0, sa = 0, sb = 0;
	if (!ST::counting && d.hx) {
		// paired HI32 table: both buckets in one record, and a bucket that cannot hold the k-mer is not looked at
		hx_bounds(d, k, want_r, want_s, ra, rb, sa, sb);
	} else {
		if (want_r) { ref_bounds(d, k >> 32, ra, rb); st.add(S_REF_QUERY, 1); }
		if (want_s) {
			// (the counting build prices the walk through the HI24 table; the HI32 table, when the index has one, bounds the same entries)
			if (!ST::counting && d.snp_jg32) jg_pair(d.snp_jg32, k >> 32, sa, sb); else jg_pair(d.snp_jg, k >> 40, sa, sb);
			st.add(S_SNP_QUERY, 1);
		}
	}
	if (ra < rb) st.add(S_REF_PROBE, ceil_log2_p1(rb - ra));
	if (sa < sb) st.add(S_SNP_PROBE, ceil_log2_p1(sb - sa));
	const uint32_t rkey = (uint32_t)k;
	const uint64_t skey = k & LO40_MASK;
#ifdef VG_DBG_SHALLOW
	if (rb - ra > 4u) rb = ra;                                      // timing experiment only (wrong results): what do the deep bisections cost?
	if (sb - sa > 4u) sb = sa;
#endif
	while (ra < rb || sa < sb) {
		const bool pr = ra < rb, ps = sa < sb;
		const uint32_t rm = ra + ((rb - ra) >> 1), sm = sa + ((sb - sa) >> 1);
		uint32_t rlo = 0;
		uint64_t sk = 0;
		if (pr) rlo = gather<uint32_t>(&d.ref[rm].lo);
		if (ps) sk = gather<uint64_t>(&d.snp[sm].key) & LO40_MASK;
		if (pr) { if (rlo == rkey) { ri = rm; ra = rb; } else if (rlo < rkey) ra = rm + 1; else rb = rm; }
		if (ps) { if (sk == skey) { si = sm; sa = sb; } else if (sk < skey) sa = sm + 1; else sb = sm; }
	}
}

// The exact look-ups of Z chunks (the four of a 150 bp read, r04; r03: two) in both dictionaries in lock step, for an index without
// the merged view: their bucket-bound gathers go out together, then every bisection step probes all 2 Z buckets at once -- 1 + max
// depth waits for what 2 Z separate queries would spend ~3 each on.  SNP buckets come from the HI32 jump table when the index has one.  The entries found
// are returned whole (every probe fetches 16 bytes).  Not for the counting build: no events are booked.
template <int Z>
__device__ inline void exact_multi_nomx(const DevIndex &d, const uint64_t (&k)[Z], const bool (&want)[Z], uint32_t (&rhit)[Z], uint32_t (&rpos)[Z], uint32_t (&shit)[Z], uint32_t (&spos)[Z])
{
	// rhit / shit: 0 = miss, 1 = hit, 3 = hit on an entry with several positions (pos is then the auxiliary row); only the
	// position and that flag leave the loop -- whole entries kept for four look-ups were what the kernel spilled (40 bytes of
	// scratch per lane in r03)
	uint32_t ra[Z], rb[Z], sa[Z], sb[Z];
	#pragma unroll
	for (int z = 0; z < Z; z++) {
		ra[z] = rb[z] = sa[z] = sb[z] = 0;
		rhit[z] = shit[z] = 0; rpos[z] = spos[z] = 0;
		if (!want[z]) continue;
		if (d.hx) { hx_bounds(d, k[z], true, true, ra[z], rb[z], sa[z], sb[z]); continue; }
		ref_bounds(d, k[z] >> 32, ra[z], rb[z]);
		if (d.snp_jg32) jg_pair(d.snp_jg32, k[z] >> 32, sa[z], sb[z]); else jg_pair(d.snp_jg, k[z] >> 40, sa[z], sb[z]);
	}
	for (;;) {
		bool pr[Z], ps[Z], any = false;
		uint32_t rm[Z], sm[Z];
		RefEnt er[Z]; SnpEnt es[Z];
		#pragma unroll
		for (int z = 0; z < Z; z++) {
			pr[z] = ra[z] < rb[z]; ps[z] = sa[z] < sb[z];
			any = any || pr[z] || ps[z];
			rm[z] = ra[z] + ((rb[z] - ra[z]) >> 1); sm[z] = sa[z] + ((sb[z] - sa[z]) >> 1);
			if (pr[z]) er[z] = d.ref[rm[z]];
			if (ps[z]) es[z] = d.snp[sm[z]];
		}
		if (!any) break;
		#pragma unroll
		for (int z = 0; z < Z; z++) {
			if (pr[z]) {
				const uint32_t key = (uint32_t)k[z];
				if (er[z].lo == key) { rhit[z] = 1u | (er[z].amb ? 2u : 0u); rpos[z] = er[z].pos; ra[z] = rb[z]; } else if (er[z].lo < key) ra[z] = rm[z] + 1; else rb[z] = rm[z];
			}
			if (ps[z]) {
				const uint64_t key = k[z] & LO40_MASK, ek = es[z].key & LO40_MASK;
				if (ek == key) { shit[z] = 1u | (((es[z].key >> 48) & 0xFFu) ? 2u : 0u); spos[z] = es[z].pos; sa[z] = sb[z]; } else if (ek < key) sa[z] = sm[z] + 1; else sb[z] = sm[z];
			}
		}
	}
}

template <class ST>
__device__ inline bool site_loose(const DevIndex &d, ST &st, uint32_t p)     // !(ref == 0 && alt == 0), qv.cc:990-991
{
	st.add(S_SITE_TEST, 1);
	if (p < d.pile_len) VG_VC(d.pile + p, 1);
	return p < d.pile_len && (d.pile[p] & 15u) != 0;
}

// pile-up walk of one supporting context (src/qv.cc:1386-1436 = :1444-1494): 32 consecutive site bytes as two
// 16-byte gathers; only bases that match ref or alt at a site need the site id (one rank-block gather).
// Saturation is applied at fetch time as min(63, sum).
// The walk is split so that a caller can batch it: load_pile_window (two gathers, no dependence on anything
// else), walk_matches (pure ALU over the window), bump_site (one rank gather + one atomic).
__device__ inline void load_pile_window(const DevIndex &d, uint32_t kpos, uint4 (&w4)[2])
{
	__builtin_memcpy(&w4[0], d.pile + kpos, 16);
	__builtin_memcpy(&w4[1], d.pile + kpos + 16, 16);
}

template <class F>
__device__ inline void walk_matches(const uint4 (&w4)[2], uint64_t kk, uint32_t kpos, uint32_t mod, F &&hit)
{
	const uint32_t w[8] = {w4[0].x, w4[0].y, w4[0].z, w4[0].w, w4[1].x, w4[1].y, w4[1].z, w4[1].w};
	// site flags of the 32 positions: bit 4 of every byte
	uint32_t sites = 0;
	#pragma unroll
	for (int g = 0; g < 8; g++) {
		const uint32_t f = (w[g] >> 4) & 0x01010101u;               // one flag per byte
		sites |= ((f | (f >> 7) | (f >> 14) | (f >> 21)) & 0xFu) << (4 * g);
	}
	if (mod < 32u) sites &= ~(1u << mod);
	while (sites) {
		const uint32_t b = (uint32_t)__ffs((int)sites) - 1;
		sites &= sites - 1;
		uint32_t by = 0;
		#pragma unroll
		for (int g = 0; g < 8; g++) if ((b >> 2) == (uint32_t)g) by = w[g];
		by = (by >> (8 * (b & 3u))) & 0xFFu;
		const uint32_t base = (uint32_t)(kk >> (2 * b)) & 3u;
		if (base == (by & 3u)) hit(kpos + b, 0u);
		else if (base == ((by >> 2) & 3u)) hit(kpos + b, 1u);
	}
}

__device__ inline uint32_t site_id(const ulonglong2 rb, uint32_t p)
{
	return (uint32_t)rb.y + (uint32_t)__popcll(rb.x & ((1ull << (p & 63u)) - 1ull));
}

__device__ inline void bump_site(const DevIndex &d, uint32_t p, uint32_t which)
{
	atomicAdd(&d.cnt[2ull * site_id(d.srank[p >> 6], p) + which], 1u);
}

template <class ST>
__device__ inline void walk_ctx(const DevIndex &d, ST &st, uint64_t kk, uint32_t kpos, uint32_t mod)
{
	st.add(S_WALKS, 1);
	if ((uint64_t)kpos + 32 > d.pile_len) return;          // cannot happen: pile_len = max position + 64
	uint4 w4[2];
	load_pile_window(d, kpos, w4);
	walk_matches(w4, kk, kpos, mod, [&](uint32_t p, uint32_t which) { bump_site(d, p, which); st.add(S_INCR, 1); });
}

// ------------------------------------------------------------------------------------------------
// Generic lane machine: one lane runs one read sequentially with its hit contexts and vote keys in
// HBM scratch.  It handles every read (any length, any number of hits) and is the tier the
// wave-cooperative kernel (vg_wave.h) falls back to for the few reads that do not fit its LDS lists.
// ------------------------------------------------------------------------------------------------

// Per-lane scratch in HBM, slot-major so that lanes of a wave touching the same slot coalesce.
struct Scratch {
	uint64_t *ctx_kmer;            // [cap][nlanes]
	uint32_t *ctx_kpos;            // [cap][nlanes]
	uint32_t *ctx_meta;            // [cap][nlanes]  mod (16 bits) | chunk << 16
	uint32_t *key_index;           // [kcap][nlanes]
	uint32_t *key_first;           // [kcap][nlanes]
	uint32_t *key_fm;              // [kcap][nlanes]  freq (8 bits) | multi << 8
	uint32_t cap, kcap, nlanes;
};

template <bool STATS>
struct Lane {
	const DevIndex &d;
	const Scratch &s;
	uint32_t lane;                 // scratch column
	uint32_t nctx, nkeys;
	int best; bool amb; bool overflow;
	LaneStats<STATS> st;

	__device__ Lane(const DevIndex &d_, const Scratch &s_, uint32_t lane_) : d(d_), s(s_), lane(lane_), nctx(0), nkeys(0), best(-1), amb(false), overflow(false) {}

	__device__ inline void reset_pass() { nctx = 0; nkeys = 0; best = -1; amb = false; }

	__device__ inline void push_ctx(uint64_t kk, uint32_t kpos, uint32_t mod, uint32_t chunk)
	{
		if (nctx >= s.cap) { overflow = true; return; }
		const uint64_t at = (uint64_t)nctx * s.nlanes + lane;
		s.ctx_kmer[at] = kk; s.ctx_kpos[at] = kpos; s.ctx_meta[at] = (mod & 0xFFFFu) | (chunk << 16);
		nctx++;
		st.add(S_CTX, 1);
	}
	// improved_index_table_add, src/qv.cc:132-178
	__device__ inline void vote(uint32_t index, uint32_t kpos, bool neigh)
	{
		int e = -1;
		for (uint32_t i = 0; i < nkeys; i++) if (s.key_index[(uint64_t)i * s.nlanes + lane] == index) { e = (int)i; break; }
		uint32_t first, fm;
		if (e < 0) {
			if (neigh) return;                                                    // :134-139
			if (nkeys >= s.kcap) { overflow = true; return; }
			e = (int)nkeys++;
			s.key_index[(uint64_t)e * s.nlanes + lane] = index;
			s.key_first[(uint64_t)e * s.nlanes + lane] = first = kpos;
			fm = 0;
		} else {
			first = s.key_first[(uint64_t)e * s.nlanes + lane];
			fm = s.key_fm[(uint64_t)e * s.nlanes + lane];
		}
		const uint32_t freq = (fm + 1) & 0xFFu;                                   // uint8_t freq, :146
		const uint32_t multi = (fm >> 8) | (kpos != first ? 1u : 0u);             // |set| >= 2, :163-165
		s.key_fm[(uint64_t)e * s.nlanes + lane] = freq | (multi << 8);
		if (!multi) return;
		if (best < 0) { best = e; amb = false; }
		else if (e == best) amb = false;
		else {
			const uint32_t bf = s.key_fm[(uint64_t)best * s.nlanes + lane] & 0xFFu;
			if (freq == bf) amb = true;
			else if (freq > bf) { best = e; amb = false; }
		}
	}
	__device__ inline void add(uint64_t kk, uint32_t pos, uint32_t chunk, uint32_t mod, bool neigh)
	{
		push_ctx(kk, pos, mod, chunk);
		vote(pos - 32u * chunk, pos, neigh);
	}

	// a ref-dict hit: exact qv.cc:850-890; neighbour :979-1047, :1131-1171, :1228-1296
	__device__ inline void ref_hit(int64_t idx, uint64_t kk, uint32_t chunk, uint32_t mod, bool neigh)
	{
		if (idx < 0) return;
		const RefEnt e = d.ref[idx];
		if (e.pos == POS_AMBIGUOUS) return;
		if (e.amb == 0) {
			if (neigh && site_loose(d, st, e.pos + mod)) return;
			add(kk, e.pos, chunk, mod, neigh);
		} else {
			const uint32_t *row = d.ref_aux + (uint64_t)e.pos * AUX_COLS;
			st.add(S_AUX_REF, 1);
			for (int j = 0; j < AUX_COLS; j++) {
				const uint32_t p = row[j];
				if (p == 0) break;
				if (neigh && site_loose(d, st, p + mod)) continue;
				add(kk, p, chunk, mod, neigh);
			}
		}
	}
	// a SNP-dict hit: exact qv.cc:897-937; neighbour :1055-1101, :1176-1207, :1308-1352
	__device__ inline void snp_hit(int64_t idx, uint64_t kk, uint32_t chunk, uint32_t mod, bool neigh)
	{
		if (idx < 0) return;
		const SnpEnt e = d.snp[idx];
		if (e.pos == POS_AMBIGUOUS) return;
		if (((e.key >> 48) & 0xFFu) == 0) {
			if (neigh && (uint32_t)((e.key >> 43) & 0x1Fu) == mod) return;        // SNP_INFO_POS, vartype.h:47
			add(kk, e.pos, chunk, mod, neigh);
		} else {
			const uint32_t *prow = d.snp_aux_pos + (uint64_t)e.pos * AUX_COLS;
			const uint8_t *irow = d.snp_aux_info + (uint64_t)e.pos * AUX_COLS;
			st.add(S_AUX_SNP, 1);
			for (int j = 0; j < AUX_COLS; j++) {
				const uint32_t p = prow[j];
				if (p == 0) break;
				if (neigh && (uint32_t)(irow[j] >> 3) == mod) continue;
				add(kk, p, chunk, mod, neigh);
			}
		}
	}

	// ---- one chunk: src/qv.cc:834-1365 ---------------------------------------------------
	__device__ inline void do_chunk(uint64_t k, uint32_t c, bool gate_open)
	{
		st.add(S_CHUNKS, 1);
		uint32_t lo, hi, slo, shi;
		ref_hit(ref_query(d, st, k, lo, hi), k, c, NOMOD, false);                // :840, :850-890
		snp_hit(snp_query(d, st, k, slo, shi), k, c, NOMOD, false);              // :841, :897-937
		if (!gate_open) return;                                                   // :943
		st.add(S_GATE_OPEN, 1);
		const uint32_t bs = hi - lo;                                              // check_block_size :242-264
		uint32_t rsb = 64, ssb = 64;                                              // :946-956
		{
			const uint64_t rp = (uint64_t)hash32((uint32_t)k) % d.ref_bf_bits;
			const uint64_t sp = hash40(k & LO40_MASK) % d.snp_bf_bits;
			if ((d.ref_bf[rp >> 6] >> (rp & 63)) & 1u) st.add(S_REFBF_POS, 1); else rsb = 32;
			if ((d.snp_bf[sp >> 6] >> (sp & 63)) & 1u) st.add(S_SNPBF_POS, 1); else ssb = 40;
		}
		if (bs >= BLOCK_THRESHOLD) {                                              // :962-1109
			st.add(S_LARGE_BLOCK, 1);
			for (uint32_t i = 0; i < 32; i += 2) {
				const uint64_t base = (k >> i) & 3u;
				for (uint64_t j = 0; j < 4; j++) {
					if (j == base) continue;
					const uint64_t nb = (k & ~(3ull << i)) | (j << i);
					uint32_t a, b;
					const int64_t r = ref_query(d, st, nb, a, b);
					const int64_t q = snp_query(d, st, nb, a, b);
					ref_hit(r, nb, c, i >> 1, true);
					snp_hit(q, nb, c, i >> 1, true);
					if (overflow) return;
				}
			}
		} else {                                                                  // :1110-1209
			// iterate_ref_dict :316-376: TEST entry lo + 9*(i-lo), RECORD entry i          (B1)
			for (uint32_t i = lo; i < hi; i++) {
				const uint64_t t = (uint64_t)lo + (uint64_t)(i - lo) * REF_STRIDE;
				uint32_t tlo = 0;
				st.add(S_SCAN_REF, 1);
				if (t < d.n_ref) tlo = d.ref[t].lo; else st.add(S_SCAN_OOB, 1);
				const int dd = onebase((uint64_t)((uint32_t)k ^ tlo));
				if (dd >= 0) ref_hit((int64_t)i, (k & 0xFFFFFFFF00000000ull) | tlo, c, (uint32_t)dd, true);
				if (overflow) return;
			}
			// iterate_snp_dict :413-464
			for (uint32_t i = slo; i < shi; i++) {
				const uint64_t t = (uint64_t)slo + (uint64_t)(i - slo) * SNP_STRIDE;
				uint64_t tlo = 0;
				st.add(S_SCAN_SNP, 1);
				if (t < d.n_snp) tlo = d.snp[t].key & LO40_MASK; else st.add(S_SCAN_OOB, 1);
				const int dd = onebase((k & LO40_MASK) ^ tlo);
				if (dd >= 0) snp_hit((int64_t)i, (k & 0xFFFFFF0000000000ull) | tlo, c, (uint32_t)dd, true);
				if (overflow) return;
			}
		}
		for (uint32_t i = 32; i < 64; i += 2) {                                   // :1213-1365
			const uint64_t base = (k >> i) & 3u;
			for (uint64_t j = 0; j < 4; j++) {
				if (j == base) continue;
				const uint64_t nb = (k & ~(3ull << i)) | (j << i);
				uint32_t a, b;
				if (i < rsb) ref_hit(ref_query(d, st, nb, a, b), nb, c, i >> 1, true);
				if ((bs >= BLOCK_THRESHOLD || i >= 40) && i < ssb) snp_hit(snp_query(d, st, nb, a, b), nb, c, i >> 1, true);
				if (overflow) return;
			}
		}
	}

	// ---- decision + pile-up walk: src/qv.cc:1375-1502 -------------------------------------
	__device__ inline bool finish_pass()
	{
		st.add(S_PASSES, 1);
		if (best < 0 || amb) return false;
		if ((s.key_fm[(uint64_t)best * s.nlanes + lane] & 0xFFu) <= 1) return false;
		st.add(S_PASSES_OK, 1);
		const uint32_t target = s.key_index[(uint64_t)best * s.nlanes + lane];
		for (uint32_t i = 0; i < nctx; i++) {
			const uint64_t at = (uint64_t)i * s.nlanes + lane;
			const uint32_t kpos = s.ctx_kpos[at];
			const uint32_t meta = s.ctx_meta[at];
			if (kpos - 32u * (meta >> 16) != target) continue;
			walk_ctx(d, st, s.ctx_kmer[at], kpos, meta & 0xFFFFu);
		}
		return true;
	}
};

}  // namespace vg
