// vg_wave.h -- the wave-cooperative read-loop kernel (gfx950, one 64-lane wavefront per workgroup).
//
// Why not one read per lane end to end: 8 % of the 32-base chunks are "gate-open" (src/qv.cc:943) and
// need ~100 more dictionary queries each, so in a lane-per-read kernel one lane of nearly every wave
// walks a 100-300 deep chain of dependent gathers while 63 lanes wait -- the first kernel of this
// repo ran at 14 G gathers/s, 29 % of the chip's measured random-gather ceiling, for that reason.
// Here a wave keeps 64 (read, pass) jobs in flight and runs each pass in three stages:
//   A  lane-parallel   exact ref/SNP look-ups of every chunk            (src/qv.cc:840-937)
//   B  wave-parallel   for each gate-open chunk in the wave, its ~100 Hamming-1 neighbour queries /
//                      strided bucket-scan probes are dealt to the 64 lanes; accepted hits are
//                      compacted with a ballot + prefix sum into the owner's list in canonical order
//                                                                         (src/qv.cc:943-1365)
//   C  lane-parallel   the order-dependent vote is replayed per read from the two short lists, then
//                      the supporting contexts walk the pile-up            (src/qv.cc:132-178, 1375-1502)
// Lists and vote keys live in LDS ([slot][lane], conflict-free).  Neighbour contexts
// whose implied read position is not the position of any exact hit of the same pass can neither
// vote (qv.cc:134-139) nor support the winner, so stage B drops them -- the lists stay tiny.
// A job that outgrows its lists touches no counter and is handed, whole, to the next tier: the same
// kernel with deeper lists, then the generic lane machine of vg_device.h (exactness is never traded).
#pragma once
#include "vg_device.h"

namespace vg {

// List capacities per job-pass (LDS, [slot][lane]).  Two instantiations: the main tier keeps 17 waves per CU
// resident; the second tier takes the reads that spill from it (repeat regions: aux rows, many keys)
// with lists 6-8x deeper at 2 waves per CU -- still wave-parallel, so a heavy read costs a few dozen
// dependent gathers instead of the thousands the sequential lane machine needs.
constexpr int W1_ECAP = 8, W1_NCAP = 4, W1_KCAP = 4;
constexpr int W2_ECAP = 48, W2_NCAP = 48, W2_KCAP = 32;

// packed reads: chunk k-mers at [offsets[r] >> 5 ...), one flag word per read
constexpr uint64_t PK_SKIP_N = 1ull << 62;     // an N inside the trimmed read: skipped (qv.cc:815-828)
constexpr uint64_t PK_INVALID = 1ull << 63;    // another character: the reference aborts (util.c:103)
constexpr uint64_t PK_LONG = 1ull << 61;       // more than 32 chunks: generic tier

// context meta word: chunk (5) | mod (5) << 5 | neighbour flag << 10 | new base (2) << 11
__device__ inline uint32_t mk_meta(uint32_t chunk, uint32_t mod, bool neigh, uint32_t nbase) { return chunk | (mod << 5) | ((neigh ? 1u : 0u) << 10) | (nbase << 11); }

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

template <bool STATS, int W_ECAP, int W_NCAP, int W_KCAP, int WPB>
__global__ __launch_bounds__(64 * WPB) void vg_wave_kernel(DevIndex d, const uint64_t *__restrict__ pk_kmer, const uint64_t *__restrict__ pk_meta,
                                                     const uint64_t *__restrict__ offsets, uint64_t n_reads_arg,
                                                     const uint32_t *__restrict__ read_ids, const uint32_t *__restrict__ n_ids,
                                                     uint32_t *overflow_list, uint32_t *overflow_count, unsigned long long *stats)
{
	// narrow element types keep a wave at 6.5 KB of LDS (24 waves per CU): an exact context only needs its
	// chunk number next to the position, a neighbour context 13 bits, a vote key 8 + 1 bits of state
	__shared__ uint32_t E_kpos[W_ECAP][64 * WPB], N_kpos[W_NCAP][64 * WPB], K_idx[W_KCAP][64 * WPB], K_first[W_KCAP][64 * WPB];
	__shared__ uint16_t N_meta[W_NCAP][64 * WPB], K_fm[W_KCAP][64 * WPB];
	__shared__ uint8_t E_meta[W_ECAP][64 * WPB];
	const uint32_t lane = threadIdx.x & 63u;             // lane in the wave
	const uint32_t col = threadIdx.x;                    // this lane's LDS column
	const uint32_t col0 = threadIdx.x & ~63u;            // first column of this wave
	const uint64_t lane_bit = 1ull << lane;
	const uint64_t n_reads = read_ids ? (uint64_t)*n_ids : n_reads_arg;
	const uint64_t wave_id = (uint64_t)blockIdx.x * WPB + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * WPB;
	uint64_t cursor = n_reads * wave_id / n_waves;
	const uint64_t end = n_reads * (wave_id + 1) / n_waves;

	bool active = false;
	uint32_t rid = 0, n = 0, gates = 0, pass = 0;
	uint64_t slot0 = 0;
	LaneStats<STATS> tot, cur;
	tot.clear(); cur.clear();

	auto chunk_kmer = [&](uint32_t c) -> uint64_t {
		const uint64_t kf = pk_kmer[slot0 + (pass ? n - 1 - c : c)];
		return pass ? revcomp64(kf) : kf;
	};

	for (;;) {
		// ------------------------------------------------------------------ refill free lanes
		{
			const uint64_t freem = __ballot(!active);
			const uint64_t avail = end - cursor;
			if (freem && avail) {
				const uint32_t nfree = (uint32_t)__popcll(freem);
				const uint32_t take = (uint32_t)(avail < nfree ? avail : nfree);
				if (!active) {
					const uint32_t rank = (uint32_t)__popcll(freem & (lane_bit - 1));
					if (rank < take) {
						rid = read_ids ? read_ids[cursor + rank] : (uint32_t)(cursor + rank);
						const uint64_t off = offsets[rid];
						const uint64_t meta = pk_meta[rid];
						n = (uint32_t)((offsets[rid + 1] - off) >> 5);
						slot0 = off >> 5;
						gates = (uint32_t)meta;
						pass = 0;
						cur.clear();
						cur.add(S_READS, 1);
						cur.add(S_INGEST, 9 * n);
						if (meta & (PK_SKIP_N | PK_INVALID)) {
							cur.add((meta & PK_INVALID) ? S_READS_INVALID : S_READS_N, 1);
							if constexpr (STATS) for (int i = 0; i < S_COUNT; i++) tot.v[i] += cur.v[i];
						} else if (meta & PK_LONG) {
							overflow_list[atomicAdd(overflow_count, 1u)] = rid;
						} else {
							active = true;
						}
					}
				}
				cursor += take;
			}
		}
		if (!__any(active)) { if (cursor >= end) break; continue; }

		// ------------------------------------------------------------------ stage A: exact look-ups
		uint32_t ecnt = 0, ncnt = 0;
		bool ovf = false;
		if (active) {
			for (uint32_t c = 0; c < n; c++) {
				const uint64_t k = chunk_kmer(c);
				cur.add(S_CHUNKS, 1);
				uint32_t lo, hi;
				const int64_t ri = ref_query(d, cur, k, lo, hi);
				if (ri >= 0) {                                                   // qv.cc:850-890
					const RefEnt e = d.ref[ri];
					if (e.pos != POS_AMBIGUOUS) {
						if (e.amb == 0) {
							cur.add(S_CTX, 1);
							if (ecnt < W_ECAP) { E_kpos[ecnt][col] = e.pos; E_meta[ecnt][col] = (uint8_t)c; ecnt++; } else ovf = true;
						} else {
							const uint32_t *row = d.ref_aux + (uint64_t)e.pos * AUX_COLS;
							cur.add(S_AUX_REF, 1);
							for (int j = 0; j < AUX_COLS; j++) {
								const uint32_t p = row[j];
								if (p == 0) break;
								cur.add(S_CTX, 1);
								if (ecnt < W_ECAP) { E_kpos[ecnt][col] = p; E_meta[ecnt][col] = (uint8_t)c; ecnt++; } else ovf = true;
							}
						}
					}
				}
				const int64_t si = snp_query(d, cur, k, lo, hi);
				if (si >= 0) {                                                   // qv.cc:897-937
					const SnpEnt e = d.snp[si];
					if (e.pos != POS_AMBIGUOUS) {
						if (((e.key >> 48) & 0xFFu) == 0) {
							cur.add(S_CTX, 1);
							if (ecnt < W_ECAP) { E_kpos[ecnt][col] = e.pos; E_meta[ecnt][col] = (uint8_t)c; ecnt++; } else ovf = true;
						} else {
							const uint32_t *prow = d.snp_aux_pos + (uint64_t)e.pos * AUX_COLS;
							cur.add(S_AUX_SNP, 1);
							for (int j = 0; j < AUX_COLS; j++) {
								const uint32_t p = prow[j];
								if (p == 0) break;
								cur.add(S_CTX, 1);
								if (ecnt < W_ECAP) { E_kpos[ecnt][col] = p; E_meta[ecnt][col] = (uint8_t)c; ecnt++; } else ovf = true;
							}
						}
					}
				}
			}
		}
		VG_WAVE_SYNC();

		// ------------------------------------------------------------------ stage B: gate-open chunks, one at a time, 64 lanes wide
		uint32_t pend = (active && !ovf) ? (n >= 32 ? gates : (gates & ((1u << n) - 1u))) : 0u;
		for (;;) {
			const uint64_t m = __ballot(pend != 0);
			if (!m) break;
			const int owner = __ffsll((long long)m) - 1;
			uint32_t c = 0, lo = 0, hi = 0, slo = 0, shi = 0, fl = 0, klo = 0, khi = 0;
			if ((int)lane == owner) {
				c = (uint32_t)__ffs((int)pend) - 1;
				const uint64_t k = chunk_kmer(c);
				klo = (uint32_t)k; khi = (uint32_t)(k >> 32);
				jg_pair(d.ref_jg, k >> 32, lo, hi);                              // check_block_size, qv.cc:242-264
				jg_pair(d.snp_jg, k >> 40, slo, shi);
				const uint64_t rp = (uint64_t)hash32((uint32_t)k) % d.ref_bf_bits;   // qv.cc:946-956
				const uint64_t sp = hash40(k & LO40_MASK) % d.snp_bf_bits;
				if ((d.ref_bf[rp >> 6] >> (rp & 63)) & 1u) fl |= 1u;
				if ((d.snp_bf[sp >> 6] >> (sp & 63)) & 1u) fl |= 2u;
				cur.add(S_GATE_OPEN, 1);
				cur.add(S_REFBF_POS, fl & 1u);
				cur.add(S_SNPBF_POS, (fl >> 1) & 1u);
				if (hi - lo >= BLOCK_THRESHOLD) cur.add(S_LARGE_BLOCK, 1);
			}
			c = __shfl(c, owner); lo = __shfl(lo, owner); hi = __shfl(hi, owner); slo = __shfl(slo, owner); shi = __shfl(shi, owner);
			fl = __shfl(fl, owner); klo = __shfl(klo, owner); khi = __shfl(khi, owner);
			const uint64_t k = ((uint64_t)khi << 32) | klo;
			const uint32_t o_ecnt = __shfl(ecnt, owner);
			uint32_t wcnt = __shfl(ncnt, owner);
			bool wovf = false;
			const bool large = hi - lo >= BLOCK_THRESHOLD;
			const uint32_t Lr = large ? 48u : hi - lo, Ls = large ? 0u : shi - slo;
			const uint32_t L = Lr + Ls, total = L + 48u;
			const uint32_t rsb = (fl & 1u) ? 64u : 32u, ssb = (fl & 2u) ? 64u : 40u;
			LaneStats<STATS> hs;
			hs.clear();

			// is `position` the implied read position of one of the owner's exact hits?
			auto in_keys = [&](uint32_t position) -> bool {
				bool f = false;
				for (uint32_t e = 0; e < o_ecnt; e++) f |= (E_kpos[e][col0 + owner] - 32u * (E_meta[e][col0 + owner] & 31u)) == position;
				return f;
			};

			for (uint32_t t0 = 0; t0 < total && !wovf; t0 += 64) {
				const uint32_t t = t0 + lane;
				int64_t ri = -1, si = -1;
				uint32_t mod = 0, nbase = 0;
				if (t < total) {
					if (t < L) {
						if (large) {                                             // qv.cc:962-1109
							const uint32_t pair = t / 3, sel = t % 3, base = (uint32_t)(k >> (2 * pair)) & 3u;
							nbase = sel + (sel >= base ? 1u : 0u);
							mod = pair;
							const uint64_t nb = (k & ~(3ull << (2 * pair))) | ((uint64_t)nbase << (2 * pair));
							uint32_t a, b;
							ri = ref_query(d, hs, nb, a, b);
							si = snp_query(d, hs, nb, a, b);
						} else if (t < Lr) {                                     // iterate_ref_dict, qv.cc:316-376   (B1)
							const uint64_t tt = (uint64_t)lo + (uint64_t)t * REF_STRIDE;
							uint32_t tlo = 0;
							hs.add(S_SCAN_REF, 1);
							if (tt < d.n_ref) tlo = d.ref[tt].lo; else hs.add(S_SCAN_OOB, 1);
							const int dd = onebase((uint64_t)(klo ^ tlo));
							if (dd >= 0) { ri = (int64_t)lo + t; mod = (uint32_t)dd; nbase = (tlo >> (2 * dd)) & 3u; }
						} else {                                                 // iterate_snp_dict, qv.cc:413-464   (B1)
							const uint32_t u = t - Lr;
							const uint64_t tt = (uint64_t)slo + (uint64_t)u * SNP_STRIDE;
							uint64_t tlo = 0;
							hs.add(S_SCAN_SNP, 1);
							if (tt < d.n_snp) tlo = d.snp[tt].key & LO40_MASK; else hs.add(S_SCAN_OOB, 1);
							const int dd = onebase((k & LO40_MASK) ^ tlo);
							if (dd >= 0) { si = (int64_t)slo + u; mod = (uint32_t)dd; nbase = (uint32_t)(tlo >> (2 * dd)) & 3u; }
						}
					} else {                                                     // qv.cc:1213-1365
						const uint32_t u = t - L, pair = 16u + u / 3, sel = u % 3, base = (uint32_t)(k >> (2 * pair)) & 3u;
						nbase = sel + (sel >= base ? 1u : 0u);
						mod = pair;
						const uint64_t nb = (k & ~(3ull << (2 * pair))) | ((uint64_t)nbase << (2 * pair));
						uint32_t a, b;
						if (2 * pair < rsb) ri = ref_query(d, hs, nb, a, b);
						if ((large || 2 * pair >= 40u) && 2 * pair < ssb) si = snp_query(d, hs, nb, a, b);
					}
				}
				// acceptance (site / SNP-base tests) + key filter -> bit j: ref candidate j kept, bit 10+j: snp candidate j kept
				uint32_t keepm = 0;
				RefEnt re{}; SnpEnt se{};
				if (ri >= 0) {
					re = d.ref[ri];
					if (re.pos != POS_AMBIGUOUS) {
						if (re.amb == 0) {
							if (!site_loose(d, hs, re.pos + mod)) { hs.add(S_CTX, 1); if (in_keys(re.pos - 32u * c)) keepm |= 1u; }
						} else {
							const uint32_t *row = d.ref_aux + (uint64_t)re.pos * AUX_COLS;
							hs.add(S_AUX_REF, 1);
							for (int j = 0; j < AUX_COLS; j++) {
								const uint32_t p = row[j];
								if (p == 0) break;
								if (site_loose(d, hs, p + mod)) continue;
								hs.add(S_CTX, 1);
								if (in_keys(p - 32u * c)) keepm |= 1u << j;
							}
						}
					}
				}
				if (si >= 0) {
					se = d.snp[si];
					if (se.pos != POS_AMBIGUOUS) {
						if (((se.key >> 48) & 0xFFu) == 0) {
							if ((uint32_t)((se.key >> 43) & 0x1Fu) != mod) { hs.add(S_CTX, 1); if (in_keys(se.pos - 32u * c)) keepm |= 1u << 10; }
						} else {
							const uint32_t *prow = d.snp_aux_pos + (uint64_t)se.pos * AUX_COLS;
							const uint8_t *irow = d.snp_aux_info + (uint64_t)se.pos * AUX_COLS;
							hs.add(S_AUX_SNP, 1);
							for (int j = 0; j < AUX_COLS; j++) {
								const uint32_t p = prow[j];
								if (p == 0) break;
								if ((uint32_t)(irow[j] >> 3) == mod) continue;
								hs.add(S_CTX, 1);
								if (in_keys(p - 32u * c)) keepm |= 1u << (10 + j);
							}
						}
					}
				}
				if (!__any(keepm != 0)) continue;
				// canonical order inside the round = lane order (item order); ref contexts of an item before its SNP contexts
				const uint32_t keep = (uint32_t)__popc(keepm);
				uint32_t incl = keep;
				for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o); if ((int)lane >= o) incl += y; }
				const uint32_t round_total = __shfl(incl, 63);
				if (wcnt + round_total > (uint32_t)W_NCAP) { wovf = true; break; }
				uint32_t at = wcnt + incl - keep;
				if (keepm & 0x3FFu) {
					if (re.amb == 0) { N_kpos[at][col0 + owner] = re.pos; N_meta[at][col0 + owner] = (uint16_t)mk_meta(c, mod, true, nbase); at++; }
					else {
						const uint32_t *row = d.ref_aux + (uint64_t)re.pos * AUX_COLS;
						for (int j = 0; j < AUX_COLS; j++) if (keepm & (1u << j)) { N_kpos[at][col0 + owner] = row[j]; N_meta[at][col0 + owner] = (uint16_t)mk_meta(c, mod, true, nbase); at++; }
					}
				}
				if (keepm >> 10) {
					if (((se.key >> 48) & 0xFFu) == 0) { N_kpos[at][col0 + owner] = se.pos; N_meta[at][col0 + owner] = (uint16_t)mk_meta(c, mod, true, nbase); at++; }
					else {
						const uint32_t *prow = d.snp_aux_pos + (uint64_t)se.pos * AUX_COLS;
						for (int j = 0; j < AUX_COLS; j++) if (keepm & (1u << (10 + j))) { N_kpos[at][col0 + owner] = prow[j]; N_meta[at][col0 + owner] = (uint16_t)mk_meta(c, mod, true, nbase); at++; }
					}
				}
				wcnt += round_total;
			}
			if constexpr (STATS) {
				const int ids[] = {S_REF_QUERY, S_SNP_QUERY, S_REF_PROBE, S_SNP_PROBE, S_SCAN_REF, S_SCAN_SNP, S_SCAN_OOB, S_AUX_REF, S_AUX_SNP, S_SITE_TEST, S_CTX};
				for (int id : ids) { const uint32_t sum = wave_sum<STATS>(hs.v[id]); if ((int)lane == owner) cur.v[id] += sum; }
			}
			if ((int)lane == owner) {
				pend &= ~(1u << c);
				ncnt = wcnt;
				if (wovf) { ovf = true; pend = 0; }
			}
		}
		VG_WAVE_SYNC();

		// ------------------------------------------------------------------ stage C: replay the vote, walk the pile-up
		if (active) {
			bool processed = false;
			if (!ovf) {
				uint32_t nkeys = 0;
				int best = -1; bool amb = false;
				// improved_index_table_add, qv.cc:132-178; keys in this lane's LDS column
				auto vote = [&](uint32_t index, uint32_t kpos, bool neigh) {
					int e = -1;
					for (uint32_t i = 0; i < nkeys; i++) if (K_idx[i][col] == index) { e = (int)i; break; }
					uint32_t first, fm;
					if (e < 0) {
						if (neigh) return;
						if (nkeys >= (uint32_t)W_KCAP) { ovf = true; return; }
						e = (int)nkeys++;
						K_idx[e][col] = index; K_first[e][col] = first = kpos; fm = 0;
					} else { first = K_first[e][col]; fm = K_fm[e][col]; }
					const uint32_t freq = (fm + 1) & 0xFFu;
					const uint32_t multi = (fm >> 8) | (kpos != first ? 1u : 0u);
					K_fm[e][col] = (uint16_t)(freq | (multi << 8));
					if (!multi) return;
					if (best < 0) { best = e; amb = false; }
					else if (e == best) amb = false;
					else {
						const uint32_t bf = K_fm[best][col] & 0xFFu;
						if (freq == bf) amb = true;
						else if (freq > bf) { best = e; amb = false; }
					}
				};
				uint32_t ei = 0, ni = 0;
				for (uint32_t c = 0; c < n && !ovf; c++) {
					while (ei < ecnt && (E_meta[ei][col] & 31u) == c) { const uint32_t p = E_kpos[ei][col]; vote(p - 32u * c, p, false); ei++; }
					while (ni < ncnt && (N_meta[ni][col] & 31u) == c) { const uint32_t p = N_kpos[ni][col]; vote(p - 32u * c, p, true); ni++; }
				}
				if (!ovf) {
					cur.add(S_PASSES, 1);
					const uint32_t bfm = best >= 0 ? K_fm[best][col] : 0u, target = best >= 0 ? K_idx[best][col] : 0u;
					processed = best >= 0 && !amb && (bfm & 0xFFu) > 1;          // qv.cc:1375
					if (processed) {
						cur.add(S_PASSES_OK, 1);
						for (uint32_t i = 0; i < ecnt; i++) {
							const uint32_t p = E_kpos[i][col], c = E_meta[i][col] & 31u;
							if (p - 32u * c == target) walk_ctx(d, cur, chunk_kmer(c), p, NOMOD);
						}
						for (uint32_t i = 0; i < ncnt; i++) {
							const uint32_t p = N_kpos[i][col], mt = N_meta[i][col], c = mt & 31u, mod = (mt >> 5) & 31u;
							if (p - 32u * c != target) continue;
							const uint64_t kk = (chunk_kmer(c) & ~(3ull << (2 * mod))) | ((uint64_t)((mt >> 11) & 3u) << (2 * mod));
							walk_ctx(d, cur, kk, p, mod);
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
	}
	if constexpr (STATS) {
		for (int i = 0; i < S_COUNT; i++) if (tot.v[i]) atomicAdd(&stats[i], (unsigned long long)tot.v[i]);
	}
}

}  // namespace vg
