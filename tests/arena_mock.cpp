// tests/arena_mock.cpp -- TEST ONLY (compiled and run by tests/test_abi_and_layout.py; never part of the product).
// The index handle's device-memory arena (vargeno_amd/csrc/vg_arena.h: one block, permanent arrays from the bottom, temporaries
// from the top; the library instantiates it with hipMalloc / hipFree) against a mock block:
//   fuzz <seed>     random takes and gives; a shadow model checks that live allocations never overlap, lie inside the block, are
//                   aligned, that a refusal only happens when no free gap could have held the request, and that the block is
//                   returned exactly once
//   replay          the loader's own sequence of allocations (vargeno_hip.hip, build_on_device, in its order) with the array
//                   sizes of BASELINE.json's configurations: in a block the size of the FINISHED index every request must fit --
//                   that is the property the construction order was chosen for
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <random>
#include <string>
#include <vector>

#include "../vargeno_amd/csrc/vg_arena.h"

static int errors = 0;
static void complain(const std::string &what) { fprintf(stderr, "mock: %s\n", what.c_str()); errors++; }

struct MockBlock {
	static int allocs, frees;
	static uint64_t last_bytes;
	static void *alloc(uint64_t bytes) { allocs++; last_bytes = bytes; return (void *)0x100000000000ull; }
	static void free(void *p) { if (p != (void *)0x100000000000ull) complain("free of another pointer"); frees++; }
};
int MockBlock::allocs = 0, MockBlock::frees = 0;
uint64_t MockBlock::last_bytes = 0;
typedef vg::DevArenaT<MockBlock> Arena;
static uint8_t *const BASE = (uint8_t *)0x100000000000ull;

static int fuzz(unsigned seed)
{
	std::mt19937_64 rng(seed);
	const uint64_t S = (64ull + rng() % 256) << 30;
	{
		Arena a;
		if (!a.init(S)) { complain("init failed"); return 1; }
		// every third seed: temporaries only in the upper part of the block (set_temp_floor: what a budget-limited index does with
		// the floor at the block's end -- here half-way, so that both kinds of request keep being served)
		const uint64_t floor = seed % 3 == 0 ? a.size() / 2 : 0;
		a.set_temp_floor(floor);
		std::map<uint64_t, uint64_t> live;                       // offset -> bytes
		for (int step = 0; step < 20000; step++) {
			if (rng() % 100 < 55 || live.empty()) {
				uint64_t bytes;
				const unsigned k = (unsigned)(rng() % 10);
				if (k < 3) bytes = 1 + rng() % 4096;
				else if (k < 8) bytes = (1ull << 20) * (1 + rng() % 8192);
				else bytes = (1ull << 30) * (1 + rng() % 48);
				const bool temp = rng() % 3 != 0;
				uint8_t *p = (uint8_t *)a.take(bytes, temp);
				if (!p) {
					// was there really no gap for it?  (gaps of the shadow model, with 2 MiB of slack for the alignment)
					// (a temporary may only use the part of a gap at or above the floor)
					const uint64_t lo_ok = temp ? floor : 0;
					uint64_t prev = 0, best = 0;
					auto gap = [&](uint64_t g0, uint64_t g1) { if (g0 < lo_ok) g0 = lo_ok; if (g1 > g0 && g1 - g0 > best) best = g1 - g0; };
					for (auto &kv : live) { gap(prev, kv.first); prev = kv.first + kv.second; }
					gap(prev, a.size());
					if (best >= bytes + (4ull << 20)) complain("a request was refused although a gap could hold it");
					continue;
				}
				const uint64_t at = (uint64_t)(p - BASE), rounded = (bytes + 255) / 256 * 256;
				if (at + rounded > a.size()) complain("allocation outside the block");
				if (temp && at < floor) complain("a temporary below the floor");
				if (at % 256) complain("allocation not 256-byte aligned");
				if (bytes >= (2ull << 20) && at % (2ull << 20)) complain("large allocation not 2 MiB aligned");
				auto nx = live.lower_bound(at);
				if (nx != live.end() && nx->first < at + rounded) complain("overlap with the allocation above");
				if (nx != live.begin()) { auto pv = std::prev(nx); if (pv->first + pv->second > at) complain("overlap with the allocation below"); }
				live[at] = rounded;
			} else {
				auto it = live.begin();
				std::advance(it, (long)(rng() % live.size()));
				if (!a.give(BASE + it->first)) complain("give refused a live allocation");
				if (a.give(BASE + it->first)) complain("give accepted the same allocation twice");
				live.erase(it);
			}
			uint64_t sum = 0;
			if (step % 64 == 0) { for (auto &kv : live) sum += kv.second; if (sum != a.in_use()) complain("in_use() disagrees with the shadow model"); }
		}
		// everything back: the block must be one free gap again
		for (auto &kv : live) a.give(BASE + kv.first);
		if (a.in_use() != 0) complain("bytes in use after everything was given back");
		if (!a.take(a.size(), false)) complain("the emptied block is fragmented: it cannot be taken whole");
	}
	if (MockBlock::allocs != 1 || MockBlock::frees != 1) complain("the block was not taken / returned exactly once");
	return 0;
}

// the loader's sequence (vargeno_hip.hip: open_impl + build_on_device) for an index of n reference k-mers, m SNP k-mers, plen genome
// positions; mx: the layout with merged view + direct table, else the one with the paired HI32 table
static uint64_t replay(const char *name, uint64_t n, uint64_t m, uint64_t aux_r, uint64_t aux_s, uint64_t plen, uint64_t sites, bool mx, uint64_t S)
{
	Arena a;
	a.init(S);
	std::map<std::string, void *> h;
	uint64_t misses = 0;
	auto P = [&](const char *w, uint64_t b) { void *p = a.take(b, false); if (!p) { misses++; if (S < (1ull << 50)) complain(std::string(name) + ": no room for the permanent array " + w); } h[w] = p; };
	auto T = [&](const char *w, uint64_t b) { void *p = a.take(b, true); if (!p) { misses++; if (S < (1ull << 50)) complain(std::string(name) + ": no room for the temporary " + w); } h[w] = p; };
	auto g = [&](const char *w) { if (h[w]) a.give(h[w]); h[w] = nullptr; };
	// (r06: tables that scale with the index -- plan_views' table_bits_for: the power of two at or above the entry count, 2^32 from 2^31 entries on)
	auto tbits = [](uint64_t entries) { uint64_t b = 16; while (b < 32 && (1ull << b) < entries) b++; return b >= 31 ? 32ull : b; };
	const uint64_t Jref = ((1ull << tbits(n)) + 1) * 4;
	T("ref_kmer", 8 * n); T("ref_pos", 4 * n); T("ref_amb", n); T("snp_kmer", 8 * m); T("snp_pos", 4 * m);
	T("snp_info", m); T("snp_amb", m); T("snp_rf", m); T("snp_af", m);
	P("ref_aux", 40 * aux_r); P("snp_aux_pos", 40 * aux_s); P("snp_aux_info", 10 * aux_s);
	T("raw", 13 * n + 40 * aux_r + 64); g("raw"); T("raw", 16 * m + 78 * aux_s + 64); g("raw");
	T("chk", 32); g("chk");
	P("ref_bf", 1ull << 29); P("snp_bf", 140000000);
	T("winner", 4 * plen); T("blk", plen / 8); P("pile", plen); P("srank", plen / 4); T("scan_tmp", 1 << 20);
	T("s_pos", 4 * sites); T("s_ref", sites); T("s_alt", sites); T("s_rf", sites); T("s_af", sites); P("site_ba", sites); P("cnt4", 16 * sites);
	g("winner"); g("blk"); g("scan_tmp"); g("s_pos"); g("s_ref"); g("s_alt"); g("s_rf"); g("s_af"); g("snp_rf"); g("snp_af");
	P("snp_jg", ((1ull << 24) + 1) * 4); P("snp", 16 * m); g("snp_pos"); g("snp_info"); g("snp_amb"); P("snp_sig", 2 * (m + 16));
	{	// r06: LO32-ordered view of the SNP dictionary (sorted while the SNP k-mers are still there)
		T("ska", 8 * m); T("sva", 4 * m); T("skb", 8 * m); T("svb", 4 * m); T("sort_tmp", 64ull << 20); g("sort_tmp"); g("skb"); g("svb");
		uint64_t sb = 14; while (sb < 30 && (1ull << sb) < m) sb++;
		P("ssec_jg", ((1ull << sb) + 1) * 4); P("ssec3", 12 * m + 16); g("ska"); g("sva");
	}
	if (!mx) g("snp_kmer");
	if (mx) P("ref_jg", Jref);
	P("ref", 16 * n); g("ref_pos"); g("ref_amb");
	T("ka", 8 * n); T("va", 4 * n);
	if (!mx) g("ref_kmer");
	T("kb", 8 * n); T("vb", 4 * n); T("sort_tmp", 64ull << 20); g("sort_tmp"); g("kb"); g("vb");
	uint64_t bits = 14; while (bits < 30 && (1ull << bits) < n) bits++;
	P("sec_jg", ((1ull << bits) + 1) * 4); P("sec3", 12 * n + 16); T("chk", 24); g("chk"); g("ka"); g("va");
	if (!mx) P("hx", ((1ull << 32) + 1) * 16);
	else {
		const uint64_t nm = n + m;
		T("ka", 8 * nm); g("ref_kmer"); g("snp_kmer"); T("va", 4 * nm); T("strand", nm / 8 + 8); T("kb", 8 * nm); T("vb", 4 * nm); T("sort_tmp", 64ull << 20); g("sort_tmp"); g("kb"); g("vb");
		P("mx", 16 * nm); g("ka"); g("va"); g("strand");
		T("big", 4); P("dx", (1ull << tbits(nm)) * 16); g("big");
	}
	P("scratch_mid", 2048ull * 256 * (64 * 16 + 32 * 12)); P("scratch_big", 4096ull * (16384 * 16 + 2048 * 12));
	const uint64_t used = a.in_use();
	if (S < (1ull << 50)) printf("replay %-14s block %.1f GB, finished index %.1f GB, at most %.1f GB alive, %llu requests without room\n", name, S / 1e9, used / 1e9, a.peak() / 1e9, (unsigned long long)misses);
	return used;
}

int main(int argc, char **argv)
{
	if (argc > 2 && !strcmp(argv[1], "fuzz")) fuzz((unsigned)atoi(argv[2]));
	else {
		struct Cfg { const char *name; uint64_t n, m, aux_r, aux_s, plen, sites; bool mx; };
		const Cfg cfgs[] = {
			{"configs[2]", 2900000000ull, 320000000ull, 25000000ull, 500000ull, 3100000000ull, 10000000ull, true},
			{"configs[4]", 2900000000ull, 3200000000ull, 25000000ull, 5000000ull, 3100000000ull, 100000000ull, false},
			{"configs[1]", 40000000ull, 32000000ull, 300000ull, 5000ull, 40000000ull, 1000000ull, true},
			{"F-tiny", 2200000ull, 9600ull, 100ull, 10ull, 2000000ull, 300ull, true},
			{"F-tiny, hx", 2200000ull, 9600ull, 100ull, 10ull, 2000000ull, 300ull, false},
		};
		for (const Cfg &c : cfgs) {
			// the finished index's size: what the sequence leaves alive in a block with room to spare; then the same sequence in a
			// block of that size + 64 MiB of alignment slack (what plan_views gives the arena)
			const uint64_t fin = replay(c.name, c.n, c.m, c.aux_r, c.aux_s, c.plen, c.sites, c.mx, 1ull << 52);
			replay(c.name, c.n, c.m, c.aux_r, c.aux_s, c.plen, c.sites, c.mx, fin + (64ull << 20));
		}
	}
	if (errors) { fprintf(stderr, "%d complaints\n", errors); return 1; }
	printf("ok\n");
	return 0;
}
