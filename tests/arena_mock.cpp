// tests/arena_mock.cpp -- TEST ONLY (compiled and run by tests/test_abi_and_layout.py; never part of the product).
// The bookkeeping of the index handle's device-memory arena (vargeno_amd/csrc/vg_arena.h, which the library instantiates with
// the HIP virtual-memory calls) run against a mock of those calls: "physical chunks" are numbered handles, the mock keeps the
// table of what is mapped where and complains about everything the driver would refuse (mapping over a mapping, mapping one
// chunk twice, unmapping what is not mapped, touching addresses outside the reservation) and about leaks.  On top of it a
// shadow model replays the loader's pattern -- permanent arrays growing from the bottom, temporaries of 1 KiB .. 64 GiB taken and
// given back in any order -- and checks after every step that live allocations do not overlap, that every byte of them is backed
// by a mapped chunk, that chunks under nothing are back in the pool, and that an out-of-memory device leaves the arena consistent.
//   usage: arena_mock <seed> [device GiB, default 288]            prints "ok ..." or complaints
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <random>
#include <set>
#include <vector>

#include "../vargeno_amd/csrc/vg_arena.h"

static int errors = 0;
static void complain(const char *what) { fprintf(stderr, "mock: %s\n", what); errors++; }

struct MockVm {
	typedef long Handle;
	static uint8_t *va; static uint64_t va_bytes;
	static std::set<long> alive;                       // chunks the "driver" has handed out
	static std::map<uint64_t, long> mapped;            // offset in the reservation -> chunk
	static std::set<long> in_use;                      // chunks currently mapped somewhere
	static long next_handle; static uint64_t device_chunks, chunk_bytes;
	static bool granularity_ok(int, uint64_t chunk) { chunk_bytes = chunk; return true; }
	static uint8_t *reserve(uint64_t bytes) { va = (uint8_t *)0x100000000000ull; va_bytes = bytes; return va; }
	static void unreserve(uint8_t *p, uint64_t bytes) { if (p != va || bytes != va_bytes) complain("unreserve of another range"); if (!mapped.empty()) complain("address range freed with chunks still mapped"); va = nullptr; }
	static bool create(int, uint64_t bytes, Handle *h)
	{
		if (bytes != chunk_bytes) complain("chunk of another size");
		if (alive.size() >= device_chunks) return false;                  // the device is full
		*h = ++next_handle; alive.insert(*h);
		return true;
	}
	static void release(Handle h) { if (!alive.erase(h)) complain("release of a chunk that is not alive"); if (in_use.count(h)) complain("release of a mapped chunk"); }
	static bool map(uint8_t *at, uint64_t bytes, Handle h, int)
	{
		const uint64_t off = (uint64_t)(at - va);
		if (at < va || off + bytes > va_bytes || off % chunk_bytes || bytes != chunk_bytes) { complain("map outside the reservation / off the chunk grid"); return false; }
		if (mapped.count(off)) { complain("map over a mapping"); return false; }
		if (!alive.count(h)) { complain("map of a dead chunk"); return false; }
		if (!in_use.insert(h).second) { complain("one chunk mapped twice"); return false; }
		mapped[off] = h;
		return true;
	}
	static void unmap(uint8_t *at, uint64_t)
	{
		const uint64_t off = (uint64_t)(at - va);
		auto it = mapped.find(off);
		if (it == mapped.end()) { complain("unmap of an address that is not mapped"); return; }
		in_use.erase(it->second);
		mapped.erase(it);
	}
};
uint8_t *MockVm::va = nullptr; uint64_t MockVm::va_bytes = 0;
std::set<long> MockVm::alive; std::map<uint64_t, long> MockVm::mapped; std::set<long> MockVm::in_use;
long MockVm::next_handle = 0; uint64_t MockVm::device_chunks = 0, MockVm::chunk_bytes = 0;

typedef vg::DevArenaT<MockVm> Arena;

struct Live { uint64_t at, bytes; bool temp; };

static void check(const Arena &a, const std::vector<Live> &live)
{
	// no two live allocations overlap; all of them lie on mapped chunks; nothing else is mapped
	std::map<uint64_t, uint64_t> spans;
	std::set<uint64_t> needed;
	for (const Live &l : live) {
		spans[l.at] = l.bytes;
		for (uint64_t s = l.at / Arena::CHUNK; s <= (l.at + l.bytes - 1) / Arena::CHUNK; s++) needed.insert(s * Arena::CHUNK);
	}
	uint64_t end = 0;
	for (auto &kv : spans) { if (kv.first < end) complain("two live allocations overlap"); end = kv.first + kv.second; }
	for (uint64_t off : needed) if (!MockVm::mapped.count(off)) complain("a live allocation lies on an unmapped chunk");
	if (MockVm::mapped.size() != needed.size()) complain("a chunk is mapped under nothing");
	if (a.mapped_bytes() != needed.size() * Arena::CHUNK) complain("mapped_bytes() disagrees with the driver's table");
	if (a.held_bytes() != MockVm::alive.size() * Arena::CHUNK) complain("held_bytes() disagrees with the chunks alive");
}

int main(int argc, char **argv)
{
	const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
	const uint64_t dev_gib = argc > 2 ? (uint64_t)atoll(argv[2]) : 288;
	std::mt19937_64 rng(seed);
	MockVm::device_chunks = dev_gib;
	uint64_t served = 0, refused = 0, peak_live = 0;
	{
		Arena a;
		if (!a.init(0, dev_gib << 30)) { complain("init failed"); return 1; }
		std::vector<Live> live;
		uint64_t live_bytes = 0;
		for (int step = 0; step < 4000; step++) {
			const unsigned what = (unsigned)(rng() % 100);
			if (what < 55 || live.empty()) {
				// sizes like the loader's: mostly large (GiB scale), some tiny
				uint64_t bytes;
				const unsigned k = (unsigned)(rng() % 10);
				if (k < 3) bytes = 1 + rng() % 4096;
				else if (k < 8) bytes = (1ull << 20) * (1 + rng() % 8192);
				else bytes = (1ull << 30) * (8 + rng() % 57);
				const bool temp = rng() % 4 != 0;
				uint8_t *p = (uint8_t *)a.take(bytes, temp);
				if (!p) {
					refused++;
					// a refusal must come from the device being full (or the permanent half's address space), and must leave everything as it was
					if (MockVm::alive.size() + (bytes + Arena::CHUNK - 1) / Arena::CHUNK + 1 < MockVm::device_chunks && live_bytes + bytes < (dev_gib << 30) / 2 && temp) complain("a temporary was refused although the device had room");
				} else {
					served++;
					if (!a.owns(p)) complain("take returned an address outside the reservation");
					if ((uint64_t)(p - MockVm::va) % 256) complain("allocation not 256-byte aligned");
					if (bytes >= (2ull << 20) && (uint64_t)(p - MockVm::va) % (2ull << 20)) complain("large allocation not 2 MiB aligned");
					live.push_back(Live{(uint64_t)(p - MockVm::va), bytes, temp});
					live_bytes += bytes;
					if (live_bytes > peak_live) peak_live = live_bytes;
				}
			} else {
				const size_t i = (size_t)(rng() % live.size());
				if (!a.give(MockVm::va + live[i].at)) complain("give refused a live allocation");
				if (a.give(MockVm::va + live[i].at)) complain("give accepted the same allocation twice");
				live_bytes -= live[i].bytes;
				live[i] = live.back(); live.pop_back();
			}
			if (step % 16 == 0) check(a, live);
		}
		check(a, live);
		// the end of construction: temporaries go back, the pool is returned, permanent arrays stay
		for (size_t i = 0; i < live.size();) if (live[i].temp) { a.give(MockVm::va + live[i].at); live[i] = live.back(); live.pop_back(); } else i++;
		a.trim();
		check(a, live);
		if (MockVm::alive.size() != MockVm::mapped.size()) complain("chunks left in the pool after trim");
		if (a.peak_bytes() > (dev_gib << 30)) complain("peak above the device");
	}
	// the arena is gone: nothing may be left with the driver
	if (!MockVm::alive.empty() || !MockVm::mapped.empty()) complain("the arena's destructor leaked chunks or mappings");
	if (errors) { fprintf(stderr, "%d complaints\n", errors); return 1; }
	printf("ok served %llu refused %llu peak_live_GiB %.1f\n", (unsigned long long)served, (unsigned long long)refused, peak_live / 1073741824.0);
	return 0;
}
