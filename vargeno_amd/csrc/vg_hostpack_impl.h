// vg_hostpack_impl.h -- what the two builds of the host packer's hot loops (vg_hostpack_impl.inc) share with vg_hostpack.cpp.
#pragma once
#include <immintrin.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "vg_hostpack.h"

namespace vgp {

constexpr uint64_t MAX_LINE = 1023;          // fgets(buf, 1024, f): at most 1023 characters per call, newline included (qv.cc:700, 760-763)

// what one thread framed -- its own cache lines (the counters move with every record), plain arrays that only ever grow
struct alignas(128) Out {
	uint64_t *kmers = nullptr, *meta = nullptr;
	uint8_t *nch = nullptr;
	uint64_t kn = 0, kcap = 0, rn = 0, rcap = 0;
	uint64_t n_invalid = 0;
	uint64_t last_rec = 0;                 // start (in the aligned text) of the last record framed here
	uint64_t stop_at = ~0ull;              // start of a record this thread found incomplete (the tail begins there)
	bool bad = false;
	Out() = default;
	Out(const Out &) = delete;                  // owns its arrays: never copied (a vector of them is sized once)
	Out &operator=(const Out &) = delete;
	void clear() { kn = rn = 0; n_invalid = 0; last_rec = 0; stop_at = ~0ull; bad = false; }
	void grow_k(uint64_t need)
	{
		uint64_t cap = kcap ? kcap * 2 : 1 << 16;
		while (cap < kn + need) cap *= 2;
		kmers = (uint64_t *)realloc(kmers, cap * 8);
		if (!kmers) abort();
		kcap = cap;
	}
	void grow_r()
	{
		const uint64_t cap = rcap ? rcap * 2 : 1 << 14;
		meta = (uint64_t *)realloc(meta, cap * 8);
		nch = (uint8_t *)realloc(nch, cap);
		if (!meta || !nch) abort();
		rcap = cap;
	}
	~Out() { free(kmers); free(meta); free(nch); }
};

enum Framed { REC_OK, REC_INCOMPLETE, REC_BAD };

struct Kernels {                             // one build of the hot loops
	uint64_t (*count_newlines)(const uint8_t *p, uint64_t n);
	void (*frame_piece)(const uint8_t *a, uint64_t N, uint64_t b0, uint64_t b1, uint64_t g0, bool first_piece, Out &o);
	void (*frame_piece_guess)(const uint8_t *a, uint64_t N, uint64_t b0, uint64_t b1, bool first_piece, Out &o, uint64_t &start, uint64_t &end);
	Framed (*frame_record)(const uint8_t *a, uint64_t N, uint64_t r, uint64_t *next, Out &o);
	const char *name;
};
const Kernels &kernels_base();
const Kernels &kernels_avx2();               // defined in vg_hostpack_avx2.cpp (compiled with -mavx2 -mbmi2); only called when the CPU has both

}  // namespace vgp
