// line_probe.hip -- measurement aid (not part of the product path): how many bytes does ONE L2 miss of a random 8-byte
// gather move on gfx950 -- a 64-byte half or the whole 128-byte line?  The read loop's roofline hangs on the answer
// (DESIGN.md §7): FETCH_SIZE tallies 64 bytes per request whatever its size.
//
// Every variant is a kernel of its own name, so `rocprofv3 --pmc ...` reports its counters per variant:
//   one       each lane gathers 8 bytes at random 128-byte-aligned addresses A                      (baseline)
//   half      ... and also A + 64: the other half of the same line, an independent load of the same lane
//   near      ... and also A + 32: the same 64-byte half
//   two       ... and also B: another random line
//   xwave     workgroup b gathers A, workgroup b + 8 (same XCD under round-robin dispatch, another CU) the same sequence + 64,
//             both in one launch
//   seq_a / seq_b   two LAUNCHES over a set of lines that fits every L2 (0.5 MiB per XCD): seq_a gathers A, seq_b A + 64 from
//             the same workgroup numbers (same XCD): is the second half still a miss?
// If a miss moves 128 bytes, `half` runs at the LINE rate of `one` (and seq_b / the partner of xwave hit in L2); if it moves
// 64 bytes, `half` runs at the LOAD rate of `two`.
// Prints one JSON line per variant: loads, distinct 128-byte lines, ms, G loads/s, G lines/s.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

// MODE 0 one, 1 half (+64), 2 near (+32), 3 two (another line), 4 xwave (group b + 8 reads + 64), 5 seq (plain, offset `delta`)
template <int MODE>
__global__ __launch_bounds__(256) void probe(const uint8_t *__restrict__ tab, uint64_t lines, int iters, uint32_t delta, uint64_t *out)
{
	uint32_t blk = blockIdx.x;
	uint32_t extra = delta;
	if constexpr (MODE == 4) { extra = ((blk >> 3) & 1u) ? 64u : 0u; blk = (blk & 7u) | ((blk >> 4) << 3); }   // groups b and b + 8 share a sequence
	const uint64_t gid = (uint64_t)blk * blockDim.x + threadIdx.x;
	uint64_t acc = 0, st = mix(gid + 1);
	for (int i = 0; i < iters; i++) {
		const uint64_t a = ((st >> 11) % lines) * 128ull;
		acc += *(const uint64_t *)(tab + a + extra);
		if constexpr (MODE == 1) acc += *(const uint64_t *)(tab + a + 64);
		if constexpr (MODE == 2) acc += *(const uint64_t *)(tab + a + 32);
		if constexpr (MODE == 3) { const uint64_t b = ((mix(st ^ 0x9e3779b97f4a7c15ull) >> 11) % lines) * 128ull; acc += *(const uint64_t *)(tab + b); }
		st = mix(st + (uint64_t)i);
	}
	if (acc == 0x1234567) out[0] = acc;
}

template <int MODE>
static void run(const char *name, const uint8_t *tab, uint64_t lines, int blocks, int iters, uint32_t delta, int loads_per_iter, double lines_per_iter, uint64_t *out, bool warm = true)
{
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	if (warm) { probe<MODE><<<blocks, 256>>>(tab, lines, 2, delta, out); CK(hipDeviceSynchronize()); }
	float best = 1e30f;
	for (int rep = 0; rep < (warm ? 3 : 1); rep++) {
		CK(hipEventRecord(a));
		probe<MODE><<<blocks, 256>>>(tab, lines, iters, delta, out);
		CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
		float ms; CK(hipEventElapsedTime(&ms, a, b));
		if (ms < best) best = ms;
	}
	const double n = (double)blocks * 256 * iters;
	printf("{\"variant\": \"%s\", \"table_GiB\": %.2f, \"lanes\": %d, \"iters\": %d, \"loads\": %.4g, \"lines\": %.4g, \"ms\": %.3f, \"Gloads_per_s\": %.2f, \"Glines_per_s\": %.2f}\n",
	       name, lines * 128.0 / (double)(1ull << 30), blocks * 256, iters, n * loads_per_iter, n * lines_per_iter, best, n * loads_per_iter / best / 1e6, n * lines_per_iter / best / 1e6);
	fflush(stdout);
}

int main(int argc, char **argv)
{
	const double gib = argc > 1 ? atof(argv[1]) : 16.0;
	const uint64_t bytes = (uint64_t)(gib * (double)(1ull << 30)), lines = bytes / 128;
	uint8_t *tab; uint64_t *out;
	CK(hipMalloc((void **)&tab, (size_t)bytes + 256)); CK(hipMalloc((void **)&out, 64));
	CK(hipMemset(tab, 0x5a, (size_t)bytes + 256));
	hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
	const int blocks = p.multiProcessorCount * 8, iters = 256;
	run<0>("one", tab, lines, blocks, iters, 0, 1, 1.0, out);
	run<1>("half", tab, lines, blocks, iters, 0, 2, 1.0, out);
	run<2>("near", tab, lines, blocks, iters, 0, 2, 1.0, out);
	run<3>("two", tab, lines, blocks, iters, 0, 2, 2.0, out);
	run<4>("xwave", tab, lines, blocks, iters, 0, 1, 0.5, out);
	// two launches over 32 768 lines (4 MiB: 0.5 MiB per XCD's L2) somewhere in the table, each line gathered once per launch
	{
		hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
		const int sb = 128;
		// flush what earlier variants left: sweep 1 GiB of other lines
		probe<0><<<blocks, 256>>>(tab, lines, 64, 0, out);
		CK(hipDeviceSynchronize());
		run<5>("seq_a", tab, lines, sb, 1, 0, 1, 1.0, out, false);
		run<5>("seq_b", tab, lines, sb, 1, 64, 1, 1.0, out, false);
		run<5>("seq_a_again", tab, lines, sb, 1, 0, 1, 1.0, out, false);
	}
	return 0;
}
