// gather_probe.hip -- measurement aid (not part of the product path): the random-gather ceiling of
// one MI355X for the access shapes the read loop is made of.  Every dictionary query of the path is
// two dependent sector-sized touches (jump table word pair -> bucket), so the number that bounds it
// is "random 8-byte gathers per second from a table much larger than the 256 MiB Infinity Cache",
// not the 8 TB/s streaming peak.  Prints one JSON line per configuration.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

// each lane issues `iters` rounds of ILP independent 8-byte loads at hashed addresses; with DEP the next
// address depends on the loaded value (pointer-chase shape of jumpgate -> bucket)
template <int ILP, bool DEP>
__global__ __launch_bounds__(256) void gather(const uint64_t *__restrict__ tab, uint64_t mask, int iters, uint64_t *out)
{
	const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t acc = 0, st[ILP];
	for (int j = 0; j < ILP; j++) st[j] = mix(gid * ILP + j + 1);
	for (int i = 0; i < iters; i++) {
		uint64_t v[ILP];
		for (int j = 0; j < ILP; j++) v[j] = tab[st[j] & mask];
		for (int j = 0; j < ILP; j++) { acc += v[j]; st[j] = mix(st[j] + (DEP ? v[j] : (uint64_t)i)); }
	}
	if (acc == 0x1234567) out[0] = acc;
}

template <int ILP, bool DEP>
static void run(const uint64_t *tab, uint64_t words, int blocks, int iters, uint64_t *out)
{
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	gather<ILP, DEP><<<blocks, 256>>>(tab, words - 1, 4, out);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(a));
	gather<ILP, DEP><<<blocks, 256>>>(tab, words - 1, iters, out);
	CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
	float ms; CK(hipEventElapsedTime(&ms, a, b));
	const double n = (double)blocks * 256 * iters * ILP;
	printf("{\"table_GiB\": %.2f, \"lanes\": %d, \"ilp\": %d, \"dependent\": %s, \"loads\": %.3g, \"ms\": %.3f, \"Gloads_per_s\": %.2f, \"GBps_at_64B\": %.0f}\n",
	       words * 8.0 / (1 << 30), blocks * 256, ILP, DEP ? "true" : "false", n, ms, n / ms / 1e6, n / ms / 1e6 * 64);
	fflush(stdout);
}

int main(int argc, char **argv)
{
	const double gib = argc > 1 ? atof(argv[1]) : 16.0;
	uint64_t words = 1; while (words * 8 < (uint64_t)(gib * (1ull << 30))) words <<= 1;
	uint64_t *tab, *out;
	CK(hipMalloc((void **)&tab, words * 8)); CK(hipMalloc((void **)&out, 64));
	CK(hipMemset(tab, 0x5a, words * 8));
	hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
	const int cus = p.multiProcessorCount;
	for (int bpc : {1, 2, 4, 8}) {
		run<1, false>(tab, words, cus * bpc, 256, out);
		run<1, true>(tab, words, cus * bpc, 256, out);
	}
	run<4, false>(tab, words, cus * 8, 128, out);
	run<4, true>(tab, words, cus * 8, 128, out);
	run<8, false>(tab, words, cus * 8, 64, out);
	run<8, true>(tab, words, cus * 4, 64, out);
	return 0;
}
