// fetch_gran_probe.hip -- measurement aid (not part of the product path).  tools/line_probe showed that an L2 miss of a random
// 8-byte gather moves a whole 128-byte line from DRAM (TCC_EA0_RDREQ_128B = 1, TCC_EA0_RDREQ_DRAM_32B = 4 per gather), so the
// chip's "random-gather ceiling" (49 G/s) is just HBM: 49 G x 128 B = 6.3 TB/s.  Can a gather be made to move LESS?
// The same gather loop over tables allocated three ways (hipMalloc; hipExtMallocWithFlags fine-grained; ... uncached) and
// loaded four ways (plain, nontemporal, relaxed atomic load at agent scope = sc1, at system scope = sc0 sc1); 8- and 16-byte
// loads.  One kernel name per (load mode, width); the allocation is in the JSON line.  Run it under
//   rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B TCC_EA0_RDREQ_DRAM_32B
// to see what each combination fetches.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// LOAD 0 plain, 1 nontemporal, 2 atomic relaxed agent scope, 3 atomic relaxed system scope; WIDTH 8 or 16 (16: plain / nt only)
template <int LOAD, int WIDTH>
__global__ __launch_bounds__(256) void fgp(const uint8_t *__restrict__ tab, uint64_t slots, int iters, uint64_t *out)
{
	const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t acc = 0, st = mix(gid + 1);
	for (int i = 0; i < iters; i++) {
		const uint8_t *p = tab + ((st >> 11) % slots) * 16ull;
		if constexpr (WIDTH == 16) {
			v4u v;
			if constexpr (LOAD == 1) v = __builtin_nontemporal_load((const v4u *)p); else v = *(const v4u *)p;
			acc += v.x + v.w;
		} else {
			if constexpr (LOAD == 0) acc += *(const uint64_t *)p;
			else if constexpr (LOAD == 1) acc += __builtin_nontemporal_load((const uint64_t *)p);
			else if constexpr (LOAD == 2) acc += __hip_atomic_load((const uint64_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			else acc += __hip_atomic_load((const uint64_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
		st = mix(st + (uint64_t)i);
	}
	if (acc == 0x1234567) out[0] = acc;
}

template <int LOAD, int WIDTH>
static void run(const char *alloc, const char *name, const uint8_t *tab, uint64_t bytes, int blocks, int iters, uint64_t *out)
{
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	fgp<LOAD, WIDTH><<<blocks, 256>>>(tab, bytes / 16, 2, out);
	CK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int rep = 0; rep < 2; rep++) {
		CK(hipEventRecord(a));
		fgp<LOAD, WIDTH><<<blocks, 256>>>(tab, bytes / 16, iters, out);
		CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
		float ms; CK(hipEventElapsedTime(&ms, a, b));
		if (ms < best) best = ms;
	}
	const double n = (double)blocks * 256 * iters;
	printf("{\"alloc\": \"%s\", \"load\": \"%s\", \"width\": %d, \"table_GiB\": %.2f, \"loads\": %.4g, \"ms\": %.3f, \"Gloads_per_s\": %.2f}\n", alloc, name, WIDTH, bytes / (double)(1ull << 30), n, best, n / best / 1e6);
	fflush(stdout);
}

int main(int argc, char **argv)
{
	const double gib = argc > 1 ? atof(argv[1]) : 8.0;
	const int only = argc > 2 ? atoi(argv[2]) : -1;              // allocation mode to run (default: all three, one after the other)
	const uint64_t bytes = (uint64_t)(gib * (double)(1ull << 30));
	hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
	const int blocks = p.multiProcessorCount * 8, iters = 128;
	uint64_t *out; CK(hipMalloc((void **)&out, 64));
	const char *names[3] = {"hipMalloc", "finegrained", "uncached"};
	for (int mode = 0; mode < 3; mode++) {
		if (only >= 0 && mode != only) continue;
		uint8_t *tab = nullptr;
		hipError_t e = mode == 0 ? hipMalloc((void **)&tab, bytes + 256) : hipExtMallocWithFlags((void **)&tab, bytes + 256, mode == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
		if (e != hipSuccess) { printf("{\"alloc\": \"%s\", \"error\": \"%s\"}\n", names[mode], hipGetErrorString(e)); (void)hipGetLastError(); continue; }
		CK(hipMemset(tab, 0x5a, bytes + 256));
		CK(hipDeviceSynchronize());
		run<0, 8>(names[mode], "plain", tab, bytes, blocks, iters, out);
		run<1, 8>(names[mode], "nontemporal", tab, bytes, blocks, iters, out);
		run<2, 8>(names[mode], "atomic relaxed agent", tab, bytes, blocks, iters, out);
		run<3, 8>(names[mode], "atomic relaxed system", tab, bytes, blocks, iters, out);
		run<0, 16>(names[mode], "plain", tab, bytes, blocks, iters, out);
		run<1, 16>(names[mode], "nontemporal", tab, bytes, blocks, iters, out);
		CK(hipFree(tab));
	}
	return 0;
}
