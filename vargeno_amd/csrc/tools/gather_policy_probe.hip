// gather_policy_probe.hip -- measurement aid (not part of the product path): does the cache policy of a load
// (sc0 / sc1 / nt bits of gfx950's memory instructions) or its width move the chip's random-gather ceiling?
// Same shape as gather_probe.hip (independent hashed gathers, every lane its own line), table of 4 GiB - 16 B
// so that one buffer resource covers it.  Prints one JSON line per variant.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

// MODE 0: plain global load; 1: __builtin_nontemporal_load; 2: raw buffer load with cache-policy AUX; WIDTH in bytes (4, 8, 16)
template <int MODE, int AUX, int WIDTH>
__global__ __launch_bounds__(256) void gather(const uint8_t *__restrict__ tab, uint32_t bytes, int iters, uint64_t *out)
{
	const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t acc = 0, st = mix(gid + 1);
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)tab, (short)0, (int)bytes, 0x00020000);
	const uint32_t slots = bytes / 16;
	for (int i = 0; i < iters; i++) {
		const uint32_t off = (uint32_t)((st >> 20) % slots) * 16u;
		if constexpr (MODE == 0) {
			if constexpr (WIDTH == 4) acc += *(const uint32_t *)(tab + off);
			else if constexpr (WIDTH == 8) acc += *(const uint64_t *)(tab + off);
			else { const v4u v = *(const v4u *)(tab + off); acc += v.x + v.w; }
		} else if constexpr (MODE == 1) {
			if constexpr (WIDTH == 4) acc += __builtin_nontemporal_load((const uint32_t *)(tab + off));
			else if constexpr (WIDTH == 8) acc += __builtin_nontemporal_load((const uint64_t *)(tab + off));
			else { const v4u v = __builtin_nontemporal_load((const v4u *)(tab + off)); acc += v.x + v.w; }
		} else {
			if constexpr (WIDTH == 4) acc += __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, AUX);
			else if constexpr (WIDTH == 8) { const v2u v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, AUX); acc += v.x + v.y; }
			else { const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, AUX); acc += v.x + v.w; }
		}
		st = mix(st + (uint64_t)i);
	}
	if (acc == 0x1234567) out[0] = acc;
}

template <int MODE, int AUX, int WIDTH>
static void run(const char *name, const uint8_t *tab, uint32_t bytes, int blocks, int iters, uint64_t *out)
{
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	gather<MODE, AUX, WIDTH><<<blocks, 256>>>(tab, bytes, 4, out);
	CK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int rep = 0; rep < 3; rep++) {
		CK(hipEventRecord(a));
		gather<MODE, AUX, WIDTH><<<blocks, 256>>>(tab, bytes, iters, out);
		CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
		float ms; CK(hipEventElapsedTime(&ms, a, b));
		if (ms < best) best = ms;
	}
	const double n = (double)blocks * 256 * iters;
	printf("{\"variant\": \"%s\", \"width\": %d, \"table_GiB\": %.2f, \"lanes\": %d, \"loads\": %.3g, \"ms\": %.3f, \"Gloads_per_s\": %.2f}\n",
	       name, WIDTH, bytes / (double)(1 << 30), blocks * 256, n, best, n / best / 1e6);
	fflush(stdout);
}

int main(int argc, char **argv)
{
	const uint32_t bytes = argc > 1 ? (uint32_t)(atof(argv[1]) * (1 << 20)) : 0xFFFFFFF0u;
	uint8_t *tab; uint64_t *out;
	CK(hipMalloc((void **)&tab, (size_t)bytes + 16)); CK(hipMalloc((void **)&out, 64));
	CK(hipMemset(tab, 0x5a, (size_t)bytes + 16));
	hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
	const int blocks = p.multiProcessorCount * 8, iters = 256;
	run<0, 0, 8>("global", tab, bytes, blocks, iters, out);
	run<0, 0, 4>("global", tab, bytes, blocks, iters, out);
	run<0, 0, 16>("global", tab, bytes, blocks, iters, out);
	run<1, 0, 8>("global nontemporal", tab, bytes, blocks, iters, out);
	run<1, 0, 16>("global nontemporal", tab, bytes, blocks, iters, out);
	run<2, 0, 8>("buffer aux=0", tab, bytes, blocks, iters, out);
	run<2, 1, 8>("buffer aux=1 (sc0)", tab, bytes, blocks, iters, out);
	run<2, 2, 8>("buffer aux=2 (nt)", tab, bytes, blocks, iters, out);
	run<2, 3, 8>("buffer aux=3 (sc0 nt)", tab, bytes, blocks, iters, out);
	run<2, 16, 8>("buffer aux=16 (sc1)", tab, bytes, blocks, iters, out);
	run<2, 17, 8>("buffer aux=17 (sc0 sc1)", tab, bytes, blocks, iters, out);
	run<2, 18, 8>("buffer aux=18 (sc1 nt)", tab, bytes, blocks, iters, out);
	run<2, 19, 8>("buffer aux=19 (sc0 sc1 nt)", tab, bytes, blocks, iters, out);
	run<2, 0, 16>("buffer aux=0", tab, bytes, blocks, iters, out);
	run<2, 2, 16>("buffer aux=2 (nt)", tab, bytes, blocks, iters, out);
	return 0;
}
