// cu_mask_probe.hip -- measurement aid (not part of the product path): which CUs a stream made with
// hipExtStreamCreateWithCUMask really gets on this chip (bit -> XCD / CU), and what a streaming copy reaches on such a subset.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void where(uint32_t *out, int spin)
{
	const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
	const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
	uint64_t t0 = __builtin_readcyclecounter();
	while (__builtin_readcyclecounter() - t0 < (uint64_t)spin) {}
	if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

__global__ __launch_bounds__(256) void copy16(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

static void placement(const char *name, const std::vector<uint32_t> &mask)
{
	hipStream_t s;
	hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
	if (e != hipSuccess) { printf("{\"mask\": \"%s\", \"error\": \"%s\"}\n", name, hipGetErrorString(e)); return; }
	const int blocks = 4096;
	uint32_t *d; CK(hipMalloc((void **)&d, blocks * 8));
	where<<<blocks, 64, 0, s>>>(d, 200000);
	CK(hipStreamSynchronize(s));
	std::vector<uint32_t> h(2 * blocks);
	CK(hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost));
	std::map<uint32_t, int> cus; std::map<uint32_t, int> per_xcc;
	for (int i = 0; i < blocks; i++) {
		const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 15u;
		const uint32_t cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
		const uint32_t key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
		if (!cus.count(key)) per_xcc[xcc]++;
		cus[key]++;
	}
	printf("{\"mask\": \"%s\", \"distinct_cus\": %zu, \"per_xcc\": [", name, cus.size());
	for (uint32_t x = 0; x < 8; x++) printf("%d%s", per_xcc.count(x) ? per_xcc[x] : 0, x < 7 ? ", " : "");
	printf("]}\n");
	// streaming copy on this subset
	const size_t bytes = (size_t)2 << 30;
	uint4 *a, *b; CK(hipMalloc((void **)&a, bytes)); CK(hipMalloc((void **)&b, bytes));
	CK(hipMemset(a, 1, bytes));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int bpc : {4, 8}) {
		const int grid = (int)cus.size() * bpc;
		copy16<<<grid, 256, 0, s>>>(a, b, bytes / 16);
		CK(hipEventRecord(e0, s));
		copy16<<<grid, 256, 0, s>>>(a, b, bytes / 16);
		CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1));
		printf("{\"mask\": \"%s\", \"copy_blocks_per_cu\": %d, \"GBps_read_plus_write\": %.0f}\n", name, bpc, 2.0 * bytes / ms / 1e6);
	}
	CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(d));
	CK(hipStreamDestroy(s));
	fflush(stdout);
}

int main()
{
	hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
	printf("{\"cus\": %d}\n", p.multiProcessorCount);
	auto bits = [](std::initializer_list<std::pair<int, int>> ranges) { std::vector<uint32_t> m(8, 0u); for (auto r : ranges) for (int i = r.first; i < r.second; i++) m[i >> 5] |= 1u << (i & 31); return m; };
	placement("all 256", bits({{0, 256}}));
	placement("bits 0-15", bits({{0, 16}}));
	placement("bits 0-31", bits({{0, 32}}));
	placement("bits 0-7", bits({{0, 8}}));
	placement("bits 16-255", bits({{16, 256}}));
	placement("bits 32-255", bits({{32, 256}}));
	{ std::vector<uint32_t> m(8, 0u); for (int i = 0; i < 256; i += 16) m[i >> 5] |= 1u << (i & 31); placement("every 16th bit", m); }
	{ std::vector<uint32_t> m(8, 0u); for (int i = 0; i < 256; i += 8) m[i >> 5] |= 1u << (i & 31); placement("every 8th bit", m); }
	return 0;
}
