// vmm_probe -- does the HIP virtual-memory API behave the way vg_arena.h uses it?  (development probe, `make probes`)
// Chunks of 1 GiB in one reserved range; every step is checked through a kernel AND through hipMemcpy:
//   1  map A at slot 0, B at slot 1; a kernel writes a pattern across the boundary; read back by kernel and by hipMemcpy
//   2  hipMemsetAsync / hipMemcpyAsync (pinned host -> device, device -> device) across the chunk boundary
//   3  unmap slot 0, map chunk C there (fresh), is C's content visible (not A's)?  then unmap, map A again elsewhere: A's content intact?
//   4  the arena's recycle: unmap A, map A at slot 5, write; unmap, map at slot 0 again ... (stale translations would show)
// One line per check: "ok" or "FAIL".
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static const uint64_t C = 1ull << 30;

__global__ void fill(uint64_t *p, uint64_t n, uint64_t tag) { for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = tag ^ (i * 0x9E3779B97F4A7C15ull); }
__global__ void check(const uint64_t *p, uint64_t n, uint64_t tag, unsigned long long *bad) { unsigned long long b = 0; for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) b += p[i] != (tag ^ (i * 0x9E3779B97F4A7C15ull)); if (b) atomicAdd(bad, b); }
__global__ void check_const(const uint64_t *p, uint64_t n, uint64_t v, unsigned long long *bad) { unsigned long long b = 0; for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) b += p[i] != v; if (b) atomicAdd(bad, b); }

static unsigned long long *d_bad;
static unsigned long long run_check(const uint64_t *p, uint64_t n, uint64_t tag)
{
	unsigned long long h = 0;
	(void)hipMemset(d_bad, 0, 8);
	check<<<4096, 256>>>(p, n, tag, d_bad);
	(void)hipDeviceSynchronize();
	(void)hipMemcpy(&h, d_bad, 8, hipMemcpyDeviceToHost);
	return h;
}
static unsigned long long run_check_const(const uint64_t *p, uint64_t n, uint64_t v)
{
	unsigned long long h = 0;
	(void)hipMemset(d_bad, 0, 8);
	check_const<<<4096, 256>>>(p, n, v, d_bad);
	(void)hipDeviceSynchronize();
	(void)hipMemcpy(&h, d_bad, 8, hipMemcpyDeviceToHost);
	return h;
}

int main()
{
	CK(hipSetDevice(0));
	CK(hipMalloc(&d_bad, 8));
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned;
	prop.location.type = hipMemLocationTypeDevice;
	prop.location.id = 0;
	hipMemAccessDesc acc = {};
	acc.location = prop.location;
	acc.flags = hipMemAccessFlagsProtReadWrite;
	void *vav = nullptr;
	CK(hipMemAddressReserve(&vav, 8 * C, 0, nullptr, 0));
	uint8_t *va = (uint8_t *)vav;
	printf("reserved at %p (mod 1 GiB: %llu)\n", vav, (unsigned long long)((uintptr_t)vav % C));
	hipMemGenericAllocationHandle_t A, B, Cc;
	CK(hipMemCreate(&A, C, &prop, 0)); CK(hipMemCreate(&B, C, &prop, 0)); CK(hipMemCreate(&Cc, C, &prop, 0));
	auto map = [&](int slot, hipMemGenericAllocationHandle_t h) -> int { CK(hipMemMap(va + slot * C, C, 0, h, 0)); CK(hipMemSetAccess(va + slot * C, C, &acc, 1)); return 0; };
	auto unmap = [&](int slot) -> int { CK(hipDeviceSynchronize()); CK(hipMemUnmap(va + slot * C, C)); return 0; };
	// 1
	if (map(0, A) || map(1, B)) return 1;
	const uint64_t W = C / 8;
	uint64_t *mid = (uint64_t *)(va + C - (64ull << 20));               // 128 MiB straddling the boundary
	fill<<<4096, 256>>>(mid, (128ull << 20) / 8, 0x1111);
	CK(hipDeviceSynchronize());
	printf("1a kernel write + kernel read across the boundary: %s\n", run_check(mid, (128ull << 20) / 8, 0x1111) == 0 ? "ok" : "FAIL");
	{
		std::vector<uint64_t> h((128ull << 20) / 8);
		CK(hipMemcpy(h.data(), mid, 128ull << 20, hipMemcpyDeviceToHost));
		uint64_t bad = 0;
		for (uint64_t i = 0; i < h.size(); i++) bad += h[i] != (0x1111ull ^ (i * 0x9E3779B97F4A7C15ull));
		printf("1b hipMemcpy D2H across the boundary: %s (%llu bad words)\n", bad == 0 ? "ok" : "FAIL", (unsigned long long)bad);
	}
	// 2
	{
		CK(hipMemsetAsync(mid, 0, 128ull << 20, 0));
		CK(hipDeviceSynchronize());
		printf("2a hipMemsetAsync across the boundary: %s\n", run_check_const(mid, (128ull << 20) / 8, 0) == 0 ? "ok" : "FAIL");
		uint64_t *hp = nullptr;
		CK(hipHostMalloc((void **)&hp, 128ull << 20, hipHostMallocDefault));
		for (uint64_t i = 0; i < (128ull << 20) / 8; i++) hp[i] = 0x2222ull ^ (i * 0x9E3779B97F4A7C15ull);
		hipStream_t s;
		CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
		CK(hipMemcpyAsync(mid, hp, 128ull << 20, hipMemcpyHostToDevice, s));
		CK(hipStreamSynchronize(s));
		printf("2b hipMemcpyAsync pinned H2D across the boundary: %s (%llu bad)\n", run_check(mid, (128ull << 20) / 8, 0x2222) == 0 ? "ok" : "FAIL", run_check(mid, (128ull << 20) / 8, 0x2222));
		uint64_t *src = nullptr;
		CK(hipMalloc((void **)&src, 128ull << 20));
		fill<<<4096, 256>>>(src, (128ull << 20) / 8, 0x3333);
		CK(hipMemcpyAsync(mid, src, 128ull << 20, hipMemcpyDeviceToDevice, 0));
		CK(hipDeviceSynchronize());
		printf("2c hipMemcpyAsync D2D (hipMalloc -> vmm) across the boundary: %s\n", run_check(mid, (128ull << 20) / 8, 0x3333) == 0 ? "ok" : "FAIL");
		std::vector<uint64_t> pg((64ull << 20) / 8);
		for (uint64_t i = 0; i < pg.size(); i++) pg[i] = 0x4444ull ^ (i * 0x9E3779B97F4A7C15ull);
		CK(hipMemcpy(va + C - (32ull << 20), pg.data(), 64ull << 20, hipMemcpyHostToDevice));
		printf("2d hipMemcpy pageable H2D across the boundary: %s\n", run_check((uint64_t *)(va + C - (32ull << 20)), (64ull << 20) / 8, 0x4444) == 0 ? "ok" : "FAIL");
		(void)hipFree(src); (void)hipHostFree(hp);
	}
	// 3: chunk contents follow the chunk, translations follow the mapping
	fill<<<4096, 256>>>((uint64_t *)va, W, 0xAAAA);                      // A's content (slot 0)
	fill<<<4096, 256>>>((uint64_t *)(va + C), W, 0xBBBB);                // B's content (slot 1)
	CK(hipDeviceSynchronize());
	if (unmap(0) || map(0, Cc)) return 1;
	fill<<<4096, 256>>>((uint64_t *)va, W, 0xCCCC);
	CK(hipDeviceSynchronize());
	printf("3a slot 0 re-mapped to a fresh chunk, written, read: %s\n", run_check((uint64_t *)va, W, 0xCCCC) == 0 ? "ok" : "FAIL");
	if (map(5, A)) return 1;
	printf("3b chunk A mapped at slot 5 still holds what was written through slot 0: %s (%llu bad)\n", run_check((uint64_t *)(va + 5 * C), W, 0xAAAA) == 0 ? "ok" : "FAIL", run_check((uint64_t *)(va + 5 * C), W, 0xAAAA));
	// 4: recycle in a loop: slot 0 alternates between chunk Cc and chunk A, slot 5 between A and Cc.  After every swap the slots must
	//    show what their NEW chunk held (a stale translation would show the old chunk's content), before anything is written
	unsigned long long bad_total = 0, bad_follow = 0;
	uint64_t tagA = 0xAAAA, tagC = 0xCCCC;                               // what each chunk holds (A sits at slot 5, Cc at slot 0 now)
	for (int it = 0; it < 6; it++) {
		if (unmap(0) || unmap(5)) return 1;
		const bool a_at_0 = it % 2 == 0;
		if (map(0, a_at_0 ? A : Cc) || map(5, a_at_0 ? Cc : A)) return 1;
		bad_follow += run_check((uint64_t *)va, W, a_at_0 ? tagA : tagC);
		bad_follow += run_check((uint64_t *)(va + 5 * C), W, a_at_0 ? tagC : tagA);
		tagA = 0xA000ull + it; tagC = 0xC000ull + it;
		fill<<<4096, 256>>>((uint64_t *)va, W, a_at_0 ? tagA : tagC);
		fill<<<4096, 256>>>((uint64_t *)(va + 5 * C), W, a_at_0 ? tagC : tagA);
		CK(hipDeviceSynchronize());
		bad_total += run_check((uint64_t *)va, W, a_at_0 ? tagA : tagC);
		bad_total += run_check((uint64_t *)(va + 5 * C), W, a_at_0 ? tagC : tagA);
		bad_total += run_check((uint64_t *)(va + C), W, 0xBBBB);         // the neighbour that never moved
	}
	printf("4a six rounds of swapping two chunks between two slots, content follows the chunk: %s (%llu bad words)\n", bad_follow == 0 ? "ok" : "FAIL", bad_follow);
	printf("4b ... written and read back in place, neighbour untouched: %s (%llu bad words)\n", bad_total == 0 ? "ok" : "FAIL", bad_total);
	// 5: hipMemcpyAsync H2D into a slot right after it was re-mapped (the loader's raw-file buffer), read by a kernel
	{
		uint64_t *hp = nullptr;
		CK(hipHostMalloc((void **)&hp, 64ull << 20, hipHostMallocDefault));
		hipStream_t s;
		CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
		unsigned long long bad5 = 0;
		for (int it = 0; it < 4; it++) {
			if (unmap(0) || unmap(5)) return 1;
			if (map(0, it % 2 ? A : Cc) || map(5, it % 2 ? Cc : A)) return 1;
			for (uint64_t i = 0; i < (64ull << 20) / 8; i++) hp[i] = (0x5000ull + it) ^ (i * 0x9E3779B97F4A7C15ull);
			CK(hipMemcpyAsync(va + (100ull << 20), hp, 64ull << 20, hipMemcpyHostToDevice, s));
			CK(hipStreamSynchronize(s));
			bad5 += run_check((uint64_t *)(va + (100ull << 20)), (64ull << 20) / 8, 0x5000ull + it);
		}
		printf("5 pinned H2D into a slot that has just been re-mapped, read by a kernel: %s (%llu bad words)\n", bad5 == 0 ? "ok" : "FAIL", bad5);
	}
	return 0;
}
