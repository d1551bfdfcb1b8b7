// alloc_probe -- what does device-memory allocation cost on this box?  (development probe, `make probes`)
//
// vg_index_open at hg38 scale runs ~1.2 s of kernels inside 11-16 s of wall time while it allocates and frees ~60 buffers of
// 1-64 GiB (columns, sort buffers, views).  This probe times, for a few sizes:
//   hipMalloc / hipFree                       (what the loader did through round 4)
//   first touch vs second touch of a fresh allocation (hipMemsetAsync: is anything populated lazily?)
//   the virtual-memory route: hipMemAddressReserve once, then hipMemCreate + hipMemMap + hipMemSetAccess per 2 GiB chunk,
//   hipMemUnmap + hipMemRelease per chunk     (an arena that can grow and give its tail back)
// One JSON line per measurement.   usage: alloc_probe [max GiB, default 64]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
	const size_t GiB = 1ull << 30;
	const size_t max_gib = argc > 1 ? (size_t)atoi(argv[1]) : 64;
	CK(hipSetDevice(0));
	CK(hipFree(nullptr));
	for (size_t g : {(size_t)1, (size_t)4, (size_t)16, (size_t)64}) {
		if (g > max_gib) break;
		for (int rep = 0; rep < 2; rep++) {
			void *p = nullptr;
			double t0 = now();
			CK(hipMalloc(&p, g * GiB));
			const double t_malloc = now() - t0;
			t0 = now();
			CK(hipMemsetAsync(p, 1, g * GiB, 0)); CK(hipDeviceSynchronize());
			const double t_touch1 = now() - t0;
			t0 = now();
			CK(hipMemsetAsync(p, 2, g * GiB, 0)); CK(hipDeviceSynchronize());
			const double t_touch2 = now() - t0;
			t0 = now();
			CK(hipFree(p));
			const double t_free = now() - t0;
			printf("{\"what\": \"hipMalloc\", \"GiB\": %zu, \"rep\": %d, \"malloc_s\": %.4f, \"memset_first_s\": %.4f, \"memset_second_s\": %.4f, \"free_s\": %.4f}\n", g, rep, t_malloc, t_touch1, t_touch2, t_free);
			fflush(stdout);
		}
	}
	// many buffers alive at once, freed in allocation order (the loader's pattern): does the cost grow with what is already mapped?
	{
		std::vector<void *> ps;
		double t0 = now();
		for (int i = 0; i < 12 && (size_t)(i + 1) * 16 <= max_gib * 4; i++) { void *p = nullptr; if (hipMalloc(&p, 16 * GiB) != hipSuccess) break; ps.push_back(p); }
		const double t_m = now() - t0;
		t0 = now();
		for (void *p : ps) (void)hipFree(p);
		printf("{\"what\": \"hipMalloc x N alive\", \"GiB_each\": 16, \"n\": %zu, \"malloc_s_total\": %.4f, \"free_s_total\": %.4f}\n", ps.size(), t_m, now() - t0);
		fflush(stdout);
	}
	// virtual-memory route
	{
		hipMemAllocationProp prop = {};
		prop.type = hipMemAllocationTypePinned;
		prop.location.type = hipMemLocationTypeDevice;
		prop.location.id = 0;
		size_t gran = 0;
		hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
		if (e != hipSuccess) { printf("{\"what\": \"vmm\", \"error\": \"hipMemGetAllocationGranularity: %s\"}\n", hipGetErrorString(e)); return 0; }
		printf("{\"what\": \"vmm granularity\", \"bytes\": %zu}\n", gran);
		for (size_t chunk_gib : {(size_t)2, (size_t)8}) {
			const size_t total = (max_gib < 64 ? max_gib : 64) * GiB, chunk = chunk_gib * GiB, n = total / chunk;
			void *va = nullptr;
			double t0 = now();
			e = hipMemAddressReserve(&va, total, 0, nullptr, 0);
			if (e != hipSuccess) { printf("{\"what\": \"vmm\", \"error\": \"hipMemAddressReserve: %s\"}\n", hipGetErrorString(e)); return 0; }
			const double t_res = now() - t0;
			std::vector<hipMemGenericAllocationHandle_t> hs(n);
			double t_create = 0, t_map = 0, t_acc = 0;
			hipMemAccessDesc acc = {};
			acc.location = prop.location;
			acc.flags = hipMemAccessFlagsProtReadWrite;
			bool ok = true;
			for (size_t i = 0; i < n && ok; i++) {
				t0 = now(); e = hipMemCreate(&hs[i], chunk, &prop, 0); t_create += now() - t0; if (e != hipSuccess) { ok = false; break; }
				t0 = now(); e = hipMemMap((char *)va + i * chunk, chunk, 0, hs[i], 0); t_map += now() - t0; if (e != hipSuccess) { ok = false; break; }
				t0 = now(); e = hipMemSetAccess((char *)va + i * chunk, chunk, &acc, 1); t_acc += now() - t0; if (e != hipSuccess) { ok = false; break; }
			}
			if (!ok) { printf("{\"what\": \"vmm\", \"error\": \"create/map/access: %s\"}\n", hipGetErrorString(e)); return 0; }
			t0 = now();
			CK(hipMemsetAsync(va, 1, total, 0)); CK(hipDeviceSynchronize());
			const double t_touch1 = now() - t0;
			t0 = now();
			CK(hipMemsetAsync(va, 2, total, 0)); CK(hipDeviceSynchronize());
			const double t_touch2 = now() - t0;
			double t_unmap = 0, t_rel = 0;
			for (size_t i = 0; i < n; i++) {
				t0 = now(); (void)hipMemUnmap((char *)va + i * chunk, chunk); t_unmap += now() - t0;
				t0 = now(); (void)hipMemRelease(hs[i]); t_rel += now() - t0;
			}
			t0 = now();
			(void)hipMemAddressFree(va, total);
			printf("{\"what\": \"vmm\", \"GiB\": %zu, \"chunk_GiB\": %zu, \"reserve_s\": %.4f, \"create_s\": %.4f, \"map_s\": %.4f, \"set_access_s\": %.4f, \"memset_first_s\": %.4f, \"memset_second_s\": %.4f, \"unmap_s\": %.4f, \"release_s\": %.4f, \"address_free_s\": %.4f}\n",
			       total / GiB, chunk_gib, t_res, t_create, t_map, t_acc, t_touch1, t_touch2, t_unmap, t_rel, now() - t0);
			fflush(stdout);
		}
	}
	// page-locked host staging (the file reader's ring: 8 x 64 MiB per file)
	{
		double t0 = now();
		void *h[8];
		for (auto &p : h) CK(hipHostMalloc(&p, 64ull << 20, hipHostMallocDefault));
		const double t_a = now() - t0;
		t0 = now();
		for (auto &p : h) (void)hipHostFree(p);
		printf("{\"what\": \"hipHostMalloc 8 x 64 MiB\", \"alloc_s\": %.4f, \"free_s\": %.4f}\n", t_a, now() - t0);
	}
	return 0;
}
