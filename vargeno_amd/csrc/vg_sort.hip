// vg_sort.hip -- device radix sort (rocPRIM) behind a plain function, in its own translation unit so that
// the kernels' TU does not pay for the rocPRIM headers.  Used at index-load time to build the LO32-ordered
// secondary view of the reference dictionary.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <stdint.h>

// sorts (key, value) pairs by key; returns a hipError_t as int.  temp storage is allocated and freed here.
int vg_dev_sort_pairs_u64_u32(const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out, size_t n, hipStream_t stream)
{
	if (n == 0) return 0;
	size_t bytes = 0;
	hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 64, stream);
	if (e != hipSuccess) return (int)e;
	void *tmp = nullptr;
	e = hipMalloc(&tmp, bytes ? bytes : 1);
	if (e != hipSuccess) return (int)e;
	e = rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 64, stream);
	hipError_t e2 = hipStreamSynchronize(stream);
	(void)hipFree(tmp);
	return (int)(e != hipSuccess ? e : e2);
}

// exclusive prefix sums (u32 -> u32, u64 -> u64); temp storage allocated and freed here
template <class T>
static int scan_impl(const T *in, T *out, size_t n, hipStream_t stream)
{
	if (n == 0) return 0;
	size_t bytes = 0;
	hipError_t e = rocprim::exclusive_scan(nullptr, bytes, in, out, T(0), n, rocprim::plus<T>(), stream);
	if (e != hipSuccess) return (int)e;
	void *tmp = nullptr;
	e = hipMalloc(&tmp, bytes ? bytes : 1);
	if (e != hipSuccess) return (int)e;
	e = rocprim::exclusive_scan(tmp, bytes, in, out, T(0), n, rocprim::plus<T>(), stream);
	hipError_t e2 = hipStreamSynchronize(stream);
	(void)hipFree(tmp);
	return (int)(e != hipSuccess ? e : e2);
}
int vg_dev_exclusive_scan_u32(const uint32_t *in, uint32_t *out, size_t n, hipStream_t stream) { return scan_impl<uint32_t>(in, out, n, stream); }
int vg_dev_exclusive_scan_u64(const uint64_t *in, uint64_t *out, size_t n, hipStream_t stream) { return scan_impl<unsigned long long>((const unsigned long long *)in, (unsigned long long *)out, n, stream); }
