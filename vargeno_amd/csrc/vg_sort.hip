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

// exclusive prefix sums (u32 -> u32, u64 -> u64) on the caller's stream with the caller's scratch: nothing is allocated,
// freed or synchronised here (a hipFree would stall the whole device -- and with it the read loop these scans run beside)
template <class T>
static size_t scan_bytes(size_t n)
{
	size_t bytes = 0;
	(void)rocprim::exclusive_scan(nullptr, bytes, (const T *)nullptr, (T *)nullptr, T(0), n ? n : 1, rocprim::plus<T>(), (hipStream_t)0);
	return bytes;
}
size_t vg_dev_scan_temp_bytes(size_t n_u32, size_t n_u64)
{
	const size_t a = scan_bytes<uint32_t>(n_u32), b = scan_bytes<unsigned long long>(n_u64);
	return (a > b ? a : b) + 256;
}
template <class T>
static int scan_impl(const T *in, T *out, size_t n, hipStream_t stream, void *tmp, size_t tmp_bytes)
{
	if (n == 0) return 0;
	size_t bytes = 0;
	hipError_t e = rocprim::exclusive_scan(nullptr, bytes, in, out, T(0), n, rocprim::plus<T>(), stream);
	if (e != hipSuccess) return (int)e;
	if (bytes > tmp_bytes) return (int)hipErrorInvalidValue;
	return (int)rocprim::exclusive_scan(tmp, bytes, in, out, T(0), n, rocprim::plus<T>(), stream);
}
int vg_dev_exclusive_scan_u32(const uint32_t *in, uint32_t *out, size_t n, hipStream_t stream, void *tmp, size_t tmp_bytes) { return scan_impl<uint32_t>(in, out, n, stream, tmp, tmp_bytes); }
int vg_dev_exclusive_scan_u64(const uint64_t *in, uint64_t *out, size_t n, hipStream_t stream, void *tmp, size_t tmp_bytes)
{
	return scan_impl<unsigned long long>((const unsigned long long *)in, (unsigned long long *)out, n, stream, tmp, tmp_bytes);
}
