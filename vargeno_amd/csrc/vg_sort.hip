// vg_sort.hip -- device radix sort (rocPRIM) behind a plain function, in its own translation unit so that
// the kernels' TU does not pay for the rocPRIM headers.  Used at index-load time to build the LO32-ordered
// secondary view of the reference dictionary and the merged view, and for the scans of the FASTQ framing.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <stdint.h>

// Sorts (key, value) pairs by key, stable, in a pair of DOUBLE BUFFERS: the passes ping-pong between (keys_a, vals_a) -- the input --
// and (keys_b, vals_b), and *result_in_b says where the sorted pairs ended up.  With rocPRIM's separate input / output arrays the
// library needs scratch for a third copy of everything (39 GB for the merged view of an hg38-scale index); this way its scratch
// is a few megabytes, which the CALLER provides: nothing is allocated, freed or synchronised here (r05: the loader carves every
// buffer out of one arena, vargeno_hip.hip).  Returns a hipError_t as int.
size_t vg_dev_sort_pairs_temp_bytes(size_t n)
{
	size_t bytes = 0;
	rocprim::double_buffer<unsigned long long> k((unsigned long long *)nullptr, (unsigned long long *)nullptr);
	rocprim::double_buffer<uint32_t> v((uint32_t *)nullptr, (uint32_t *)nullptr);
	(void)rocprim::radix_sort_pairs(nullptr, bytes, k, v, n ? n : 1, 0, 64, (hipStream_t)0);
	return bytes + 256;
}
int vg_dev_sort_pairs_u64_u32(uint64_t *keys_a, uint64_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, size_t n, hipStream_t stream, void *tmp, size_t tmp_bytes, bool *result_in_b)
{
	*result_in_b = false;
	if (n == 0) return 0;
	rocprim::double_buffer<unsigned long long> k((unsigned long long *)keys_a, (unsigned long long *)keys_b);
	rocprim::double_buffer<uint32_t> v(vals_a, vals_b);
	size_t bytes = 0;
	hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, k, v, n, 0, 64, stream);
	if (e != hipSuccess) return (int)e;
	if (bytes > tmp_bytes) return (int)hipErrorInvalidValue;
	e = rocprim::radix_sort_pairs(tmp, bytes, k, v, n, 0, 64, stream);
	*result_in_b = (uint64_t *)k.current() == keys_b;
	return (int)e;
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
