// vg_hostpack_avx2.cpp -- the host packer's hot loops compiled for AVX2 + BMI2 (32 bases per compare, pext for the 2-bit gather);
// vg_hostpack.cpp calls into this build only on a CPU that has both.
#include "vg_hostpack_impl.h"

namespace vgp {
#define VGP_AVX2 1
#define VGP_NS avx2
#include "vg_hostpack_impl.inc"
#undef VGP_NS

static Framed frame_record_avx2(const uint8_t *a, uint64_t N, uint64_t r, uint64_t *next, Out &o) { return avx2::frame_record(a, N, r, *next, o); }
const Kernels &kernels_avx2()
{
	static const Kernels k{avx2::count_newlines, avx2::frame_piece, avx2::frame_piece_guess, frame_record_avx2, "avx2+bmi2"};
	return k;
}
}  // namespace vgp
