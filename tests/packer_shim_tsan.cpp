// TEST INFRASTRUCTURE (tests/test_host_tools.py, the ThreadSanitizer run of the once-only FASTQ route): the library's vg_packer_*
// entry points (vargeno_amd/csrc/vargeno_hip.hip, "host-side framing + packing alone") restated over vgp::Packer, so that a
// -fsanitize=thread build of the command line can link an INSTRUMENTED packer -- libvargeno_hip.so is built by hipcc without the
// sanitizer, and a thread pool whose hand-over ThreadSanitizer cannot see reports every staged byte as a race.  The executable's
// definitions take precedence over the library's; everything else the command line references still comes from the library.
#include <cstdint>

#include "vargeno_hip.h"
#include "vg_hostpack.h"

struct vg_packer { vgp::Packer p; explicit vg_packer(int t) : p(t) {} };
extern "C" int vg_packer_create(int host_threads, vg_packer **out)
{
	if (!out) return VG_EINVAL;
	int t = host_threads <= 0 ? 2 : host_threads;
	if (t > 256) t = 256;
	*out = new vg_packer(t);
	(*out)->p.begin();
	return VG_OK;
}
extern "C" void vg_packer_destroy(vg_packer *pk) { delete pk; }
extern "C" int vg_packer_begin(vg_packer *pk) { if (!pk) return VG_EINVAL; pk->p.begin(); return VG_OK; }
extern "C" uint64_t vg_packer_reads_cap(uint64_t nbytes) { return vgp::Packer::reads_cap(nbytes) + 1; }
extern "C" uint64_t vg_packer_kmers_cap(uint64_t nbytes) { return vgp::Packer::kmers_cap(nbytes); }
extern "C" int vg_packer_push(vg_packer *pk, const uint8_t *text, uint64_t nbytes, uint64_t *kmers, uint64_t kmers_cap, uint64_t *meta, uint64_t *chunk_offsets, uint64_t reads_cap,
                              uint64_t *n_reads, uint64_t *n_chunks, uint64_t *n_invalid)
{
	if (!pk || (!text && nbytes) || !kmers || !meta || !chunk_offsets || !n_reads || !n_chunks) return VG_EINVAL;
	vgp::Staging st;
	st.kmers = kmers; st.kmers_cap = kmers_cap; st.meta = meta; st.offsets = chunk_offsets; st.reads_cap = reads_cap;
	const vgp::ChunkResult r = pk->p.push(text, nbytes, st);
	*n_reads = r.n_reads; *n_chunks = r.n_chunks;
	if (n_invalid) *n_invalid = r.n_invalid;
	for (uint64_t i = 0; i <= r.n_reads && r.n_reads; i++) chunk_offsets[i] >>= 5;       // (flat-batch offsets -> chunk offsets, as the library does)
	return VG_OK;
}
extern "C" int vg_packer_end(vg_packer *pk, uint64_t *n_records, uint64_t *consumed, uint64_t *last_record_start, int *refused)
{
	if (!pk) return VG_EINVAL;
	if (n_records) *n_records = pk->p.records();
	if (consumed) *consumed = pk->p.consumed();
	if (last_record_start) *last_record_start = pk->p.last_record_start();
	if (refused) *refused = pk->p.poisoned() ? 1 : 0;
	return VG_OK;
}
