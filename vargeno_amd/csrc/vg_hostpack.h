// vg_hostpack.h -- FASTQ framing + 2-bit packing on HOST threads (no device code): the other half of the FASTQ stream of
// include/vargeno_hip.h.  Replaces the four fgets() + strlen of src/qv.cc:760-784 and encode_kmer (src/util.c:89-111) for a
// stream of text chunks cut anywhere, with the same rules the device-side framing applies (vargeno_hip.hip, vg_fq_*): a record is
// four lines counted from the start of the stream; the read is its second line without the newline (strlen(read) - 1,
// qv.cc:778), trimmed to whole 32-base chunks; chunk c is gate-open iff character c of the fourth line is below '8'
// (qv.cc:836, 943); an N in the trimmed read skips it (qv.cc:815-828), any other character outside ACGTacgt makes it invalid
// (util.c:103: the reference aborts).  A chunk with a line beyond fgets' 1023 characters or a quality line shorter than the
// read's chunk count is REFUSED, and with it everything after it: the caller's host reader, which reproduces the reference's
// stale-buffer behaviour, takes over from `consumed`.
// What crosses the link is 8 bytes per chunk + 16 per read (48 bytes per 150 bp read where the FASTQ text has ~315).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace vgp {

// flag bits of a read's meta word (the same word the device's pack kernel writes, vg_wave.h)
constexpr uint64_t META_SKIP_N = 1ull << 62, META_INVALID = 1ull << 63;

struct Staging {              // caller-owned (pinned) output buffers of one chunk
	uint64_t *kmers = nullptr;   uint64_t kmers_cap = 0;     // chunk k-mers, read after read
	uint64_t *meta = nullptr;                                // per read: gate bits (low 32) | flags
	uint64_t *offsets = nullptr; uint64_t reads_cap = 0;     // per read 32 x (chunks before it): the flat-batch offsets of a batch of trimmed reads; [n_reads] = 32 x chunks
};

struct ChunkResult {
	uint64_t n_reads = 0, n_chunks = 0, n_invalid = 0;
	bool refused = false;        // this chunk (and the stream from here on) needs the host reader
};

class Packer {
public:
	explicit Packer(int threads);
	~Packer();
	Packer(const Packer &) = delete;
	Packer &operator=(const Packer &) = delete;
	int threads() const;
	const char *isa() const;                                   // which build of the hot loops this CPU gets ("avx2+bmi2" / "sse")
	void begin();                                              // a new stream
	// worst-case sizes of the staging buffers for a chunk of nbytes (lines of at least 8 bytes on average, like the device framing)
	static uint64_t reads_cap(uint64_t nbytes) { return nbytes / 32 + 64; }
	static uint64_t kmers_cap(uint64_t nbytes) { return (nbytes + 65536) / 32 + 64; }
	// frame + pack the complete records of (what the previous chunk left unfinished) + text[0, nbytes)
	ChunkResult push(const uint8_t *text, uint64_t nbytes, const Staging &out);
	// stream totals, as vg_fastq_stream_end reports them
	uint64_t records() const;
	uint64_t consumed() const;
	uint64_t last_record_start() const;
	bool poisoned() const;
private:
	struct Impl;
	Impl *p;
};

}  // namespace vgp
