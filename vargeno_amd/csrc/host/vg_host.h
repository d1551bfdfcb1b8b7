// vg_host.h -- host side of the vargeno drop-in (C++17): `vargeno index` (file producers of the hot
// path's inputs), FASTQ framing, genotype caller and VCF writer.  None of this is device code; it is
// what sits on either side of the C-ABI in include/vargeno_hip.h.
#pragma once
#include <stdint.h>

#include <string>
#include <utility>
#include <vector>

namespace vgh {

// ---- errors: the reference asserts / exit()s; the host library throws, main() turns it into exit(1)
struct Error {
	std::string msg;
};

// ---- `vargeno index` (reference src/qv.cc:2239-2389) ------------------------------------------
struct IndexOptions {
	bool write_lite = true;      // <prefix>.ref.bf.lite.bf: written by the reference, read by nothing in `geno`
	int threads = 0;             // 0 = all
	bool quiet = false;
};
void build_index(const std::string &fasta, const std::string &vcf, const std::string &prefix, const IndexOptions &opt);

// ---- FASTQ framing (reference src/qv.cc:760-784) ----------------------------------------------
struct ReadBatch {
	std::vector<uint8_t> bases, quals;     // flat, same offsets
	std::vector<uint64_t> offsets;         // n + 1
	uint64_t n() const { return offsets.empty() ? 0 : offsets.size() - 1; }
	void clear() { bases.clear(); quals.clear(); offsets.assign(1, 0); }
};
class FastqReader {
public:
	explicit FastqReader(const std::string &path);
	// a stream that can be read only once (a pipe: the reference fopen()s whatever path it is given and fgets its way through,
	// qv.cc:2182, 760-763): bytes [base, base + the spans' lengths) of it are in the caller's memory (kept alive by the caller),
	// everything after them is read from fd.  seek() works inside those bytes only.
	FastqReader(int fd, uint64_t base, std::vector<std::pair<const uint8_t *, size_t>> spans);
	~FastqReader();
	// appends up to max_reads records; returns the number appended (0 at end of file)
	uint64_t next(ReadBatch &out, uint64_t max_reads);
	// continue from byte `off` of the file; the four line buffers keep their content (what a truncated record sees)
	void seek(uint64_t off);
private:
	struct Impl;
	Impl *p;
};

// ---- caller + VCF writer (behaviour of reference src/qv.cc:1573-1747, 1789-1848) ---------------
enum : uint8_t { GT_NONE = 0, GT_HOM_REF = 1, GT_HOM_ALT = 2, GT_HET = 3 };     // numbering of the reference's GTYPE_* (vartype.h)
struct Genotype {
	uint8_t gt;            // GT_*
	double confidence;     // posterior of the winning genotype x Poisson(7.1) mass of the depth
};
// counts are the 6-bit saturated pile-up counters; frequencies are the dictionary's n/255 encodings
Genotype call_genotype(unsigned ref_cnt, unsigned alt_cnt, uint8_t ref_freq, uint8_t alt_freq);
int genotype_quality(double confidence);                                        // the GQ column: (int)(-10 ln c)

struct ChrLen { std::string name; uint64_t len; };
std::vector<ChrLen> read_chrlens(const std::string &path);

struct SiteCounts {
	std::vector<uint32_t> pos;                     // 1-based over the concatenated genome, ascending
	std::vector<uint8_t> ref_freq, alt_freq, ref_cnt, alt_cnt;
};
// returns {ref calls, alt calls, het calls}
struct CallSummary { uint64_t ref = 0, alt = 0, het = 0; };
// vcf_text: the SNP list's bytes if the caller has read them already (the command line reads them while the index is being opened)
CallSummary write_genotyped_vcf(const SiteCounts &s, const std::vector<ChrLen> &chrlens,
                                const std::string &vcf_in, const std::string &vcf_out, const std::string *vcf_text = nullptr);
bool read_whole_file(const std::string &path, std::string &text);

}  // namespace vgh
