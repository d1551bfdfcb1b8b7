// index_build.cpp -- `vargeno index <ref.fa> <snps.vcf> <prefix>`: writes <prefix>.ref.bf,
// .ref.bf.lite.bf, .snp.bf, .chrlens, .snp.dict, .ref.dict byte-for-byte as the reference does
// (src/qv.cc:2239-2389), including its accidental behaviours:
//   * two different FASTA readers: the bit-vector side keeps the WHOLE header line as the sequence
//     name and does not fold case (src/generate_bf.cc:18-73); the dictionary side cuts the name at
//     '|'/whitespace/64 chars, upper-cases and maps non-ACGT to N (src/fasta_parser.c:35-133);
//   * B2: the SNP bit vector receives LO40 of the 32-mer PRECEDING each SNP, because shift_kmer's
//     result is discarded (src/generate_bf.cc:257);
//   * a VCF chromosome the bit-vector side cannot find leaves the PREVIOUS chromosome's sequence in
//     use (src/generate_bf.cc:214-222);
//   * `freq_index` is sticky across VCF lines (src/dictgen.c:716-735).
// Written from scratch: rolling 2-bit encoding, two-level radix partition + per-bucket sorts, positioned parallel writes.
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include <omp.h>

#include "vg_host.h"

namespace vgh {

static const uint64_t REF_BF_BITS = 1200000000ull * 8;        // generate_bf.h:199-201
static const uint64_t REF_LITE_BF_BITS = 2300000000ull * 8;
static const uint64_t SNP_BF_BITS = 140000000ull * 8;
static const uint32_t POS_AMBIGUOUS = 0xFFFFFFFFu;
static const int AUX_COLS = 10;

[[noreturn]] static void die(const std::string &m) { throw Error{m}; }

// VARGENO_VERBOSE=1: wall time of the phases on stderr
struct PhaseTimer {
	const bool on = getenv("VARGENO_VERBOSE") != nullptr && atoi(getenv("VARGENO_VERBOSE")) != 0;
	double t0 = omp_get_wtime();
	void lap(const char *what) { if (!on) return; const double t = omp_get_wtime(); fprintf(stderr, "[vargeno index] %-40s %.2f s\n", what, t - t0); t0 = t; }
};

static std::string slurp(const std::string &path)
{
	FILE *f = fopen(path.c_str(), "rb");
	if (!f) die("Error opening: " + path);
	std::string s;
	fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
	s.resize((size_t)sz);
	if (sz && fread(&s[0], 1, (size_t)sz, f) != (size_t)sz) { fclose(f); die("short read on " + path); }
	fclose(f);
	return s;
}

static inline int base_code(unsigned char c)     // 0..3 ACGT (either case), 4 = N/n, 7 = anything else (util.c:66-87)
{
	switch (c) {
	case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3;
	case 'N': case 'n': return 4; default: return 7;
	}
}
static inline uint32_t hash32(uint32_t x) { x = ((x >> 16) ^ x) * 0x45d9f3bu; x = ((x >> 16) ^ x) * 0x45d9f3bu; return (x >> 16) ^ x; }
static inline uint64_t hash40(uint64_t x) { x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull; return x ^ (x >> 31); }

struct Seq { std::string name, seq; };

// src/fasta_parser.c:35-133: a record starts at a '>' (wherever it stands); its name ends at '|', white space or 64 characters and
// the rest of that line is skipped; its sequence is every character up to the next '>' except newlines, ACGT folded to upper
// case, anything else mapped to N.  The bodies (3.1 GB for a human genome) are converted by all threads: newlines are counted
// per 16 MiB piece first, so that every piece knows where its bases land.
static std::vector<Seq> parse_fasta_dict(const std::string &buf)
{
	std::vector<Seq> out;
	std::vector<std::pair<size_t, size_t>> body;                  // [begin, end) of every record's sequence text
	const size_t n = buf.size();
	size_t i = 0;
	while (i < n) {
		const char *gt = (const char *)memchr(buf.data() + i, '>', n - i);
		if (!gt) break;
		i = (size_t)(gt - buf.data()) + 1;
		Seq s;
		bool newline = false;
		while (i < n) {
			const char c = buf[i++];
			if (c == '|' || isspace((unsigned char)c) || s.name.size() == 64) { newline = (c == '\n'); break; }
			s.name.push_back(c);
		}
		if (!newline) while (i < n && buf[i++] != '\n') {}
		const char *nx = i < n ? (const char *)memchr(buf.data() + i, '>', n - i) : nullptr;
		const size_t j = nx ? (size_t)(nx - buf.data()) : n;
		body.push_back({i, j});
		out.push_back(std::move(s));
		i = j;
	}
	struct Piece { size_t rec, lo, hi, bases, at; };
	std::vector<Piece> pieces;
	for (size_t r = 0; r < body.size(); r++)
		for (size_t lo = body[r].first; lo < body[r].second || lo == body[r].first; lo += (size_t)16 << 20) {
			pieces.push_back(Piece{r, lo, std::min(body[r].second, lo + ((size_t)16 << 20)), 0, 0});
			if (lo >= body[r].second) break;
		}
	#pragma omp parallel for schedule(dynamic, 1)
	for (long p = 0; p < (long)pieces.size(); p++) {
		Piece &pc = pieces[(size_t)p];
		size_t nl = 0;
		for (const char *q = buf.data() + pc.lo, *e = buf.data() + pc.hi; q < e;) {
			const char *f = (const char *)memchr(q, '\n', (size_t)(e - q));
			if (!f) break;
			nl++; q = f + 1;
		}
		pc.bases = (pc.hi - pc.lo) - nl;
	}
	{
		size_t r = (size_t)-1, at = 0;
		for (Piece &pc : pieces) { if (pc.rec != r) { if (r != (size_t)-1) out[r].seq.resize(at); r = pc.rec; at = 0; } pc.at = at; at += pc.bases; }
		if (r != (size_t)-1) out[r].seq.resize(at);
	}
	#pragma omp parallel for schedule(dynamic, 1)
	for (long p = 0; p < (long)pieces.size(); p++) {
		const Piece &pc = pieces[(size_t)p];
		char *dst = &out[pc.rec].seq[0] + pc.at;
		for (size_t k = pc.lo; k < pc.hi; k++) {
			const char c = buf[k];
			if (c == '\n') continue;
			const int code = base_code((unsigned char)c);
			*dst++ = code < 4 ? "ACGT"[code] : 'N';
		}
	}
	return out;
}

// src/generate_bf.cc:18-73
static std::vector<Seq> parse_fasta_bf(const std::string &buf)
{
	std::vector<Seq> out;
	std::string id, dna;
	size_t i = 0;
	const size_t n = buf.size();
	while (i < n) {
		size_t e = buf.find('\n', i);
		if (e == std::string::npos) e = n;
		if (e > i) {
			if (buf[i] == '>') {
				if (!id.empty()) out.push_back(Seq{id, dna});
				id.assign(buf, i + 1, e - i - 1);
				dna.clear();
			} else {
				dna.append(buf, i, e - i);
			}
		}
		i = e + 1;
	}
	if (!id.empty()) out.push_back(Seq{id, dna});
	return out;
}

// ---- bit vectors ---------------------------------------------------------------------------------
struct BitVec {
	// Lazily-zeroed anonymous mapping + a dirty flag per MiB: the reference's vectors are 1.2 / 2.3 /
	// 0.14 GB whatever the genome size, and for small genomes almost every page stays untouched.
	static constexpr size_t BLK_WORDS = 1 << 17;           // 1 MiB
	uint64_t bits; size_t nwords; uint64_t *w; std::vector<uint8_t> dirty;
	// dense: nearly every page will be touched (a genome of hundreds of Mbp) -- populate the mapping up front, in one go, instead
	// of taking a page fault per 4 KiB from every thread at once
	explicit BitVec(uint64_t b, bool dense = false) : bits(b), nwords((size_t)((b + 63) / 64)), w(nullptr), dirty((nwords + BLK_WORDS - 1) / BLK_WORDS, 0)
	{
		void *p = mmap(nullptr, nwords * 8, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | (dense ? MAP_POPULATE : MAP_NORESERVE), -1, 0);
		if (p == MAP_FAILED) die("cannot map a bit vector");
		w = (uint64_t *)p;
	}
	~BitVec() { if (w) munmap(w, nwords * 8); }
	BitVec(const BitVec &) = delete;
	inline void set_atomic(uint64_t p)
	{
		const size_t i = (size_t)(p >> 6);
		__atomic_fetch_or(&w[i], 1ull << (p & 63), __ATOMIC_RELAXED);
		if (!dirty[i / BLK_WORDS]) dirty[i / BLK_WORDS] = 1;      // idempotent byte store; racing writers agree
	}
	uint64_t count() const
	{
		uint64_t c = 0;
		for (size_t b = 0; b < dirty.size(); b++) {
			if (!dirty[b]) continue;
			const size_t e = std::min(nwords, (b + 1) * BLK_WORDS);
			for (size_t k = b * BLK_WORDS; k < e; k++) c += (uint64_t)__builtin_popcountll(w[k]);
		}
		return c;
	}
	// sdsl int_vector<1>::serialize: u64 size in bits, then the words (int_vector.hpp:1563-1595).
	// Untouched megabytes are left as holes in the file (they read back as zeros).
	void save(const std::string &path) const
	{
		int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
		if (fd < 0) die("cannot write " + path);
		const uint64_t total = 8 + 8 * (uint64_t)nwords;
		if (ftruncate(fd, (off_t)total) != 0) { close(fd); die("cannot size " + path); }
		if (pwrite(fd, &bits, 8, 0) != 8) { close(fd); die("write failed: " + path); }
		for (size_t b = 0; b < dirty.size(); b++) {
			if (!dirty[b]) continue;
			const size_t k0 = b * BLK_WORDS, e = std::min(nwords, k0 + BLK_WORDS);
			const char *src = (const char *)&w[k0];
			size_t left = (e - k0) * 8; off_t at = (off_t)(8 + 8 * k0);
			while (left) { ssize_t r = pwrite(fd, src, left, at); if (r <= 0) { close(fd); die("write failed: " + path); } src += r; left -= (size_t)r; at += r; }
		}
		close(fd);
	}
};

// all N-free 32-mers of s, in position order: f(kmer, start offset).  The rolling window reproduces
// ref_to_constituent_kmers (dictgen.c:12-54) and constructBfFromGenomeseq's loop (generate_bf.cc:107-146).
template <class F>
static void for_each_kmer(const std::string &s, size_t lo, size_t hi, bool strict, F &&f)
{
	// windows starting in [lo, hi); strict: a non-ACGTN base is an error (encode_kmer asserts, util.c:103)
	if (s.size() < 32) return;
	const size_t last = std::min(hi, s.size() - 31);
	if (lo >= last) return;
	uint64_t k = 0;
	size_t valid = 0;                            // number of consecutive good bases ending at the current base
	for (size_t p = lo; p < last + 31; p++) {
		const int c = base_code((unsigned char)s[p]);
		if (c < 4) { k = (k >> 2) | ((uint64_t)c << 62); valid++; }
		else { if (c == 7 && strict) die(std::string("invalid base '") + s[p] + "' in reference sequence"); valid = 0; }
		if (p >= lo + 31 && valid >= 32) f(k, p - 31);
	}
}


// ---- dictionary construction at genome scale ------------------------------------------------------
// Both dictionaries are "all k-mers, sorted by k-mer, equal k-mers in input order".  hg38 has 2.9 G of them, so the
// sort is a two-level radix partition instead of one comparison sort over 50 GB: producers (chunks of the input, in
// input order) are run twice -- once to count their k-mers per bucket (top PART_BITS bits of the k-mer), once to drop
// every record at its final bucket slot -- and each bucket (a few MB, cache resident, already in input order) is then
// sorted on its own.  Buckets are also the unit of record emission and of the (parallel, positioned) file writes, so
// no second copy of the dictionary is ever assembled in memory.
static const int PART_BITS = 12;
static const size_t N_PART = (size_t)1 << PART_BITS;

template <class T>
struct Partitioned {
	T *data = nullptr;                       // uninitialised storage, first touched by the threads that fill it
	size_t n = 0;
	std::vector<size_t> begin;               // N_PART + 1 bucket bounds
	~Partitioned() { free(data); }
};

struct ByKmerTop { template <class T> size_t operator()(const T &r) const { return (size_t)(r.kmer >> (64 - PART_BITS)); } };

// produce(chunk, sink): calls sink(const T &) for every record of the chunk, in input order; must be repeatable.
// bucket_of(record) < N_PART.
template <class T, class Produce, class Bucket = ByKmerTop>
static void partition_records(size_t n_chunks, Produce &&produce, Partitioned<T> &out, Bucket bucket_of = Bucket())
{
	std::vector<uint32_t> cnt(n_chunks * N_PART, 0);                 // a chunk holds < 2^32 records
	#pragma omp parallel for schedule(dynamic, 1)
	for (long c = 0; c < (long)n_chunks; c++) {
		uint32_t *row = &cnt[(size_t)c * N_PART];
		produce((size_t)c, [&](const T &r) { row[bucket_of(r)]++; });
	}
	// slot of (bucket b, chunk c) = records in earlier buckets + records of bucket b in earlier chunks
	out.begin.assign(N_PART + 1, 0);
	std::vector<size_t> at(n_chunks * N_PART);
	size_t total = 0;
	for (size_t b = 0; b < N_PART; b++) {
		out.begin[b] = total;
		for (size_t c = 0; c < n_chunks; c++) { at[c * N_PART + b] = total; total += cnt[c * N_PART + b]; }
	}
	out.begin[N_PART] = total;
	out.n = total;
	out.data = (T *)malloc(std::max<size_t>(total, 1) * sizeof(T));
	if (!out.data) die("out of memory while building a dictionary");
	#pragma omp parallel for schedule(dynamic, 1)
	for (long c = 0; c < (long)n_chunks; c++) {
		size_t *row = &at[(size_t)c * N_PART];
		T *dst = out.data;
		produce((size_t)c, [&](const T &r) { dst[row[bucket_of(r)]++] = r; });
	}
}

// Sort one bucket of a partitioned dictionary by k-mer, equal k-mers keeping their order (the bucket is in input order).  The
// records of a bucket share the top PART_BITS bits of the k-mer: one stable counting pass on the next SUB_BITS bits into a
// scratch buffer leaves groups of a few hundred records, which are finished by a stable comparison sort in cache -- about
// three sweeps over the bucket where a comparison sort of the whole 12 MB bucket makes twenty.
template <class T>
static void sort_bucket(T *lo, T *hi, std::vector<T> &tmp, std::vector<uint32_t> &cnt)
{
	const size_t n = (size_t)(hi - lo);
	constexpr int SUB_BITS = 12;
	constexpr size_t NS = (size_t)1 << SUB_BITS;
	const auto less = [](const T &a, const T &b) { return a.kmer < b.kmer; };
	if (n < 4 * NS) { std::stable_sort(lo, hi, less); return; }
	const auto digit = [](const T &r) { return (size_t)((r.kmer >> (64 - PART_BITS - SUB_BITS)) & (NS - 1)); };
	cnt.assign(NS + 1, 0);
	for (const T *p = lo; p < hi; p++) cnt[digit(*p) + 1]++;
	for (size_t d = 0; d < NS; d++) cnt[d + 1] += cnt[d];
	if (tmp.size() < n) tmp.resize(n);
	{
		std::vector<uint32_t> at(cnt.begin(), cnt.end() - 1);
		for (const T *p = lo; p < hi; p++) tmp[at[digit(*p)]++] = *p;
	}
	for (size_t d = 0; d < NS; d++) {
		T *a = tmp.data() + cnt[d], *b = tmp.data() + cnt[d + 1];
		if (b - a > 1) std::stable_sort(a, b, less);
	}
	memcpy((void *)lo, (const void *)tmp.data(), n * sizeof(T));
}

// A dictionary file under construction: fixed-size records region + auxiliary rows region, produced bucket by bucket by many
// threads.  How the bytes reach the file (VARGENO_WRITE_MODE), measured on the MI355X host (256 threads, overlayfs and
// tmpfs alike, 5.2 GB dictionary; profiles/io_probe.sh):
//   pwrite  (default)     every producer writes its own buffers at their offsets: 2.4 GB/s (the file's inode lock serialises the
//                         copies, but nothing else waits);
//   stream                producers hand buffers to ONE writer thread: 1.0 GB/s;
//   mmap                  producers copy into a shared mapping of the pre-sized file: 0.4 GB/s (page-fault bound; 150 s for the
//                         43 GB hg38 dictionary).
struct DictFile {
	enum Mode { STREAM, PWRITE, MMAP };
	int fd = -1; std::string path; uint8_t *map = nullptr; size_t size = 0; Mode mode = PWRITE;
	struct Piece { std::vector<uint8_t> data; uint64_t off; };
	std::deque<Piece> queue; size_t queued = 0; bool closing = false; std::string error;
	std::mutex mu; std::condition_variable cv_push, cv_pop; std::thread writer;
	static constexpr size_t QUEUE_BYTES = (size_t)2 << 30;

	DictFile(const std::string &p, size_t bytes) : path(p), size(bytes)
	{
		fd = open(p.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
		if (fd < 0) die("cannot write " + p);
		if (ftruncate(fd, (off_t)bytes) != 0) die("cannot size " + p);
		if (const char *m = getenv("VARGENO_WRITE_MODE")) mode = !strcmp(m, "mmap") ? MMAP : !strcmp(m, "pwrite") ? PWRITE : !strcmp(m, "stream") ? STREAM : mode;
		if (mode == MMAP) {
			void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
			if (m != MAP_FAILED) map = (uint8_t *)m; else mode = PWRITE;
		}
		if (mode == STREAM) writer = std::thread([this] { drain(); });
	}
	~DictFile() { finish_nothrow(); if (map) munmap(map, size); if (fd >= 0) close(fd); }
	void pwrite_all(const void *src, size_t n, uint64_t off)
	{
		const char *p = (const char *)src;
		while (n) {
			const ssize_t w = pwrite(fd, p, n, (off_t)off);
			if (w <= 0) { std::lock_guard<std::mutex> g(mu); if (error.empty()) error = "write failed: " + path; return; }
			p += w; n -= (size_t)w; off += (uint64_t)w;
		}
	}
	void drain()
	{
		for (;;) {
			Piece pc;
			{
				std::unique_lock<std::mutex> g(mu);
				cv_pop.wait(g, [&] { return !queue.empty() || closing; });
				if (queue.empty()) return;
				pc = std::move(queue.front()); queue.pop_front();
				queued -= pc.data.size();
			}
			cv_push.notify_all();
			pwrite_all(pc.data.data(), pc.data.size(), pc.off);
		}
	}
	// takes the buffer (left empty)
	void write_at(std::vector<uint8_t> &buf, uint64_t off)
	{
		if (buf.empty()) return;
		if (mode == MMAP) { memcpy(map + off, buf.data(), buf.size()); return; }
		if (mode == PWRITE) { pwrite_all(buf.data(), buf.size(), off); return; }
		std::unique_lock<std::mutex> g(mu);
		cv_push.wait(g, [&] { return queued < QUEUE_BYTES; });
		queued += buf.size();
		queue.push_back(Piece{std::move(buf), off});
		g.unlock();
		cv_pop.notify_one();
	}
	void finish_nothrow()
	{
		if (writer.joinable()) {
			{ std::lock_guard<std::mutex> g(mu); closing = true; }
			cv_pop.notify_all();
			writer.join();
		}
	}
	void finish() { finish_nothrow(); if (!error.empty()) die(error); }
};

struct KP { uint64_t kmer; uint32_t pos; uint32_t pad; };
struct SK { uint64_t kmer; uint32_t pos; uint8_t snp, rf, af, pad; };

// ---- VCF line access with the reference's pointer semantics ------------------------------------
struct Fields {
	const std::string *line; std::vector<size_t> start;
	char at(size_t f, size_t j) const { size_t p = start[f] + j; return p < line->size() ? (*line)[p] : '\0'; }
};
static void split_line_ref(const std::string &line, Fields &fl)   // util.c:190-200
{
	fl.line = &line; fl.start.clear();
	size_t p = 0;
	while (p < line.size()) {
		fl.start.push_back(p);
		while (p < line.size() && line[p] != '\t' && line[p] != '\n') p++;
		p++;
	}
}
static bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n'; }

void build_index(const std::string &fasta, const std::string &vcf, const std::string &prefix, const IndexOptions &opt)
{
	// `index` only accepts SNP lists whose file name ends in ".vcf" (qv.cc:2244, 2315)
	{
		const size_t dot = vcf.rfind('.');
		if (dot == std::string::npos || vcf.substr(dot + 1) != "vcf") { printf("Unrecongized SNP list file format.\n"); die("Unrecongized SNP list file format."); }
	}
	const int nthreads = opt.threads > 0 ? opt.threads : omp_get_max_threads();
	omp_set_num_threads(nthreads);
	PhaseTimer pt;
	// the SNP list is parsed in pieces of this many bytes (cut at line ends), in parallel; VARGENO_PARSE_PIECE: tests make it tiny
	const size_t parse_piece_bytes = getenv("VARGENO_PARSE_PIECE") ? (size_t)std::max(1, atoi(getenv("VARGENO_PARSE_PIECE"))) : (size_t)4 << 20;
	const std::string fa = slurp(fasta);
	const std::string vcf_text = slurp(vcf);
	pt.lap("inputs read");

	// =============================== bit vectors (BFGenerator) ===============================
	{
		std::vector<Seq> g = parse_fasta_bf(fa);
		pt.lap("FASTA parsed (bit-vector side)");
		const bool dense = fa.size() > ((size_t)256 << 20);
		BitVec bf(REF_BF_BITS, dense);
		BitVec lite(opt.write_lite ? REF_LITE_BF_BITS : 64, dense && opt.write_lite);
		// Bits to set: hash32(first 16 bases) of every N-free 32-mer -- 3.1 G scattered bits at hg38 scale.  Setting them with
		// atomics from all threads took 22 s there (every bit a cache miss on a line other cores are writing); instead the bit
		// numbers are radix-partitioned by their top 12 bits first, and each 128 KiB slice of the vector is then filled by one
		// thread out of its own list.  (The lite vector, which `geno` never reads, keeps the atomic path.)
		struct Chunk { size_t seq, lo, hi; };
		std::vector<Chunk> chunks;
		for (size_t si = 0; si < g.size(); si++) {
			if (g[si].seq.size() < 32) die("reference sequence shorter than 32 bases: " + g[si].name);      // assert, generate_bf.cc:104
			const size_t nwin = g[si].seq.size() - 31;
			for (size_t lo = 0; lo < nwin; lo += (size_t)1 << 22) chunks.push_back(Chunk{si, lo, std::min(nwin, lo + ((size_t)1 << 22))});
		}
		std::string err;
		struct Bit { uint32_t at; };
		Partitioned<Bit> part;
		partition_records<Bit>(chunks.size(), [&](size_t c, auto &&sink) {
			try {
				for_each_kmer(g[chunks[c].seq].seq, chunks[c].lo, chunks[c].hi, true, [&](uint64_t k, size_t) {
					sink(Bit{hash32((uint32_t)k)});                      // (hash32 < 2^32 <= REF_BF_BITS: the reference's modulo never fires)
					if (opt.write_lite) lite.set_atomic(hash40(k & 0xFFFFFFFFFFull) % REF_LITE_BF_BITS);
				});
			} catch (const Error &e) {
				#pragma omp critical
				err = e.msg;
			}
		}, part, [](const Bit &b) { return (size_t)(b.at >> (32 - PART_BITS)); });
		if (!err.empty()) die(err);
		#pragma omp parallel for schedule(dynamic, 4)
		for (long b = 0; b < (long)N_PART; b++) {
			const Bit *lo = part.data + part.begin[(size_t)b], *hi = part.data + part.begin[(size_t)b + 1];
			if (lo == hi) continue;
			for (const Bit *p = lo; p < hi; p++) bf.w[p->at >> 6] |= 1ull << (p->at & 63);
			bf.dirty[((size_t)b << (32 - PART_BITS - 6)) / BitVec::BLK_WORDS] = 1;       // a slice lies inside one dirty block
		}
		if (!opt.quiet) {
			printf("[BloomFilter constructBfFromGenomeseq] bit vector: %llu/%llu\n", (unsigned long long)bf.count(), (unsigned long long)REF_BF_BITS);
			if (opt.write_lite) printf("[BloomFilter constructBfFromGenomeseq] lite bit vector: %llu/%llu\n", (unsigned long long)lite.count(), (unsigned long long)REF_LITE_BF_BITS);
		}
		pt.lap("reference bit vector filled");
		bf.save(prefix + ".ref.bf");
		if (opt.write_lite) lite.save(prefix + ".ref.bf.lite.bf");
		pt.lap("reference bit vector written");

		// constructBfFromVcf, generate_bf.cc:179-277
		BitVec sbf(SNP_BF_BITS);
		static const std::string empty;
		// (100 M lines at hg38 + full dbSNP scale: parsed in place -- no per-line vector or string --, the k-mer never shifting
		// (B2), one hash per SNP instead of 32 of the same; and in parallel over pieces of the text.  The reference's loop carries
		// one thing from line to line: the sequence of the last chromosome name it FOUND in the FASTA (a name it does not find
		// leaves the previous sequence in place, generate_bf.cc:204-215; before the first one found there is no sequence and
		// every record is out of range).  A piece that starts with names it cannot find does not know that sequence yet: those
		// records wait for a serial sweep over the pieces, which hands each piece the last sequence found before it.)
		auto atoi_span = [](const char *a, const char *b) -> int {        // atoi() of the text [a, b)
			while (a < b && isspace((unsigned char)*a)) a++;
			bool neg = false;
			if (a < b && (*a == '+' || *a == '-')) { neg = *a == '-'; a++; }
			long v = 0;
			while (a < b && *a >= '0' && *a <= '9') { v = v * 10 + (*a - '0'); a++; }
			return (int)(neg ? -v : v);
		};
		const char *const text = vcf_text.data();
		struct NameCache { const char *c0 = nullptr; size_t len = 0; const std::string *found = nullptr; };     // column 0 of the last record looked up, and what it named
		// one record, text [ls, le).  seq: the sequence in force (nullptr = not known yet); returns false if the record has to wait
		auto record = [&](size_t ls, size_t le, const std::string *&seq, NameCache &nc_) -> bool {
			if (le == ls || text[ls] == '#') return true;
			// split(line, '\t'): the first five columns
			const char *cs[5], *ce[5];
			int nc = 0;
			for (const char *a = text + ls, *end = text + le; nc < 5;) {
				const char *tb = (const char *)memchr(a, '\t', (size_t)(end - a));
				cs[nc] = a; ce[nc] = tb ? tb : end; nc++;
				if (!tb) break;
				a = tb + 1;
			}
			if (nc < 5) return true;                                       // the reference would index past the vector
			const int pos = atoi_span(cs[1], ce[1]) - 1;
			const size_t rl = (size_t)(ce[3] - cs[3]), al = (size_t)(ce[4] - cs[4]);
			if (rl > 1 || al > 1) return true;
			const size_t c0l = (size_t)(ce[0] - cs[0]);
			if (!(nc_.c0 && c0l == nc_.len && memcmp(cs[0], nc_.c0, c0l) == 0)) {
				std::string chr(cs[0], c0l);
				if (chr.empty() || chr[0] != 'c') chr = "chr" + chr;
				nc_.found = nullptr;
				for (const Seq &s : g) if (s.name == chr) { nc_.found = &s.seq; break; }
				nc_.c0 = cs[0]; nc_.len = c0l;
			}
			if (nc_.found) seq = nc_.found;                                // not found: the previous sequence stays
			else if (!seq) return false;
			if (pos < 32 || (size_t)(pos + 32) > seq->size()) return true;
			if (rl == 0 || al == 0) return true;
			const char ref_nt = *cs[3], alt_nt = *cs[4];
			if (ref_nt != (*seq)[(size_t)pos] || ref_nt == alt_nt) return true;
			uint64_t k = 0; bool has_n = false;
			for (int j = 31; j >= 0 && !has_n; j--) {                        // encode_kmer scans from base 31 down
				const int c = base_code((unsigned char)(*seq)[(size_t)pos - 32 + (size_t)j]);
				if (c == 4) has_n = true;
				else if (c == 7) die("invalid base in reference sequence near a SNP");
				else k = (k << 2) | (uint64_t)c;
			}
			if (has_n) return true;
			bool any = false;
			for (unsigned tt = 0; tt < 32; tt++) {
				const char nb = tt ? (*seq)[(size_t)pos + tt] : alt_nt;
				const int c = base_code((unsigned char)nb);
				if (c == 4) break;
				if (c == 7) die("invalid base while building the SNP bit vector");            // shift_kmer asserts, util.c:121
				any = true;                                                                      // B2: k never shifts -- the same bit every time
			}
			if (any) sbf.set_atomic(hash40(k & 0xFFFFFFFFFFull) % SNP_BF_BITS);
			return true;
		};
		struct Piece { size_t lo, hi; const std::string *last_found = nullptr; std::vector<std::pair<size_t, size_t>> waiting; std::string err; };
		std::vector<Piece> pieces;
		for (size_t lo = 0; lo < vcf_text.size();) {
			size_t hi = std::min(vcf_text.size(), lo + parse_piece_bytes);
			if (hi < vcf_text.size()) {
				const char *nl = (const char *)memchr(text + hi, '\n', vcf_text.size() - hi);
				hi = nl ? (size_t)(nl - text) + 1 : vcf_text.size();
			}
			Piece pc; pc.lo = lo; pc.hi = hi;
			pieces.push_back(std::move(pc));
			lo = hi;
		}
		#pragma omp parallel for schedule(dynamic, 1)
		for (long pi = 0; pi < (long)pieces.size(); pi++) {
			Piece &pc = pieces[(size_t)pi];
			const std::string *seq = nullptr;
			NameCache cache;
			try {
				for (size_t i = pc.lo; i < pc.hi;) {
					const char *nl = (const char *)memchr(text + i, '\n', pc.hi - i);
					const size_t e = nl ? (size_t)(nl - text) : pc.hi;
					if (!record(i, e, seq, cache)) pc.waiting.emplace_back(i, e);
					i = e + 1;
				}
			} catch (const Error &e) { pc.err = e.msg; }
			pc.last_found = seq;
		}
		const std::string *carry = &empty;
		for (Piece &pc : pieces) {
			NameCache cache;
			for (const auto &w : pc.waiting) { const std::string *seq = carry; record(w.first, w.second, seq, cache); }
			if (!pc.err.empty()) die(pc.err);
			if (pc.last_found) carry = pc.last_found;
		}
		if (!opt.quiet) printf("[BloomFilter constructBfFromVCF] bit vector: %llu/%llu\n", (unsigned long long)sbf.count(), (unsigned long long)SNP_BF_BITS);
		sbf.save(prefix + ".snp.bf");
		pt.lap("SNP bit vector");
	}

	// =============================== dictionaries (dictgen.c) ===============================
	std::vector<Seq> ref = parse_fasta_dict(fa);
	pt.lap("FASTA parsed (dictionary side)");
	{
		FILE *f = fopen((prefix + ".chrlens").c_str(), "w");
		if (!f) die("cannot write " + prefix + ".chrlens");
		for (const Seq &s : ref) fprintf(f, "%s %lu\n", s.name.c_str(), (unsigned long)s.seq.size());     // qv.cc:2343-2345
		fclose(f);
	}
	if (ref.empty()) die("no sequences in " + fasta);

	// ---- SNP dictionary: make_snp_dict_from_vcf, dictgen.c:561-794
	{
		struct SnpRec { const Seq *chrom; uint32_t index, start_index; uint8_t ref_u, alt, f1, f2; };     // one accepted VCF line
		const bool ref_has_chr = !ref[0].name.empty() && ref[0].name[0] == 'c';
		// The reference's loop over the lines carries two things from line to line, both about the INFO column: `has_freq`
		// (cleared for good if the first record that gets that far has no CAF key) and `freq_index` (the token after the last
		// "CAF..." key seen so far: a record without one reads the token at the place the last record with one had it).  The
		// chromosome look-up repeats for every record whose name is not found, so it carries nothing.  So: the text is
		// parsed in pieces, in parallel; the first piece runs alone up to the first record that reaches the INFO column, which
		// settles `has_freq`; a record without a CAF key that comes before any record with one IN ITS PIECE waits for a serial
		// sweep over the pieces, which knows `freq_index` at the start of each piece; messages to stderr are kept per piece
		// and printed in file order; the first failure in file order is the one reported.
		struct Piece {
			size_t lo, hi;
			std::vector<SnpRec> recs;
			std::vector<std::pair<size_t, std::pair<size_t, size_t>>> waiting;      // (record index, its line's text [a, b))
			int freq_index_out = -1;                                               // freq_index after the piece (-1: no CAF key in it)
			std::string msgs, err;
		};
		// tokens of the INFO column (vcf_split_line, dictgen.c:538-553: split on ';' and '=', running on into what follows)
		auto info_tokens = [](const std::string &line, size_t p, std::vector<size_t> &tok) {
			tok.clear();
			auto ch = [&](size_t q) { return q < line.size() ? line[q] : '\0'; };
			while (ch(p) && !is_ws(ch(p))) {
				tok.push_back(p);
				while (ch(p) != ';' && ch(p) != '=') { if (ch(p) && !is_ws(ch(p))) ++p; else break; }
				++p;
			}
		};
		auto freqs_at = [](const std::string &line, const std::vector<size_t> &tok, int freq_index, uint8_t &f1, uint8_t &f2) {
			// the reference reads info_split[freq_index], which may be a stale pointer when this line has fewer
			// tokens than the line that set freq_index; defined here as 0.5/0.5 for that case.
			float freq1 = 0.5f, freq2 = 0.5f;
			if (freq_index >= 0 && (size_t)freq_index < tok.size()) {
				const char *p = line.c_str() + tok[(size_t)freq_index];
				freq1 = (float)atof(p);
				while (*p && *p != ',') p++;
				if (*p == ',') p++;
				freq2 = (float)atof(p);
			}
			f1 = (uint8_t)(freq1 * 0xff); f2 = (uint8_t)(freq2 * 0xff);
		};
		// mode 0: the whole piece, `has_freq` known.  mode 1: from pc.lo up to and including the first record that reaches the
		// INFO column (the very first piece, `has_freq` still open); pc.lo is moved past what was read.
		bool has_freq = true;
		auto parse_piece = [&](Piece &pc, int mode) {
			const Seq *chrom = nullptr; uint32_t start_index = 1;
			int freq_index = -1;                                               // of this piece so far
			std::string line; Fields fl; std::vector<size_t> tok;
			size_t i = pc.lo;
			try {
				while (i < pc.hi) {
					// fgets(line, 6000): at most 5999 characters per piece
					const char *nl = (const char *)memchr(vcf_text.data() + i, '\n', pc.hi - i);
					size_t le = nl ? (size_t)(nl - vcf_text.data()) + 1 : pc.hi;
					if (le - i > 5999) le = i + 5999;
					const size_t ls = i;
					line.assign(vcf_text, i, le - i);
					i = le;
					if (mode == 1) pc.lo = i;
					if (line[0] == '#' || line[0] == '\n') continue;
					split_line_ref(line, fl);
					if (fl.start.size() < 8) continue;                                  // the reference dereferences NULL here
					char chrom_name[50]; size_t ci;
					if (fl.at(0, 0) != 'c' && ref_has_chr) {
						chrom_name[0] = 'c'; chrom_name[1] = 'h'; chrom_name[2] = 'r';
						for (ci = 0; !isspace((unsigned char)fl.at(0, ci)) && fl.at(0, ci) && ci + 3 < 49; ci++) chrom_name[ci + 3] = fl.at(0, ci);
						chrom_name[ci + 3] = '\0';
					} else {
						for (ci = 0; !isspace((unsigned char)fl.at(0, ci)) && fl.at(0, ci) && ci < 49; ci++) chrom_name[ci] = fl.at(0, ci);
						chrom_name[ci] = '\0';
					}
					const char ref_base = (char)toupper((unsigned char)fl.at(3, 0));
					const int ref_u = base_code((unsigned char)ref_base);
					if (ref_u == 7) continue;
					if (!isspace((unsigned char)fl.at(3, 1))) continue;
					if (!isspace((unsigned char)fl.at(4, 1))) continue;
					if (chrom == nullptr || chrom->name != chrom_name) {
						chrom = nullptr; start_index = 0;
						uint32_t si = 1;
						for (const Seq &s : ref) { if (s.name == chrom_name) { chrom = &s; start_index = si; break; } si += (uint32_t)s.seq.size(); }
						if (chrom == nullptr) {
							pc.msgs += std::string("[Error] chromosome name ") + chrom_name + " in VCF file not found in reference genome FASTA file\n. Usually this is because the FASTA file has chromesome name as \"chr1\" while the VCF file has chromosome name as \"1\" without the \"chr\"\n";
							continue;
						}
					}
					const unsigned index = (unsigned)atoi(line.c_str() + fl.start[1]) - 1u;
					if (index >= chrom->seq.size() || toupper((unsigned char)chrom->seq[index]) != ref_base) {
						char msg[256];
						snprintf(msg, sizeof msg, "Mismatch found between reference sequence and SNP file at 0-based index %u in %s.", index, chrom->name.c_str());
						pc.msgs += msg; pc.msgs += "\n";
						die(msg);
					}
					if (index < 32 || (size_t)index + 32 > chrom->seq.size()) continue;
					const char a2 = (char)toupper((unsigned char)fl.at(4, 0));
					if (!(ref_base == 'A' || ref_base == 'C' || ref_base == 'G' || ref_base == 'T')) continue;
					if (!(a2 == 'A' || a2 == 'C' || a2 == 'G' || a2 == 'T')) continue;
					// allele frequencies: the token after the LAST one starting with "CAF" is "ref,alt"
					uint8_t f1 = (uint8_t)(0.5f * 0xff), f2 = f1;
					bool wait = false;
					if (has_freq) {
						info_tokens(line, fl.start[7], tok);
						for (size_t tt = 0; tt < tok.size(); tt++) if (line.compare(tok[tt], 3, "CAF") == 0) freq_index = (int)tt + 1;
						if (freq_index == -1) {
							if (mode == 1) has_freq = false;                             // the first such record of the file: no frequencies at all
							else wait = true;                                            // the place of the last CAF key before this piece
						} else freqs_at(line, tok, freq_index, f1, f2);
					}
					const bool stop = mode == 1;
					do {
						if (a2 == ref_base) break;
						// the 32 k-mers over the SNP exist only if the 63 bases around it are ACGT (dictgen.c:751-790)
						const std::string &seq = chrom->seq;
						bool clean = true;
						for (unsigned j = 0; j < 63 && clean; j++) if (j != 31 && base_code((unsigned char)seq[index - 31 + j]) >= 4) clean = false;
						if (base_code((unsigned char)seq[index - 32]) >= 4) clean = false;
						if (!clean) break;
						if (wait) pc.waiting.push_back({pc.recs.size(), {ls, le}});
						pc.recs.push_back(SnpRec{chrom, index, start_index, (uint8_t)ref_u, (uint8_t)base_code((unsigned char)a2), f1, f2});
					} while (false);
					if (stop) break;
				}
			} catch (const Error &e) { pc.err = e.msg; }
			pc.freq_index_out = freq_index;
		};
		std::vector<Piece> pieces;
		{
			Piece head; head.lo = 0; head.hi = vcf_text.size();
			parse_piece(head, 1);                                                  // settles has_freq; head.lo: where the rest starts
			const size_t rest = head.lo;
			head.lo = 0; head.hi = rest;
			const bool failed = !head.err.empty();
			pieces.push_back(std::move(head));
			for (size_t lo = rest; !failed && lo < vcf_text.size();) {
				size_t hi = std::min(vcf_text.size(), lo + parse_piece_bytes);
				if (hi < vcf_text.size()) {
					const char *nl = (const char *)memchr(vcf_text.data() + hi, '\n', vcf_text.size() - hi);
					hi = nl ? (size_t)(nl - vcf_text.data()) + 1 : vcf_text.size();
				}
				Piece pc; pc.lo = lo; pc.hi = hi;
				pieces.push_back(std::move(pc));
				lo = hi;
			}
		}
		#pragma omp parallel for schedule(dynamic, 1)
		for (long pi = 1; pi < (long)pieces.size(); pi++) parse_piece(pieces[(size_t)pi], 0);
		std::vector<size_t> rec_begin(pieces.size() + 1, 0);
		{
			int freq_index = -1;                                               // at the start of the piece
			std::string line; std::vector<size_t> tok; Fields fl;
			for (size_t pi = 0; pi < pieces.size(); pi++) {
				Piece &pc = pieces[pi];
				for (const auto &w : pc.waiting) {
					line.assign(vcf_text, w.second.first, w.second.second - w.second.first);
					split_line_ref(line, fl);
					info_tokens(line, fl.start[7], tok);
					freqs_at(line, tok, freq_index, pc.recs[w.first].f1, pc.recs[w.first].f2);
				}
				if (!pc.msgs.empty()) fputs(pc.msgs.c_str(), stderr);
				if (!pc.err.empty()) die(pc.err);
				if (pc.freq_index_out != -1) freq_index = pc.freq_index_out;
				rec_begin[pi + 1] = rec_begin[pi] + pc.recs.size();
			}
		}
		std::vector<SnpRec> snps(rec_begin.back());
		#pragma omp parallel for schedule(dynamic, 1)
		for (long pi = 0; pi < (long)pieces.size(); pi++) {
			Piece &pc = pieces[(size_t)pi];
			if (!pc.recs.empty()) memcpy(snps.data() + rec_begin[(size_t)pi], pc.recs.data(), pc.recs.size() * sizeof(SnpRec));
			std::vector<SnpRec>().swap(pc.recs);
		}
		pt.lap("SNP list parsed");
		// k-mer t of a SNP covers bases [index - 31 + t, index + t], the SNP's alt base at offset 31 - t
		const size_t SNP_CHUNK = 1 << 15;
		const size_t n_chunks = (snps.size() + SNP_CHUNK - 1) / SNP_CHUNK;
		Partitioned<SK> part;
		partition_records<SK>(n_chunks, [&](size_t c, auto &&sink) {
			const size_t hi = std::min(snps.size(), (c + 1) * SNP_CHUNK);
			for (size_t i = c * SNP_CHUNK; i < hi; i++) {
				const SnpRec &r = snps[i];
				const std::string &seq = r.chrom->seq;
				uint64_t k = 0;
				for (int j = 31; j >= 0; j--) k = (k << 2) | (uint64_t)base_code((unsigned char)seq[r.index - 32 + (unsigned)j]);
				for (unsigned t = 0; t < 32; t++) {
					const uint64_t c2 = t ? (uint64_t)base_code((unsigned char)seq[r.index + t]) : (uint64_t)r.alt;
					k = (k >> 2) | (c2 << 62);
					sink(SK{k, r.start_index + r.index - 32 + 1 + t, (uint8_t)((((31 - t) & 0x1F) << 3) | (r.ref_u & 7u)), r.f1, r.f2, 0});
				}
			}
		}, part);
		pt.lap("SNP k-mers made and partitioned");
		// per bucket: stable sort by k-mer (qsort is glibc's stable merge sort: ties keep VCF order), then count its records and rows
		std::vector<uint64_t> n_rec(N_PART + 1, 0), n_aux(N_PART + 1, 0);
		#pragma omp parallel
		{
		std::vector<SK> tmp; std::vector<uint32_t> cnt;
		#pragma omp for schedule(dynamic, 1)
		for (long b = 0; b < (long)N_PART; b++) {
			SK *lo = part.data + part.begin[(size_t)b], *hi = part.data + part.begin[(size_t)b + 1];
			sort_bucket(lo, hi, tmp, cnt);
			uint64_t rec = 0, aux = 0;
			for (SK *p = lo; p < hi;) { SK *q = p + 1; while (q < hi && q->kmer == p->kmer) q++; rec++; aux += (q - p >= 2 && q - p <= AUX_COLS); p = q; }
			n_rec[(size_t)b + 1] = rec; n_aux[(size_t)b + 1] = aux;
		}
		}
		for (size_t b = 0; b < N_PART; b++) { n_rec[b + 1] += n_rec[b]; n_aux[b + 1] += n_aux[b]; }
		const uint64_t written = n_rec[N_PART], aux_count = n_aux[N_PART];
		pt.lap("SNP k-mers sorted");
		// records: k-mer u64, pos u32, snp u8, ambig u8, ref_freq u8, alt_freq u8; rows: k-mer u64 + 10 x {pos u32, snp, rf, af}
		DictFile out(prefix + ".snp.dict", 16 + 16 * written + 78 * aux_count);
		{ std::vector<uint8_t> head(16); memcpy(&head[0], &written, 8); memcpy(&head[8], &aux_count, 8); out.write_at(head, 0); }
		uint64_t unamb = 0, amb_unique = 0, amb_total = 0;
		#pragma omp parallel for schedule(dynamic, 8) reduction(+ : unamb, amb_unique, amb_total)
		for (long b = 0; b < (long)N_PART; b++) {
			const SK *lo = part.data + part.begin[(size_t)b], *hi = part.data + part.begin[(size_t)b + 1];
			std::vector<uint8_t> rec((size_t)(n_rec[(size_t)b + 1] - n_rec[(size_t)b]) * 16), rows((size_t)(n_aux[(size_t)b + 1] - n_aux[(size_t)b]) * 78, 0);
			uint8_t *w = rec.data(), *x = rows.data();
			uint64_t ax = n_aux[(size_t)b];
			for (const SK *p = lo; p < hi;) {
				const SK *q = p + 1;
				while (q < hi && q->kmer == p->kmer) q++;
				const size_t L = (size_t)(q - p);
				memcpy(w, &p->kmer, 8);
				if (L == 1) {
					unamb++;
					memcpy(w + 8, &p->pos, 4); w[12] = p->snp; w[13] = 0; w[14] = p->rf; w[15] = p->af;
				} else {
					amb_unique++; amb_total += L;
					uint32_t posv = POS_AMBIGUOUS;
					if (L <= (size_t)AUX_COLS) {
						posv = (uint32_t)ax++;
						memcpy(x, &p->kmer, 8);
						for (size_t j = 0; j < L; j++) { uint8_t *e = x + 8 + 7 * j; memcpy(e, &p[j].pos, 4); e[4] = p[j].snp; e[5] = p[j].rf; e[6] = p[j].af; }
						x += 78;
					}
					memcpy(w + 8, &posv, 4); w[12] = 0; w[13] = 1; w[14] = 0; w[15] = 0;
				}
				w += 16;
				p = q;
			}
			out.write_at(rec, 16 + 16 * n_rec[(size_t)b]);
			out.write_at(rows, 16 + 16 * written + 78 * n_aux[(size_t)b]);
		}
		out.finish();
		pt.lap("SNP dictionary written");
		if (!opt.quiet) {
			printf("SNP Dictionary\nTotal k-mers:        %lu\nUnambig k-mers:      %lu\nAmbig unique k-mers: %lu\nAmbig total k-mers:  %lu\n",
			       (unsigned long)part.n, (unsigned long)unamb, (unsigned long)amb_unique, (unsigned long)amb_total);
		}
	}

	// ---- reference dictionary: make_ref_dict, dictgen.c:277-301
	{
		for (const Seq &s : ref) if (s.seq.size() < 32) die("reference sequence shorter than 32 bases: " + s.name);   // assert, dictgen.c:17
		struct Chunk { size_t seq, lo, hi; uint32_t base; };
		std::vector<Chunk> chunks;
		uint32_t base = 1;
		for (size_t si = 0; si < ref.size(); si++) {
			const size_t nwin = ref[si].seq.size() - 31;
			const size_t step = 1 << 22;
			for (size_t lo = 0; lo < nwin; lo += step) chunks.push_back(Chunk{si, lo, std::min(nwin, lo + step), base});
			base += (uint32_t)ref[si].seq.size();
		}
		Partitioned<KP> part;
		partition_records<KP>(chunks.size(), [&](size_t c, auto &&sink) {
			const Chunk &ch = chunks[c];
			for_each_kmer(ref[ch.seq].seq, ch.lo, ch.hi, false, [&](uint64_t k, size_t off) { sink(KP{k, ch.base + (uint32_t)off, 0}); });
		}, part);
		if (part.n >= (1ull << 32)) die("more than 2^32 32-mers in the reference");
		pt.lap("reference k-mers made and partitioned");
		// per bucket: stable sort by k-mer = order by (k-mer, position), a bucket being in position order already
		std::vector<uint64_t> n_rec(N_PART + 1, 0), n_aux(N_PART + 1, 0);
		#pragma omp parallel
		{
		std::vector<KP> tmp; std::vector<uint32_t> cnt;
		#pragma omp for schedule(dynamic, 1)
		for (long b = 0; b < (long)N_PART; b++) {
			KP *lo = part.data + part.begin[(size_t)b], *hi = part.data + part.begin[(size_t)b + 1];
			sort_bucket(lo, hi, tmp, cnt);
			uint64_t rec = 0, aux = 0;
			for (KP *p = lo; p < hi;) { KP *q = p + 1; while (q < hi && q->kmer == p->kmer) q++; rec++; aux += (q - p >= 2 && q - p <= AUX_COLS); p = q; }
			n_rec[(size_t)b + 1] = rec; n_aux[(size_t)b + 1] = aux;
		}
		}
		for (size_t b = 0; b < N_PART; b++) { n_rec[b + 1] += n_rec[b]; n_aux[b + 1] += n_aux[b]; }
		const uint64_t written = n_rec[N_PART], aux_count = n_aux[N_PART];
		pt.lap("reference k-mers sorted");
		// write_kmers (dictgen.c:63-154): records k-mer u64, pos u32, ambig u8; a k-mer with 2..10 copies points at a row of 10 u32
		// positions, one with more gets POS_AMBIGUOUS
		DictFile out(prefix + ".ref.dict", 16 + 13 * written + 40 * aux_count);
		{ std::vector<uint8_t> head(16); memcpy(&head[0], &written, 8); memcpy(&head[8], &aux_count, 8); out.write_at(head, 0); }
		uint64_t unamb = 0, amb_unique = 0, amb_total = 0;
		#pragma omp parallel for schedule(dynamic, 8) reduction(+ : unamb, amb_unique, amb_total)
		for (long b = 0; b < (long)N_PART; b++) {
			const KP *lo = part.data + part.begin[(size_t)b], *hi = part.data + part.begin[(size_t)b + 1];
			std::vector<uint8_t> rec((size_t)(n_rec[(size_t)b + 1] - n_rec[(size_t)b]) * 13), rows((size_t)(n_aux[(size_t)b + 1] - n_aux[(size_t)b]) * 40, 0);
			uint8_t *w = rec.data(), *x = rows.data();
			uint64_t ax = n_aux[(size_t)b];
			for (const KP *p = lo; p < hi;) {
				const KP *q = p + 1;
				while (q < hi && q->kmer == p->kmer) q++;
				const size_t L = (size_t)(q - p);
				uint32_t posv = p->pos;
				if (L == 1) unamb++;
				else {
					amb_unique++; amb_total += L;
					posv = POS_AMBIGUOUS;
					if (L <= (size_t)AUX_COLS) {
						posv = (uint32_t)ax++;
						for (size_t j = 0; j < L; j++) memcpy(x + 4 * j, &p[j].pos, 4);
						x += 40;
					}
				}
				memcpy(w, &p->kmer, 8); memcpy(w + 8, &posv, 4); w[12] = L > 1;
				w += 13;
				p = q;
			}
			out.write_at(rec, 16 + 13 * n_rec[(size_t)b]);
			out.write_at(rows, 16 + 13 * written + 40 * n_aux[(size_t)b]);
		}
		out.finish();
		pt.lap("reference dictionary written");
		if (!opt.quiet) {
			printf("Ref Dictionary\nTotal k-mers:        %lu\nUnambig k-mers:      %lu\nAmbig unique k-mers: %lu\nAmbig total k-mers:  %lu\n",
			       (unsigned long)part.n, (unsigned long)unamb, (unsigned long)amb_unique, (unsigned long)amb_total);
		}
	}
}

}  // namespace vgh
