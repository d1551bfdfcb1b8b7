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
// Written from scratch: rolling 2-bit encoding, parallel stable sort, one buffered write per file.
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <parallel/algorithm>
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

// src/fasta_parser.c:35-133
static std::vector<Seq> parse_fasta_dict(const std::string &buf)
{
	std::vector<Seq> out;
	const size_t n = buf.size();
	size_t i = 0;
	while (i < n) {
		if (buf[i++] != '>') continue;
		Seq s;
		bool newline = false;
		while (i < n) {
			const char c = buf[i++];
			if (c == '|' || isspace((unsigned char)c) || s.name.size() == 64) { newline = (c == '\n'); break; }
			s.name.push_back(c);
		}
		if (!newline) while (i < n && buf[i++] != '\n') {}
		size_t j = i, cnt = 0;
		while (j < n && buf[j] != '>') { cnt += buf[j] != '\n'; j++; }
		s.seq.resize(cnt);
		size_t k = 0;
		for (; i < j; i++) {
			const char c = buf[i];
			if (c == '\n') continue;
			const int code = base_code((unsigned char)c);
			s.seq[k++] = code < 4 ? "ACGT"[code] : 'N';
		}
		out.push_back(std::move(s));
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
	explicit BitVec(uint64_t b) : bits(b), nwords((size_t)((b + 63) / 64)), w(nullptr), dirty((nwords + BLK_WORDS - 1) / BLK_WORDS, 0)
	{
		void *p = mmap(nullptr, nwords * 8, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
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

struct KP { uint64_t kmer; uint32_t pos; uint32_t pad; };
struct SK { uint64_t kmer; uint32_t pos; uint8_t snp, rf, af, pad; };

static void put(std::vector<uint8_t> &o, const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; o.insert(o.end(), b, b + n); }
static void write_file(const std::string &path, const std::vector<uint8_t> &data)
{
	FILE *f = fopen(path.c_str(), "wb");
	if (!f) die("cannot write " + path);
	if (!data.empty() && fwrite(data.data(), 1, data.size(), f) != data.size()) { fclose(f); die("write failed: " + path); }
	fclose(f);
}

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
	const std::string fa = slurp(fasta);
	const std::string vcf_text = slurp(vcf);

	// =============================== bit vectors (BFGenerator) ===============================
	{
		std::vector<Seq> g = parse_fasta_bf(fa);
		BitVec bf(REF_BF_BITS);
		BitVec lite(opt.write_lite ? REF_LITE_BF_BITS : 64);
		for (const Seq &s : g) {
			if (s.seq.size() < 32) die("reference sequence shorter than 32 bases: " + s.name);      // assert, generate_bf.cc:104
			const size_t nwin = s.seq.size() - 31;
			const size_t chunk = std::max<size_t>(1 << 20, (nwin + nthreads - 1) / nthreads);
			const long nchunks = (long)((nwin + chunk - 1) / chunk);
			std::string err;
			#pragma omp parallel for schedule(dynamic, 1)
			for (long c = 0; c < nchunks; c++) {
				try {
					for_each_kmer(s.seq, (size_t)c * chunk, std::min(nwin, (size_t)(c + 1) * chunk), true, [&](uint64_t k, size_t) {
						bf.set_atomic((uint64_t)hash32((uint32_t)k) % REF_BF_BITS);
						if (opt.write_lite) lite.set_atomic(hash40(k & 0xFFFFFFFFFFull) % REF_LITE_BF_BITS);
					});
				} catch (const Error &e) {
					#pragma omp critical
					err = e.msg;
				}
			}
			if (!err.empty()) die(err);
		}
		if (!opt.quiet) {
			printf("[BloomFilter constructBfFromGenomeseq] bit vector: %llu/%llu\n", (unsigned long long)bf.count(), (unsigned long long)REF_BF_BITS);
			if (opt.write_lite) printf("[BloomFilter constructBfFromGenomeseq] lite bit vector: %llu/%llu\n", (unsigned long long)lite.count(), (unsigned long long)REF_LITE_BF_BITS);
		}
		bf.save(prefix + ".ref.bf");
		if (opt.write_lite) lite.save(prefix + ".ref.bf.lite.bf");

		// constructBfFromVcf, generate_bf.cc:179-277
		BitVec sbf(SNP_BF_BITS);
		std::string pre_chr = "XO";
		const std::string *seq = nullptr;
		static const std::string empty;
		seq = &empty;
		size_t i = 0;
		while (i < vcf_text.size()) {
			size_t e = vcf_text.find('\n', i);
			if (e == std::string::npos) e = vcf_text.size();
			const size_t ls = i, le = e;
			i = e + 1;
			if (le == ls || vcf_text[ls] == '#') continue;
			// split(line, '\t')
			std::vector<std::pair<size_t, size_t>> col;
			for (size_t a = ls;;) {
				size_t b = vcf_text.find('\t', a);
				if (b == std::string::npos || b > le) { col.push_back({a, le}); break; }
				col.push_back({a, b}); a = b + 1;
			}
			if (col.size() < 5) continue;                                  // the reference would index past the vector
			std::string chr(vcf_text, col[0].first, col[0].second - col[0].first);
			if (chr.empty() || chr[0] != 'c') chr = "chr" + chr;
			const int pos = atoi(std::string(vcf_text, col[1].first, col[1].second - col[1].first).c_str()) - 1;
			const size_t rl = col[3].second - col[3].first, al = col[4].second - col[4].first;
			if (rl > 1 || al > 1) continue;
			if (chr != pre_chr) {
				for (const Seq &s : g) if (s.name == chr) { seq = &s.seq; break; }     // not found: previous sequence stays
				pre_chr = chr;
			}
			if (pos < 32 || (size_t)(pos + 32) > seq->size()) continue;
			if (rl == 0 || al == 0) continue;
			const char ref_nt = vcf_text[col[3].first], alt_nt = vcf_text[col[4].first];
			if (ref_nt != (*seq)[(size_t)pos] || ref_nt == alt_nt) continue;
			uint64_t k = 0; bool has_n = false;
			for (int j = 31; j >= 0 && !has_n; j--) {                        // encode_kmer scans from base 31 down
				const int c = base_code((unsigned char)(*seq)[(size_t)pos - 32 + (size_t)j]);
				if (c == 4) has_n = true;
				else if (c == 7) die("invalid base in reference sequence near a SNP");
				else k = (k << 2) | (uint64_t)c;
			}
			if (has_n) continue;
			for (unsigned t = 0; t < 32; t++) {
				const char nb = t ? (*seq)[(size_t)pos + t] : alt_nt;
				const int c = base_code((unsigned char)nb);
				if (c == 4) break;
				if (c == 7) die("invalid base while building the SNP bit vector");            // shift_kmer asserts, util.c:121
				sbf.set_atomic(hash40(k & 0xFFFFFFFFFFull) % SNP_BF_BITS);                       // B2: k never shifts
			}
		}
		if (!opt.quiet) printf("[BloomFilter constructBfFromVCF] bit vector: %llu/%llu\n", (unsigned long long)sbf.count(), (unsigned long long)SNP_BF_BITS);
		sbf.save(prefix + ".snp.bf");
	}

	// =============================== dictionaries (dictgen.c) ===============================
	std::vector<Seq> ref = parse_fasta_dict(fa);
	{
		FILE *f = fopen((prefix + ".chrlens").c_str(), "w");
		if (!f) die("cannot write " + prefix + ".chrlens");
		for (const Seq &s : ref) fprintf(f, "%s %lu\n", s.name.c_str(), (unsigned long)s.seq.size());     // qv.cc:2343-2345
		fclose(f);
	}
	if (ref.empty()) die("no sequences in " + fasta);

	// ---- SNP dictionary: make_snp_dict_from_vcf, dictgen.c:561-794
	{
		std::vector<SK> kmers;
		const bool ref_has_chr = !ref[0].name.empty() && ref[0].name[0] == 'c';
		int freq_index = -1; bool has_freq = true;
		const Seq *chrom = nullptr; uint32_t start_index = 1;
		std::string line; Fields fl; std::vector<size_t> tok;
		size_t i = 0;
		while (i < vcf_text.size()) {
			// fgets(line, 6000): at most 5999 characters per piece
			size_t e = vcf_text.find('\n', i);
			size_t le = (e == std::string::npos) ? vcf_text.size() : e + 1;
			if (le - i > 5999) le = i + 5999;
			line.assign(vcf_text, i, le - i);
			i = le;
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
					fprintf(stderr, "[Error] chromosome name %s in VCF file not found in reference genome FASTA file\n. Usually this is because the FASTA file has chromesome name as \"chr1\" while the VCF file has chromosome name as \"1\" without the \"chr\"\n", chrom_name);
					continue;
				}
			}
			const unsigned index = (unsigned)atoi(line.c_str() + fl.start[1]) - 1u;
			if (index >= chrom->seq.size() || toupper((unsigned char)chrom->seq[index]) != ref_base) {
				char msg[256];
				snprintf(msg, sizeof msg, "Mismatch found between reference sequence and SNP file at 0-based index %u in %s.", index, chrom->name.c_str());
				fprintf(stderr, "%s\n", msg);
				die(msg);
			}
			if (index < 32 || (size_t)index + 32 > chrom->seq.size()) continue;
			const char a2 = (char)toupper((unsigned char)fl.at(4, 0));
			if (!(ref_base == 'A' || ref_base == 'C' || ref_base == 'G' || ref_base == 'T')) continue;
			if (!(a2 == 'A' || a2 == 'C' || a2 == 'G' || a2 == 'T')) continue;
			// allele frequencies: vcf_split_line (dictgen.c:538-553) tokenises INFO on ';' and '=' and keeps going
			// into whatever follows the field; the token after the LAST one starting with "CAF" is "ref,alt".
			float freq1 = 0.5f, freq2 = 0.5f;
			if (has_freq) {
				tok.clear();
				size_t p = fl.start[7];
				auto ch = [&](size_t q) { return q < line.size() ? line[q] : '\0'; };
				while (ch(p) && !is_ws(ch(p))) {
					tok.push_back(p);
					while (ch(p) != ';' && ch(p) != '=') { if (ch(p) && !is_ws(ch(p))) ++p; else break; }
					++p;
				}
				for (size_t t = 0; t < tok.size(); t++) if (line.compare(tok[t], 3, "CAF") == 0) freq_index = (int)t + 1;
				if (freq_index == -1) has_freq = false;
			}
			if (has_freq) {
				// the reference reads info_split[freq_index], which may be a stale pointer when this line has fewer
				// tokens than the line that set freq_index; defined here as 0.5/0.5 for that case.
				if ((size_t)freq_index < tok.size()) {
					const char *p = line.c_str() + tok[(size_t)freq_index];
					freq1 = (float)atof(p);
					while (*p && *p != ',') p++;
					if (*p == ',') p++;
					freq2 = (float)atof(p);
				}
			}
			const uint8_t f1 = (uint8_t)(freq1 * 0xff), f2 = (uint8_t)(freq2 * 0xff);
			if (a2 == ref_base) continue;
			const std::string &seq = chrom->seq;
			uint64_t k = 0; bool had_n = false;
			for (int j = 31; j >= 0; j--) {
				const int c = base_code((unsigned char)seq[index - 32 + (unsigned)j]);
				if (c >= 4) { had_n = true; break; }
				k = (k << 2) | (uint64_t)c;
			}
			if (had_n) continue;
			SK tmp[32]; bool ok = true;
			for (unsigned t = 0; t < 32; t++) {
				const char nb = t ? seq[index + t] : a2;
				const int c = base_code((unsigned char)nb);
				if (c >= 4) { ok = false; break; }
				k = (k >> 2) | ((uint64_t)c << 62);
				tmp[t] = SK{k, start_index + index - 32 + 1 + t, (uint8_t)((((31 - t) & 0x1F) << 3) | ((unsigned)ref_u & 7)), f1, f2, 0};
			}
			if (!ok) continue;
			kmers.insert(kmers.end(), tmp, tmp + 32);
		}
		// qsort (glibc merge sort: stable) by k-mer; ties keep VCF order
		__gnu_parallel::stable_sort(kmers.begin(), kmers.end(), [](const SK &a, const SK &b) { return a.kmer < b.kmer; });
		std::vector<uint8_t> out, aux;
		out.reserve(16 + kmers.size() * 16);
		uint64_t zero = 0; put(out, &zero, 8); put(out, &zero, 8);
		uint64_t written = 0, aux_count = 0, unamb = 0, amb_unique = 0, amb_total = 0;
		for (size_t a = 0; a < kmers.size();) {
			size_t b = a + 1;
			while (b < kmers.size() && kmers[b].kmer == kmers[a].kmer) b++;
			put(out, &kmers[a].kmer, 8);
			const uint8_t z8 = 0, one = 1;
			if (b - a == 1) {
				unamb++;
				put(out, &kmers[a].pos, 4); put(out, &kmers[a].snp, 1); put(out, &z8, 1); put(out, &kmers[a].rf, 1); put(out, &kmers[a].af, 1);
			} else {
				amb_unique++; amb_total += b - a;
				uint32_t posv;
				if (b - a > (size_t)AUX_COLS) posv = POS_AMBIGUOUS;
				else {
					posv = (uint32_t)aux_count++;
					put(aux, &kmers[a].kmer, 8);
					for (int j = 0; j < AUX_COLS; j++) {
						if (a + (size_t)j < b) { const SK &s = kmers[a + (size_t)j]; put(aux, &s.pos, 4); put(aux, &s.snp, 1); put(aux, &s.rf, 1); put(aux, &s.af, 1); }
						else { const uint32_t z = 0; put(aux, &z, 4); put(aux, &z8, 1); put(aux, &z8, 1); put(aux, &z8, 1); }
					}
				}
				put(out, &posv, 4); put(out, &z8, 1); put(out, &one, 1); put(out, &z8, 1); put(out, &z8, 1);
			}
			written++;
			a = b;
		}
		memcpy(&out[0], &written, 8); memcpy(&out[8], &aux_count, 8);
		out.insert(out.end(), aux.begin(), aux.end());
		write_file(prefix + ".snp.dict", out);
		if (!opt.quiet) {
			printf("SNP Dictionary\nTotal k-mers:        %lu\nUnambig k-mers:      %lu\nAmbig unique k-mers: %lu\nAmbig total k-mers:  %lu\n",
			       (unsigned long)kmers.size(), (unsigned long)unamb, (unsigned long)amb_unique, (unsigned long)amb_total);
		}
	}

	// ---- reference dictionary: make_ref_dict, dictgen.c:277-301
	{
		for (const Seq &s : ref) if (s.seq.size() < 32) die("reference sequence shorter than 32 bases: " + s.name);   // assert, dictgen.c:17
		// count, then fill in parallel, chunk by chunk in position order
		struct Chunk { size_t seq, lo, hi; uint32_t base; size_t count, at; };
		std::vector<Chunk> chunks;
		uint32_t base = 1;
		for (size_t si = 0; si < ref.size(); si++) {
			const size_t nwin = ref[si].seq.size() - 31;
			const size_t step = 1 << 22;
			for (size_t lo = 0; lo < nwin; lo += step) chunks.push_back(Chunk{si, lo, std::min(nwin, lo + step), base, 0, 0});
			base += (uint32_t)ref[si].seq.size();
		}
		#pragma omp parallel for schedule(dynamic, 1)
		for (long c = 0; c < (long)chunks.size(); c++) {
			size_t cnt = 0;
			for_each_kmer(ref[chunks[(size_t)c].seq].seq, chunks[(size_t)c].lo, chunks[(size_t)c].hi, false, [&](uint64_t, size_t) { cnt++; });
			chunks[(size_t)c].count = cnt;
		}
		size_t total = 0;
		for (Chunk &c : chunks) { c.at = total; total += c.count; }
		std::vector<KP> kmers(total);
		#pragma omp parallel for schedule(dynamic, 1)
		for (long c = 0; c < (long)chunks.size(); c++) {
			const Chunk &ch = chunks[(size_t)c];
			size_t at = ch.at;
			for_each_kmer(ref[ch.seq].seq, ch.lo, ch.hi, false, [&](uint64_t k, size_t off) { kmers[at++] = KP{k, ch.base + (uint32_t)off, 0}; });
		}
		// stable by k-mer == ascending position within equal k-mers
		__gnu_parallel::sort(kmers.begin(), kmers.end(), [](const KP &a, const KP &b) { return a.kmer < b.kmer || (a.kmer == b.kmer && a.pos < b.pos); });
		// write_kmers (dictgen.c:63-154) in parallel: run starts -> unique index, runs of 2..10 -> aux row index
		const size_t nk = kmers.size();
		const int T = std::max(1, nthreads);
		std::vector<size_t> c_runs((size_t)T + 1, 0), c_aux((size_t)T + 1, 0);
		auto chunk_lo = [&](int t) { return nk * (size_t)t / (size_t)T; };
		#pragma omp parallel for schedule(static, 1)
		for (int t = 0; t < T; t++) {
			size_t runs = 0;
			for (size_t i = chunk_lo(t); i < chunk_lo(t + 1); i++) runs += (i == 0 || kmers[i].kmer != kmers[i - 1].kmer);
			c_runs[(size_t)t + 1] = runs;
		}
		for (int t = 0; t < T; t++) c_runs[(size_t)t + 1] += c_runs[(size_t)t];
		const uint64_t written = c_runs[(size_t)T];
		std::vector<uint32_t> S(written + 1);                           // start of run r (nk < 2^32 is checked by geno's loader anyway)
		S[written] = (uint32_t)nk;
		#pragma omp parallel for schedule(static, 1)
		for (int t = 0; t < T; t++) {
			size_t r = c_runs[(size_t)t];
			for (size_t i = chunk_lo(t); i < chunk_lo(t + 1); i++) if (i == 0 || kmers[i].kmer != kmers[i - 1].kmer) S[r++] = (uint32_t)i;
		}
		if (nk >= (1ull << 32)) die("more than 2^32 32-mers in the reference");
		auto run_lo = [&](int t) { return (size_t)written * (size_t)t / (size_t)T; };
		#pragma omp parallel for schedule(static, 1)
		for (int t = 0; t < T; t++) {
			size_t na = 0;
			for (size_t r = run_lo(t); r < run_lo(t + 1); r++) { const uint32_t L = S[r + 1] - S[r]; na += (L >= 2 && L <= (uint32_t)AUX_COLS); }
			c_aux[(size_t)t + 1] = na;
		}
		for (int t = 0; t < T; t++) c_aux[(size_t)t + 1] += c_aux[(size_t)t];
		const uint64_t aux_count = c_aux[(size_t)T];
		std::vector<uint8_t> out(16 + written * 13 + aux_count * 40);
		memcpy(&out[0], &written, 8); memcpy(&out[8], &aux_count, 8);
		uint64_t unamb = 0, amb_unique = 0, amb_total = 0;
		#pragma omp parallel for schedule(static, 1) reduction(+ : unamb, amb_unique, amb_total)
		for (int t = 0; t < T; t++) {
			size_t ax = c_aux[(size_t)t];
			for (size_t r = run_lo(t); r < run_lo(t + 1); r++) {
				const size_t a = S[r], bnd = S[r + 1], L = bnd - a;
				uint8_t *rec = &out[16 + r * 13];
				memcpy(rec, &kmers[a].kmer, 8);
				uint32_t posv; uint8_t flag;
				if (L == 1) { unamb++; posv = kmers[a].pos; flag = 0; }
				else {
					amb_unique++; amb_total += L; flag = 1;
					if (L > (size_t)AUX_COLS) posv = POS_AMBIGUOUS;
					else {
						posv = (uint32_t)ax;
						uint8_t *row = &out[16 + written * 13 + ax * 40];
						for (int j = 0; j < AUX_COLS; j++) { const uint32_t v = (size_t)j < L ? kmers[a + (size_t)j].pos : 0u; memcpy(row + 4 * j, &v, 4); }
						ax++;
					}
				}
				memcpy(rec + 8, &posv, 4); rec[12] = flag;
			}
		}
		write_file(prefix + ".ref.dict", out);
		if (!opt.quiet) {
			printf("Ref Dictionary\nTotal k-mers:        %lu\nUnambig k-mers:      %lu\nAmbig unique k-mers: %lu\nAmbig total k-mers:  %lu\n",
			       (unsigned long)total, (unsigned long)unamb, (unsigned long)amb_unique, (unsigned long)amb_total);
		}
	}
}

}  // namespace vgh
