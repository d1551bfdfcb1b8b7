// caller_vcf.cpp -- genotype caller and VCF annotator of the drop-in `vargeno geno`.
//
// Behaviour to reproduce (reference src/qv.cc:1573-1747 caller loop + VCF pass, :1789-1848 posterior):
//   * per SNP site: likelihood of (ref_cnt, alt_cnt) under hom-ref / het / hom-alt with a 1 % error rate, Hardy-Weinberg
//     prior from the two allele frequencies stored as n/255, times a Poisson(7.1) depth term; a site with no reads or
//     with both counters saturated is not called; GQ = (int)(-10 ln(confidence));
//   * the SNP list is echoed with GT:GQ for every record whose "chr<CHROM>$<POS>" names a called site; other records
//     are dropped; FORMAT header lines are injected unless the input declares them.
// The arithmetic is IEEE double through libm in the reference's operation order (GQ truncates, so a last-bit
// difference can show).  Everything else is organised for a 10 M-record dbSNP file: calls live in per-chromosome sorted
// arrays instead of a string-keyed hash map, the input is scanned in place, and record chunks are annotated by all
// host threads.
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "vg_host.h"

namespace vgh {

// ------------------------------------------------------------------------------------------------
// posterior
// ------------------------------------------------------------------------------------------------
namespace {

constexpr unsigned kCap = 63;            // counter saturation, src/vartype.h:27
constexpr double kErr = 0.01;            // src/vartype.h:13
constexpr double kMeanDepth = 7.1;       // src/vartype.h:14

// Per-count factors.  The reference tabulates the three likelihoods per (ref_cnt, alt_cnt) pair; a product of two
// tabulated powers is the same pair of libm calls and the same multiply, so 3 x 64 numbers replace 3 x 4096.
struct Factors {
	double keep[kCap + 1];               // (1 - e)^k
	double flip[kCap + 1];               // e^k
	double half[2 * kCap + 1];           // 0.5^k
	double depth[2 * kCap + 1];          // Poisson(7.1) mass at k
	Factors()
	{
		for (unsigned k = 0; k <= kCap; k++) { keep[k] = pow(1.0 - kErr, (int)k); flip[k] = pow(kErr, (int)k); }
		const double m = exp(-kMeanDepth);
		for (unsigned k = 0; k <= 2 * kCap; k++) {
			half[k] = pow(0.5, (int)k);
			depth[k] = (m * pow(kMeanDepth, (int)k)) / exp(lgamma(k + 1.0));
		}
	}
};
const Factors &factors() { static const Factors f; return f; }

}  // namespace

Genotype call_genotype(unsigned ref_cnt, unsigned alt_cnt, uint8_t ref_freq, uint8_t alt_freq)
{
	Genotype out{GT_NONE, 0.0};
	if (ref_cnt > kCap) ref_cnt = kCap;
	if (alt_cnt > kCap) alt_cnt = kCap;
	if ((ref_cnt | alt_cnt) == 0 || (ref_cnt == kCap && alt_cnt == kCap)) return out;
	const Factors &f = factors();
	const double pr = ref_freq / 255.0, pa = alt_freq / 255.0;
	const double rr = pr * pr, aa = pa * pa;
	const double w[3] = {
		rr * (f.keep[ref_cnt] * f.flip[alt_cnt]),            // hom-ref
		(1.0 - rr - aa) * f.half[ref_cnt + alt_cnt],         // het
		aa * (f.flip[ref_cnt] * f.keep[alt_cnt]),            // hom-alt
	};
	const double sum = w[0] + w[1] + w[2];
	// strict maximum wins, hom-ref tested before het; anything else (ties included) is hom-alt
	int best = 2;
	if (w[0] > w[1] && w[0] > w[2]) best = 0;
	else if (w[1] > w[0] && w[1] > w[2]) best = 1;
	out.gt = best == 0 ? GT_HOM_REF : best == 1 ? GT_HET : GT_HOM_ALT;
	out.confidence = (w[best] / sum) * f.depth[ref_cnt + alt_cnt];
	return out;
}

int genotype_quality(double confidence) { return (int)(-10 * log(confidence)); }

// ------------------------------------------------------------------------------------------------
// chromosome table (<prefix>.chrlens: "name length" per line; a name is at most 32 non-space characters)
// ------------------------------------------------------------------------------------------------
std::vector<ChrLen> read_chrlens(const std::string &path)
{
	FILE *f = fopen(path.c_str(), "r");
	if (!f) throw Error{"cannot open " + path};
	std::vector<ChrLen> table;
	char line[256];
	while (fgets(line, sizeof line, f)) {
		const char *p = line;
		while (*p && !isspace((unsigned char)*p) && p - line < 32) p++;
		ChrLen c{std::string((const char *)line, (size_t)(p - line)), 0};
		while (isspace((unsigned char)*p)) p++;
		c.len = (uint64_t)atol(p);
		table.push_back(c);
	}
	fclose(f);
	return table;
}

// ------------------------------------------------------------------------------------------------
// called sites, addressable the way the reference's "name$position" keys are
// ------------------------------------------------------------------------------------------------
namespace {

struct CalledSite {
	uint64_t local;        // 1-based position inside its chromosome
	uint32_t order;        // rank in genome order (a later site replaces an earlier one under the same key)
	uint8_t gt;
	int gq;
};

class CallBook {
public:
	CallBook(const SiteCounts &s, const std::vector<ChrLen> &chrs, CallSummary &sum)
	{
		// the calls themselves are independent of one another: threads, a slice of the sites each (r05: 10 M sites at 30-fold coverage
		// are all genotyped; one thread took 0.3 s of the job's tail for them)
		const size_t ns = s.pos.size();
		std::vector<uint8_t> gts(ns);
		std::vector<int> gqs(ns);
		{
			unsigned nt = std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
			if (ns < (1u << 16)) nt = 1;
			auto work = [&](unsigned t) {
				for (size_t i = ns * t / nt; i < ns * (t + 1) / nt; i++) {
					const Genotype k = call_genotype(s.ref_cnt[i], s.alt_cnt[i], s.ref_freq[i], s.alt_freq[i]);
					gts[i] = k.gt;
					gqs[i] = k.gt == GT_NONE ? 0 : genotype_quality(k.confidence);
				}
			};
			if (nt == 1) work(0);
			else { std::vector<std::thread> th; for (unsigned t = 0; t < nt; t++) th.emplace_back(work, t); for (auto &x : th) x.join(); }
		}
		// sites ascend over the concatenated genome: walk the chromosome table alongside them
		size_t c = 0, c_of_list = SIZE_MAX;
		std::vector<CalledSite> *list = nullptr;                 // by_name_[chrs[c].name], looked up once per chromosome
		uint64_t before = 0;                                     // bases in chromosomes 0 .. c-1
		for (size_t i = 0; i < ns; i++) {
			const uint64_t g = s.pos[i];
			while (c < chrs.size() && g - before > chrs[c].len) before += chrs[c++].len;
			struct { uint8_t gt; } k{gts[i]};
			if (k.gt == GT_NONE) continue;
			(k.gt == GT_HOM_REF ? sum.ref : k.gt == GT_HOM_ALT ? sum.alt : sum.het)++;
			if (c == chrs.size()) continue;                      // beyond the last chromosome: no name to print it under
			if (c != c_of_list) { list = &by_name_[chrs[c].name]; c_of_list = c; list->reserve(list->size() + (ns - i) / 4); }
			list->push_back(CalledSite{g - before, (uint32_t)i, (uint8_t)k.gt, gqs[i]});
		}
		// two chromosomes with one name share a key space; keep, per position, the site that comes last in genome order
		for (auto &kv : by_name_) {
			std::vector<CalledSite> &v = kv.second;
			if (std::is_sorted(v.begin(), v.end(), [](const CalledSite &a, const CalledSite &b) { return a.local < b.local; }) &&
			    std::adjacent_find(v.begin(), v.end(), [](const CalledSite &a, const CalledSite &b) { return a.local == b.local; }) == v.end())
				continue;
			std::stable_sort(v.begin(), v.end(), [](const CalledSite &a, const CalledSite &b) { return a.local != b.local ? a.local < b.local : a.order < b.order; });
			size_t w = 0;
			for (size_t r = 0; r < v.size(); r++) { if (w && v[w - 1].local == v[r].local) v[w - 1] = v[r]; else v[w++] = v[r]; }
			v.resize(w);
		}
	}

	// The reference compares the strings  name + "$" + decimal(position)  and  chrom + "$" + POS-column.  A decimal number
	// holds no '$', so the two are equal exactly when the text after the LAST '$' of the right-hand side is the canonical
	// decimal form of the position and the text before it is the name.
	// (a SNP list is nearly always sorted: the caller's Hint remembers the list of the last name and the place of the last hit in
	// it, and the search gallops forward from there; any other order merely falls back to the full search)
	struct Hint { std::string name; const std::vector<CalledSite> *list = nullptr; size_t at = 0; };
	const CalledSite *find(const std::string &key, Hint &h) const
	{
		const size_t cut = key.rfind('$');
		const char *d = key.c_str() + cut + 1;
		const size_t nd = key.size() - cut - 1;
		if (nd == 0 || nd > 19 || (d[0] == '0' && nd > 1)) return nullptr;
		uint64_t v = 0;
		for (size_t i = 0; i < nd; i++) { if (d[i] < '0' || d[i] > '9') return nullptr; v = v * 10 + (uint64_t)(d[i] - '0'); }
		if (!(h.list && h.name.size() == cut && key.compare(0, cut, h.name) == 0)) {
			h.name.assign(key, 0, cut);
			const auto it = by_name_.find(h.name);
			h.list = it == by_name_.end() ? nullptr : &it->second;
			h.at = 0;
		}
		if (!h.list) return nullptr;
		const std::vector<CalledSite> &list = *h.list;
		auto lo = list.begin(), hi = list.end();
		if (h.at < list.size() && list[h.at].local <= v) {
			// gallop: the answer is at or after the last hit
			size_t step = 1, a = h.at;
			while (a + step < list.size() && list[a + step].local < v) { a += step; step *= 2; }
			lo = list.begin() + (ptrdiff_t)a;
			hi = list.begin() + (ptrdiff_t)std::min(list.size(), a + step + 1);
		}
		const auto at = std::lower_bound(lo, hi, v, [](const CalledSite &a, uint64_t x) { return a.local < x; });
		if (at == list.end() || at->local != v) return nullptr;
		h.at = (size_t)(at - list.begin());
		return &*at;
	}

private:
	std::map<std::string, std::vector<CalledSite>> by_name_;
};

// ------------------------------------------------------------------------------------------------
// the VCF pass
// ------------------------------------------------------------------------------------------------
const char kGtDecl[] = "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n";
const char kGqDecl[] = "##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype Quality\">\n";

struct Span {
	const char *b, *e;
	size_t size() const { return (size_t)(e - b); }
	bool is(const char *lit) const { return size() == strlen(lit) && memcmp(b, lit, size()) == 0; }
};

// What the header decided for every record that follows it.
struct Layout {
	bool declares_gt = false, declares_gq = false;   // the input already has "ID=GT," / "ID=GQ," meta lines
	bool sample_columns = true;                      // the #CHROM line has FORMAT + a sample column (>= 10 fields)
	int gt_slot = -1, gq_slot = -1;                  // sub-field of the sample column to overwrite; settled by the first annotated record
};

size_t count_fields(Span s, char sep) { return (size_t)std::count(s.b, s.e, sep) + 1; }
Span field(Span s, char sep, size_t k)            // k-th sep-separated field (must exist)
{
	const char *p = s.b;
	for (; k; k--) p = (const char *)memchr(p, sep, (size_t)(s.e - p)) + 1;
	const char *q = (const char *)memchr(p, sep, (size_t)(s.e - p));
	return Span{p, q ? q : s.e};
}
int slot_of(Span format, const char *tag)
{
	const size_t n = count_fields(format, ':');
	for (size_t k = 0; k < n; k++) if (field(format, ':', k).is(tag)) return (int)k;
	return -1;
}

const char *gt_text(uint8_t gt) { return gt == GT_HET ? "0/1" : gt == GT_HOM_ALT ? "1/1" : "0/0"; }

// One data line -> `out` (nothing if its key names no called site).  Returns false if the layout's slots are still open
// and this record would have to settle them (the caller does that on one thread, in file order).
struct Scratch { std::string key, fmt, smp; CallBook::Hint hint; };           // one per thread: no allocation per line
bool annotate(Span line, const CallBook &book, const Layout &lay, bool may_settle, Layout *settled, std::string &out, Scratch &sc)
{
	const size_t nf = count_fields(line, '\t');
	if (nf < 2) return true;
	const Span chrom = field(line, '\t', 0), pos = field(line, '\t', 1);
	std::string &key = sc.key;
	key.clear();
	if (chrom.size() == 0 || chrom.b[0] != 'c') key = "chr";
	key.append(chrom.b, chrom.e).push_back('$');
	key.append(pos.b, pos.e);
	const CalledSite *site = book.find(key, sc.hint);
	if (!site) return true;

	const bool in_place = lay.sample_columns && nf >= 10;      // rewrite fields 8 and 9; otherwise append two fields
	int gt_slot = lay.gt_slot, gq_slot = lay.gq_slot;
	if ((lay.declares_gt && gt_slot < 0) || (lay.declares_gq && gq_slot < 0)) {
		if (!may_settle) return false;
		const Span fmt = in_place ? field(line, '\t', 8) : Span{line.e, line.e};
		if (lay.declares_gt && gt_slot < 0) gt_slot = slot_of(fmt, "GT");
		if (lay.declares_gq && gq_slot < 0) gq_slot = slot_of(fmt, "GQ");
		// the reference asserts that a declared GT is present in the first annotated record's FORMAT (qv.cc:1701); a declared
		// GQ drives it into an out-of-range write (it never looks the column up), so there the intent is followed instead
		if ((lay.declares_gt && gt_slot < 0) || (lay.declares_gq && gq_slot < 0))
			throw Error{"the SNP list declares GT/GQ in its header but the FORMAT column of a genotyped record lacks it"};
		settled->gt_slot = gt_slot; settled->gq_slot = gq_slot;
	}
	char gq_text[16];
	snprintf(gq_text, sizeof gq_text, "%d", site->gq);

	std::string &fmt = sc.fmt, &smp = sc.smp;
	fmt.clear(); smp.clear();
	size_t n_sub = 0;
	if (in_place) {
		const Span f8 = field(line, '\t', 8), f9 = field(line, '\t', 9);
		fmt.assign(f8.b, f8.e);
		n_sub = count_fields(f9, ':');
		for (size_t k = 0; k < n_sub; k++) {
			if (k) smp.push_back(':');
			if (lay.declares_gt && (int)k == gt_slot) smp += gt_text(site->gt);
			else if (lay.declares_gq && (int)k == gq_slot) smp += gq_text;
			else { const Span v = field(f9, ':', k); smp.append(v.b, v.e); }
		}
		if ((lay.declares_gt && (size_t)gt_slot >= n_sub) || (lay.declares_gq && (size_t)gq_slot >= n_sub))
			throw Error{"a genotyped record's sample column has fewer sub-fields than its FORMAT"};
	}
	auto push = [&](const char *tag, const char *val) {
		if (!fmt.empty() || n_sub) { fmt.push_back(':'); smp.push_back(':'); }
		fmt += tag; smp += val;
		n_sub++;
	};
	if (!lay.declares_gt) push("GT", gt_text(site->gt));
	if (!lay.declares_gq) push("GQ", gq_text);

	if (in_place) {
		const Span f8 = field(line, '\t', 8), f9 = field(line, '\t', 9);
		out.append(line.b, f8.b).append(fmt).push_back('\t');
		out.append(smp).append(f9.e, line.e);
	} else {
		out.append(line.b, line.e).push_back('\t');
		out.append(fmt).push_back('\t');
		out.append(smp);
	}
	out.push_back('\n');
	return true;
}

std::string slurp(const std::string &path, bool &ok)
{
	std::string text;
	FILE *f = fopen(path.c_str(), "rb");
	ok = f != nullptr;
	if (!f) return text;
	char buf[1 << 16];
	size_t got;
	while ((got = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, got);
	fclose(f);
	return text;
}

inline Span next_line(const char *&p, const char *end)
{
	const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
	Span s{p, nl ? nl : end};
	p = nl ? nl + 1 : end;
	return s;
}

}  // namespace

// Lines that are read by one thread in file order: meta lines ("##..."), column headers ("#..."), and -- while the input
// declares GT/GQ but no annotated record has shown yet which sub-field holds them -- data lines.  Stops in front of the
// first data line that can be annotated independently of its neighbours.
static void ordered_lines(const char *&p, const char *end, Layout &lay, const CallBook &book, std::string &dst)
{
	while (p < end) {
		const char *at = p;
		const Span ln = next_line(p, end);
		if (ln.size() == 0) continue;
		if (ln.b[0] != '#') {
			const bool slots_open = (lay.declares_gt && lay.gt_slot < 0) || (lay.declares_gq && lay.gq_slot < 0);
			if (!slots_open) { p = at; return; }
			Scratch sc;
			annotate(ln, book, lay, true, &lay, dst, sc);
		} else if (ln.size() > 1 && ln.b[1] == '#') {
			dst.append(ln.b, ln.e).push_back('\n');
			const std::string meta(ln.b, ln.e);
			if (meta.find("ID=GT,") != std::string::npos) lay.declares_gt = true;
			else if (meta.find("ID=GQ,") != std::string::npos) lay.declares_gq = true;
		} else {
			if (!lay.declares_gt) dst += kGtDecl;
			if (!lay.declares_gq) dst += kGqDecl;
			dst.append(ln.b, ln.e);
			if (count_fields(ln, '\t') < 10) { lay.sample_columns = false; dst += "\tFORMAT\tDONOR"; }
			dst.push_back('\n');
		}
	}
}

bool read_whole_file(const std::string &path, std::string &text)
{
	bool ok = false;
	text = slurp(path, ok);
	return ok;
}

CallSummary write_genotyped_vcf(const SiteCounts &s, const std::vector<ChrLen> &chrlens, const std::string &vcf_in, const std::string &vcf_out, const std::string *vcf_text)
{
	CallSummary sum;
	const bool clocks = getenv("VARGENO_VCF_CLOCKS") != nullptr;
	struct timespec c0, c1, c2; clock_gettime(CLOCK_MONOTONIC, &c0);
	double t_annot = 0, t_write = 0;
	auto lap = [](const timespec &a, const timespec &b) { return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec); };
	const CallBook book(s, chrlens, sum);
	clock_gettime(CLOCK_MONOTONIC, &c1);
	bool ok = vcf_text != nullptr;
	std::string own;
	if (!vcf_text) own = slurp(vcf_in, ok);
	const std::string &text = vcf_text ? *vcf_text : own;
	struct timespec c1b; clock_gettime(CLOCK_MONOTONIC, &c1b);
	if (!ok) { fprintf(stderr, "Error opening: %s . You have failed.\n", vcf_in.c_str()); return sum; }
	FILE *out = fopen(vcf_out.c_str(), "wb");
	if (!out) throw Error{"cannot write " + vcf_out};
	unsigned threads = std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
	if (const char *e = getenv("VARGENO_THREADS")) if (atoi(e) > 0) threads = (unsigned)atoi(e);

	Layout lay;
	const char *p = text.data(), *const end = p + text.size();
	while (p < end) {
		std::string seq;
		ordered_lines(p, end, lay, book, seq);
		fwrite(seq.data(), 1, seq.size(), out);
		// a run of data lines: up to the next line that starts with '#' (the reference re-reads header lines wherever they
		// stand, so they fence the run), cut at line starts into one piece per thread
		const char *stop = end;
		for (const char *q = p; q < end;) {
			const char *h = (const char *)memchr(q, '#', (size_t)(end - q));
			if (!h) break;
			if (h == text.data() || h[-1] == '\n') { stop = h; break; }
			q = h + 1;
		}
		const size_t bytes = (size_t)(stop - p);
		const unsigned nt = bytes < (1u << 20) ? 1u : threads;
		std::vector<const char *> cut(nt + 1, stop);
		cut[0] = p;
		for (unsigned t = 1; t < nt; t++) {
			const char *c = p + bytes / nt * t;
			const char *nl = (const char *)memchr(c, '\n', (size_t)(stop - c));
			cut[t] = nl ? nl + 1 : stop;
		}
		std::vector<std::string> piece(nt), err(nt);
		auto work = [&](unsigned t) {
			const char *q = cut[t], *const e = cut[t + 1];
			piece[t].reserve((size_t)(e - q) + (size_t)(e - q) / 4 + 4096);          // (a line grows by ~":GT:GQ" + the two values)
			Scratch sc;
			try {
				while (q < e) { const Span ln = next_line(q, e); if (ln.size()) annotate(ln, book, lay, false, nullptr, piece[t], sc); }
			} catch (const Error &x) { err[t] = x.msg; }
		};
		if (nt == 1) work(0);
		else {
			std::vector<std::thread> th;
			for (unsigned t = 0; t < nt; t++) th.emplace_back(work, t);
			for (auto &x : th) x.join();
		}
		for (unsigned t = 0; t < nt; t++) if (!err[t].empty()) { fclose(out); throw Error{err[t]}; }
		clock_gettime(CLOCK_MONOTONIC, &c2);
		for (unsigned t = 0; t < nt; t++) fwrite(piece[t].data(), 1, piece[t].size(), out);          // (positioned writes by several threads are slower: one inode lock)
		struct timespec c3; clock_gettime(CLOCK_MONOTONIC, &c3);
		t_write += lap(c2, c3);
		p = stop;
	}
	if (fclose(out) != 0) throw Error{"cannot write " + vcf_out};
	if (clocks) { struct timespec c4; clock_gettime(CLOCK_MONOTONIC, &c4); t_annot = lap(c1b, c4) - t_write; fprintf(stderr, "vcf: calls + book %.3f s, SNP list read %.3f s, lines %.3f s, write %.3f s\n", lap(c0, c1), lap(c1, c1b), t_annot, t_write); }
	return sum;
}

}  // namespace vgh
