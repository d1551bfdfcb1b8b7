// vg_hostpack.cpp -- see vg_hostpack.h.  Plain C++17 host code (no HIP).
//
// One chunk = two parallel sweeps over its text.  A record is four lines COUNTED FROM THE START OF THE STREAM (that is what
// four fgets() per record amount to on a well-formed file, and what the device-side framing does), so a thread that starts in
// the middle of the chunk must know how many lines lie before its piece: sweep 1 counts the line starts of every piece
// (16 bytes per compare), a prefix sum gives every piece the stream-wide number of its first line, and sweep 2 lets every
// thread frame and pack the records that START in its piece -- straight into per-thread buffers that a third, short sweep
// copies to their place in the caller's (pinned) staging arrays.  The record that straddles two chunks is put together from
// the carried bytes and handled by the calling thread.
#include "vg_hostpack.h"

#include <emmintrin.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace vgp {

namespace {

constexpr uint64_t MAX_LINE = 1023;          // fgets(buf, 1024, f): at most 1023 characters per call, newline included (qv.cc:700, 760-763)
constexpr uint64_t MAX_CARRY = 1u << 16;     // an unfinished record of more than 64 KiB: refused (as on the device)

// ---- a small pool: the caller is thread 0; workers spin briefly for the next job before they sleep (a chunk is ~1 ms of work) ----
class Pool {
public:
	explicit Pool(int n) : n_(n < 1 ? 1 : n)
	{
		for (int i = 1; i < n_; i++) th_.emplace_back([this, i] { worker(i); });
	}
	~Pool()
	{
		{ std::lock_guard<std::mutex> g(mu_); stop_ = true; gen_.fetch_add(1, std::memory_order_release); }
		cv_.notify_all();
		for (auto &t : th_) t.join();
	}
	int size() const { return n_; }
	void run(const std::function<void(int)> &f)
	{
		job_ = &f;
		pending_.store(n_ - 1, std::memory_order_release);
		{ std::lock_guard<std::mutex> g(mu_); gen_.fetch_add(1, std::memory_order_release); }
		cv_.notify_all();
		f(0);
		for (int spin = 0; pending_.load(std::memory_order_acquire) != 0; spin++) { if (spin > 2000) std::this_thread::yield(); }
	}
private:
	void worker(int id)
	{
		uint64_t seen = 0;
		for (;;) {
			uint64_t g = gen_.load(std::memory_order_acquire);
			for (int spin = 0; g == seen && spin < 20000; spin++) g = gen_.load(std::memory_order_acquire);
			if (g == seen) {
				std::unique_lock<std::mutex> lk(mu_);
				cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
				g = gen_.load(std::memory_order_acquire);
			}
			seen = g;
			if (stop_) return;
			(*job_)(id);
			pending_.fetch_sub(1, std::memory_order_acq_rel);
		}
	}
	const int n_;
	std::vector<std::thread> th_;
	std::mutex mu_;
	std::condition_variable cv_;
	std::atomic<uint64_t> gen_{0};
	std::atomic<int> pending_{0};
	const std::function<void(int)> *job_ = nullptr;
	bool stop_ = false;
};

// ---- text helpers ------------------------------------------------------------------------------------------------------------
inline uint64_t count_newlines(const uint8_t *p, uint64_t n)
{
	uint64_t c = 0, i = 0;
	const __m128i nl = _mm_set1_epi8('\n');
	for (; i + 64 <= n; i += 64) {
		const unsigned m0 = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(p + i)), nl));
		const unsigned m1 = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(p + i + 16)), nl));
		const unsigned m2 = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(p + i + 32)), nl));
		const unsigned m3 = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(p + i + 48)), nl));
		c += (uint64_t)__builtin_popcountll((uint64_t)m0 | ((uint64_t)m1 << 16) | ((uint64_t)m2 << 32) | ((uint64_t)m3 << 48));
	}
	for (; i < n; i++) c += p[i] == '\n';
	return c;
}

// 8 ASCII bases (little-endian in v) -> 16 bits, base 0 in bits 0-1 (encode_kmer, util.c:89-111: A0 C1 G2 T3, either case);
// `bad` collects a non-zero value when a byte is not one of ACGTacgt
inline uint32_t pack8(uint64_t v, uint64_t &bad)
{
	const uint64_t K01 = 0x0101010101010101ull, K7F = 0x7F7F7F7F7F7F7F7Full;
	const uint64_t u = v & 0xDFDFDFDFDFDFDFDFull;
	auto eq = [&](uint64_t c) { const uint64_t z = u ^ (c * K01); return ~(((z & K7F) + K7F) | z | K7F); };   // 0x80 in each byte equal to c
	bad |= (eq(0x41) | eq(0x43) | eq(0x47) | eq(0x54)) ^ 0x8080808080808080ull;
	uint64_t x = (v >> 1) & 0x0303030303030303ull;     // A0 C1 G3 T2
	x ^= (x >> 1) & K01;                               // A0 C1 G2 T3
	x = (x | (x >> 6)) & 0x000F000F000F000Full;
	x = (x | (x >> 12)) & 0x000000FF000000FFull;
	x = (x | (x >> 24)) & 0xFFFFull;
	return (uint32_t)x;
}
inline uint64_t pack32(const uint8_t *p, uint64_t &bad)
{
	uint64_t w[4];
	memcpy(w, p, 32);
	return (uint64_t)pack8(w[0], bad) | ((uint64_t)pack8(w[1], bad) << 16) | ((uint64_t)pack8(w[2], bad) << 32) | ((uint64_t)pack8(w[3], bad) << 48);
}
// the FIRST offending character in the reference's scan order decides (chunk 0 .. n-1, each from base 31 down to 0):
// N / n -> the read is skipped (qv.cc:815-828), anything else -> assert(0) (util.c:103)
inline uint64_t classify_bad(const uint8_t *p, uint32_t n)
{
	for (uint32_t c = 0; c < n; c++)
		for (int j = 31; j >= 0; j--) {
			const uint8_t ch = p[32 * c + j] & 0xDF;
			if (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T') continue;
			return ch == 'N' ? META_SKIP_N : META_INVALID;
		}
	return 0;
}

struct alignas(128) Out {                   // what one thread framed (its own cache lines: the vectors' end pointers move with every record)
	std::vector<uint64_t> kmers, meta;
	std::vector<uint8_t> nch;
	uint64_t n_invalid = 0;
	uint64_t last_rec = 0;                 // start (in the aligned text) of the last record framed here
	uint64_t stop_at = ~0ull;              // start of a record this thread found incomplete (the tail begins there)
	bool bad = false;
	void clear() { kmers.clear(); meta.clear(); nch.clear(); n_invalid = 0; last_rec = 0; stop_at = ~0ull; bad = false; }
};

enum Framed { REC_OK, REC_INCOMPLETE, REC_BAD };

// the record starting at a[r]: its four newlines (each line at most MAX_LINE characters, newline included), then its packed form
inline Framed frame_record(const uint8_t *a, uint64_t N, uint64_t r, uint64_t &next, Out &o)
{
	uint64_t e[4], pos = r;
	for (int l = 0; l < 4; l++) {
		const uint64_t room = N - pos, lim = room < MAX_LINE ? room : MAX_LINE;
		const uint8_t *q = lim ? (const uint8_t *)memchr(a + pos, '\n', (size_t)lim) : nullptr;
		if (!q) return room >= MAX_LINE ? REC_BAD : REC_INCOMPLETE;      // 1023 characters without a newline: beyond one fgets(); fewer: the text ends inside the record
		e[l] = (uint64_t)(q - a);
		pos = e[l] + 1;
	}
	next = pos;
	const uint64_t s1 = e[0] + 1, s3 = e[2] + 1;
	const uint64_t len = e[1] - s1, qlen = e[3] - s3;          // without their newlines
	const uint32_t n = (uint32_t)(len >> 5);                   // qv.cc:778-779: ((strlen(read) - 1) / 32) chunks
	if (qlen < n) return REC_BAD;                              // qual[c] would show the reference's stale buffer (qv.cc:836): host reader
	uint64_t bad = 0, meta = 0;
	for (uint32_t c = 0; c < n; c++) o.kmers.push_back(pack32(a + s1 + 32 * c, bad));
	for (uint32_t c = 0; c < n; c++) if ((int)(int8_t)a[s3 + c] - '8' < 0) meta |= 1ull << c;     // n <= 31: a line holds at most 1022 bases
	if (bad) { meta |= classify_bad(a + s1, n); if (meta & META_INVALID) o.n_invalid++; }
	o.meta.push_back(meta);
	o.nch.push_back((uint8_t)n);
	o.last_rec = r;
	return REC_OK;
}

}  // namespace

struct Packer::Impl {
	Pool pool;
	std::vector<Out> outs;                 // one per thread
	Out head;                              // the record put together from the carried bytes
	std::vector<uint8_t> carry, tmp;
	std::vector<uint64_t> starts, r0, c0;  // per piece: owned line starts; prefix sums of reads / chunks
	uint64_t stream_pos = 0, n_records = 0, n_consumed = 0, last_record = 0;
	bool poison = false;
	explicit Impl(int t) : pool(t), outs((size_t)(t < 1 ? 1 : t)) {}
};

Packer::Packer(int threads) : p(new Impl(threads)) {}
Packer::~Packer() { delete p; }
int Packer::threads() const { return p->pool.size(); }
uint64_t Packer::records() const { return p->n_records; }
uint64_t Packer::consumed() const { return p->n_consumed; }
uint64_t Packer::last_record_start() const { return p->last_record; }
bool Packer::poisoned() const { return p->poison; }

void Packer::begin()
{
	p->carry.clear();
	p->stream_pos = p->n_records = p->n_consumed = p->last_record = 0;
	p->poison = false;
}

ChunkResult Packer::push(const uint8_t *text, uint64_t nbytes, const Staging &out)
{
	Impl &s = *p;
	ChunkResult res;
	if (s.poison) { res.refused = true; return res; }
	auto refuse = [&]() { s.poison = true; res = ChunkResult(); res.refused = true; return res; };
	const uint64_t carry_len = s.carry.size();
	// ---- the record that began in an earlier chunk
	uint64_t head_len = 0;                                      // bytes of `text` that complete it
	s.head.clear();
	bool have_head = false;
	if (carry_len) {
		const uint64_t take = nbytes < 4 * (MAX_LINE + 1) ? nbytes : 4 * (MAX_LINE + 1);
		s.tmp.assign(s.carry.begin(), s.carry.end());
		s.tmp.insert(s.tmp.end(), text, text + take);
		uint64_t next = 0;
		const Framed f = frame_record(s.tmp.data(), s.tmp.size(), 0, next, s.head);
		if (f == REC_BAD) return refuse();
		if (f == REC_INCOMPLETE) {
			if (take < nbytes) return refuse();                 // four lines do not fit 4 x 1024 bytes: a line beyond fgets' reach
			s.carry.insert(s.carry.end(), text, text + nbytes); // still unfinished: keep collecting
			s.stream_pos += nbytes;
			if (s.carry.size() > MAX_CARRY) return refuse();
			return res;
		}
		head_len = next - carry_len;
		have_head = true;
	}
	// ---- the aligned rest: it starts at a record start
	const uint8_t *a = text + head_len;
	const uint64_t N = nbytes - head_len;
	const int T = s.pool.size();
	int np = (int)((N + 65535) / 65536);                        // pieces of at least 64 KiB
	if (np > T) np = T;
	if (np < 1) np = 1;
	s.starts.assign((size_t)np + 1, 0);
	auto bound = [&](int i) { return (uint64_t)i * N / (uint64_t)np; };
	// sweep 1: line starts owned by every piece (a line start at q is owned by the piece holding q; q = 0 belongs to piece 0)
	s.pool.run([&](int t) {
		for (int i = t; i < np; i += T) {
			const uint64_t b0 = bound(i), b1 = bound(i + 1);
			const uint64_t lo = b0 ? b0 - 1 : 0, hi = b1 ? b1 - 1 : 0;       // newlines at [lo, hi) open the line starts in [b0, b1) (b0 = 0: but for the start itself)
			s.starts[(size_t)i + 1] = (i == 0 ? 1u : 0u) + (hi > lo ? count_newlines(a + lo, hi - lo) : 0);
		}
	});
	for (int i = 0; i < np; i++) s.starts[(size_t)i + 1] += s.starts[(size_t)i];
	// sweep 2: every piece frames the records that start in it
	s.pool.run([&](int t) {
		Out &o = s.outs[(size_t)t];
		o.clear();
		if (t >= np || N == 0) return;
		// (np <= T: piece t is thread t's)
		const uint64_t b0 = bound(t), b1 = bound(t + 1);
		const uint64_t g0 = s.starts[(size_t)t];                // stream-wide number (within this aligned text) of the first line start owned here
		uint64_t skip = (4 - (g0 & 3)) & 3;                     // owned line starts to pass before one that opens a record
		uint64_t r;
		if (t == 0) r = 0;
		else {
			// owned line starts = positions after the newlines at [b0 - 1, b1 - 1)
			uint64_t pos = b0 - 1;
			r = ~0ull;
			while (pos < b1 - 1) {
				const uint8_t *q = (const uint8_t *)memchr(a + pos, '\n', (size_t)(b1 - 1 - pos));
				if (!q) break;
				const uint64_t at = (uint64_t)(q - a) + 1;        // a line start in [b0, b1)
				if (skip == 0) { r = at; break; }
				skip--;
				pos = at;
			}
			if (r == ~0ull) return;                             // no record starts in this piece
		}
		while (r < b1 && r < N) {
			uint64_t next = 0;
			const Framed f = frame_record(a, N, r, next, o);
			if (f == REC_BAD) { o.bad = true; return; }
			if (f == REC_INCOMPLETE) { o.stop_at = r; return; }
			r = next;
		}
	});
	uint64_t tail = N;                                          // where the unfinished last record begins (N: the text ends with a complete record)
	for (int t = 0; t < np; t++) {
		const Out &o = s.outs[(size_t)t];
		if (o.bad) return refuse();
		if (o.stop_at < tail) tail = o.stop_at;
	}
	if (N - tail > MAX_CARRY) return refuse();
	// ---- totals, then every thread copies its part to its place
	s.r0.assign((size_t)np + 1, 0); s.c0.assign((size_t)np + 1, 0);
	const uint64_t hr = have_head ? s.head.meta.size() : 0, hc = have_head ? s.head.kmers.size() : 0;
	s.r0[0] = hr; s.c0[0] = hc;
	for (int t = 0; t < np; t++) { s.r0[(size_t)t + 1] = s.r0[(size_t)t] + s.outs[(size_t)t].meta.size(); s.c0[(size_t)t + 1] = s.c0[(size_t)t] + s.outs[(size_t)t].kmers.size(); }
	const uint64_t R = s.r0[(size_t)np], C = s.c0[(size_t)np];
	if (R + 1 > out.reads_cap || C > out.kmers_cap) return refuse();     // lines of fewer than 8 bytes on average: not this path's business
	auto place = [&](const Out &o, uint64_t rr, uint64_t cc) {
		if (!o.kmers.empty()) memcpy(out.kmers + cc, o.kmers.data(), o.kmers.size() * 8);
		if (!o.meta.empty()) memcpy(out.meta + rr, o.meta.data(), o.meta.size() * 8);
		uint64_t c = cc;
		for (size_t k = 0; k < o.nch.size(); k++) { out.offsets[rr + k] = 32 * c; c += o.nch[k]; }
	};
	if (have_head) place(s.head, 0, 0);
	s.pool.run([&](int t) { if (t < np) place(s.outs[(size_t)t], s.r0[(size_t)t], s.c0[(size_t)t]); });
	out.offsets[R] = 32 * C;
	res.n_reads = R; res.n_chunks = C;
	res.n_invalid = have_head ? s.head.n_invalid : 0;
	uint64_t last_in_a = ~0ull;
	for (int t = 0; t < np; t++) { res.n_invalid += s.outs[(size_t)t].n_invalid; if (!s.outs[(size_t)t].meta.empty()) last_in_a = s.outs[(size_t)t].last_rec; }
	// ---- stream state
	if (last_in_a != ~0ull) s.last_record = s.stream_pos + head_len + last_in_a;
	else if (have_head) s.last_record = s.stream_pos - carry_len;
	s.carry.assign(a + tail, a + N);
	s.stream_pos += nbytes;
	s.n_consumed = s.stream_pos - s.carry.size();
	s.n_records += R;
	return res;
}

}  // namespace vgp
