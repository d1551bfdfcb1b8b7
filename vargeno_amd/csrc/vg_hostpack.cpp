// vg_hostpack.cpp -- see vg_hostpack.h.  Plain C++17 host code (no HIP).
//
// One chunk = two parallel sweeps over its text.  A record is four lines COUNTED FROM THE START OF THE STREAM (that is what
// four fgets() per record amount to on a well-formed file, and what the device-side framing does), so a thread that starts in
// the middle of the chunk must know how many lines lie before its piece: sweep 1 counts the line starts of every piece
// (16 bytes per compare), a prefix sum gives every piece the stream-wide number of its first line, and sweep 2 lets every
// thread frame and pack the records that START in its piece -- straight into per-thread buffers that a third, short sweep
// copies to their place in the caller's (pinned) staging arrays.  The record that straddles two chunks is put together from
// the carried bytes and handled by the calling thread.
#include "vg_hostpack_impl.h"

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace vgp {

namespace {

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
		for (int spin = 0; pending_.load(std::memory_order_acquire) != 0; spin++) { if (spin > 2000) std::this_thread::yield(); else __builtin_ia32_pause(); }
	}
private:
	void worker(int id)
	{
		uint64_t seen = 0;
		for (;;) {
			uint64_t g = gen_.load(std::memory_order_acquire);
			for (int spin = 0; g == seen && spin < 4000; spin++) { g = gen_.load(std::memory_order_acquire); __builtin_ia32_pause(); }
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

}  // namespace

// the baseline build of the hot loops lives in this file
#define VGP_NS base
#include "vg_hostpack_impl.inc"
#undef VGP_NS
static Framed frame_record_base(const uint8_t *a, uint64_t N, uint64_t r, uint64_t *next, Out &o) { return base::frame_record(a, N, r, *next, o); }
const Kernels &kernels_base()
{
	static const Kernels k{base::count_newlines, base::frame_piece, base::frame_piece_guess, frame_record_base, "sse"};
	return k;
}
static const Kernels &pick_kernels()
{
	__builtin_cpu_init();
	const char *e = getenv("VG_PACK_ISA");                      // "sse": the baseline build whatever the CPU (tests compare the two)
	if (e && !strcmp(e, "sse")) return kernels_base();
	return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") ? kernels_avx2() : kernels_base();
}

struct Packer::Impl {
	Pool pool;
	const Kernels &K;
	std::vector<Out> outs;                 // one per thread
	Out head;                              // the record put together from the carried bytes
	std::vector<uint8_t> carry, tmp;
	std::vector<uint64_t> starts, r0, c0;  // per piece: owned line starts; prefix sums of reads / chunks
	std::vector<uint64_t> gs, ge;          // per piece: where its guessed framing began / ended
	uint64_t two_sweep_blocks = 0;         // blocks whose guesses did not line up (framed the exact way)
	uint64_t stream_pos = 0, n_records = 0, n_consumed = 0, last_record = 0;
	bool poison = false;
	explicit Impl(int t) : pool(t), K(pick_kernels()), outs((size_t)(t < 1 ? 1 : t)) {}
	// one block: the complete records of carry + text[0, nbytes), placed at reads R0.. / chunks C0.. of the staging arrays
	bool block(const uint8_t *text, uint64_t nbytes, const Staging &out, uint64_t R0, uint64_t C0, ChunkResult &res);
};

Packer::Packer(int threads) : p(new Impl(threads)) {}
Packer::~Packer() { delete p; }
int Packer::threads() const { return p->pool.size(); }
const char *Packer::isa() const { return p->K.name; }
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

// A chunk goes through in BLOCKS of threads x 512 KiB, one piece per thread (should a block need the exact two sweeps, a piece
// stays in its thread's L2 in between).  A block the framing rules refuse poisons
// the stream from its first byte on; the records of the blocks before it stay framed (the host reader takes over at `consumed`).
ChunkResult Packer::push(const uint8_t *text, uint64_t nbytes, const Staging &out)
{
	Impl &s = *p;
	ChunkResult res;
	if (s.poison) { res.refused = true; return res; }
	const uint64_t BLOCK = (uint64_t)s.pool.size() << 19;
	for (uint64_t o = 0; o < nbytes && !s.poison; o += BLOCK) {
		const uint64_t len = nbytes - o < BLOCK ? nbytes - o : BLOCK;
		ChunkResult r;
		if (!s.block(text + o, len, out, res.n_reads, res.n_chunks, r)) { s.poison = true; break; }
		res.n_reads += r.n_reads; res.n_chunks += r.n_chunks; res.n_invalid += r.n_invalid;
	}
	out.offsets[res.n_reads] = 32 * res.n_chunks;
	res.refused = s.poison;
	return res;
}

bool Packer::Impl::block(const uint8_t *text, uint64_t nbytes, const Staging &out, uint64_t R0, uint64_t C0, ChunkResult &res)
{
	Impl &s = *this;
	const uint64_t carry_len = s.carry.size();
	// ---- the record that began in an earlier block
	uint64_t head_len = 0;                                      // bytes of `text` that complete it
	s.head.clear();
	bool have_head = false;
	if (carry_len) {
		const uint64_t take = nbytes < 4 * (MAX_LINE + 1) ? nbytes : 4 * (MAX_LINE + 1);
		s.tmp.assign(s.carry.begin(), s.carry.end());
		s.tmp.insert(s.tmp.end(), text, text + take);
		s.tmp.resize(s.tmp.size() + 64, 0);                     // (the vector compares read whole words: slack behind the text)
		uint64_t next = 0;
		const Framed f = s.K.frame_record(s.tmp.data(), carry_len + take, 0, &next, s.head);
		if (f == REC_BAD) return false;
		if (f == REC_INCOMPLETE) {
			if (take < nbytes) return false;                    // four lines do not fit 4 x 1024 bytes: a line beyond fgets' reach
			s.carry.insert(s.carry.end(), text, text + nbytes); // still unfinished: keep collecting
			s.stream_pos += nbytes;
			return s.carry.size() <= MAX_CARRY;
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
	s.gs.assign((size_t)np, 0); s.ge.assign((size_t)np, 0);
	auto bound = [&](int i) { return (uint64_t)i * N / (uint64_t)np; };
	// ONE sweep: every piece guesses where its first record starts and frames on from there; the guesses are then checked against
	// each other -- piece t must have begun exactly where the pieces before it ended (or framed nothing when they reach past it),
	// which by induction from piece 0 (it starts at a record start by construction) makes the framing the sequential one
	bool exact = getenv("VG_PACK_TWO_SWEEPS") != nullptr;
	if (!exact) {
		s.pool.run([&](int t) {
			Out &o = s.outs[(size_t)t];
			o.clear();
			if (t >= np || N == 0) return;
			s.K.frame_piece_guess(a, N, bound(t), bound(t + 1), t == 0, o, s.gs[(size_t)t], s.ge[(size_t)t]);
		});
		uint64_t cur = N ? s.ge[0] : 0;                             // where the next record starts, as far as the pieces checked so far go
		bool stopped = N == 0 || s.outs[0].stop_at != ~0ull || s.outs[0].bad;
		for (int t = 1; t < np && !exact; t++) {
			const Out &o = s.outs[(size_t)t];
			const bool owns = !stopped && cur < bound(t + 1);         // the sequential framing has a record start inside this piece
			if (owns) {
				if (s.gs[(size_t)t] != cur) exact = true;              // a wrong guess: count lines instead
				else { cur = s.ge[(size_t)t]; stopped = o.stop_at != ~0ull || o.bad; }
			} else if (o.rn != 0 || o.bad) exact = true;               // it framed something the sequential framing does not start here
		}
	}
	if (exact) {
		// the exact way: sweep 1 counts the line starts of every piece, a prefix sum numbers them, sweep 2 frames from the first line
		// start whose number is a multiple of four
		s.pool.run([&](int t) {
			if (t >= np) return;
			const uint64_t b0 = bound(t), b1 = bound(t + 1);
			const uint64_t lo = b0 ? b0 - 1 : 0, hi = b1 ? b1 - 1 : 0;          // newlines at [lo, hi) open the line starts in [b0, b1) (b0 = 0: but for the start itself)
			s.starts[(size_t)t + 1] = (t == 0 ? 1u : 0u) + (hi > lo ? s.K.count_newlines(a + lo, hi - lo) : 0);
		});
		for (int i = 0; i < np; i++) s.starts[(size_t)i + 1] += s.starts[(size_t)i];
		s.pool.run([&](int t) {
			Out &o = s.outs[(size_t)t];
			o.clear();
			if (t >= np || N == 0) return;
			s.K.frame_piece(a, N, bound(t), bound(t + 1), s.starts[(size_t)t], t == 0, o);
		});
		s.two_sweep_blocks++;
	}
	uint64_t tail = N;                                          // where the unfinished last record begins (N: the text ends with a complete record)
	for (int t = 0; t < np; t++) {
		const Out &o = s.outs[(size_t)t];
		if (o.bad) return false;
		if (o.stop_at < tail) tail = o.stop_at;
	}
	if (N - tail > MAX_CARRY) return false;
	// ---- totals, then every thread copies its part to its place
	s.r0.assign((size_t)np + 1, 0); s.c0.assign((size_t)np + 1, 0);
	s.r0[0] = R0 + (have_head ? s.head.rn : 0); s.c0[0] = C0 + (have_head ? s.head.kn : 0);
	for (int t = 0; t < np; t++) { s.r0[(size_t)t + 1] = s.r0[(size_t)t] + s.outs[(size_t)t].rn; s.c0[(size_t)t + 1] = s.c0[(size_t)t] + s.outs[(size_t)t].kn; }
	const uint64_t R1 = s.r0[(size_t)np], C1 = s.c0[(size_t)np];
	if (R1 + 1 > out.reads_cap || C1 > out.kmers_cap) return false;      // lines of fewer than 8 bytes on average: not this path's business
	auto place = [&](const Out &o, uint64_t rr, uint64_t cc) {
		if (o.kn) memcpy(out.kmers + cc, o.kmers, o.kn * 8);
		if (o.rn) memcpy(out.meta + rr, o.meta, o.rn * 8);
		uint64_t c = cc;
		for (uint64_t k = 0; k < o.rn; k++) { out.offsets[rr + k] = 32 * c; c += o.nch[k]; }
	};
	if (have_head) place(s.head, R0, C0);
	s.pool.run([&](int t) { if (t < np) place(s.outs[(size_t)t], s.r0[(size_t)t], s.c0[(size_t)t]); });
	res.n_reads = R1 - R0; res.n_chunks = C1 - C0;
	res.n_invalid = have_head ? s.head.n_invalid : 0;
	uint64_t last_in_a = ~0ull;
	for (int t = 0; t < np; t++) { res.n_invalid += s.outs[(size_t)t].n_invalid; if (s.outs[(size_t)t].rn) last_in_a = s.outs[(size_t)t].last_rec; }
	// ---- stream state
	if (last_in_a != ~0ull) s.last_record = s.stream_pos + head_len + last_in_a;
	else if (have_head) s.last_record = s.stream_pos - carry_len;
	s.carry.assign(a + tail, a + N);
	s.stream_pos += nbytes;
	s.n_consumed = s.stream_pos - s.carry.size();
	s.n_records += res.n_reads;
	return true;
}

}  // namespace vgp
