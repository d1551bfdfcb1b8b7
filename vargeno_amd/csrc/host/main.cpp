// main.cpp -- the `vargeno` command line of the drop-in (reference front-end: src/qv.cc:1853-2395).
//   vargeno index <ref.fa> <snps.vcf> <prefix>
//   vargeno geno  <prefix> <reads.fq> <snps.vcf> <out.vcf>
// Same positional arguments, file names, messages and exit codes as upstream.  `geno` drives the
// HIP library through the C-ABI of include/vargeno_hip.h only.  Extra knobs come from the
// environment so that the argument list stays the reference's:
//   VARGENO_GPUS=n        n index replicas, one per GPU of this node (default 1): each streams its own record-aligned range of
//                         the FASTQ file, counters summed with RCCL
//   VARGENO_SHARE_DEVICES=1  allow more replicas than GPUs (replica g on device g % GPUs): small indexes, one-GPU test boxes
//   VARGENO_BATCH=n       reads per batch of the host-framed path (default 4194304)
//   VARGENO_CHUNK_MB=n    FASTQ bytes per chunk handed to the library (default 64; 256 when the host packs)
//   VARGENO_PACK_THREADS=n  host threads (per replica) that frame and 2-bit pack the FASTQ text, so that 48 bytes per read cross the
//                         link instead of ~315 of text (default: the CPUs the process may use -- a cgroup quota counts -- less two,
//                         shared among the replicas, at most 96).  They start BEFORE the index is opened and pack beside it; when the
//                         index is ready their measured rate is compared with the link's (vg_link_rate) and the faster route --
//                         host packing, or the text framed on the device -- takes the rest of the file.  0: device framing only
//   VARGENO_PREPACK=0     do not pack ahead of the index (and do not measure: host packing if VARGENO_PACK_THREADS > 0, else device framing)
//   VARGENO_PREPACK_GB=n  device memory the packed-ahead reads may take per device (the read store; default 16, at most an eighth of the device)
//   VARGENO_PREPACK_BYTES=n  the read store's size in bytes, exactly (tests: a store that fills up in the middle of a small file)
//   VARGENO_PREPACK_MMAP=0  the pre-packer reads the file with pread into buffers of its own instead of mapping it
//   VARGENO_ORDERLY_EXIT=1  close the handles and let the runtime shut down before the process ends (default: it ends when the VCF is closed)
//   VARGENO_VCF_CLOCKS=1  stderr: the seconds of the caller / VCF pass, phase by phase
//   VARGENO_READERS=n     threads reading the FASTQ file into pinned chunk buffers (default: an eighth of the hardware threads, 8 to 32)
//   VARGENO_MAX_DEVICE_GB=x  device-memory budget per replica (vg_index_open_ex): which re-laid-out views the replica holds follows
//                         from the index and this number alone (default: the whole device); VARGENO_VERBOSE=1 prints the plan
//   VARGENO_DUMP_COUNTS=path  also write the per-site counters the caller is given (ref counts then alt counts, one byte per site, site order of the index)
//   VARGENO_HOST_FASTQ=1  frame the FASTQ on the host (the reference's four fgets per record) instead of on the device
//   VARGENO_NO_LITE=1     index: skip <prefix>.ref.bf.lite.bf (2.3 GB, read by nothing in geno)
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/vargeno_hip.h"
#include "vg_host.h"

static void print_help()
{
	fprintf(stderr, "Usage: vargeno <option> [option parameters ...]\n");
	fprintf(stderr, "Option  Description                   Parameters\n");
	fprintf(stderr, "------  -----------                   ----------\n");
	fprintf(stderr, "index   Generate index            <input FASTA> <input SNPs in VCF> <index_prefix>\n");
	fprintf(stderr, "geno    Perform genotyping        <index_prefix> <input FASTQ> <input SNPs in VCF> <output file in VCF>\n");
}
static void arg_check(int argc, int expected)
{
	if (argc - 2 != expected) { print_help(); exit(EXIT_FAILURE); }
}
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e && *e ? atoi(e) : dflt; }
// CPUs this process may use: the hardware threads, capped by a cgroup CPU quota (cpu.max: "<quota> <period>"; this pool's GPU boxes
// give a container 16 CPUs' worth of time on a 256-thread host)
static int usable_cpus()
{
	int h = (int)std::thread::hardware_concurrency();
	if (h <= 0) h = 1;
	if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
		char a[64] = ""; long long per = 100000;
		if (fscanf(f, "%63s %lld", a, &per) >= 1 && strcmp(a, "max") != 0 && per > 0) { const long long q = atoll(a); const int c = (int)((q + per - 1) / per); if (q > 0 && c > 0 && c < h) h = c; }
		fclose(f);
	}
	return h;
}

#define VG_CHECK(call)                                                                           \
	do {                                                                                         \
		int rc_ = (call);                                                                        \
		if (rc_ != VG_OK) { fprintf(stderr, "vargeno: %s failed (%d): %s\n", #call, rc_, vg_last_error()); exit(EXIT_FAILURE); } \
	} while (0)

// Bytes [lo, hi) of the FASTQ file as a stream to one replica: reader threads pread() the range piecewise into a ring of pinned
// chunk buffers; this thread pushes the chunks in order (vg_fastq_stream_push returns as soon as a chunk is on the device) and
// learns what was framed only at the end.  Offsets in the result are relative to lo.
struct StreamResult {
	uint64_t nrec = 0, used = 0, last = 0;
	int refused = 0;
	std::string error;                                               // empty: fine
};
static StreamResult stream_range(vg_index *ix, int fd, uint64_t lo, uint64_t hi, uint64_t chunk, int n_readers, int pack_threads)
{
	StreamResult res;
	const uint64_t fsize = hi - lo;                                  // the stream's length
	const uint64_t piece = std::min<uint64_t>(chunk, 8ull << 20);
	const uint64_t n_chunks = (fsize + chunk - 1) / chunk;
	const int NBUF = 4;
	std::vector<uint8_t *> ring((size_t)NBUF, nullptr);
	std::vector<std::vector<uint8_t>> pageable((size_t)NBUF);
	for (int i = 0; i < NBUF; i++) {
		ring[(size_t)i] = (uint8_t *)vg_host_alloc_pinned((size_t)chunk);
		if (!ring[(size_t)i]) { pageable[(size_t)i].resize((size_t)chunk); ring[(size_t)i] = pageable[(size_t)i].data(); }
	}
	std::mutex mu; std::condition_variable cv;
	std::vector<uint32_t> left((size_t)n_chunks);                   // pieces of chunk i still to be read
	for (uint64_t i = 0; i < n_chunks; i++) { const uint64_t len = std::min(chunk, fsize - i * chunk); left[(size_t)i] = (uint32_t)((len + piece - 1) / piece); }
	uint64_t pushed = 0;                                            // chunks handed to the device (their buffers are free again)
	std::atomic<uint64_t> next_piece{0};
	const uint64_t ppc = (chunk + piece - 1) / piece;               // pieces per (full) chunk
	bool io_error = false;
	std::vector<std::thread> readers;
	for (int t = 0; t < n_readers; t++) readers.emplace_back([&] {
		for (;;) {
			const uint64_t p = next_piece.fetch_add(1);
			const uint64_t ci = p / ppc, off = ci * chunk + (p % ppc) * piece;
			if (ci >= n_chunks) return;
			if (off >= std::min(fsize, (ci + 1) * chunk)) continue;
			{ std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return ci < pushed + (uint64_t)NBUF || io_error; }); if (io_error) return; }
			uint64_t n = std::min(piece, std::min(fsize, (ci + 1) * chunk) - off), done = 0;
			uint8_t *dst = ring[(size_t)(ci % NBUF)] + (off - ci * chunk);
			while (done < n) {
				const ssize_t g = pread(fd, dst + done, (size_t)(n - done), (off_t)(lo + off + done));
				if (g <= 0) break;
				done += (uint64_t)g;
			}
			std::lock_guard<std::mutex> g(mu);
			if (done < n) io_error = true;
			left[(size_t)ci]--;
			cv.notify_all();
		}
	});
	int rc = pack_threads > 0 ? vg_fastq_stream_begin_packed(ix, pack_threads) : vg_fastq_stream_begin(ix);
	for (uint64_t i = 0; i < n_chunks && rc == VG_OK; i++) {
		{ std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return left[(size_t)i] == 0 || io_error; }); if (io_error) break; }
		rc = vg_fastq_stream_push(ix, ring[(size_t)(i % NBUF)], std::min(chunk, fsize - i * chunk));
		{ std::lock_guard<std::mutex> g(mu); pushed = i + 1; }
		cv.notify_all();
	}
	{ std::lock_guard<std::mutex> g(mu); if (rc != VG_OK) io_error = true; pushed = n_chunks; }
	cv.notify_all();
	for (auto &t : readers) t.join();
	if (rc != VG_OK) res.error = std::string("FASTQ stream failed: ") + vg_last_error();
	else if (io_error) res.error = "error reading the FASTQ file";
	else {
		rc = vg_fastq_stream_end(ix, &res.nrec, &res.used, &res.last, &res.refused);
		if (rc != VG_OK) res.error = std::string("vg_fastq_stream_end failed: ") + vg_last_error();
	}
	for (int i = 0; i < NBUF; i++) if (pageable[(size_t)i].empty()) vg_host_free_pinned(ring[(size_t)i]);
	return res;
}

// ---- packing ahead of the index ------------------------------------------------------------------------------------------------
// Framing + 2-bit packing need no device (vg_packer_*), and vg_index_open takes seconds during which the host would otherwise
// idle: bytes [lo, hi) of the FASTQ file are read and packed WHILE the replica's index is being built (r05; the r04 command line
// started to pack only when the handle existed).  What is packed goes up to the device at once, into a read store
// (vg_read_store_*: device memory taken before the index is planned; the link is idle two thirds of the open's time), out of two
// page-locked staging sets -- a first version kept the batches in page-locked HOST memory until the handle existed: 11 GB for
// 200 M reads, 0.15 s per GB to lock and 0.1 s per GB for the operating system to take back at exit, more than the read loop
// itself takes (profiles/job_tail_r05.txt).  The packed form is ~56 bytes per 150 bp read, so a 30x file (620 M reads) is 35 GB:
// VARGENO_PREPACK_GB (default 16) bounds the store; what does not fit is framed after the open like the rest of the range.
// The pre-packer also MEASURES its rate (text bytes framed + packed per second with the threads it was given, on this host, now);
// when the index is ready the caller compares it with the link's rate -- what the device-side framing of the remaining text
// would run at -- and lets the faster one finish.
class PrePacker {
public:
	PrePacker(int fd, uint64_t lo, uint64_t hi, uint64_t chunk, int n_readers, int pack_threads, vg_read_store *store)
		: fd_(fd), lo_(lo), hi_(hi), chunk_(std::min(chunk, std::max<uint64_t>(hi - lo, 1))), n_readers_(n_readers), pack_threads_(pack_threads), store_(store)
	{
		clock_gettime(CLOCK_MONOTONIC, &born_);
		th_ = std::thread([this] { run(); });
	}
	~PrePacker()
	{
		stop();
		if (th_.joinable()) th_.join();
	}
	void stop() { stop_.store(true); }
	bool done() const { return done_.load(); }
	void join() { if (th_.joinable()) th_.join(); }
	// valid after join(): what vg_fastq_stream_end would have said about the bytes [lo, lo + consumed) -- the batches in the store
	uint64_t records() const { return records_; }
	uint64_t consumed() const { return consumed_; }
	uint64_t last_record_start() const { return last_; }
	bool refused() const { return refused_; }
	bool store_full() const { return full_; }
	double text_bytes_per_s() const { const double t = pack_s_.load(); return t > 0 ? (double)packed_text_.load() / t : 0.0; }
	uint64_t text_bytes_done() const { return packed_text_.load(); }
	double finished_after_s() const { return finished_s_.load(); }           // seconds from construction to the last batch (0: still running)
	std::string error;
private:
	void run()
	{
		const uint64_t fsize = hi_ - lo_, n_chunks = (fsize + chunk_ - 1) / chunk_;
		vg_packer *pk = nullptr;
		if (vg_packer_create(pack_threads_, &pk) != VG_OK) { error = vg_last_error(); finish(); return; }
		// two page-locked staging sets of a chunk's worst case (a 256 MiB chunk: 3 x 67 MB each): the store copies out of one
		// while the packer fills the other
		const uint64_t rcap = vg_packer_reads_cap(chunk_), kcap = vg_packer_kmers_cap(chunk_);
		uint64_t *stage[2] = {nullptr, nullptr};
		for (int k = 0; k < 2; k++) {
			stage[k] = (uint64_t *)vg_host_alloc_pinned((size_t)(kcap + 2 * rcap + 2) * 8);
			if (!stage[k]) { error = "page-locked staging for the pre-packer: allocation failed"; if (stage[0]) vg_host_free_pinned(stage[0]); vg_packer_destroy(pk); finish(); return; }
		}
		// The text: the file mapped (the packer's threads read the page cache / tmpfs pages themselves: no copy into buffers of
		// ours, which cost as many CPU seconds as the packing itself -- and the job shares a CPU quota with the index's start-up),
		// the chunks after the current one asked for ahead (MADV_WILLNEED); VARGENO_PREPACK_MMAP=0 or a file that cannot be
		// mapped: reader threads fill three chunk buffers with pread, a chunk ahead of the packer
		const uint64_t page = (uint64_t)sysconf(_SC_PAGESIZE), map_lo = lo_ / page * page;
		const uint8_t *map = nullptr;
		if (env_int("VARGENO_PREPACK_MMAP", 1)) {
			void *m = mmap(nullptr, (size_t)(hi_ - map_lo), PROT_READ, MAP_SHARED, fd_, (off_t)map_lo);
			if (m != MAP_FAILED) { map = (const uint8_t *)m; (void)madvise(m, (size_t)(hi_ - map_lo), MADV_SEQUENTIAL); }
		}
		const int NBUF = 3;
		std::vector<std::vector<uint8_t>> text((size_t)NBUF);
		if (!map) for (auto &t : text) t.resize((size_t)std::min(chunk_, fsize));
		const uint64_t piece = std::min<uint64_t>(chunk_, 8ull << 20), ppc = (chunk_ + piece - 1) / piece;
		std::vector<uint32_t> left((size_t)n_chunks);
		for (uint64_t i = 0; i < n_chunks; i++) { const uint64_t len = std::min(chunk_, fsize - i * chunk_); left[(size_t)i] = map ? 0u : (uint32_t)((len + piece - 1) / piece); }
		std::mutex rmu; std::condition_variable rcv;
		uint64_t packed = 0;                                                     // chunks the packer is done with (their buffers are free)
		uint64_t n_pushes = 0;                                                   // batches handed to the store
		std::atomic<uint64_t> next_piece{0};
		bool io_error = false, quit = false;
		std::vector<std::thread> readers;
		for (int t = 0; t < (map ? 0 : n_readers_); t++) readers.emplace_back([&] {
			for (;;) {
				const uint64_t p = next_piece.fetch_add(1);
				const uint64_t ci = p / ppc, off = ci * chunk_ + (p % ppc) * piece;
				if (ci >= n_chunks) return;
				if (off >= std::min(fsize, (ci + 1) * chunk_)) continue;
				{ std::unique_lock<std::mutex> g(rmu); rcv.wait(g, [&] { return ci < packed + (uint64_t)NBUF || io_error || quit; }); if (io_error || quit) return; }
				uint64_t n = std::min(piece, std::min(fsize, (ci + 1) * chunk_) - off), done = 0;
				uint8_t *dst = text[(size_t)(ci % NBUF)].data() + (off - ci * chunk_);
				while (done < n) {
					const ssize_t g = pread(fd_, dst + done, (size_t)(n - done), (off_t)(lo_ + off + done));
					if (g <= 0) break;
					done += (uint64_t)g;
				}
				std::lock_guard<std::mutex> g(rmu);
				if (done < n) io_error = true;
				left[(size_t)ci]--;
				rcv.notify_all();
			}
		});
		for (uint64_t i = 0; i < n_chunks && !stop_.load(); i++) {
			{ std::unique_lock<std::mutex> g(rmu); rcv.wait(g, [&] { return left[(size_t)i] == 0 || io_error; }); if (io_error) break; }
			const uint64_t len = std::min(chunk_, fsize - i * chunk_);
			// (the sets alternate on the number of PUSHES: the store waits for the copies of the push before at its next push, so a
			// chunk that framed nothing -- no push -- must not hand the set of a push still in flight to the chunk after it)
			uint64_t *sk = stage[n_pushes & 1], *sm = sk + kcap, *so = sm + rcap;
			uint64_t nr = 0, nc = 0, ninv = 0;
			struct timespec a, b; clock_gettime(CLOCK_MONOTONIC, &a);
			const uint8_t *src = map ? map + (lo_ - map_lo) + i * chunk_ : text[(size_t)(i % NBUF)].data();
			if (map && i + 1 < n_chunks) (void)madvise((void *)(map + ((lo_ - map_lo) + (i + 1) * chunk_) / page * page), (size_t)std::min(2 * chunk_, hi_ - lo_ - (i + 1) * chunk_), MADV_WILLNEED);
			const int rc = vg_packer_push(pk, src, len, sk, kcap, sm, so, rcap, &nr, &nc, &ninv);
			if (map && i > 0) (void)madvise((void *)(map + ((lo_ - map_lo) + (i - 1) * chunk_ + page - 1) / page * page), (size_t)(chunk_ / page * page - page), MADV_DONTNEED);      // (the mapping of the chunk before: its pages stay in the page cache, the page tables go)
			clock_gettime(CLOCK_MONOTONIC, &b);
			if (rc != VG_OK) { error = vg_last_error(); break; }
			pack_s_.store(pack_s_.load() + (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec));
			{ std::lock_guard<std::mutex> g(rmu); packed = i + 1; }
			rcv.notify_all();
			if (nr) {
				// (the push waits for the copies of the chunk before -- the other staging set -- and enqueues this chunk's)
				const int prc = vg_read_store_push(store_, sk, sm, so, nr);
				if (prc == VG_ENOMEM) { full_ = true; break; }                      // (this chunk's records are dropped with it: the stream is re-framed from the last batch that was kept)
				if (prc != VG_OK) { error = vg_last_error(); break; }
				n_pushes++;
				uint64_t rec = 0, cons = 0, last = 0; int ref = 0;
				(void)vg_packer_end(pk, &rec, &cons, &last, &ref);               // (a query: the stream's totals so far)
				records_ = rec; consumed_ = cons; last_ = last;
			}
			packed_text_.fetch_add(len);
			int ref = 0;
			(void)vg_packer_end(pk, nullptr, nullptr, nullptr, &ref);
			if (ref) { refused_ = true; break; }
		}
		{ std::lock_guard<std::mutex> g(rmu); quit = true; }
		rcv.notify_all();
		for (auto &t : readers) t.join();
		if (io_error && error.empty()) error = "error reading the FASTQ file";
		if (vg_read_store_flush(store_) != VG_OK && error.empty()) error = vg_last_error();      // the staging sets are free
		if (map) (void)munmap((void *)map, (size_t)(hi_ - map_lo));
		for (int k = 0; k < 2; k++) vg_host_free_pinned(stage[k]);
		vg_packer_destroy(pk);
		finish();
	}
	void finish()
	{
		struct timespec now; clock_gettime(CLOCK_MONOTONIC, &now);
		finished_s_.store((double)(now.tv_sec - born_.tv_sec) + 1e-9 * (double)(now.tv_nsec - born_.tv_nsec));
		done_.store(true);
	}
	const int fd_; const uint64_t lo_, hi_, chunk_; const int n_readers_, pack_threads_;
	vg_read_store *const store_;
	std::thread th_;
	std::atomic<bool> stop_{false}, done_{false};
	uint64_t records_ = 0, consumed_ = 0, last_ = 0;
	bool refused_ = false, full_ = false;
	std::atomic<double> pack_s_{0.0}; std::atomic<uint64_t> packed_text_{0};
	struct timespec born_; std::atomic<double> finished_s_{0.0};
};

// ---- a FASTQ "file" that can be read only once -------------------------------------------------------------------------------
// The reference fopen()s whatever path it is given and fgets its way through it (qv.cc:2182, 760-763): a FIFO, /dev/stdin, bash's
// <(zcat reads.fq.gz) all work there by construction.  Here the file routes above pread / mmap ranges of the file from many
// threads, cut it at record starts per replica, and re-open it for the host reader -- none of which a pipe allows: a FIFO
// loses its only reader between two open()s, a /dev/fd/N substitution has size 0.  So a path that is not a regular file takes THIS
// route: the one descriptor is drained by one thread (whole chunks into a small ring, from the moment the command line
// starts -- beside the index open; the pipe's pages are moved on to a few copier threads, see read_loop), the chunks are framed + packed by the host packer (vg_packer_*: bytes cut anywhere), and the
// packed batches go to the replicas round robin -- into their read stores while the index is still opening, straight into the
// read loop (vg_reads_submit_packed) afterwards.  No ranges, no seek.  What the packer refuses (a line beyond fgets' 1023
// characters ...) and the possibly truncated tail go through the host reader like on the file routes: it is given the bytes
// still in memory (from the start of the last framed record on, to prime the reference's stale line buffers) and the descriptor.
class PipeIngest {
public:
	// sink(replica, kmers, meta, chunk_offsets, n_reads): a packed batch for a replica's read loop, once attach() has been called;
	// blocking (the arrays are free when it returns); returns an error text or ""
	typedef std::function<std::string(size_t, const uint64_t *, const uint64_t *, const uint64_t *, uint64_t)> Sink;
	PipeIngest(int fd, uint64_t chunk, int pack_threads, const std::vector<vg_read_store *> &stores, Sink sink)
		: fd_(fd), chunk_(chunk), pack_threads_(pack_threads < 1 ? 1 : pack_threads), stores_(stores), sink_(std::move(sink))
	{
		clock_gettime(CLOCK_MONOTONIC, &born_);
		for (auto &b : ring_) b.data.resize((size_t)chunk_);
		reader_ = std::thread([this] { read_loop(); });
		worker_ = std::thread([this] { work_loop(); });
	}
	~PipeIngest() { finish(); }
	// the handles exist: the stores' batches are submitted by the caller; from now on batches go straight to the handles
	void attach() { { std::lock_guard<std::mutex> g(mu_); attached_ = true; } cv_.notify_all(); }
	void finish() { if (worker_.joinable()) worker_.join(); if (reader_.joinable()) reader_.join(); }
	// valid after finish(): the stream's totals, and what the host reader needs
	uint64_t records = 0, consumed = 0, last = 0, bytes_read = 0;
	bool refused = false;
	uint64_t to_store = 0, direct = 0;                       // reads that went through a read store / straight into the read loop
	double seconds = 0.0;                                    // from construction to the end of the stream
	int copiers = 0;                                         // copier threads the reader dealt the stream to (0: it read() the descriptor itself)
	std::string error;
	// bytes [span_base, span_base + the spans) of the stream are still in memory (every chunk from the one that holds `last` on);
	// the descriptor continues behind them
	uint64_t span_base = 0;
	std::vector<std::pair<const uint8_t *, size_t>> spans;
private:
	struct Buf { std::vector<uint8_t> data; uint64_t len = 0, off = 0; bool last = false; };
	static constexpr int NB = 4;                             // the chunk before the packer's (its tail may hold the last framed record), the packer's, two read ahead
	// A pipe is one copy stream per reader: read() copies page by page under the pipe's lock (~5 GB/s against a producer that
	// write()s, ~9 GB/s against one that lends its pages, profiles/pipe_ab_r06.jsonl).  splice() between two pipes MOVES page
	// references instead: the reader thread deals the stream in segments to a few private pipes, and a copier thread per private
	// pipe read()s its segments into their places in the chunk -- the copies run side by side ($VARGENO_PIPE_COPIERS, default 4;
	// 0: the plain read() loop).  Everything taken from the descriptor has reached the ring when a chunk is handed on, so the
	// host reader still continues at the descriptor.  A descriptor that cannot be spliced from (EINVAL on the first call): read().
	struct Copier {
		int r = -1, w = -1;
		std::thread th;
		std::mutex mu; std::condition_variable cv;
		std::vector<std::pair<uint8_t *, size_t>> q; size_t head = 0;      // segments of the current chunk, in order
		bool stop = false, failed = false;
		void run()
		{
			for (;;) {
				std::pair<uint8_t *, size_t> job;
				{ std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return head < q.size() || stop; }); if (head >= q.size()) return; job = q[head]; }
				size_t at = 0;
				bool bad = false;
				while (at < job.second) {
					const ssize_t got = read(r, job.first + at, job.second - at);
					if (got < 0 && errno == EINTR) continue;
					if (got <= 0) { bad = true; break; }
					at += (size_t)got;
				}
				{ std::lock_guard<std::mutex> g(mu); head++; if (bad) failed = true; }
				cv.notify_all();
			}
		}
		void push(uint8_t *dest, size_t n) { { std::lock_guard<std::mutex> g(mu); q.emplace_back(dest, n); } cv.notify_all(); }
		bool drain() { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return head >= q.size(); }); q.clear(); head = 0; return !failed; }
	};
	void read_loop()
	{
		const int want = std::min(16, std::max(0, env_int("VARGENO_PIPE_COPIERS", 4)));
		std::vector<std::unique_ptr<Copier>> cop;
		for (int k = 0; k < want; k++) {
			int p[2];
			if (pipe(p) != 0) break;
			(void)fcntl(p[1], F_SETPIPE_SZ, 1 << 20);               // (fails harmlessly when the user's pipe pages are used up: 64 KiB then)
			cop.emplace_back(new Copier());
			cop.back()->r = p[0]; cop.back()->w = p[1];
			Copier *c = cop.back().get();
			c->th = std::thread([c] { c->run(); });
		}
		bool fan = !cop.empty();
		uint64_t seg = 0, spliced = 0;
		for (uint64_t i = 0;; i++) {
			{ std::unique_lock<std::mutex> g(mu_); cv_.wait(g, [&] { return i + 2 <= done_ + (uint64_t)NB || stop_reading_; }); if (stop_reading_) break; }
			Buf &b = ring_[i % NB];
			uint64_t got = 0;
			bool eof = false, bad = false;
			while (got < chunk_) {
				ssize_t g;
				if (fan) {
					Copier &c = *cop[(size_t)(seg % cop.size())];
					g = splice(fd_, nullptr, c.w, nullptr, (size_t)std::min<uint64_t>(chunk_ - got, 1u << 20), SPLICE_F_MOVE);
					if (g < 0 && errno == EINTR) continue;
					if (g < 0 && spliced == 0 && (errno == EINVAL || errno == ENOSYS || errno == EBADF)) { fan = false; continue; }
					if (g > 0) { c.push(b.data.data() + got, (size_t)g); seg++; spliced += (uint64_t)g; }
				} else {
					g = read(fd_, b.data.data() + got, (size_t)(chunk_ - got));
					if (g < 0 && errno == EINTR) continue;
				}
				if (g < 0) { bad = true; eof = true; break; }
				if (g == 0) { eof = true; break; }
				got += (uint64_t)g;
				// (a refusal: the packer will read no further -- hand over what has arrived, the host reader reads on from the descriptor)
				{ std::lock_guard<std::mutex> l(mu_); if (stop_reading_) break; }
			}
			for (auto &c : cop) if (!c->drain()) { bad = true; eof = true; }      // (every byte taken from the descriptor is in the chunk now)
			{ std::lock_guard<std::mutex> l(mu_); if (bad) io_error_ = true; b.len = got; b.off = total_read_; b.last = eof; total_read_ += got; filled_ = i + 1; if (eof) eof_ = true; }
			cv_.notify_all();
			if (eof) break;
		}
		for (auto &c : cop) {
			{ std::lock_guard<std::mutex> g(c->mu); c->stop = true; }
			c->cv.notify_all();
			c->th.join();
			close(c->r); close(c->w);
		}
		{ std::lock_guard<std::mutex> l(mu_); reader_done_ = true; copiers_ = fan ? (int)cop.size() : 0; }
		cv_.notify_all();
	}
	void work_loop()
	{
		vg_packer *pk = nullptr;
		const uint64_t rcap = vg_packer_reads_cap(chunk_), kcap = vg_packer_kmers_cap(chunk_);
		uint64_t *stage[2] = {nullptr, nullptr};
		int set_store[2] = {-1, -1};                         // the store whose copies may still read staging set k
		auto bail = [&](const std::string &e) { if (error.empty()) error = e; };
		if (vg_packer_create(pack_threads_, &pk) != VG_OK) bail(vg_last_error());
		bool pinned[2] = {true, true};                       // (page-locked when a device is there to lock it for; any memory works)
		for (int k = 0; k < 2 && error.empty(); k++) {
			stage[k] = (uint64_t *)vg_host_alloc_pinned((size_t)(kcap + 2 * rcap + 2) * 8);
			if (!stage[k]) { pinned[k] = false; stage[k] = (uint64_t *)malloc((size_t)(kcap + 2 * rcap + 2) * 8); }
			if (!stage[k]) bail("staging for the FASTQ stream: allocation failed");
		}
		const size_t nrep = stores_.size();
		size_t rr = 0;
		uint64_t n_sets = 0;
		uint64_t i = 0;
		for (; error.empty(); i++) {
			{ std::unique_lock<std::mutex> g(mu_); cv_.wait(g, [&] { return filled_ > i || reader_done_; }); if (filled_ <= i) break; }
			Buf &b = ring_[i % NB];
			if (b.len) {
				const int k = (int)(n_sets & 1);
				if (set_store[k] >= 0) { if (vg_read_store_flush(stores_[(size_t)set_store[k]]) != VG_OK) { bail(vg_last_error()); break; } set_store[k] = -1; }
				uint64_t *sk = stage[k], *sm = sk + kcap, *so = sm + rcap;
				uint64_t nr = 0, nc = 0, ninv = 0;
				if (vg_packer_push(pk, b.data.data(), b.len, sk, kcap, sm, so, rcap, &nr, &nc, &ninv) != VG_OK) { bail(vg_last_error()); break; }
				if (nr) {
					bool sent = false;
					bool att; { std::lock_guard<std::mutex> g(mu_); att = attached_; }
					if (!att && stores_[rr]) {
						const int prc = vg_read_store_push(stores_[rr], sk, sm, so, nr);
						if (prc == VG_OK) { sent = true; set_store[k] = (int)rr; n_sets++; to_store += nr; }
						else if (prc != VG_ENOMEM) { bail(vg_last_error()); break; }
					}
					if (!sent) {
						// the store is full (or there is none): this batch waits here for the handle -- the reader thread keeps filling the ring,
						// then the pipe's writer waits
						{ std::unique_lock<std::mutex> g(mu_); cv_.wait(g, [&] { return attached_; }); }
						const std::string e = sink_(rr, sk, sm, so, nr);
					if (!e.empty()) { bail(e); break; }
					direct += nr;
					}
					rr = (rr + 1) % nrep;
				}
				int ref = 0;
				(void)vg_packer_end(pk, &records, &consumed, &last, &ref);     // (a query: the stream's totals so far)
				if (ref) { refused = true; i++; break; }
			}
			{ std::lock_guard<std::mutex> g(mu_); done_ = i + 1; }           // (chunk i stays in the ring until chunk i + 1 is done with: the last framed record may begin in it)
			cv_.notify_all();
			if (b.last) { i++; break; }
		}
		// the end of the stream, a refusal or an error: the reader stops after the read() it is in; every chunk from the one before the
		// packer's last on is handed to the host reader
		{ std::lock_guard<std::mutex> g(mu_); stop_reading_ = true; }
		cv_.notify_all();
		if (reader_.joinable()) reader_.join();
		for (int k = 0; k < 2; k++) if (set_store[k] >= 0) (void)vg_read_store_flush(stores_[(size_t)set_store[k]]);
		{
			std::lock_guard<std::mutex> g(mu_);
			if (io_error_) bail("error reading the FASTQ stream");
			bytes_read = total_read_;
			copiers = copiers_;
			// chunks [first, filled_) are intact in the ring: first = the chunk before the last one the packer saw (or 0)
			const uint64_t seen = i;                                             // chunks the packer has been given
			const uint64_t first = seen >= 2 ? seen - 2 : 0;
			span_base = filled_ > first ? ring_[first % NB].off : total_read_;
			for (uint64_t c = first; c < filled_; c++) spans.emplace_back(ring_[c % NB].data.data(), (size_t)ring_[c % NB].len);
		}
		for (int k = 0; k < 2; k++) if (stage[k]) { if (pinned[k]) vg_host_free_pinned(stage[k]); else free(stage[k]); }
		if (pk) vg_packer_destroy(pk);
		struct timespec now; clock_gettime(CLOCK_MONOTONIC, &now);
		seconds = (double)(now.tv_sec - born_.tv_sec) + 1e-9 * (double)(now.tv_nsec - born_.tv_nsec);
	}
	const int fd_; const uint64_t chunk_; const int pack_threads_;
	std::vector<vg_read_store *> stores_;
	Sink sink_;
	Buf ring_[NB];
	std::mutex mu_; std::condition_variable cv_;
	uint64_t filled_ = 0, done_ = 0, total_read_ = 0;
	bool eof_ = false, reader_done_ = false, stop_reading_ = false, io_error_ = false, attached_ = false;
	int copiers_ = 0;
	std::thread reader_, worker_;
	struct timespec born_;
};

// The first record start at or after `from`: the start of a line that begins with '@' whose next-but-one line begins with '+'
// (a quality line may begin with '@', but then the line two below it is a sequence line, and no sequence begins with '+').
// Returns fsize when there is none; UINT64_MAX on a line too long to be a FASTQ line of this tool (the caller falls back).
static uint64_t find_record_start(int fd, uint64_t from, uint64_t fsize)
{
	if (from == 0) return 0;
	const uint64_t WIN = 1 << 20;
	std::vector<char> buf((size_t)WIN);
	uint64_t base = from - 1;                                        // one byte back: is `from` itself the start of a line?
	const uint64_t n = std::min(WIN, fsize - base);
	uint64_t got = 0;
	while (got < n) { const ssize_t g = pread(fd, buf.data() + got, (size_t)(n - got), (off_t)(base + got)); if (g <= 0) break; got += (uint64_t)g; }
	std::vector<uint64_t> starts;                                    // line starts inside the window
	for (uint64_t i = 0; i + 1 < got; i++) if (buf[(size_t)i] == '\n') starts.push_back(i + 1);
	for (size_t k = 0; k + 2 < starts.size(); k++)
		if (buf[(size_t)starts[k]] == '@' && buf[(size_t)starts[k + 2]] == '+') return base + starts[k];
	// the window reaches the end of the file and holds no record start: the split point fell inside the last record (its header
	// line included) -- what is left belongs to the range before
	if (base + got >= fsize) return fsize;
	return UINT64_MAX;
}

// Where n replicas split the file: cut[0] = 0 <= cut[1] <= ... <= cut[n] = fsize, every inner cut a record start.  false: no
// record start where one should be (the caller frames the whole file on the host).
static bool range_cuts(int fd, uint64_t fsize, int n, std::vector<uint64_t> &cut)
{
	cut.assign((size_t)n + 1, fsize);
	cut[0] = 0;
	for (int g = 1; g < n; g++) {
		const uint64_t at = find_record_start(fd, std::max(cut[(size_t)g - 1], fsize / (uint64_t)n * (uint64_t)g), fsize);
		if (at == UINT64_MAX) return false;
		cut[(size_t)g] = at;
	}
	return true;
}

[[maybe_unused]] static uint64_t mem_available_bytes()
{
	uint64_t kb = 0;
	if (FILE *f = fopen("/proc/meminfo", "r")) {
		char line[256];
		while (fgets(line, sizeof line, f)) if (sscanf(line, "MemAvailable: %lu kB", &kb) == 1) break;
		fclose(f);
	}
	return kb * 1024;
}

static int run_geno(const std::string &prefix, const std::string &fastq, const std::string &vcf_in, const std::string &vcf_out)
{
	const clock_t begin = clock();
	struct timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
	auto secs = [](const struct timespec &a, const struct timespec &b) { return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec); };
	std::vector<vgh::ChrLen> chrlens = vgh::read_chrlens(prefix + ".chrlens");
	int ngpu = env_int("VARGENO_GPUS", 1);
	const int have = vg_device_count();
	if (have <= 0) { fprintf(stderr, "vargeno: no HIP device found (this build has no CPU path)\n"); return EXIT_FAILURE; }
	const bool share = env_int("VARGENO_SHARE_DEVICES", 0) != 0;     // more replicas than devices: replica g sits on device g % have
	if (ngpu > have && !share) ngpu = have;
	if (ngpu < 1) ngpu = 1;
	const uint64_t batch = (uint64_t)env_int("VARGENO_BATCH", 1 << 22);
	const bool verbose = env_int("VARGENO_VERBOSE", 0) != 0;

	fprintf(stderr, "Initializing...\n");
	// ---- the FASTQ file first: where the replicas' ranges are cut, and -- what needs no device -- framing + packing of their text on
	//      host threads, started BEFORE the index is opened and running beside it
	const bool host_framing = env_int("VARGENO_HOST_FASTQ", 0) != 0;
	int fd = -1;
	uint64_t fsize = 0;
	std::vector<uint64_t> cut;
	bool cuts_ok = false;
	const int hw = usable_cpus();
	// host threads that frame + pack (per replica); 0: the text is framed on the device, nothing is packed ahead
	int pack_threads = env_int("VARGENO_PACK_THREADS", -1);
	const bool pack_threads_given = pack_threads >= 0;
	if (pack_threads < 0) pack_threads = std::max(2, std::min(hw - 2, 96) / ngpu);
	const int n_readers = std::max(1, std::min(env_int("VARGENO_READERS", std::max(8, std::min(32, hw / 8))), 64));
	std::vector<std::unique_ptr<PrePacker>> pre((size_t)ngpu);
	std::vector<vg_read_store *> store((size_t)ngpu, nullptr);
	std::vector<vg_index *> ix((size_t)ngpu, nullptr);
	std::unique_ptr<PipeIngest> pipe_in;
	bool once_only = false;
	if (!host_framing) {
		fd = open(fastq.c_str(), O_RDONLY);
		if (fd < 0) { fprintf(stderr, "vargeno: cannot open %s\n", fastq.c_str()); return EXIT_FAILURE; }
		struct stat sb;
		if (fstat(fd, &sb) != 0) { close(fd); fprintf(stderr, "vargeno: cannot stat %s\n", fastq.c_str()); return EXIT_FAILURE; }
		fsize = (uint64_t)sb.st_size;
		once_only = !S_ISREG(sb.st_mode);                               // a FIFO, /dev/stdin, <(...): one descriptor, read once, no ranges (PipeIngest)
		if (once_only) {
			(void)fcntl(fd, F_SETPIPE_SZ, 1 << 20);                     // (a pipe: the largest buffer an unprivileged process may ask for; fails harmlessly on anything else)
			const uint64_t want = (uint64_t)std::max(1, env_int("VARGENO_PREPACK_GB", 16)) << 30;
			if (env_int("VARGENO_PREPACK", 1)) for (int g = 0; g < ngpu; g++) {
				int on = 0;
				for (int k = 0; k < ngpu; k++) on += k % have == g % have;
				uint64_t bytes = std::min<uint64_t>(want, vg_device_memory(g % have) / 8) / (uint64_t)on;
				if (const char *e = getenv("VARGENO_PREPACK_BYTES")) if (atoll(e) > 0) bytes = (uint64_t)atoll(e);
				if (vg_read_store_create(g % have, bytes, &store[(size_t)g]) != VG_OK) store[(size_t)g] = nullptr;      // (no store: its batches wait for the handle)
			}
			pipe_in.reset(new PipeIngest(fd, (uint64_t)std::max(1, env_int("VARGENO_CHUNK_MB", 64)) << 20, std::max(1, pack_threads * ngpu), store,
			                             [&ix](size_t g, const uint64_t *k, const uint64_t *m, const uint64_t *o, uint64_t n) -> std::string {
				                             return vg_reads_submit_packed(ix[g], k, m, o, n) == VG_OK ? std::string() : std::string("vg_reads_submit_packed failed: ") + vg_last_error();
			                             }));
		}
		cuts_ok = !once_only && range_cuts(fd, fsize, ngpu, cut);
		if (cuts_ok && pack_threads > 0 && env_int("VARGENO_PREPACK", 1)) {
			// a read store per replica, on its device, taken NOW (the index is planned with what is left): as large as the range's
			// packed form (~1/5.5 of its text, and room for a chunk's worst case is not needed: a push that does not fit ends the
			// pre-packing), at most VARGENO_PREPACK_GB per device and never more than an eighth of the device
			const uint64_t want = (uint64_t)std::max(1, env_int("VARGENO_PREPACK_GB", 16)) << 30;
			const uint64_t pchunk = (uint64_t)std::max(1, env_int("VARGENO_CHUNK_MB", 256)) << 20;
			for (int g = 0; g < ngpu; g++) {
				if (cut[(size_t)g] >= cut[(size_t)g + 1]) continue;
				int on = 0;
				for (int k = 0; k < ngpu; k++) on += k % have == g % have;
				const uint64_t range = cut[(size_t)g + 1] - cut[(size_t)g];
				uint64_t bytes = std::min<uint64_t>(std::min<uint64_t>(want, vg_device_memory(g % have) / 8) / (uint64_t)on, range / 5 + (8ull << 20));
				if (const char *e = getenv("VARGENO_PREPACK_BYTES")) if (atoll(e) > 0) bytes = (uint64_t)atoll(e);          // (tests: a store that fills up after a chunk or two)
				if (vg_read_store_create(g % have, bytes, &store[(size_t)g]) != VG_OK) { fprintf(stderr, "vargeno: no read store on device %d (%s): its range is framed after the index is open\n", g % have, vg_last_error()); continue; }
				pre[(size_t)g].reset(new PrePacker(fd, cut[(size_t)g], cut[(size_t)g + 1], pchunk, std::max(2, n_readers / ngpu), pack_threads, store[(size_t)g]));
			}
		}
	}
	// the SNP list is read now, beside the index open (the VCF pass at the end of the job starts from its bytes)
	std::string vcf_text;
	bool vcf_ok = false;
	std::thread vcf_reader([&] { vcf_ok = vgh::read_whole_file(vcf_in, vcf_text); });
	struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } vcf_joiner{vcf_reader};
	{
		std::vector<std::thread> th;
		std::vector<int> rcs((size_t)ngpu, 0);
		std::vector<std::string> errs((size_t)ngpu);
		const char *bgt = getenv("VARGENO_MAX_DEVICE_GB");
		const uint64_t budget = bgt && *bgt ? (uint64_t)(atof(bgt) * 1e9) : 0ull;
		// replicas that share a device (VARGENO_SHARE_DEVICES) share its memory too: without a budget of its own each of them gets
		// an equal part -- planned for the whole device, the third or fourth one would fail where it could have run on fewer views
		std::vector<uint64_t> budgets((size_t)ngpu, budget);
		if (!budget && ngpu > have) for (int g = 0; g < ngpu; g++) { int on = 0; for (int k = 0; k < ngpu; k++) on += k % have == g % have; if (on > 1) budgets[(size_t)g] = vg_share_budget(g % have, on); }
		// a read store was taken on the device before the index is planned: the plan must be told, or it plans for the device's TOTAL
		// less 12 GiB while the store (up to 16 GiB) already holds part of that -- an index whose views fill the budget would then
		// fail in its allocations where it could have run on fewer views.  vg_share_budget looks at what is free NOW (all budgets are
		// computed here, before any replica opens)
		if (!budget) for (int g = 0; g < ngpu; g++) if (!budgets[(size_t)g] && store[(size_t)g]) { int on = 0; for (int k = 0; k < ngpu; k++) on += k % have == g % have; budgets[(size_t)g] = vg_share_budget(g % have, on); }
		for (int g = 0; g < ngpu; g++) th.emplace_back([&, g] { rcs[(size_t)g] = vg_index_open_ex(prefix.c_str(), g % have, budgets[(size_t)g], &ix[(size_t)g]); if (rcs[(size_t)g]) errs[(size_t)g] = vg_last_error(); });
		for (auto &t : th) t.join();
		for (int g = 0; g < ngpu; g++) if (rcs[(size_t)g]) {
			fprintf(stderr, "vargeno: cannot load index %s on GPU %d (%d): %s\n", prefix.c_str(), g, rcs[(size_t)g], errs[(size_t)g].c_str());
			if (pipe_in) exit(EXIT_FAILURE);                            // (its threads hold the pipe: no unwinding)
			return EXIT_FAILURE;
		}
	}
	for (auto *h : ix) VG_CHECK(vg_set_stats(h, env_int("VARGENO_STATS", 0)));
	if (verbose) { fprintf(stderr, "index replica: %s\n", vg_index_plan(ix[0])); fprintf(stderr, "index start-up: %s\n", vg_index_open_report(ix[0])); }

	fprintf(stderr, "Processing...\n");
	struct timespec t_loaded; clock_gettime(CLOCK_MONOTONIC, &t_loaded);
	uint64_t total = 0; int next_gpu = 0;
	// Default ingest: the file is a byte stream to the device(s).  One replica takes all of it; several take one contiguous range
	// each, cut at record starts, all at once.  A range's text is framed + packed by host threads (what was packed while the index
	// was being opened is submitted first) or copied up as it is and framed on the device -- whichever runs faster HERE: the
	// pre-packer has measured its rate, vg_link_rate() the link's.  Whatever the stream refuses (from the first chunk with a line
	// beyond fgets' 1023 characters on) and the (possibly truncated) tail of the file go through the host reader below, which
	// reproduces the reference's four-fgets framing exactly, stale buffers included; with several replicas a refusal anywhere but
	// in the last range means the ranges after it were framed out of step with the reference, so everything is reset and framed
	// on the host.
	uint64_t host_from = 0;                    // file offset the host reader takes over from
	uint64_t prime_from = UINT64_MAX;          // start of the last record the device framed (to prime the stale buffers)
	if (pipe_in) {
		// a stream that is read once: what went into the read stores while the index opened first, then the packer's batches
		// straight into the read loops until the stream ends (or is refused)
		for (int g = 0; g < ngpu; g++) if (store[(size_t)g]) { VG_CHECK(vg_read_store_flush(store[(size_t)g])); }
		pipe_in->attach();
		pipe_in->finish();
		for (int g = 0; g < ngpu; g++) if (store[(size_t)g] && vg_read_store_reads(store[(size_t)g])) VG_CHECK(vg_reads_submit_store(ix[(size_t)g], store[(size_t)g]));
		if (!pipe_in->error.empty()) { fprintf(stderr, "vargeno: %s\n", pipe_in->error.c_str()); exit(EXIT_FAILURE); }
		total += pipe_in->records;
		if (verbose) fprintf(stderr, "ingest, replica 0: the FASTQ is not a regular file: one descriptor read once, %lu reads framed + packed by %d host threads (%lu into the read stores while the index opened, %lu straight into the read loop), "
		                             "%.2f GB of text in %.2f s (%.2f GB/s), dealt to %d copier threads%s\n", (unsigned long)pipe_in->records, std::max(1, pack_threads * ngpu), (unsigned long)pipe_in->to_store, (unsigned long)pipe_in->direct,
		                     (double)pipe_in->bytes_read / 1e9, pipe_in->seconds, pipe_in->seconds > 0 ? (double)pipe_in->bytes_read / 1e9 / pipe_in->seconds : 0.0, pipe_in->copiers, pipe_in->refused ? "; the stream framing refused a chunk: the host reader takes the rest" : "");
	} else if (!host_framing) {
		if (!cuts_ok) {
			host_from = 0;                                              // no record start found where one should be: the host reader takes the file
			fprintf(stderr, "vargeno: no FASTQ record start within 1 MiB of a range boundary: the whole file is framed on the host (slower)\n");
		} else {
			const double link = pack_threads > 0 ? vg_link_rate(0) : 0.0;          // bytes/s of text the device-side framing can be fed at
			std::vector<StreamResult> res((size_t)ngpu);
			std::vector<std::string> route((size_t)ngpu);
			std::vector<std::thread> th;
			for (int g = 0; g < ngpu; g++)
				if (cut[(size_t)g] < cut[(size_t)g + 1] || g == 0)
					th.emplace_back([&, g] {
						StreamResult &r = res[(size_t)g];
						uint64_t done_to = 0;                                  // bytes of the range framed so far
						// the rest of a range (what the pre-packer did not take -- it was stopped, its store filled up, or it never ran): host
						// packing or device framing, whichever is faster HERE.  With a pre-packer its measured rate decides, whether it is
						// still running or done (a 30x file overruns a 16 GB store: the larger part of the file is "the rest"); without one
						// (VARGENO_PREPACK=0, no store) the number of CPUs does, as in r04, unless VARGENO_PACK_THREADS says what is wanted
						bool pack_rest = pack_threads > 0 && (pack_threads_given || hw >= 32);
						if (pre[(size_t)g]) {
							PrePacker &pp = *pre[(size_t)g];
							pack_rest = pack_threads > 0;
							// it keeps the rest of the range unless the device-side framing would finish it at least a second earlier
							// (changing horses costs about that much: a new stream, its readers starting cold)
							{
								const double rp = pp.text_bytes_per_s();
								const double left = (double)(cut[(size_t)g + 1] - cut[(size_t)g]) - (double)pp.text_bytes_done();
								if (rp > 0 && link > 0 && left > 0) pack_rest = rp >= link || left * (1.0 / rp - 1.0 / link) < 1.0;
								if (!pp.done() && !pack_rest) pp.stop();
							}
							pp.join();
							if (!pp.error.empty()) r.error = pp.error;
							const uint64_t submitted = vg_read_store_reads(store[(size_t)g]);
							if (r.error.empty() && vg_reads_submit_store(ix[(size_t)g], store[(size_t)g]) != VG_OK) r.error = std::string("vg_reads_submit_store failed: ") + vg_last_error();
							r.nrec = pp.records(); r.used = pp.consumed(); r.last = pp.last_record_start(); r.refused = pp.refused() ? 1 : 0;
							done_to = pp.consumed();
							char line[320];
							snprintf(line, sizeof line, "%lu reads packed ahead of / beside the index into %.1f GB of device memory (%.1f GB/s of text on %d threads, done %.2f s after the command line started them%s; link %.1f GB/s)",
							         (unsigned long)submitted, (double)vg_read_store_bytes_used(store[(size_t)g]) / 1e9, pp.text_bytes_per_s() / 1e9, pack_threads, pp.finished_after_s(), pp.store_full() ? ", when the store was full" : "", link / 1e9);
							route[(size_t)g] = line;
						}
						const uint64_t lo = cut[(size_t)g] + done_to, hi = cut[(size_t)g + 1];
						if (r.error.empty() && !r.refused && lo < hi) {
							const uint64_t chunk = (uint64_t)std::max(1, env_int("VARGENO_CHUNK_MB", pack_rest ? 256 : 64)) << 20;
							const StreamResult r2 = stream_range(ix[(size_t)g], fd, lo, hi, chunk, std::max(2, n_readers / ngpu), pack_rest ? pack_threads : 0);
							route[(size_t)g] += pack_rest ? "; rest of the range: framed + packed by host threads" : "; rest of the range: framed on the device";
							if (!r2.error.empty()) r.error = r2.error;
							if (r2.nrec) r.last = done_to + r2.last;
							r.nrec += r2.nrec; r.used = done_to + r2.used; r.refused = r2.refused;
						}
					});
			for (auto &t : th) t.join();
			for (int g = 0; g < ngpu; g++) if (!res[(size_t)g].error.empty()) { fprintf(stderr, "vargeno: %s\n", res[(size_t)g].error.c_str()); exit(EXIT_FAILURE); }
			if (verbose) for (int g = 0; g < ngpu; g++) if (!route[(size_t)g].empty()) fprintf(stderr, "ingest, replica %d: %s\n", g, route[(size_t)g].c_str());
			int last_range = 0;                                         // the last range that holds bytes
			for (int g = 0; g < ngpu; g++) if (cut[(size_t)g] < cut[(size_t)g + 1]) last_range = g;
			bool in_step = true;
			for (int g = 0; g < last_range; g++) if (res[(size_t)g].used != cut[(size_t)g + 1] - cut[(size_t)g]) in_step = false;
			if (in_step) {
				for (int g = 0; g <= last_range; g++) total += res[(size_t)g].nrec;
				for (int g = last_range; g >= 0; g--) if (res[(size_t)g].nrec) { prime_from = cut[(size_t)g] + res[(size_t)g].last; break; }
				host_from = cut[(size_t)last_range] + res[(size_t)last_range].used;    // the incomplete tail, or everything from a refused chunk on
				next_gpu = last_range;
			} else {
				// (the file has been streamed once already: a 2x or larger slowdown that must not pass silently)
				fprintf(stderr, "vargeno: a FASTQ range before the last one was refused by the stream framing (a line beyond 1023 characters?): "
				                "counters reset, the whole file is framed on the host\n");
				for (auto *h : ix) VG_CHECK(vg_counts_reset(h));
				host_from = 0;
			}
		}
		pre.clear();
		close(fd);
	}
	{
		// the host reader: the file from host_from on -- or, for a stream that is read once, the bytes still in memory and then the
		// descriptor (never a second open: a FIFO has lost its writer by then)
		if (pipe_in) { host_from = pipe_in->consumed; if (pipe_in->records) prime_from = pipe_in->last; }
		std::unique_ptr<vgh::FastqReader> rdp(pipe_in ? new vgh::FastqReader(fd, pipe_in->span_base, pipe_in->spans) : new vgh::FastqReader(fastq));
		vgh::FastqReader &rd = *rdp;
		vgh::ReadBatch rb;
		if (!host_framing && prime_from != UINT64_MAX) {   // re-read the last framed record: it only fills the line buffers
			rd.seek(prime_from);
			rb.clear();
			(void)rd.next(rb, 1);
		}
		if (!host_framing) rd.seek(host_from);
		for (;;) {
			rb.clear();
			const uint64_t n = rd.next(rb, batch);
			if (!n) break;
			total += n;
			VG_CHECK(vg_reads_submit(ix[(size_t)next_gpu], rb.bases.data(), rb.quals.data(), rb.offsets.data(), n));
			next_gpu = (next_gpu + 1) % ngpu;
		}
	}
	for (auto *h : ix) VG_CHECK(vg_sync(h));
	struct timespec t_reads; clock_gettime(CLOCK_MONOTONIC, &t_reads);
	{
		// util.c:103: the reference aborts on a read with a character other than ACGTN (and writes no VCF); the library counts
		// such reads whether or not event counting is on
		vg_stats st;
		uint64_t invalid = 0;
		for (auto *h : ix) { VG_CHECK(vg_stats_get(h, &st)); invalid += st.reads_invalid; }
		if (invalid) { fprintf(stderr, "vargeno: %lu reads contain a character other than ACGTN (the reference aborts on these)\n", (unsigned long)invalid); return EXIT_FAILURE; }
	}
	// one process, n devices: one RCCL all-reduce of the per-site counters over xGMI (VARGENO_FORCE_RCCL=1 also sends a
	// single device through it, which is the identity)
	if (ngpu > 1 || env_int("VARGENO_FORCE_RCCL", 0)) VG_CHECK(vg_counts_allreduce_devices(ix.data(), ngpu));
	vgh::SiteCounts sc;
	const uint64_t ns = vg_num_sites(ix[0]);
	sc.pos.resize(ns); sc.ref_freq.resize(ns); sc.alt_freq.resize(ns); sc.ref_cnt.resize(ns); sc.alt_cnt.resize(ns);
	VG_CHECK(vg_sites_fetch(ix[0], sc.pos.data(), nullptr, nullptr, sc.ref_freq.data(), sc.alt_freq.data()));
	VG_CHECK(vg_counts_fetch(ix[0], sc.ref_cnt.data(), sc.alt_cnt.data()));
	if (const char *dump = getenv("VARGENO_DUMP_COUNTS")) {              // the saturated counters as the caller gets them: ref counts, then alt counts, one byte per site
		FILE *f = fopen(dump, "wb");
		if (!f || fwrite(sc.ref_cnt.data(), 1, ns, f) != ns || fwrite(sc.alt_cnt.data(), 1, ns, f) != ns) { fprintf(stderr, "vargeno: cannot write %s\n", dump); return EXIT_FAILURE; }
		fclose(f);
	}
	if (vcf_reader.joinable()) vcf_reader.join();
	vgh::write_genotyped_vcf(sc, chrlens, vcf_in, vcf_out, vcf_ok ? &vcf_text : nullptr);
	struct timespec t_vcf; clock_gettime(CLOCK_MONOTONIC, &t_vcf);
	// The output is complete and closed.  What is left is giving back ~240 GB of device memory and the page-locked buffers, which the
	// operating system does for a process that ends anyway: an orderly vg_index_close + runtime shut-down took 0.7 + 0.9 s of an
	// 7 s job at hg38 scale (profiles/job_tail_r05.txt), so the command line ends here unless VARGENO_ORDERLY_EXIT=1 asks for the
	// full tear-down (tests that look for leaks, sanitizer runs).
	const bool orderly = env_int("VARGENO_ORDERLY_EXIT", 0) != 0;
	if (orderly) { for (auto *h : ix) vg_index_close(h); for (auto *rs : store) vg_read_store_destroy(rs); }
	const double cpu = (double)(clock() - begin) / CLOCKS_PER_SEC;
	printf("Time: %f sec\n", cpu);                                       // qv.cc:1749-1751 prints CPU seconds
	if (verbose) {
		struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
		fprintf(stderr, "reads: %lu  gpus: %d  wall: %.3f s = index load %.3f + FASTQ->counters %.3f (%.2f M reads/s) + call/VCF %.3f + close %.3f\n", (unsigned long)total, ngpu,
		        secs(t0, t1), secs(t0, t_loaded), secs(t_loaded, t_reads), (double)total / secs(t_loaded, t_reads) / 1e6, secs(t_reads, t_vcf), secs(t_vcf, t1));
		// how long the process has existed (its start time in /proc/self/stat, 10 ms ticks, against the boot clock): what the loader
		// and the HIP runtime's static start-up took before main() is that minus the wall time above
		if (FILE *f = fopen("/proc/self/stat", "r")) {
			char buf[2048]; const size_t n = fread(buf, 1, sizeof buf - 1, f); buf[n] = 0; fclose(f);
			const char *q = strrchr(buf, ')');
			unsigned long long start = 0; int field = 2;
			for (q = q ? q + 1 : buf; q && *q && field < 22; ) { q = strchr(q + 1, ' '); field++; if (field == 21 && q) start = strtoull(q + 1, nullptr, 10); }
			struct timespec bt; clock_gettime(CLOCK_BOOTTIME, &bt);
			if (start) fprintf(stderr, "process: alive for %.2f s at this point\n", (double)bt.tv_sec + 1e-9 * (double)bt.tv_nsec - (double)start / (double)sysconf(_SC_CLK_TCK));
		}
	}
	if (!orderly) { fflush(stdout); fflush(stderr); _exit(EXIT_SUCCESS); }
	return EXIT_SUCCESS;
}

int main(int argc, const char *argv[])
{
	if (argc < 2) { print_help(); return 0; }
	const std::string opt = argv[1];
	try {
		if (opt == "index") {
			arg_check(argc, 3);
			vgh::IndexOptions io;
			io.write_lite = !env_int("VARGENO_NO_LITE", 0);
			io.threads = env_int("VARGENO_THREADS", 0);
			vgh::build_index(argv[2], argv[3], argv[4], io);
			return EXIT_SUCCESS;
		} else if (opt == "geno") {
			arg_check(argc, 4);
			return run_geno(argv[2], argv[3], argv[4], argv[5]);
		} else if (opt == "fqcheck") {
			// hidden: the host FASTQ framing alone -- one line per record: read length, then the read and the quality
			// characters the path can see (no device needed; tests/test_host_tools.py)
			arg_check(argc, 1);
			vgh::FastqReader rd(argv[2]);
			vgh::ReadBatch rb;
			for (;;) {
				rb.clear();
				if (!rd.next(rb, 1000)) break;
				for (uint64_t i = 0; i < rb.n(); i++) {
					const uint64_t o = rb.offsets[i], len = rb.offsets[i + 1] - o;
					printf("%lu ", (unsigned long)len);
					fwrite(rb.bases.data() + o, 1, len, stdout);
					printf(" ");
					for (uint64_t j = 0; j < len / 32; j++) printf("%02x", rb.quals[o + j]);
					printf("\n");
				}
			}
			return EXIT_SUCCESS;
		} else if (opt == "fqpipe") {
			// hidden: the once-only FASTQ route of `geno` (PipeIngest + the host reader behind it) without a device -- <path> is
			// opened ONCE, whatever it is (a FIFO, /dev/stdin, a regular file).  One line per record in the form both halves can
			// produce: "<chunks> <trimmed read, upper case> <gate bits, hex>", "N" / "X" for a read the reference skips / aborts on
			// (tests/test_host_tools.py compares with `fqcheck` on the same bytes).  [chunk bytes] [threads]
			if (argc < 3 || argc > 5) { print_help(); return EXIT_FAILURE; }
			const int fd = open(argv[2], O_RDONLY);
			if (fd < 0) throw vgh::Error{std::string("cannot open ") + argv[2]};
			const uint64_t chunk = argc > 3 ? (uint64_t)atoll(argv[3]) : (1ull << 20);
			std::vector<std::string> lines;
			const bool quiet = env_int("VARGENO_FQPIPE_QUIET", 0) != 0;          // (a rate probe of the route: count, print nothing)
			uint64_t counted = 0;
			auto one = [&](uint64_t nch, const uint64_t *km, uint64_t meta) {
				if (quiet) { counted += 1 + (nch & 0); return; }
				if (meta >> 63) { lines.push_back("X"); return; }
				if ((meta >> 62) & 1u) { lines.push_back("N"); return; }
				std::string l = std::to_string(nch) + " ";
				for (uint64_t c = 0; c < nch; c++) for (int b = 0; b < 32; b++) l.push_back("ACGT"[(km[c] >> (2 * b)) & 3u]);
				char hx[32]; snprintf(hx, sizeof hx, " %x", (unsigned)(meta & 0xFFFFFFFFu));
				lines.push_back(l + hx);
			};
			std::vector<vg_read_store *> none(1, nullptr);
			PipeIngest pin(fd, chunk, argc > 4 ? atoi(argv[4]) : 2, none, [&](size_t, const uint64_t *k, const uint64_t *m, const uint64_t *o, uint64_t n) -> std::string {
				for (uint64_t r = 0; r < n; r++) one(o[r + 1] - o[r], k + o[r], m[r]);
				return std::string();
			});
			pin.attach();
			pin.finish();
			if (!pin.error.empty()) throw vgh::Error{pin.error};
			fprintf(stderr, "fqpipe: %lu records framed by the packer, %lu bytes consumed of %lu read, refused %d\n", (unsigned long)pin.records, (unsigned long)pin.consumed, (unsigned long)pin.bytes_read, pin.refused ? 1 : 0);
			if (quiet) fprintf(stderr, "fqpipe: %lu reads through the sink, %d copier threads, %.3f s, %.3f GB/s of text\n", (unsigned long)counted, pin.copiers, pin.seconds, pin.seconds > 0 ? (double)pin.bytes_read / pin.seconds / 1e9 : 0.0);
			vgh::FastqReader rd(fd, pin.span_base, pin.spans);
			vgh::ReadBatch rb;
			if (pin.records) { rd.seek(pin.last); rb.clear(); (void)rd.next(rb, 1); }
			rd.seek(pin.consumed);
			for (;;) {
				rb.clear();
				if (!rd.next(rb, 1000)) break;
				for (uint64_t i = 0; i < rb.n(); i++) {
					const uint64_t o = rb.offsets[i], nch = (rb.offsets[i + 1] - o) / 32;
					std::vector<uint64_t> km((size_t)nch, 0);
					uint64_t meta = 0;
					for (uint64_t c = 0; c < nch && !(meta >> 62); c++)
						for (int b = 31; b >= 0; b--) {                       // (the reference's scan order: the first offending character decides, qv.cc:815-828)
							const char ch = (char)(rb.bases[o + 32 * c + (uint64_t)b] & 0xDF);
							const int code = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : -1;
							if (code < 0) { meta |= ch == 'N' ? 1ull << 62 : 1ull << 63; break; }
							km[(size_t)c] |= (uint64_t)code << (2 * b);
						}
					for (uint64_t c = 0; c < nch && c < 32; c++) if ((int)(int8_t)rb.quals[o + c] - '8' < 0) meta |= 1ull << c;
					one(nch, km.data(), meta);
				}
			}
			for (const auto &l : lines) puts(l.c_str());
			return EXIT_SUCCESS;
		} else if (opt == "fqcuts") {
			// hidden: where `geno` with n replicas would cut the FASTQ file (no device needed; tests/test_host_tools.py)
			arg_check(argc, 2);
			const int fd = open(argv[2], O_RDONLY);
			struct stat sb;
			if (fd < 0 || fstat(fd, &sb) != 0) throw vgh::Error{std::string("cannot open ") + argv[2]};
			std::vector<uint64_t> cut;
			const bool ok = range_cuts(fd, (uint64_t)sb.st_size, std::max(1, atoi(argv[3])), cut);
			close(fd);
			if (!ok) { printf("none\n"); return EXIT_SUCCESS; }
			for (uint64_t c : cut) printf("%lu\n", (unsigned long)c);
			return EXIT_SUCCESS;
		} else if (opt == "callvcf") {
			// hidden (like the reference's vcfd/ucscd/filt): caller + VCF writer alone, from a counts table
			// "pos ref_freq alt_freq ref_cnt alt_cnt" per line: <chrlens> <counts.txt> <snps.vcf> <out.vcf>
			arg_check(argc, 4);
			vgh::SiteCounts sc;
			FILE *f = fopen(argv[3], "r");
			if (!f) throw vgh::Error{std::string("cannot open ") + argv[3]};
			unsigned long p; unsigned rf, af, rc, ac;
			while (fscanf(f, "%lu %u %u %u %u", &p, &rf, &af, &rc, &ac) == 5) {
				sc.pos.push_back((uint32_t)p); sc.ref_freq.push_back((uint8_t)rf); sc.alt_freq.push_back((uint8_t)af);
				sc.ref_cnt.push_back((uint8_t)rc); sc.alt_cnt.push_back((uint8_t)ac);
			}
			fclose(f);
			vgh::write_genotyped_vcf(sc, vgh::read_chrlens(argv[2]), argv[4], argv[5]);
			return EXIT_SUCCESS;
		} else if (opt == "version") {
			// hidden: the build ids (sha256 prefixes of the sources) of this binary and of the HIP library it loaded
#ifndef VG_HOST_BUILD_ID
#define VG_HOST_BUILD_ID "unknown"
#endif
			printf("host %s\nlib %s\n", VG_HOST_BUILD_ID, vg_build_id());
			return EXIT_SUCCESS;
		} else if (opt == "help") {
			print_help();
			return EXIT_SUCCESS;
		}
	} catch (const vgh::Error &e) {
		fprintf(stderr, "vargeno: %s\n", e.msg.c_str());
		return EXIT_FAILURE;
	}
	print_help();
	return EXIT_FAILURE;
}
