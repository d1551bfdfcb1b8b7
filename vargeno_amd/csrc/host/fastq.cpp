// fastq.cpp -- FASTQ framing with the reference's semantics (src/qv.cc:699-784): four fgets() calls
// of at most 1023 characters per record (id, read, separator, quality); the read length is
// strlen(read) - 1 whatever the last character is; chunk c is gated by the c-th character of the
// quality LINE.  Records are packed into the flat batch layout the C-ABI takes.
#include <errno.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include "vg_host.h"

namespace vgh {

struct FastqReader::Impl {
	FILE *f = nullptr;
	// a stream that can be read only once (a pipe, qv.cc:2182 fopen()s whatever it is given): bytes [base, base + the spans' lengths) of
	// it are in the caller's memory, what follows comes from the descriptor
	int rawfd = -1;
	uint64_t base = 0;
	std::vector<std::pair<const uint8_t *, size_t>> spans;
	size_t span_i = 0;
	std::vector<char> buf;
	const char *cur = nullptr;                    // [pos, end) of the bytes at hand
	size_t pos = 0, end = 0;
	bool eof = false;
	char line[4][1024];
	void refill()
	{
		pos = end = 0;
		if (f) { end = fread(buf.data(), 1, buf.size(), f); cur = buf.data(); return; }
		while (span_i < spans.size() && spans[span_i].second == 0) span_i++;
		if (span_i < spans.size()) { cur = (const char *)spans[span_i].first; end = spans[span_i].second; span_i++; return; }
		if (rawfd >= 0) {
			ssize_t g;
			do g = read(rawfd, buf.data(), buf.size()); while (g < 0 && errno == EINTR);
			end = g > 0 ? (size_t)g : 0;
			cur = buf.data();
		}
	}
	// fgets(dst, 1024, f): up to 1023 chars, stops after '\n'; false at end of file with nothing read
	bool gets(char *dst)
	{
		size_t n = 0;
		for (;;) {
			if (pos == end) {
				if (eof) break;
				refill();
				if (end == 0) { eof = true; break; }
			}
			const char *s = cur + pos;
			const size_t avail = end - pos, room = 1023 - n;
			const size_t take = avail < room ? avail : room;
			const char *nl = (const char *)memchr(s, '\n', take);
			const size_t m = nl ? (size_t)(nl - s) + 1 : take;
			memcpy(dst + n, s, m);
			n += m; pos += m;
			if (nl || n == 1023) break;
		}
		if (n == 0) return false;                 // fgets() returning NULL leaves the buffer untouched (qv.cc:761-763 rely on it)
		dst[n] = '\0';
		return true;
	}
};

FastqReader::FastqReader(const std::string &path) : p(new Impl)
{
	p->f = fopen(path.c_str(), "r");
	if (!p->f) { delete p; throw Error{"cannot open " + path}; }
	p->buf.resize(1 << 24);
	for (auto &l : p->line) memset(l, 0, sizeof l);
}
FastqReader::FastqReader(int fd, uint64_t base, std::vector<std::pair<const uint8_t *, size_t>> spans) : p(new Impl)
{
	p->rawfd = fd;
	p->base = base;
	p->spans = std::move(spans);
	p->buf.resize(1 << 22);
	for (auto &l : p->line) memset(l, 0, sizeof l);
}
FastqReader::~FastqReader() { if (p->f) fclose(p->f); delete p; }

void FastqReader::seek(uint64_t off)
{
	p->pos = p->end = 0;
	p->eof = false;
	if (p->f) { fseeko(p->f, (off_t)off, SEEK_SET); return; }
	// a once-only stream: only offsets inside the bytes held in memory (or their end) can be gone back to
	uint64_t at = p->base;
	p->span_i = 0;
	if (off < at) throw Error{"FASTQ stream: cannot go back before the bytes kept in memory"};
	for (; p->span_i < p->spans.size(); p->span_i++) {
		const uint64_t len = p->spans[p->span_i].second;
		if (off < at + len) {
			p->cur = (const char *)p->spans[p->span_i].first;
			p->pos = (size_t)(off - at); p->end = (size_t)len;
			p->span_i++;
			return;
		}
		at += len;
	}
	if (off != at) throw Error{"FASTQ stream: cannot skip ahead on a stream that is read once"};
}

uint64_t FastqReader::next(ReadBatch &out, uint64_t max_reads)
{
	if (out.offsets.empty()) out.offsets.assign(1, 0);
	uint64_t got = 0;
	while (got < max_reads) {
		if (!p->gets(p->line[0])) break;                        // while (fgets(id, ...)), qv.cc:760
		// a NULL return leaves the previous record's buffer in place in the reference (qv.cc:761-763);
		// a truncated final record is therefore processed with stale lines.  Reproduced as is.
		(void)p->gets(p->line[1]);
		(void)p->gets(p->line[2]);
		(void)p->gets(p->line[3]);
		const size_t sl = strlen(p->line[1]);
		const size_t rlen = sl ? sl - 1 : 0;                    // strlen(read) - 1, qv.cc:778 (size_t wrap on an empty buffer is not reproduced)
		const size_t at = out.bases.size();
		out.bases.insert(out.bases.end(), p->line[1], p->line[1] + rlen);
		out.quals.resize(at + rlen, 0);
		// qual[c] for c < rlen/32 is all the path reads, straight out of the 1024-byte buffer: a quality line shorter than that
		// shows its newline, its NUL, and then whatever earlier lines left there -- exactly what the reference sees (qv.cc:836)
		memcpy(out.quals.data() + at, p->line[3], rlen);
		out.offsets.push_back(out.bases.size());
		got++;
	}
	return got;
}

}  // namespace vgh
