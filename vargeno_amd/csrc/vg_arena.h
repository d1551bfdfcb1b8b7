// vg_arena.h -- the device memory of one index handle: ONE block taken from the driver, carved up here (host code; the driver
// calls sit behind a backend type so that the bookkeeping runs against a mock on the CPU: tests/arena_mock.cpp).
//
// Why.  vg_index_open at hg38 scale ran ~1.2 s of kernels inside 10-16 s of wall time (round 4).  The rest was hipMalloc:
// the loader allocated and freed ~60 buffers of 1-64 GiB -- columns, sort buffers, views -- 600+ GB in all on a 309 GB device,
// so most allocations were handed memory that something had just freed, and that costs 13-17 GB/s (tools/alloc_probe,
// profiles/alloc_probe_r05.jsonl: the first 16 GiB hipMalloc of a process takes 0.3 ms, the next one, of the memory the first
// has just freed, 1.26 s; 64 GiB: 2.4-3.9 s; hipFree itself costs nothing.  profiles/index_open_phases_r05.txt: 10.7 of the
// 13.1 s of an open were spent inside allocation calls).  Recycling physical chunks under a reserved address range
// (hipMemCreate / hipMemMap / hipMemUnmap) does not work on this stack: a chunk that has been unmapped once is not usable
// under a second mapping (tools/vmm_probe, profiles/vmm_probe_r05.txt).
//
// What.  The block is as large as the index will be when it is finished (the plan knows every array's size before anything is
// allocated), and construction is ordered so that what is alive at any moment -- the permanent arrays built so far plus the
// temporaries of the current step -- never exceeds that.  Permanent arrays are placed from the bottom upwards,
// temporaries from the top downwards (highest fit), so that a temporary's space is under a permanent array a few steps
// later and long-lived temporaries never sit in a permanent array's way.  A request the block cannot serve returns nullptr and the
// caller falls back to hipMalloc (correct, merely slower): the arena changes how long the start-up takes, never what is built.
#pragma once
#include <stdint.h>

#include <iterator>
#include <map>

namespace vg {

// Backend B:  static void *alloc(uint64_t bytes);  static void free(void *p);
template <class B>
class DevArenaT {
public:
	DevArenaT() = default;
	DevArenaT(const DevArenaT &) = delete;
	DevArenaT &operator=(const DevArenaT &) = delete;
	~DevArenaT() { destroy(); }

	bool init(uint64_t bytes)
	{
		destroy();
		bytes = (bytes + ALIGN_BIG - 1) / ALIGN_BIG * ALIGN_BIG;
		if (bytes == 0) return false;
		void *p = B::alloc(bytes);
		if (!p) return false;
		base_ = (uint8_t *)p; size_ = bytes;
		free_.clear(); live_.clear();
		free_[0] = bytes;
		in_use_ = peak_ = 0;
		return true;
	}
	bool ready() const { return base_ != nullptr; }
	// Temporaries only at or above this offset (default 0: anywhere).  A block sized for a budget-limited index -- views left out
	// -- is smaller than what construction has alive at its peak: there the temporaries stay out of the block (the caller's
	// hipMalloc fallback serves them, transiently) so that every permanent array finds its place and the finished handle holds
	// exactly what was planned.
	void set_temp_floor(uint64_t off) { temp_floor_ = off; }
	uint64_t size() const { return size_; }
	uint64_t in_use() const { return in_use_; }
	uint64_t peak() const { return peak_; }
	bool owns(const void *p) const { return base_ && (const uint8_t *)p >= base_ && (const uint8_t *)p < base_ + size_; }

	// bytes of device memory, or nullptr when the block has no room for them where they belong
	void *take(uint64_t bytes, bool temporary)
	{
		if (!base_) return nullptr;
		bytes = (bytes + ALIGN_SMALL - 1) / ALIGN_SMALL * ALIGN_SMALL;
		if (bytes == 0) bytes = ALIGN_SMALL;
		const uint64_t al = bytes >= ALIGN_BIG ? ALIGN_BIG : ALIGN_SMALL;
		uint64_t at = 0;
		bool found = false;
		if (!temporary) {
			for (auto it = free_.begin(); it != free_.end() && !found; ++it) {                         // lowest fit
				const uint64_t b0 = it->first, b1 = b0 + it->second, a = (b0 + al - 1) / al * al;
				if (a + bytes <= b1) { at = a; found = true; carve(b0, b1, a, bytes); break; }
			}
		} else {
			for (auto it = free_.rbegin(); it != free_.rend() && !found; ++it) {                       // highest fit
				const uint64_t b0 = it->first, b1 = b0 + it->second;
				if (b1 - b0 < bytes) continue;
				const uint64_t a = (b1 - bytes) / al * al;
				if (a >= b0 && a >= temp_floor_) { at = a; found = true; carve(b0, b1, a, bytes); break; }
			}
		}
		if (!found) return nullptr;
		live_[at] = bytes;
		in_use_ += bytes;
		if (in_use_ > peak_) peak_ = in_use_;
		return base_ + at;
	}
	// the caller has made sure the device is done with it
	bool give(void *p)
	{
		if (!owns(p)) return false;
		const uint64_t at = (uint64_t)((uint8_t *)p - base_);
		auto it = live_.find(at);
		if (it == live_.end()) return false;
		uint64_t bytes = it->second;
		live_.erase(it);
		in_use_ -= bytes;
		uint64_t a = at;
		auto nx = free_.lower_bound(a);
		if (nx != free_.end() && nx->first == a + bytes) { bytes += nx->second; nx = free_.erase(nx); }
		if (nx != free_.begin()) {
			auto pv = std::prev(nx);
			if (pv->first + pv->second == a) { pv->second += bytes; return true; }
		}
		free_[a] = bytes;
		return true;
	}
	void destroy()
	{
		if (base_) B::free(base_);
		base_ = nullptr; size_ = 0; free_.clear(); live_.clear(); in_use_ = 0; temp_floor_ = 0;
	}

private:
	static constexpr uint64_t ALIGN_SMALL = 256, ALIGN_BIG = 2ull << 20;
	void carve(uint64_t b0, uint64_t b1, uint64_t a, uint64_t bytes)
	{
		free_.erase(b0);
		if (a > b0) free_[b0] = a - b0;
		if (a + bytes < b1) free_[a + bytes] = b1 - (a + bytes);
	}
	uint8_t *base_ = nullptr;
	uint64_t size_ = 0, in_use_ = 0, peak_ = 0, temp_floor_ = 0;
	std::map<uint64_t, uint64_t> free_, live_;                  // offset -> bytes
};

}  // namespace vg
