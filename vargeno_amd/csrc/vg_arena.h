// vg_arena.h -- the device memory of one index handle (host code over the HIP virtual-memory API; the API sits behind a
// backend type so that the bookkeeping runs against a mock on the CPU: tests/arena_mock.cpp).
//
// Why.  vg_index_open at hg38 scale ran ~1.2 s of kernels inside 10-16 s of wall time (round 4).  The rest was hipMalloc:
// the loader allocated and freed ~60 buffers of 1-64 GiB -- columns, sort buffers, views -- 600+ GB in all on a 309 GB device,
// so most allocations were handed memory that something had just freed, and the driver clears such memory before it hands it
// out, at 13-17 GB/s (tools/alloc_probe, profiles/alloc_probe_r05.jsonl: the first 16 GiB hipMalloc of a process takes 0.3 ms,
// the next one, of the memory the first has just freed, 1.26 s; 64 GiB: 2.4-3.9 s; hipFree itself costs nothing).
//
// What.  Physical memory is taken from the driver ONCE, in chunks (hipMemCreate), and recycled here: a chunk that a freed
// temporary no longer needs goes to a pool and is mapped under the next allocation (hipMemUnmap / hipMemMap: ~30 us a chunk)
// instead of back to the driver.  Addresses come from one reserved range twice the device's size (hipMemAddressReserve):
// permanent arrays from its bottom upwards, temporaries from its top downwards, so the two never meet and nothing fragments --
// address space is free, only mapped chunks are memory.  When construction is over the pooled chunks are released (trim): the
// handle then holds its permanent arrays and nothing else, to the chunk.
//
// A handle whose arena cannot be set up (no virtual-memory support, address space refused) allocates with hipMalloc / hipFree
// as before: the arena changes how long the start-up takes, never what is built.
#pragma once
#include <stdint.h>

#include <iterator>
#include <map>
#include <vector>

namespace vg {

// Backend B: `typedef ... Handle;`  bool granularity_ok(int device, uint64_t chunk);  uint8_t *reserve(uint64_t bytes);
// void unreserve(uint8_t *va, uint64_t bytes);  bool create(int device, uint64_t bytes, Handle *h);  void release(Handle h);
// bool map(uint8_t *at, uint64_t bytes, Handle h, int device);  void unmap(uint8_t *at, uint64_t bytes);
template <class B>
class DevArenaT {
	typedef typename B::Handle Handle;
public:
	static constexpr uint64_t CHUNK = 1ull << 30;          // physical granule: what trim() can give back
	DevArenaT() = default;
	DevArenaT(const DevArenaT &) = delete;
	DevArenaT &operator=(const DevArenaT &) = delete;
	~DevArenaT() { destroy(); }

	// reserve the address range (no memory yet).  false: this handle allocates the old way.
	bool init(int device, uint64_t device_total_bytes)
	{
		if (va_) return true;
		device_ = device;
		if (!B::granularity_ok(device, CHUNK)) return false;
		const uint64_t half = ((device_total_bytes ? device_total_bytes : (320ull << 30)) + 2 * CHUNK - 1) / CHUNK * CHUNK;
		uint8_t *p = B::reserve(2 * half);
		if (!p) return false;
		va_ = p; size_ = 2 * half; perm_top_ = 0;
		slot_h_.assign(size_ / CHUNK, Handle{});
		slot_ref_.assign(size_ / CHUNK, 0u);
		temp_free_.clear();
		temp_free_[half] = half;                               // temporaries: [half, 2 half), first fit from the top
		return true;
	}
	bool ready() const { return va_ != nullptr && !broken_; }
	bool owns(const void *p) const { return va_ && (const uint8_t *)p >= va_ && (const uint8_t *)p < va_ + size_; }

	// bytes of device memory; nullptr when the arena cannot serve it (out of device memory, or not set up): the caller falls back
	void *take(uint64_t bytes, bool temporary)
	{
		if (!ready()) return nullptr;
		bytes = (bytes + 255) & ~255ull;
		if (bytes == 0) bytes = 256;
		const uint64_t al = bytes >= (2ull << 20) ? (2ull << 20) : 256ull;
		uint64_t at = 0;
		if (!temporary) {
			at = (perm_top_ + al - 1) / al * al;
			if (at + bytes > size_ / 2) return nullptr;
		} else {
			bool found = false;
			for (auto it = temp_free_.rbegin(); it != temp_free_.rend(); ++it) {                       // highest block first
				const uint64_t b0 = it->first, b1 = b0 + it->second;
				if (b1 - b0 < bytes) continue;
				const uint64_t a = (b1 - bytes) / al * al;
				if (a < b0) continue;
				at = a;
				// carve [a, a + bytes) out of [b0, b1)
				const uint64_t tail0 = a + bytes, tail = b1 - tail0;
				temp_free_.erase(b0);
				if (a > b0) temp_free_[b0] = a - b0;
				if (tail) temp_free_[tail0] = tail;
				found = true;
				break;
			}
			if (!found) return nullptr;
		}
		if (!back(at, bytes)) {
			if (temporary) put_back(at, bytes);
			return nullptr;
		}
		if (!temporary) perm_top_ = at + bytes;
		live_[at] = bytes;
		in_use_ += bytes;
		return va_ + at;
	}
	// a temporary (or a permanent array that is not needed any more) goes back; the caller has made sure the device is done with it
	bool give(void *p)
	{
		if (!owns(p)) return false;
		const uint64_t at = (uint64_t)((uint8_t *)p - va_);
		auto it = live_.find(at);
		if (it == live_.end()) return false;
		const uint64_t bytes = it->second;
		live_.erase(it);
		in_use_ -= bytes;
		unback(at, bytes);
		if (at >= size_ / 2) put_back(at, bytes);          // (addresses of the permanent half are not reused: there is no shortage of them)
		return true;
	}
	// construction is over: pooled chunks go back to the driver
	void trim()
	{
		for (auto h : pool_) B::release(h);
		pool_.clear();
	}
	void destroy()
	{
		if (!va_) return;
		for (size_t i = 0; i < slot_ref_.size(); i++) if (slot_ref_[i]) { B::unmap(va_ + i * CHUNK, CHUNK); B::release(slot_h_[i]); slot_ref_[i] = 0; }
		trim();
		B::unreserve(va_, size_);
		va_ = nullptr; size_ = 0; live_.clear(); temp_free_.clear(); mapped_ = 0; in_use_ = 0;
	}
	uint64_t mapped_bytes() const { return mapped_ * CHUNK; }                       // memory under live allocations
	uint64_t held_bytes() const { return (mapped_ + pool_.size()) * CHUNK; }        // ... plus pooled chunks
	uint64_t peak_bytes() const { return peak_ * CHUNK; }
	uint64_t created_chunks() const { return created_; }
	uint64_t remaps() const { return remaps_; }

private:
	void put_back(uint64_t at, uint64_t bytes)
	{
		auto nx = temp_free_.lower_bound(at);
		if (nx != temp_free_.end() && nx->first == at + bytes) { bytes += nx->second; nx = temp_free_.erase(nx); }
		if (nx != temp_free_.begin()) {
			auto pv = std::prev(nx);
			if (pv->first + pv->second == at) { pv->second += bytes; return; }
		}
		temp_free_[at] = bytes;
	}
	bool back(uint64_t at, uint64_t bytes)
	{
		const uint64_t s0 = at / CHUNK, s1 = (at + bytes - 1) / CHUNK;
		for (uint64_t s = s0; s <= s1; s++) {
			if (slot_ref_[s]++ != 0) continue;
			Handle h;
			bool fresh = false;
			if (!pool_.empty()) { h = pool_.back(); pool_.pop_back(); remaps_++; }
			else { if (!B::create(device_, CHUNK, &h)) { slot_ref_[s]--; undo(s0, s); return false; } fresh = true; created_++; }
			if (!B::map(va_ + s * CHUNK, CHUNK, h, device_)) {
				broken_ = true;                                       // mapping a chunk we hold must not fail: stop using the arena
				if (fresh) B::release(h); else pool_.push_back(h);
				slot_ref_[s]--; undo(s0, s);
				return false;
			}
			slot_h_[s] = h;
			mapped_++;
			if (mapped_ + pool_.size() > peak_) peak_ = mapped_ + pool_.size();
		}
		return true;
	}
	void undo(uint64_t s0, uint64_t s_end)                       // slots [s0, s_end) were referenced by a failed back()
	{
		for (uint64_t s = s0; s < s_end; s++) drop(s);
	}
	void drop(uint64_t s)
	{
		if (--slot_ref_[s] != 0) return;
		B::unmap(va_ + s * CHUNK, CHUNK);
		pool_.push_back(slot_h_[s]);
		mapped_--;
	}
	void unback(uint64_t at, uint64_t bytes)
	{
		const uint64_t s0 = at / CHUNK, s1 = (at + bytes - 1) / CHUNK;
		for (uint64_t s = s0; s <= s1; s++) drop(s);
	}

	uint8_t *va_ = nullptr;
	uint64_t size_ = 0, perm_top_ = 0;
	bool broken_ = false;
	int device_ = 0;
	std::vector<Handle> slot_h_, pool_;
	std::vector<uint32_t> slot_ref_;
	std::map<uint64_t, uint64_t> temp_free_, live_;             // offset -> bytes
	uint64_t mapped_ = 0, peak_ = 0, in_use_ = 0, created_ = 0, remaps_ = 0;
};

}  // namespace vg
