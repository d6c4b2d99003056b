// keymap.h -- host key index of the column store (plain C++, no HIP): key -> (segment, row).
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace iskhost {

struct Key {
    uint64_t hi, lo;
    bool operator==(const Key& o) const { return hi == o.hi && lo == o.lo; }
};
struct KeyHash {
    // Full avalanche on both words (splitmix64 finaliser).  Linear probing needs it: composite simprint keys are
    // (small consecutive asset ids, a handful of offset|size values), which a multiply-xor hash maps to long runs
    // of consecutive slots (a 4 M-key build took 35 s before this).
    static uint64_t mix(uint64_t x) {
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
        x ^= x >> 27; x *= 0x94D049BB133111EBULL;
        return x ^ (x >> 31);
    }
    size_t operator()(const Key& k) const {
        return (size_t)mix(mix(k.lo + 0x9E3779B97F4A7C15ULL) ^ (k.hi * 0xD6E8FEB86659FD93ULL + 0x2545F4914F6CDD1DULL));
    }
};
struct Loc {
    uint32_t seg;   // segment = code length in bytes
    uint64_t row;
};

// Host key index: open addressing with linear probing and backward-shift deletion (no tombstones).
// 16 bytes per slot for 64-bit keys, 24 for 128-bit keys, load factor <= 0.6: about 3x smaller than a node-based
// std::unordered_map and free of per-entry allocations, which matters at 10^8 keys.
class KeyMap {
  public:
    void reset(bool wide) { wide_ = wide; lo_.clear(); hi_.clear(); loc_.clear(); mask_ = 0; size_ = 0; }
    size_t size() const { return size_; }
    void reserve(size_t n) { if (need_cap(n) > loc_.size()) rehash(need_cap(n)); }
    bool find(const Key& k, Loc& out) const {
        if (!size_) return false;
        for (size_t i = slot(k);; i = (i + 1) & mask_) {
            if (loc_[i] == EMPTY) return false;
            if (lo_[i] == k.lo && (!wide_ || hi_[i] == k.hi)) { out = unpack(loc_[i]); return true; }
        }
    }
    bool contains(const Key& k) const { Loc l; return find(k, l); }
    void set(const Key& k, const Loc& l) {
        if (need_cap(size_ + 1) > loc_.size()) rehash(need_cap(size_ + 1) * 2);
        for (size_t i = slot(k);; i = (i + 1) & mask_) {
            if (loc_[i] == EMPTY) { lo_[i] = k.lo; if (wide_) hi_[i] = k.hi; loc_[i] = pack(l); ++size_; return; }
            if (lo_[i] == k.lo && (!wide_ || hi_[i] == k.hi)) { loc_[i] = pack(l); return; }
        }
    }
    bool erase(const Key& k) {
        if (!size_) return false;
        size_t i = slot(k);
        for (;; i = (i + 1) & mask_) {
            if (loc_[i] == EMPTY) return false;
            if (lo_[i] == k.lo && (!wide_ || hi_[i] == k.hi)) break;
        }
        // backward-shift: pull later entries of the probe run into the hole
        size_t hole = i;
        for (size_t j = (i + 1) & mask_; loc_[j] != EMPTY; j = (j + 1) & mask_) {
            const size_t home = slot(Key{wide_ ? hi_[j] : 0, lo_[j]});
            // entry j may move to `hole` iff its home is not in the cyclic interval (hole, j]
            const bool in_between = hole <= j ? (home > hole && home <= j) : (home > hole || home <= j);
            if (!in_between) { lo_[hole] = lo_[j]; if (wide_) hi_[hole] = hi_[j]; loc_[hole] = loc_[j]; hole = j; }
        }
        loc_[hole] = EMPTY;
        --size_;
        return true;
    }

  private:
    static constexpr uint64_t EMPTY = ~0ULL;
    static uint64_t pack(const Loc& l) { return ((uint64_t)l.seg << 56) | l.row; }
    static Loc unpack(uint64_t v) { return Loc{(uint32_t)(v >> 56), v & ((1ULL << 56) - 1)}; }
    static size_t need_cap(size_t n) { size_t c = 16; while (c * 6 < n * 10) c <<= 1; return c; }
    size_t slot(const Key& k) const { return KeyHash()(k) & mask_; }
    void rehash(size_t cap) {
        std::vector<uint64_t> lo(cap), hi(wide_ ? cap : 0), loc(cap, EMPTY);
        lo.swap(lo_); hi.swap(hi_); loc.swap(loc_);
        mask_ = cap - 1;
        size_ = 0;
        for (size_t i = 0; i < loc.size(); ++i)
            if (loc[i] != EMPTY) set(Key{wide_ ? hi[i] : 0, lo[i]}, unpack(loc[i]));
    }
    std::vector<uint64_t> lo_, hi_, loc_;
    size_t mask_ = 0, size_ = 0;
    bool wide_ = false;
};

}  // namespace iskhost
