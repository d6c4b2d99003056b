// Randomised check of the host key index (iscc_search_amd/csrc/keymap.h) against std::unordered_map.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <unordered_map>

#include "../iscc_search_amd/csrc/keymap.h"

using namespace iskhost;

int run(bool wide, uint64_t seed, int ops, uint64_t key_space) {
    KeyMap m;
    m.reset(wide);
    std::unordered_map<Key, Loc, KeyHash> ref;
    std::mt19937_64 rng(seed);
    for (int i = 0; i < ops; ++i) {
        // clustered keys (small space, shared low bits) make long probe runs and wrap-arounds likely
        Key k{wide ? rng() % 3 : 0, (rng() % key_space) * 0x100000000ULL + (rng() % 4)};
        const int op = (int)(rng() % 10);
        if (op < 5) {
            Loc l{(uint32_t)(rng() % 33), rng() & ((1ULL << 56) - 1)};
            m.set(k, l);
            ref[k] = l;
        } else if (op < 8) {
            const bool a = m.erase(k);
            const bool b = ref.erase(k) > 0;
            if (a != b) { printf("erase mismatch at op %d\n", i); return 1; }
        } else {
            Loc l;
            const bool a = m.find(k, l);
            auto it = ref.find(k);
            if (a != (it != ref.end())) { printf("find mismatch at op %d\n", i); return 1; }
            if (a && (l.seg != it->second.seg || l.row != it->second.row)) { printf("value mismatch at op %d\n", i); return 1; }
        }
        if (m.size() != ref.size()) { printf("size mismatch at op %d: %zu vs %zu\n", i, m.size(), ref.size()); return 1; }
    }
    for (auto& kv : ref) {
        Loc l;
        if (!m.find(kv.first, l) || l.seg != kv.second.seg || l.row != kv.second.row) { printf("final lookup mismatch\n"); return 1; }
    }
    return 0;
}

int main() {
    for (uint64_t seed = 1; seed <= 6; ++seed)
        for (int wide = 0; wide < 2; ++wide) {
            if (run(wide != 0, seed, 200000, 50)) return 1;          // tiny key space: constant insert/erase churn
            if (run(wide != 0, seed + 100, 300000, 20000)) return 1; // growth through several rehashes
        }
    // bulk: reserve + 2M inserts + erase every other
    KeyMap m;
    m.reset(false);
    m.reserve(2000000);
    for (uint64_t i = 0; i < 2000000; ++i) m.set(Key{0, i * 2654435761ULL}, Loc{8, i});
    for (uint64_t i = 0; i < 2000000; i += 2) if (!m.erase(Key{0, i * 2654435761ULL})) { printf("bulk erase failed\n"); return 1; }
    for (uint64_t i = 0; i < 2000000; ++i) {
        Loc l;
        const bool f = m.find(Key{0, i * 2654435761ULL}, l);
        if (f != (i % 2 == 1) || (f && l.row != i)) { printf("bulk find mismatch at %llu\n", (unsigned long long)i); return 1; }
    }
    // composite simprint keys: consecutive asset ids x few (offset|size) values; must stay fast (no clustering)
    {
        KeyMap w;
        w.reset(true);
        const uint64_t assets = 100000, chunks = 40;
        w.reserve(assets * chunks);
        for (uint64_t a = 1; a <= assets; ++a)
            for (uint64_t c = 0; c < chunks; ++c) w.set(Key{a, (c * 100) << 32 | 100}, Loc{16, a * chunks + c});
        if (w.size() != assets * chunks) { printf("composite size mismatch\n"); return 1; }
        Loc l;
        for (uint64_t a = 1; a <= assets; a += 997)
            if (!w.find(Key{a, (7 * 100ULL) << 32 | 100}, l) || l.row != a * chunks + 7) { printf("composite find mismatch\n"); return 1; }
    }
    printf("keymap ok\n");
    return 0;
}
