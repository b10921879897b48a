// Host-only stress test of hast_amd/csrc/ingest.h (no GPU): the sharded barcode dictionary hands out dense,
// consistent ids under concurrent lookups/inserts (built with -fsanitize=thread by tests/test_ingest_cpu.py),
// and the fork-join pool runs every worker exactly once per call.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "../../hast_amd/csrc/ingest.h"

int main(int argc, char **argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 8;
    const int n_keys = 20000, rounds = 200000;
    hast::BarcodeDict dict;
    std::vector<std::vector<uint32_t>> seen(T, std::vector<uint32_t>(n_keys, 0xFFFFFFFFu));
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            hast::BarcodeDict::Cache cache;
            uint64_t x = 0x9E3779B97F4A7C15ull * (uint64_t)(t + 1);
            for (int i = 0; i < rounds; i++) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                int k = (i % 5 == 0) ? 0 : (int)(x % n_keys);                 // key 0 is hot, like 0_0_0
                std::string name = k == 0 ? "0_0_0" : std::to_string(k % 1536 + 1) + "_" + std::to_string(k / 1536 + 1) + "_" + std::to_string(k);
                uint32_t id = dict.get(name, cache);
                if (seen[t][k] != 0xFFFFFFFFu && seen[t][k] != id) { fprintf(stderr, "id changed\n"); exit(1); }
                seen[t][k] = id;
            }
        });
    for (auto &x : th) x.join();
    // every thread saw the same id for the same key; ids are dense
    std::vector<std::string_view> names = dict.names();
    std::vector<char> used(names.size(), 0);
    for (int k = 0; k < n_keys; k++) {
        uint32_t id = 0xFFFFFFFFu;
        for (int t = 0; t < T; t++)
            if (seen[t][k] != 0xFFFFFFFFu) {
                if (id != 0xFFFFFFFFu && id != seen[t][k]) { fprintf(stderr, "threads disagree on key %d\n", k); return 1; }
                id = seen[t][k];
            }
        if (id == 0xFFFFFFFFu) continue;
        if (id >= names.size() || used[id]) { fprintf(stderr, "id %u not dense/unique\n", id); return 1; }
        used[id] = 1;
        std::string expect = k == 0 ? "0_0_0" : std::to_string(k % 1536 + 1) + "_" + std::to_string(k / 1536 + 1) + "_" + std::to_string(k);
        if (names[id] != expect) { fprintf(stderr, "name mismatch for id %u\n", id); return 1; }
    }
    for (size_t i = 0; i < used.size(); i++)
        if (!used[i]) { fprintf(stderr, "hole at id %zu\n", i); return 1; }
    // pool: each worker index exactly once per run, many runs
    hast::WorkerPool pool(T);
    std::vector<int> hits(T, 0);
    for (int r = 0; r < 2000; r++) pool.run([&](int w) { hits[w]++; });
    for (int w = 0; w < T; w++)
        if (hits[w] != 2000) { fprintf(stderr, "pool worker %d ran %d times\n", w, hits[w]); return 1; }
    printf("ok %zu barcodes\n", names.size());
    return 0;
}
