// host_path — SURVEY.md §8d's wall time through the C ABI alone: host arrays (parsed from a .faa by pdl_scan_faa) ->
// pdl_preprocess -> pdl_compute_scores for every genome, the G calls made from a pool of host threads as the reference's are
// (Executors.newFixedThreadPool, Pangenes.java:54-66) -> every Scores block in host memory (pdl_free_scores'd again: the Java
// side copies them into its own arrays).  bench.py runs it beside its Python-binding figure, which pays ten numpy copies per call.
// usage: host_path <in.faa> <k> <threads> <iterations>      -> one JSON line
#include "../../include/pandelos_amd.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: host_path <in.faa> <k> <threads> <iterations>\n"); return 2; }
    const char *path = argv[1];
    const int k = atoi(argv[2]), threads = std::max(1, atoi(argv[3])), iters = std::max(1, atoi(argv[4]));
    pdl_ingest in;
    if (pdl_scan_faa(path, &in, nullptr, 0, nullptr, nullptr, 0) != PDL_OK) { fprintf(stderr, "%s\n", pdl_last_error(nullptr)); return 1; }
    std::vector<uint8_t> res(in.residues + 16);
    std::vector<uint64_t> off(in.sequences + 1);
    std::vector<uint32_t> gen(in.sequences);
    if (pdl_scan_faa(path, &in, res.data(), in.residues, off.data(), gen.data(), in.sequences) != PDL_OK) { fprintf(stderr, "%s\n", pdl_last_error(nullptr)); return 1; }
    pdl_ctx *ctx = pdl_create(nullptr);
    if (!ctx) { fprintf(stderr, "%s\n", pdl_last_error(nullptr)); return 1; }
    std::vector<double> ms;
    unsigned long long cells = 0;
    for (int it = 0; it <= iters; it++) {                  // (the first pass allocates: not counted)
        const auto t0 = std::chrono::steady_clock::now();
        if (pdl_preprocess(ctx, res.data(), off.data(), gen.data(), in.sequences, k, 0, nullptr) != PDL_OK) { fprintf(stderr, "%s\n", pdl_last_error(ctx)); return 1; }
        std::atomic<uint32_t> next{0};
        std::atomic<unsigned long long> z{0};
        std::atomic<int> failed{0};
        auto work = [&]() {
            for (;;) {
                const uint32_t g = next.fetch_add(1);
                if (g >= in.genomes) break;
                pdl_scores s;
                if (pdl_compute_scores(ctx, g, &s) != PDL_OK) { failed = 1; break; }
                z += s.scoresCount;
                pdl_free_scores(&s);
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; t++) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
        if (failed) { fprintf(stderr, "%s\n", pdl_last_error(ctx)); return 1; }
        cells = z;
        if (it) ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    pdl_destroy(ctx);
    std::sort(ms.begin(), ms.end());
    double mean = 0; for (double v : ms) mean += v; mean /= (double) ms.size();
    printf("{\"ms\": %.4f, \"ms_min\": %.4f, \"ms_mean\": %.4f, \"iterations\": %zu, \"threads\": %d, \"cells\": %llu, \"genes\": %u, \"genomes\": %u}\n",
           ms[ms.size() / 2], ms.front(), mean, ms.size(), threads, cells, in.sequences, in.genomes);
    return 0;
}
