// pangenes_main.cpp — native host over the C ABI: what `java … infoasys.cli.pangenes.Pangenes -i in.faa -k K -o out.net`
// does (pandelos.sh:73), for machines without a JVM.
//
//   ig/infoasys/cli/pangenes/Cli.java:13-57          flags -i/--input -k/--kvalue -o/--output (required), -c/--complexity,
//                                                    -j/--threads (accepted, unused: the device pass is not threaded), -h
//   ig/infoasys/cli/pangenes/PangeneIData.java:30-75 .faa reader: the library's streaming ingest (pdl_ingest_faa, pdl_ingest.hip)
//   ig/infoasys/cli/pangenes/Pangenes.java:60-183    per-genome task: bidirectional-best-hit filter, both phases
//   ig/infoasys/cli/pangenes/PangeneNet.java:49-62,159-179   first insert per (src,dst) wins; undirected save in
//                                                    java.util.HashMap iteration order, edges by ascending destination
// Single-thread order (-j 1).  The same logic exists as array code in pandelos_amd/pangenes.py; tests compare both with
// oracle/pangenes_host.py.  Nothing of the reference pins the Java host (no JVM in the build image, no reference tests).
#include "../../include/pandelos_amd.h"

#include <algorithm>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

namespace {

// Double.toString of a float widened to double: shortest round-trip digits; decimal layout for 1e-3 <= x < 1e7,
// "d.dddE-n" otherwise
std::string java_double(double x) {
    if (x == 0) return "0.0";
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), x, std::chars_format::scientific);
    std::string sci(buf, r.ptr);                     // d[.ddd]e[+-]XX
    const size_t epos = sci.find('e');
    std::string mant = sci.substr(0, epos);
    const int e10 = atoi(sci.c_str() + epos + 1);
    std::string digits;
    for (char ch : mant) if (ch != '.' && ch != '-') digits.push_back(ch);
    const std::string sign = x < 0 ? "-" : "";
    const double ax = std::fabs(x);
    if (ax >= 1e-3 && ax < 1e7) {
        std::string ip, fp;
        if (e10 >= 0) {
            ip = digits.substr(0, std::min<size_t>(digits.size(), (size_t) e10 + 1));
            ip.append((size_t) e10 + 1 - ip.size(), '0');
            fp = digits.size() > (size_t) e10 + 1 ? digits.substr((size_t) e10 + 1) : "0";
        } else {
            ip = "0";
            fp = std::string((size_t) (-e10 - 1), '0') + digits;
        }
        return sign + ip + "." + fp;
    }
    return sign + digits.substr(0, 1) + "." + (digits.size() > 1 ? digits.substr(1) : "0") + "E" + std::to_string(e10);
}

// PangeneNet (PangeneNet.java:38-62,159-179) as flat arrays.  The Java container is HashMap<Integer, TreeSet<Edge>>: per source a set
// keyed by destination (the first insert per (src, dst) wins), saved undirected (src <= dst) in the map's iteration order — buckets
// of the final power-of-two table, insertion order inside a bucket — with a source's edges by ascending destination.  The same
// order comes out of three counting sorts over the edges in insertion order (a tree node and a hash node per edge made the
// container + text 54 ms of a 66-ms .faa -> .net run on the 64-genome set; this form: a few ms).
struct Net {
    std::vector<int32_t> src, dst;
    std::vector<float> score;
    void add(int32_t s, int32_t d, float sc) { src.push_back(s); dst.push_back(d); score.push_back(sc); }

    // -> the text of saveToFile(file, false), in pieces that follow each other (one per host thread: the sources in map order
    //    are cut into stretches, every thread formats its own)
    std::vector<std::string> text(unsigned threads = 8) const {
        const size_t E = src.size();
        std::vector<std::string> pieces;
        if (!E) return pieces;
        int32_t max_id = 0;
        for (size_t i = 0; i < E; i++) max_id = std::max(max_id, std::max(src[i], dst[i]));
        const size_t N = (size_t) max_id + 1;
        // sources in first-insertion order (= insertion order of the map's keys) and their edge counts
        std::vector<uint32_t> cnt(N + 1, 0), first_seen(N, 0xffffffffu);
        std::vector<int32_t> keys;
        for (size_t i = 0; i < E; i++) {
            if (first_seen[src[i]] == 0xffffffffu) { first_seen[src[i]] = (uint32_t) keys.size(); keys.push_back(src[i]); }
            cnt[src[i] + 1]++;
        }
        for (size_t g = 0; g < N; g++) cnt[g + 1] += cnt[g];
        // edges grouped by source, insertion order kept inside a group (stable counting sort)
        std::vector<uint32_t> at(cnt.begin(), cnt.end() - 1), by_src(E);
        for (size_t i = 0; i < E; i++) by_src[at[src[i]]++] = (uint32_t) i;
        // HashMap iteration order of the keys: bucket (h ^ h >>> 16) & (cap - 1) of the final table, insertion order inside
        size_t cap = 16;
        while ((double) keys.size() > 0.75 * (double) cap) cap *= 2;
        auto bucket = [&](int32_t key) { const uint32_t h = (uint32_t) key; return (size_t) ((h ^ (h >> 16)) & (uint32_t) (cap - 1)); };
        std::vector<uint32_t> bcnt(cap + 1, 0);
        for (int32_t k : keys) bcnt[bucket(k) + 1]++;
        for (size_t b = 0; b < cap; b++) bcnt[b + 1] += bcnt[b];
        std::vector<int32_t> ordered(keys.size());
        for (int32_t k : keys) ordered[bcnt[bucket(k)]++] = k;           // (keys are visited in insertion order: stable)
        const unsigned T = (unsigned) std::max<size_t>(1, std::min<size_t>(threads, ordered.size() / 4096 + 1));
        pieces.resize(T);
        auto format = [&](unsigned t) {
            std::string &out = pieces[t];
            const size_t k0 = ordered.size() * t / T, k1 = ordered.size() * (t + 1) / T;
            out.reserve((size_t) ((double) E / T * 20));
            std::vector<std::pair<int32_t, uint32_t>> seg;               // {destination, edge} of one source
            char buf[64];
            for (size_t ki = k0; ki < k1; ki++) {
                const int32_t s_id = ordered[ki];
                seg.clear();
                for (uint32_t q = cnt[s_id]; q < cnt[s_id + 1]; q++) seg.emplace_back(dst[by_src[q]], by_src[q]);
                std::stable_sort(seg.begin(), seg.end(), [](const auto &x, const auto &y) { return x.first < y.first; });   // (a handful per source)
                for (size_t q = 0; q < seg.size(); q++) {
                    if (q && seg[q].first == seg[q - 1].first) continue;     // TreeSet keyed by destination: the first insert stays
                    if (s_id > seg[q].first) continue;                       // undirected save
                    auto r = std::to_chars(buf, buf + sizeof(buf), s_id);
                    out.append(buf, r.ptr); out.push_back('\t');
                    r = std::to_chars(buf, buf + sizeof(buf), seg[q].first);
                    out.append(buf, r.ptr); out.push_back('\t');
                    const double x = (double) score[seg[q].second];
                    if (x >= 1e-3 && x < 1e7) {                              // Double.toString's plain decimal range: shortest round-trip digits, at least one after the point
                        r = std::to_chars(buf, buf + sizeof(buf), x, std::chars_format::fixed);
                        out.append(buf, r.ptr);
                        if (!memchr(buf, '.', (size_t) (r.ptr - buf))) out.append(".0");
                    } else out.append(java_double(x));
                    out.push_back('\n');
                }
            }
        };
        std::vector<std::thread> team;
        for (unsigned t = 1; t < T; t++) team.emplace_back(format, t);
        format(0);
        for (auto &th : team) th.join();
        return pieces;
    }
};

void usage() {
    printf("usage: PanDelos [OPTIONS]\n"
           " -c,--complexity      Compute the required number of operations without computing the network (fast)\n"
           " -h,--help            Print this help message\n"
           " -i,--input <arg>     Input file (.faa) to process\n"
           " -j,--threads <arg>   Number of threads to use for the computation, defaults to # of processors\n"
           " -k,--kvalue <arg>    Length of the kmers used by the algorithm\n"
           " -o,--output <arg>    Output file for the network\n");
}

}  // namespace

int main(int argc, char **argv) {
    std::string input, output;
    int k = 0;
    bool have_k = false, complexity = false, k_auto = false;
    std::string timings_path;      // --timings FILE (not in the reference): wall time of every stage of the pipeline, as one JSON line
    int repeat = 1;                // --repeat N: the whole pipeline N times in this process (the first pass allocates), medians reported
    int genome_batch = 0;          // --genome-batch N (not in the reference): score N genomes at a time on one dictionary — sets whose maxima,
                                   // staging and cells do not fit the device together (pdl_set_genome_shard on an existing dictionary)
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * { return i + 1 < argc ? argv[++i] : nullptr; };
        if (a == "-h" || a == "--help") { usage(); return 0; }
        else if (a == "-c" || a == "--complexity") complexity = true;
        else if (a == "-i" || a == "--input") { const char *v = val(); if (v) input = v; }
        else if (a == "-o" || a == "--output") { const char *v = val(); if (v) output = v; }
        else if (a == "-k" || a == "--kvalue") {
            const char *v = val();
            if (v) { k_auto = strcmp(v, "auto") == 0; k = k_auto ? 0 : atoi(v); have_k = true; }      // "auto" (not in the reference): calculate_k.py's value, computed by the ingest pass
        }
        else if (a == "-j" || a == "--threads") { (void) val(); }
        else if (a == "--timings") { const char *v = val(); if (v) timings_path = v; }
        else if (a == "--repeat") { const char *v = val(); if (v) repeat = std::max(1, atoi(v)); }
        else if (a == "--genome-batch") { const char *v = val(); if (v) genome_batch = std::max(0, atoi(v)); }
        else { input.clear(); break; }
    }
    if (input.empty() || output.empty() || !have_k) {           // Cli.java:83-87
        printf("Error while parsing cli arguments!\n");
        usage();
        return 1;
    }

    // ---- PangeneIData.readFromFile: the library's streaming ingest (pdl_ingest.hip) — the file goes to HBM as it is parsed ----
    pdl_ctx *ctx = pdl_create(nullptr);
    if (!ctx) { fprintf(stderr, "pandelos_amd: %s\n", pdl_last_error(nullptr)); return 1; }
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
    std::vector<std::vector<double>> stage(6);       // ingest | dictionary | scores | edges | net (container + text) | all
    uint64_t n_edges_total = 0, file_bytes = 0, n_cells = 0;
    uint32_t n_genes = 0, n_genomes = 0;
  for (int pass = 0; pass < repeat; pass++) {
    const bool quiet = pass > 0;
    const auto t_all = clk::now();
    auto t0 = clk::now();
    pdl_ingest ing;
    if (pdl_ingest_faa(ctx, input.c_str(), &ing) != PDL_OK) {
        // Pangenes.java:26-31: the reader's exception is printed and main returns
        fprintf(stderr, "java.io.FileNotFoundException / reader: %s\n", pdl_last_error(ctx));
        pdl_destroy(ctx);
        return 0;
    }
    stage[0].push_back(ms_since(t0));
    file_bytes = ing.file_bytes; n_genes = ing.sequences; n_genomes = ing.genomes;
    const uint32_t G = ing.genomes;
    if (k_auto) { k = ing.k_suggested; if (!quiet) printf("k = %d\n", k); }                // calculate_k.py:30 (the line pandelos.sh:67-68 greps for)
    if (k <= 0) { printf("K value must be greater than 0."); return 1; }      // library.cpp:90-93
    t0 = clk::now();
    pdl_cost cost;
    const bool batched = genome_batch > 0 && (uint32_t) genome_batch < G && !complexity;
    auto set_batch = [&](uint32_t g0) {
        std::vector<uint32_t> ids;
        for (uint32_t g = g0; g < std::min<uint32_t>(G, g0 + (uint32_t) genome_batch); g++) ids.push_back(g);
        return pdl_set_genome_shard(ctx, ids.data(), (uint32_t) ids.size());
    };
    if (batched) {
        (void) pdl_set_option(ctx, "low_memory", 1);
        if (set_batch(0) != PDL_OK) { fprintf(stderr, "pandelos_amd: %s\n", pdl_last_error(ctx)); return 1; }
    }
    if (pdl_preprocess_ingested(ctx, k, complexity ? 1 : 0, &cost) != PDL_OK) {
        fprintf(stderr, "pandelos_amd: %s\n", pdl_last_error(ctx));
        return 1;
    }
    stage[1].push_back(ms_since(t0));
    if (!quiet) {
        if (cost.hash_fallback) printf("Hashing fallback!\n");
        if (batched) printf("------------\nCOMPUTATIONAL COSTS: \n(genome batches of %d: the lookups are counted batch by batch)\n------------\n\n", genome_batch);
        else printf("------------\nCOMPUTATIONAL COSTS: \nTotal cost: %llu lookups\nLinear ratio: %g\n------------\n\n",
                    (unsigned long long) cost.total_cost, (double) cost.linear_ratio);
    }
    if (complexity) { pdl_destroy(ctx); return 0; }                           // Pangenes.java:33-36

    // ---- per-genome tasks, Pangenes.java:60-183 ------------------------------------------------------------
    // The best-hit filter of the task (:98-176) runs on the device, over the cells where they are (pdl_compute_edges,
    // pdl_bbh.hip); what arrives here are the task's addConnection calls in order.
    t0 = clk::now();
    if (pdl_score_all(ctx) != PDL_OK) { fprintf(stderr, "pandelos_amd: %s\n", pdl_last_error(ctx)); return 1; }
    stage[2].push_back(ms_since(t0));
    t0 = clk::now();
    Net net;
    std::vector<uint32_t> counts(G, 0), bcounts(G, 0);
    std::vector<uint64_t> gcosts(G, 0);
    std::vector<pdl_edges> per_genome(G);
    uint64_t n_edges = 0;
    n_cells = 0;
    for (uint32_t g0 = 0; g0 < G; g0 += batched ? (uint32_t) genome_batch : G) {
        const uint32_t g1 = batched ? std::min<uint32_t>(G, g0 + (uint32_t) genome_batch) : G;
        if (batched && g0 && set_batch(g0) != PDL_OK) { fprintf(stderr, "pandelos_amd: %s\n", pdl_last_error(ctx)); return 1; }      // (its range lists are built before it is scored)
        if (pdl_scores_counts(ctx, bcounts.data()) != PDL_OK) { fprintf(stderr, "pandelos_amd: %s\n", pdl_last_error(ctx)); return 1; }
        for (uint32_t g = g0; g < g1; g++) {
            counts[g] = bcounts[g]; n_cells += counts[g];
            if (pdl_compute_edges(ctx, g, &per_genome[g]) != PDL_OK) { fprintf(stderr, "pandelos_amd: %s\n", pdl_last_error(ctx)); return 1; }
            n_edges += per_genome[g].count;
            (void) pdl_genome_cost(ctx, g, &gcosts[g]);
        }
    }
    stage[3].push_back(ms_since(t0));
    n_edges_total = n_edges;
    t0 = clk::now();
    net.src.reserve(n_edges); net.dst.reserve(n_edges); net.score.reserve(n_edges);
    for (uint32_t g = 0; g < G; g++) {
        pdl_edges &e = per_genome[g];
        if (!quiet) printf("Genome %u cost = %llu\nFiltered count: %u\n", g, (unsigned long long) gcosts[g], counts[g]);   // library.cpp:535-538, Pangenes.java:68
        for (uint32_t i = 0; i < e.count; i++) net.add(e.src[i], e.dst[i], e.score[i]);
        pdl_free_edges(&e);
    }

    // ---- PangeneNet.saveToFile(file, false) ---------------------------------------------------------------
    if (!quiet) printf("----------\nwriting into %s\n", output.c_str());
    const std::vector<std::string> text = net.text();
    FILE *f = fopen(output.c_str(), "w");
    if (!f) { perror(output.c_str()); return 0; }
    for (const std::string &piece : text) fwrite(piece.data(), 1, piece.size(), f);
    fclose(f);
    stage[4].push_back(ms_since(t0));
    stage[5].push_back(ms_since(t_all));
  }
    pdl_destroy(ctx);
    if (!timings_path.empty()) {
        auto median = [](std::vector<double> v) { if (v.size() > 1) v.erase(v.begin()); std::sort(v.begin(), v.end()); return v[v.size() / 2]; };   // (first pass allocates: left out when there are more)
        FILE *t = fopen(timings_path.c_str(), "w");
        if (t) {
            fprintf(t, "{\"passes\": %d, \"file_bytes\": %llu, \"genes\": %u, \"genomes\": %u, \"cells\": %llu, \"edges_added\": %llu, "
                       "\"faa_to_hbm_ms\": %.4f, \"dictionary_ms\": %.4f, \"scores_ms\": %.4f, \"bbh_edges_to_host_ms\": %.4f, \"net_ms\": %.4f, \"faa_to_net_ms\": %.4f}\n",
                    repeat, (unsigned long long) file_bytes, n_genes, n_genomes, (unsigned long long) n_cells, (unsigned long long) n_edges_total,
                    median(stage[0]), median(stage[1]), median(stage[2]), median(stage[3]), median(stage[4]), median(stage[5]));
            fclose(t);
        }
    }
    return 0;
}
