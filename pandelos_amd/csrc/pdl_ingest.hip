// pdl_ingest.hip — K-ingest: a `.faa` file goes from the page cache to HBM in one pass.
//
// What it follows, line for line of behaviour:
//   ig/infoasys/cli/pangenes/PangeneIData.java:30-75   readFromFile: BufferedReader.readLine (terminators \n, \r, \r\n),
//                                                      String.trim (chars <= U+0020 at both ends), blank lines skipped, the
//                                                      others alternate header / sequence; header = genome \t gene \t product
//                                                      (cc[1], cc[2] are indexed, so a header with fewer than three fields
//                                                      throws); genome ids dense, in first-seen order
//   calculate_k.py:9-30                                k from the residues of every ODD RAW line (no blank skipping: raw line
//                                                      parity, str.strip), letters in first-seen order for the entropy sum
// The file is mapped, lines are found with memchr, sequence bytes are copied ONCE — into one of two pinned staging buffers —
// and each full buffer leaves for the device on a stream of its own while the parser fills the other; offsets and genome ids
// follow at the end.  Nothing is kept on the host but the per-gene arrays (8 + 4 bytes per gene).  pdl_scan_faa is the same
// parser without a device (host buffers or counting only): what the CPU tests compare with the Python reader.
#include "pdl_common.h"

#include <cmath>
#include <cstring>
#include <functional>
#include <string_view>
#include <thread>
#include <unordered_map>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

void pdl_set_create_error(const std::string &msg);                                                     // pdl_api.hip
int pdl_preprocess_common(pdl_ctx *c, uint32_t n, uint64_t n_res, int k, int only_complexity, pdl_cost *out_cost);

namespace {

struct MappedFile {
    int fd = -1;
    const uint8_t *p = nullptr;
    size_t n = 0;
    explicit MappedFile(const char *path) {
        fd = open(path, O_RDONLY);
        if (fd < 0) PDL_FAIL(PDL_ERR_ARGUMENT, "%s: %s", path, strerror(errno));      // (the Java host prints FileNotFoundException, Pangenes.java:26-31)
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); fd = -1; PDL_FAIL(PDL_ERR_ARGUMENT, "%s: not a regular file", path); }
        n = (size_t) st.st_size;
        if (n) {
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { close(fd); fd = -1; PDL_FAIL(PDL_ERR_ARGUMENT, "%s: mmap: %s", path, strerror(errno)); }
            (void) madvise(m, n, MADV_SEQUENTIAL);
            p = static_cast<const uint8_t *>(m);
        }
    }
    ~MappedFile() {
        if (p) munmap(const_cast<uint8_t *>(p), n);
        if (fd >= 0) close(fd);
    }
};

struct FaaTables {
    std::vector<uint64_t> off;
    std::vector<uint32_t> gen;
    std::vector<std::string> names;
    uint64_t R = 0;
    // calculate_k.py's view of the file
    uint64_t k_total = 0, cnt[4][256];
    uint8_t order[256];
    uint32_t letters = 0;
    uint8_t seen[256];
};

inline bool py_space(uint8_t ch) { return ch == ' ' || (ch >= 9 && ch <= 13) || (ch >= 0x1c && ch <= 0x1f); }   // str.strip() on ASCII text

// calculate_k.py:9-30 from the letter counts (first-seen order = the order the script's dict is summed in); 0 = undefined
// (the script divides by log(1) or log(0) there and dies)
int k_from_counts(const FaaTables &t) {
    if (t.letters < 2 || t.k_total == 0) return 0;
    const double a = (double) t.letters, size = (double) t.k_total;
    double h = 0.0;
    for (uint32_t i = 0; i < t.letters; i++) {
        const uint8_t ch = t.order[i];
        const double c = (double) (t.cnt[0][ch] + t.cnt[1][ch] + t.cnt[2][ch] + t.cnt[3][ch]);
        h += -(std::log(c / size) / std::log(a)) * (c / size);        // math.log(x, base) = log(x) / log(base)
    }
    return (int) std::floor((std::log(size) / std::log(a)) / h);
}

// ---- the parser, in parallel ------------------------------------------------------------------------------------------------
// What makes the reader sequential is four running values: the parity of the raw line number (calculate_k.py reads odd raw
// lines), the header / sequence alternation of the NON-BLANK lines, the position of a sequence among the sequences (and of its
// residues among the residues), and the genome ids in first-seen order.  All four are prefix sums (or a merge in file order) of
// per-chunk values, so the file is cut into chunks at line starts and parsed twice by a team of host threads:
//   pass 1   per chunk: raw lines, non-blank lines, and — for both parities the alternation may arrive with — sequences and
//            sequence bytes, and the last header line
//   (one thread: prefixes over the chunks; every chunk now knows the state it starts in)
//   pass 2   per chunk: the real parse — residues copied once to their final place, offsets, each sequence's genome as an index
//            into the chunk's own first-seen list, calculate_k.py's letter counts
//   (one thread: the chunks' genome lists and letter orders merged in file order = global first-seen order)
//   pass 3   per chunk: local genome indices -> global ids
// One thread and one chunk for small files.  Same results as the serial reader, byte for byte (tests/test_ingest.py).
template <class Fn>
void for_each_line(const uint8_t *p, size_t b, size_t e, Fn &&line) {      // [b, e): from a line start to just behind a terminator (or EOF)
    size_t pos = b;
    while (pos < e) {
        const uint8_t *nl = static_cast<const uint8_t *>(memchr(p + pos, '\n', e - pos));
        const size_t seg_end = nl ? (size_t) (nl - p) : e;               // [pos, seg_end) holds no \n
        size_t s = pos;
        bool consumed = false;
        for (;;) {                                                       // a lone \r ends a line too (readLine; universal newlines)
            const uint8_t *cr = s < seg_end ? static_cast<const uint8_t *>(memchr(p + s, '\r', seg_end - s)) : nullptr;
            if (!cr) break;
            const size_t le = (size_t) (cr - p);
            line(s, le);
            s = le + 1;
            if (s == seg_end && nl) { consumed = true; break; }          // \r\n is one terminator
        }
        if (!consumed && (s < seg_end || nl)) line(s, seg_end);
        pos = nl ? seg_end + 1 : e;
    }
}

struct FaaChunk {
    size_t b = 0, e = 0;
    // pass 1
    uint64_t raw = 0, nonblank = 0, seqs[2] = {0, 0}, bytes[2] = {0, 0};      // [parity of the non-blank line's index inside the chunk]
    size_t last_s[2] = {0, 0}, last_e[2] = {0, 0};                            // last non-blank line of either parity (trimmed)
    bool have_last[2] = {false, false};
    // state the chunk starts in (prefix)
    uint64_t raw0 = 0, seq0 = 0, res0 = 0;
    bool name_line0 = true, have_header0 = false;
    std::string_view genome0;
    // pass 2
    std::vector<std::string_view> names;            // genomes in the chunk's first-seen order
    std::vector<uint32_t> to_global;
    uint64_t k_total = 0, cnt[4][256];
    uint8_t order[256];
    uint32_t letters = 0;
    std::string error;                              // first failure inside the chunk
    int error_code = 0;
};

class SpinBarrier {
    std::atomic<uint32_t> arrived{0}, phase{0};
    const uint32_t n;
public:
    explicit SpinBarrier(uint32_t n_) : n(n_) {}
    void wait() {
        const uint32_t ph = phase.load(std::memory_order_acquire);
        if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == n) { arrived.store(0, std::memory_order_relaxed); phase.store(ph + 1, std::memory_order_release); }
        else while (phase.load(std::memory_order_acquire) == ph) __builtin_ia32_pause();
    }
};

inline std::string_view header_genome(const uint8_t *p, size_t s, size_t e, bool &ok) {      // genome = the text in front of the first tab; three fields needed
    const uint8_t *tab = static_cast<const uint8_t *>(memchr(p + s, '\t', e - s));
    const uint8_t *tab2 = tab ? static_cast<const uint8_t *>(memchr(tab + 1, '\t', (p + e) - (tab + 1))) : nullptr;
    ok = tab2 != nullptr;        // (trimmed: no leading or trailing tab, so fields = tabs + 1 and Java's split drops nothing)
    return ok ? std::string_view(reinterpret_cast<const char *>(p + s), (size_t) (tab - (p + s))) : std::string_view();
}

constexpr size_t ING_PAR_MIN = (size_t) 1 << 20;      // smaller files: one thread

// residues go to dst (may be null: counting only) at their final positions; cap = bytes dst holds
void faa_parse(const uint8_t *p, size_t n, const char *path, uint8_t *dst, size_t cap, FaaTables &t,
               const std::function<void(uint64_t, uint64_t)> &chunk_done = nullptr /* (first residue, residues) of a chunk whose residues are in place */,
               unsigned max_threads = 16) {
    memset(t.cnt, 0, sizeof(t.cnt));
    memset(t.seen, 0, sizeof(t.seen));
    unsigned T = 1;
    if (n >= ING_PAR_MIN) T = std::max(1u, std::min({max_threads, std::thread::hardware_concurrency(), (unsigned) (n / (ING_PAR_MIN / 4))}));
    std::vector<FaaChunk> ch(T);
    // chunk borders at line starts: the first line start at or behind i * n / T
    ch[0].b = 0;
    for (unsigned i = 1; i < T; i++) {
        size_t q = (size_t) ((unsigned __int128) n * i / T);
        q = std::max(q, ch[i - 1].b);
        size_t tpos = q ? q - 1 : 0;                                        // a terminator at q - 1 makes q itself a line start
        while (tpos < n && p[tpos] != '\n' && p[tpos] != '\r') tpos++;
        size_t b = tpos < n ? tpos + 1 : n;
        if (tpos < n && p[tpos] == '\r' && b < n && p[b] == '\n') b++;      // \r\n is one terminator
        if (q == 0) b = 0;
        ch[i].b = b;
    }
    for (unsigned i = 0; i < T; i++) ch[i].e = i + 1 < T ? ch[i + 1].b : n;
    SpinBarrier bar(T);
    std::unordered_map<std::string, uint32_t> genome_id;
    std::atomic<bool> failed{false};         // some chunk has met a malformed line (the others still finish their own: the FIRST failure in file order is reported)
    bool prefix_failed = false;             // (written by one thread between two barriers)

    auto pass1 = [&](FaaChunk &c) {
        for_each_line(p, c.b, c.e, [&](size_t s, size_t e) {
            c.raw++;
            while (s < e && p[s] <= ' ') s++;                                // String.trim
            while (e > s && p[e - 1] <= ' ') e--;
            if (s == e) return;                                              // PangeneIData.java:42-44
            const unsigned par = (unsigned) (c.nonblank & 1);
            c.nonblank++;
            c.seqs[par]++; c.bytes[par] += e - s;
            c.last_s[par] = s; c.last_e[par] = e; c.have_last[par] = true;
        });
    };
    auto prefix = [&]() {
        uint64_t raw = 0, seq = 0, res = 0;
        bool name_line = true, have_header = false;
        std::string_view genome;
        for (unsigned i = 0; i < T; i++) {
            FaaChunk &c = ch[i];
            c.raw0 = raw; c.seq0 = seq; c.res0 = res; c.name_line0 = name_line; c.have_header0 = have_header; c.genome0 = genome;
            // inside the chunk the non-blank line of index j is a header iff name_line0 == (j even)
            const unsigned seq_par = name_line ? 1u : 0u, hdr_par = seq_par ^ 1u;
            raw += c.raw; seq += c.seqs[seq_par]; res += c.bytes[seq_par];
            if (c.have_last[hdr_par]) {
                bool ok;
                const std::string_view g = header_genome(p, c.last_s[hdr_par], c.last_e[hdr_par], ok);
                if (ok) { genome = g; have_header = true; }                  // (a malformed header fails in pass 2, in its own chunk)
            }
            if (c.nonblank & 1) name_line = !name_line;
        }
        t.R = res;
        if (seq >= 0xfffffffeull) { ch[0].error = std::string(path) + ": more than 2^32 - 2 sequences"; ch[0].error_code = PDL_ERR_UNSUPPORTED; prefix_failed = true; return; }
        if (dst && res > cap) { ch[0].error = std::string(path) + ": more residues than the buffer holds"; ch[0].error_code = PDL_ERR_ARGUMENT; prefix_failed = true; return; }
        t.off.assign(seq + 1, 0);
        t.gen.assign(seq, 0);
    };
    auto pass2 = [&](FaaChunk &c) {
        memset(c.cnt, 0, sizeof(c.cnt));
        uint8_t seen[256];
        memset(seen, 0, sizeof(seen));
        uint64_t raw = c.raw0, seq = c.seq0, res = c.res0;
        bool name_line = c.name_line0, have_header = c.have_header0;
        std::string_view genome = c.genome0, last_name;
        uint32_t last_local = 0;
        bool have_last = false;
        std::unordered_map<std::string_view, uint32_t> local;
        for_each_line(p, c.b, c.e, [&](size_t s, size_t e) {
            if (c.error_code) return;
            const uint64_t i = raw++;
            if (i & 1) {                                                     // calculate_k.py:9-17
                size_t a = s, b = e;
                while (a < b && py_space(p[a])) a++;
                while (b > a && py_space(p[b - 1])) b--;
                c.k_total += b - a;
                size_t j = a;
                for (; j + 4 <= b; j += 4) {
                    const uint8_t c0 = p[j], c1 = p[j + 1], c2 = p[j + 2], c3 = p[j + 3];
                    c.cnt[0][c0]++; c.cnt[1][c1]++; c.cnt[2][c2]++; c.cnt[3][c3]++;
                    if (!(seen[c0] & seen[c1] & seen[c2] & seen[c3]))        // first-seen order (rare after the first lines)
                        for (int q = 0; q < 4; q++) if (!seen[p[j + q]]) { seen[p[j + q]] = 1; c.order[c.letters++] = p[j + q]; }
                }
                for (; j < b; j++) {
                    c.cnt[0][p[j]]++;
                    if (!seen[p[j]]) { seen[p[j]] = 1; c.order[c.letters++] = p[j]; }
                }
            }
            while (s < e && p[s] <= ' ') s++;                                // String.trim
            while (e > s && p[e - 1] <= ' ') e--;
            if (s == e) return;                                              // PangeneIData.java:42-44
            if (name_line) {
                bool ok;
                genome = header_genome(p, s, e, ok);
                if (!ok) {
                    char b[256];
                    snprintf(b, sizeof(b), "%s: line %llu: a header needs genome<TAB>gene<TAB>product (PangeneIData.java:49-51 indexes all three)", path, (unsigned long long) (i + 1));
                    c.error = b; c.error_code = PDL_ERR_ARGUMENT; failed = true;
                    return;
                }
                have_header = true;
            } else {
                if (!have_header) { c.error = std::string(path) + ": sequence without a header"; c.error_code = PDL_ERR_ARGUMENT; failed = true; return; }
                if (dst) memcpy(dst + res, p + s, e - s);
                res += e - s;
                t.off[seq + 1] = res;
                uint32_t id;
                if (have_last && genome == last_name) id = last_local;
                else {
                    auto it = local.find(genome);
                    if (it == local.end()) { it = local.emplace(genome, (uint32_t) c.names.size()).first; c.names.push_back(genome); }
                    id = it->second; last_local = id; last_name = genome; have_last = true;
                }
                t.gen[seq] = id;
                seq++;
            }
            name_line = !name_line;
        });
    };
    auto merge = [&]() {
        for (unsigned i = 0; i < T; i++) {
            FaaChunk &c = ch[i];
            c.to_global.resize(c.names.size());
            for (size_t q = 0; q < c.names.size(); q++) {
                auto it = genome_id.find(std::string(c.names[q]));
                if (it == genome_id.end()) { it = genome_id.emplace(std::string(c.names[q]), (uint32_t) genome_id.size()).first; t.names.emplace_back(c.names[q]); }
                c.to_global[q] = it->second;
            }
            t.k_total += c.k_total;
            for (int q = 0; q < 4; q++) for (int x = 0; x < 256; x++) t.cnt[q][x] += c.cnt[q][x];
            for (uint32_t q = 0; q < c.letters; q++) if (!t.seen[c.order[q]]) { t.seen[c.order[q]] = 1; t.order[t.letters++] = c.order[q]; }
        }
    };
    auto pass3 = [&](FaaChunk &c, unsigned idx) {
        const uint64_t s1 = idx + 1 < T ? ch[idx + 1].seq0 : (uint64_t) t.gen.size();
        for (uint64_t q = c.seq0; q < s1; q++) t.gen[q] = c.to_global[t.gen[q]];
    };
    auto worker = [&](unsigned idx) {
        pass1(ch[idx]);
        bar.wait();
        if (idx == 0) prefix();
        bar.wait();
        if (!prefix_failed) {
            pass2(ch[idx]);
            if (chunk_done && !ch[idx].error_code) {
                const uint64_t r1 = idx + 1 < T ? ch[idx + 1].res0 : t.R;
                if (r1 > ch[idx].res0) chunk_done(ch[idx].res0, r1 - ch[idx].res0);
            }
        }
        bar.wait();
        if (idx == 0 && !failed && !prefix_failed) merge();
        bar.wait();
        if (!failed && !prefix_failed) pass3(ch[idx], idx);
    };
    if (T == 1) worker(0);
    else {
        std::vector<std::thread> team;
        for (unsigned i = 1; i < T; i++) team.emplace_back(worker, i);
        worker(0);
        for (auto &th : team) th.join();
    }
    for (unsigned i = 0; i < T; i++)                                         // the failure the serial reader would have met first
        if (ch[i].error_code) throw pdl_error{ch[i].error_code, ch[i].error};
}

void fill_ingest(pdl_ingest *out, const FaaTables &t, size_t sequences, size_t genomes, size_t file_bytes, double ms) {
    memset(out, 0, sizeof(*out));
    out->file_bytes = file_bytes;
    out->residues = t.R;
    out->sequences = (uint32_t) sequences;
    out->genomes = (uint32_t) genomes;
    out->k_suggested = k_from_counts(t);
    out->parse_ms = ms;
}

double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

extern "C" {

int pdl_scan_faa(const char *path, pdl_ingest *out, uint8_t *residues, uint64_t cap_residues, uint64_t *offsets, uint32_t *genome_of,
                 uint32_t cap_sequences) {
    if (!path || !out) return PDL_ERR_ARGUMENT;
    try {
        const auto t0 = std::chrono::steady_clock::now();
        MappedFile f(path);
        FaaTables t;
        faa_parse(f.p, f.n, path, residues, residues ? cap_residues : 0, t);
        if ((offsets || genome_of) && t.gen.size() > cap_sequences)
            PDL_FAIL(PDL_ERR_ARGUMENT, "%s: %zu sequences, the buffers hold %u", path, t.gen.size(), cap_sequences);
        if (offsets) memcpy(offsets, t.off.data(), t.off.size() * 8);
        if (genome_of && !t.gen.empty()) memcpy(genome_of, t.gen.data(), t.gen.size() * 4);
        fill_ingest(out, t, t.gen.size(), t.names.size(), f.n, ms_since(t0));
        return PDL_OK;
    } catch (const pdl_error &e) { pdl_set_create_error(e.msg); return e.code;
    } catch (const std::bad_alloc &) { pdl_set_create_error("host allocation failed"); return PDL_ERR_DEVICE; }
}

int pdl_ingest_faa(pdl_ctx *c, const char *path, pdl_ingest *out) {
    if (!c || !path || !out) return PDL_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lk(c->mu);
    try {
        const auto t0 = std::chrono::steady_clock::now();
        PDL_HIP(hipSetDevice(c->device));
        c->ingested = false;
        MappedFile f(path);
        if (!c->ing_stream) PDL_HIP(hipStreamCreateWithFlags(&c->ing_stream, hipStreamNonBlocking));
        // one pinned staging buffer the size of the file (an upper bound of the residues), kept for the next ingest: the parser's
        // threads put every chunk's residues at their final offsets, and each chunk leaves for the device as soon as it is complete
        if (c->ing_pin_bytes < f.n + 16) {
            if (c->ing_pin[0]) { (void) hipHostFree(c->ing_pin[0]); c->ing_pin[0] = nullptr; c->ing_pin_bytes = 0; }
            const size_t want = f.n + f.n / 8 + 4096;
            PDL_HIP(hipHostMalloc((void **) &c->ing_pin[0], want, hipHostMallocDefault));
            c->ing_pin_bytes = want;
        }
        // a dictionary built from the previous ingest still reads these buffers: wait for whatever the context has queued
        PDL_HIP(hipStreamSynchronize(c->stream));
        c->preprocessed = false; c->scored = false; c->tasks_ready = false;
        c->ing_res.alloc(f.n + 16);
        FaaTables t;
        uint8_t *stage = c->ing_pin[0], *d_dst = c->ing_res.as<uint8_t>();
        std::mutex copy_mu;
        std::string copy_err;
        faa_parse(f.p, f.n, path, stage, f.n, t, [&](uint64_t r0, uint64_t bytes) {
            std::lock_guard<std::mutex> g(copy_mu);                          // (one queue: the copies of the chunks follow each other on the ingest stream)
            if (hipSetDevice(c->device) != hipSuccess || hipMemcpyAsync(d_dst + r0, stage + r0, bytes, hipMemcpyHostToDevice, c->ing_stream) != hipSuccess)
                copy_err = hipGetErrorString(hipGetLastError());
        });
        if (!copy_err.empty()) PDL_FAIL(PDL_ERR_DEVICE, "ingest: copy to the device failed: %s", copy_err.c_str());
        const size_t n = t.gen.size();
        c->ing_off.alloc((n + 1) * 8); c->ing_gen.alloc(n * 4 + 4);
        c->ing_h_off.swap(t.off); c->ing_h_gen.swap(t.gen); c->ing_genome_names.swap(t.names);
        PDL_HIP(hipMemcpyAsync(c->ing_off.p, c->ing_h_off.data(), (n + 1) * 8, hipMemcpyHostToDevice, c->ing_stream));
        if (n) PDL_HIP(hipMemcpyAsync(c->ing_gen.p, c->ing_h_gen.data(), n * 4, hipMemcpyHostToDevice, c->ing_stream));
        PDL_HIP(hipStreamSynchronize(c->ing_stream));
        fill_ingest(out, t, n, c->ing_genome_names.size(), f.n, ms_since(t0));
        out->offsets = c->ing_h_off.data(); out->genome_of = c->ing_h_gen.data();
        out->d_residues = c->ing_res.as<uint8_t>(); out->d_offsets = c->ing_off.as<uint64_t>(); out->d_genome_of = c->ing_gen.as<uint32_t>();
        c->ing_R = out->residues;
        c->ingested = true;
        return PDL_OK;
    } catch (const pdl_error &e) { c->err = e.msg; return e.code;
    } catch (const std::bad_alloc &) { c->err = "host allocation failed"; return PDL_ERR_DEVICE; }
}

const char *pdl_ingest_genome_name(const pdl_ctx *c, uint32_t genome) {
    if (!c || !c->ingested || genome >= c->ing_genome_names.size()) return nullptr;
    return c->ing_genome_names[genome].c_str();
}

int pdl_preprocess_ingested(pdl_ctx *c, int k, int only_complexity, pdl_cost *out_cost) {
    if (!c) return PDL_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->ingested) { c->err = "pdl_preprocess_ingested before pdl_ingest_faa"; return PDL_ERR_STATE; }
    const uint32_t n = (uint32_t) c->ing_h_gen.size();
    c->d_res = c->ing_res.as<uint8_t>(); c->d_off = c->ing_off.as<uint64_t>(); c->d_gen = c->ing_gen.as<uint32_t>();
    c->h_genome_of = c->ing_h_gen;                 // the genome layout is built from the host copy: nothing comes back from the device
    c->layout_deferred = false;
    return pdl_preprocess_common(c, n, c->ing_R, k, only_complexity, out_cost);
}

}  // extern "C"
