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
#include <string_view>
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

template <class Sink>
void faa_parse(const uint8_t *p, size_t n, const char *path, Sink &&put, FaaTables &t) {
    memset(t.cnt, 0, sizeof(t.cnt));
    memset(t.seen, 0, sizeof(t.seen));
    t.off.assign(1, 0);
    std::unordered_map<std::string, uint32_t> genome_id;
    std::string_view genome;              // of the header in force (points into the mapping)
    bool have_header = false, name_line = true;
    uint64_t raw = 0;
    uint32_t last_id = 0;
    std::string_view last_name;
    bool have_last = false;

    auto line = [&](size_t s, size_t e) {
        const uint64_t i = raw++;
        if (i & 1) {                                                     // calculate_k.py:9-17
            size_t a = s, b = e;
            while (a < b && py_space(p[a])) a++;
            while (b > a && py_space(p[b - 1])) b--;
            t.k_total += b - a;
            size_t j = a;
            for (; j + 4 <= b; j += 4) {
                const uint8_t c0 = p[j], c1 = p[j + 1], c2 = p[j + 2], c3 = p[j + 3];
                t.cnt[0][c0]++; t.cnt[1][c1]++; t.cnt[2][c2]++; t.cnt[3][c3]++;
                if (!(t.seen[c0] & t.seen[c1] & t.seen[c2] & t.seen[c3]))        // first-seen order (rare after the first lines)
                    for (int q = 0; q < 4; q++) if (!t.seen[p[j + q]]) { t.seen[p[j + q]] = 1; t.order[t.letters++] = p[j + q]; }
            }
            for (; j < b; j++) {
                t.cnt[0][p[j]]++;
                if (!t.seen[p[j]]) { t.seen[p[j]] = 1; t.order[t.letters++] = p[j]; }
            }
        }
        while (s < e && p[s] <= ' ') s++;                                // String.trim
        while (e > s && p[e - 1] <= ' ') e--;
        if (s == e) return;                                              // PangeneIData.java:42-44
        if (name_line) {
            const uint8_t *tab = static_cast<const uint8_t *>(memchr(p + s, '\t', e - s));
            const uint8_t *tab2 = tab ? static_cast<const uint8_t *>(memchr(tab + 1, '\t', (p + e) - (tab + 1))) : nullptr;
            if (!tab2)         // (trimmed: no leading or trailing tab, so fields = tabs + 1 and Java's split drops nothing)
                PDL_FAIL(PDL_ERR_ARGUMENT, "%s: line %llu: a header needs genome<TAB>gene<TAB>product (PangeneIData.java:49-51 indexes all three)",
                         path, (unsigned long long) (i + 1));
            genome = std::string_view(reinterpret_cast<const char *>(p + s), (size_t) (tab - (p + s)));
            have_header = true;
        } else {
            if (!have_header) PDL_FAIL(PDL_ERR_ARGUMENT, "%s: sequence without a header", path);
            if (t.gen.size() >= 0xfffffffeull) PDL_FAIL(PDL_ERR_UNSUPPORTED, "%s: more than 2^32 - 2 sequences", path);
            put(p + s, e - s);
            t.R += e - s;
            t.off.push_back(t.R);
            uint32_t id;
            if (have_last && genome == last_name) id = last_id;
            else {
                auto it = genome_id.find(std::string(genome));
                if (it == genome_id.end()) { it = genome_id.emplace(std::string(genome), (uint32_t) genome_id.size()).first; t.names.emplace_back(genome); }
                id = it->second; last_id = id; last_name = genome; have_last = true;
            }
            t.gen.push_back(id);
        }
        name_line = !name_line;
    };

    size_t pos = 0;
    while (pos < n) {
        const uint8_t *nl = static_cast<const uint8_t *>(memchr(p + pos, '\n', n - pos));
        const size_t seg_end = nl ? (size_t) (nl - p) : n;               // [pos, seg_end) holds no \n
        size_t s = pos;
        bool consumed = false;
        for (;;) {                                                       // a lone \r ends a line too (readLine; universal newlines)
            const uint8_t *cr = s < seg_end ? static_cast<const uint8_t *>(memchr(p + s, '\r', seg_end - s)) : nullptr;
            if (!cr) break;
            const size_t e = (size_t) (cr - p);
            line(s, e);
            s = e + 1;
            if (s == seg_end && nl) { consumed = true; break; }          // \r\n is one terminator
        }
        if (!consumed && (s < seg_end || nl)) line(s, seg_end);
        pos = nl ? seg_end + 1 : n;
    }
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

constexpr size_t ING_CHUNK = (size_t) 8 << 20;

// residues -> pinned staging -> device, two buffers in turn
struct DeviceSink {
    pdl_ctx *c;
    uint8_t *d_dst;
    size_t cap;
    size_t fill = 0, sent = 0;
    int cur = 0;
    bool busy[2] = {false, false};
    void flush() {
        if (!fill) return;
        if (sent + fill > cap) PDL_FAIL(PDL_ERR_DEVICE, "ingest: more residues than file bytes");     // (cannot happen: residues are a subset of the file)
        PDL_HIP(hipMemcpyAsync(d_dst + sent, c->ing_pin[cur], fill, hipMemcpyHostToDevice, c->ing_stream));
        PDL_HIP(hipEventRecord(c->ing_ev[cur], c->ing_stream));
        busy[cur] = true;
        sent += fill; fill = 0; cur ^= 1;
        if (busy[cur]) { PDL_HIP(hipEventSynchronize(c->ing_ev[cur])); busy[cur] = false; }
    }
    void operator()(const uint8_t *src, size_t len) {
        while (len) {
            const size_t take = std::min(len, ING_CHUNK - fill);
            memcpy(c->ing_pin[cur] + fill, src, take);
            fill += take; src += take; len -= take;
            if (fill == ING_CHUNK) flush();
        }
    }
};

}  // namespace

extern "C" {

int pdl_scan_faa(const char *path, pdl_ingest *out, uint8_t *residues, uint64_t cap_residues, uint64_t *offsets, uint32_t *genome_of,
                 uint32_t cap_sequences) {
    if (!path || !out) return PDL_ERR_ARGUMENT;
    try {
        const auto t0 = std::chrono::steady_clock::now();
        MappedFile f(path);
        FaaTables t;
        uint64_t at = 0;
        faa_parse(f.p, f.n, path, [&](const uint8_t *src, size_t len) {
            if (residues) {
                if (at + len > cap_residues) PDL_FAIL(PDL_ERR_ARGUMENT, "%s: more than the %llu residues the buffer holds", path, (unsigned long long) cap_residues);
                memcpy(residues + at, src, len);
            }
            at += len;
        }, t);
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
        for (int i = 0; i < 2; i++) {       // (each piece checked by itself: a call that failed half-way leaves the rest for the next one)
            if (!c->ing_pin[i]) PDL_HIP(hipHostMalloc((void **) &c->ing_pin[i], ING_CHUNK, hipHostMallocDefault));
            if (!c->ing_ev[i]) PDL_HIP(hipEventCreateWithFlags(&c->ing_ev[i], hipEventDisableTiming));
        }
        // a dictionary built from the previous ingest still reads these buffers: wait for whatever the context has queued
        PDL_HIP(hipStreamSynchronize(c->stream));
        c->preprocessed = false; c->scored = false; c->tasks_ready = false;
        c->ing_res.alloc(f.n + 16);
        FaaTables t;
        DeviceSink sink{c, c->ing_res.as<uint8_t>(), f.n};
        faa_parse(f.p, f.n, path, sink, t);
        sink.flush();
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
