// pdl_api.hip — the C ABI of include/pandelos_amd.h over the two device stages
// (pdl_dict.hip: preprocessSequences; pdl_join.hip: computeScores).
#include "pdl_common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <limits>

static thread_local std::string g_create_error = "";

#define PDL_GUARD_BEGIN try {
#define PDL_GUARD_END(ctx_)                                                       \
    } catch (const pdl_error &e) {                                                \
        if (ctx_) (ctx_)->err = e.msg; else g_create_error = e.msg;               \
        return e.code;                                                            \
    } catch (const std::bad_alloc &) {                                            \
        if (ctx_) (ctx_)->err = "host allocation failed";                         \
        return PDL_ERR_DEVICE;                                                    \
    }

template <class T> static T *xalloc(size_t n) {
    T *p = static_cast<T *>(malloc((n ? n : 1) * sizeof(T)));
    if (!p) throw std::bad_alloc();
    return p;
}

extern "C" {

const char *pdl_version(void) { return "pandelos_amd 0.1 (HIP, gfx950)"; }

// The host side of the small-read protocol on plain memory (no device, no context): see PinRead in pdl_common.h.
int pdl_pin_arrived(const uint32_t *pin, const uint32_t *dst_word, const uint32_t *words, uint32_t n, uint32_t flag_word, uint32_t epoch) {
    if (!pin || !dst_word || !words) return -1;
    return pin_arrived(pin, dst_word, words, n, flag_word, epoch) ? 1 : 0;
}
uint32_t pdl_pin_checksum(const uint32_t *payload, const uint32_t *dst_word, const uint32_t *words, uint32_t n) {
    uint32_t sum = 0, at = 0;        // payload: the segments' words one after the other, as the kernel reads them from the device
    for (uint32_t s = 0; s < n; s++) for (uint32_t i = 0; i < words[s]; i++) sum += payload[at++] * pin_weight(dst_word[s] + i);
    return sum;
}

pdl_ctx *pdl_create(const pdl_config *cfg) {
    pdl_ctx *c = nullptr;
    try {
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev == 0)
            PDL_FAIL(PDL_ERR_DEVICE, "no HIP device available (%s): pandelos_amd has no CPU path", hipGetErrorString(e));
        c = new pdl_ctx();
        int dev = cfg ? cfg->device : -1;
        if (dev < 0) PDL_HIP(hipGetDevice(&dev));
        if (dev >= ndev) PDL_FAIL(PDL_ERR_ARGUMENT, "device %d out of range (%d devices)", dev, ndev);
        c->device = dev;
        PDL_HIP(hipSetDevice(dev));
        c->flags = cfg ? cfg->flags : 0;
        if (cfg && cfg->stream) { c->stream = (hipStream_t) cfg->stream; c->own_stream = false; }
        else { PDL_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
        // (coherent = fine-grained: what a kernel stores there is visible to the host while the kernel still runs — PinRead's flag)
        if (hipHostMalloc((void **) &c->pin, 1 << 20, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
            c->pin_bytes = 1 << 20; memset(c->pin + c->pin_bytes - PDL_PIN_FLAG_BYTES, 0, PDL_PIN_FLAG_BYTES);
            void *dp = nullptr;      // kernels store into the buffer through the device's alias of it, not through the host pointer
            if (hipHostGetDevicePointer(&dp, c->pin, 0) == hipSuccess && dp) c->pin_dev = static_cast<uint8_t *>(dp);
            else (void) hipGetLastError();
        } else { c->pin = nullptr; (void) hipGetLastError(); }
        return c;
    } catch (const pdl_error &e) {
        g_create_error = e.msg;
        delete c;
        return nullptr;
    }
}

void pdl_destroy(pdl_ctx *c) {
    if (!c) return;
    (void) hipSetDevice(c->device);
    (void) hipStreamSynchronize(c->stream);
    for (auto &e : c->ev) { if (e.a) (void) hipEventDestroy(e.a); if (e.b) (void) hipEventDestroy(e.b); }
    if (c->own_stream && c->stream) (void) hipStreamDestroy(c->stream);
    if (c->pin) (void) hipHostFree(c->pin);
    if (c->mirror) (void) hipHostFree(c->mirror);
    if (c->edge_mirror) (void) hipHostFree(c->edge_mirror);
    if (c->task_pin) (void) hipHostFree(c->task_pin);
    if (c->ev_tasks) (void) hipEventDestroy(c->ev_tasks);
    if (c->ev_entry) (void) hipEventDestroy(c->ev_entry);
    if (c->copy_stream) (void) hipStreamDestroy(c->copy_stream);
    if (c->gen_pin) (void) hipHostFree(c->gen_pin);
    if (c->ev_gen) (void) hipEventDestroy(c->ev_gen);
    for (int i = 0; i < 2; i++) { if (c->ing_pin[i]) (void) hipHostFree(c->ing_pin[i]); if (c->ing_ev[i]) (void) hipEventDestroy(c->ing_ev[i]); }
    if (c->ing_stream) (void) hipStreamDestroy(c->ing_stream);
    delete c;
}

const char *pdl_last_error(const pdl_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

static void fill_cost(const pdl_ctx *c, pdl_cost *out) {
    if (!out) return;
    memset(out, 0, sizeof(*out));
    out->residues = c->R; out->kmer_occurrences = c->M; out->dictionary_records = c->U;
    out->shared_records = c->Ushared; out->groups = c->NG; out->total_cost = c->P;
    out->linear_ratio = c->sum_kseq ? (float) c->P / (float) c->sum_kseq : 0.f;   // library.cpp:350
    out->sequences = c->N; out->genomes = c->G; out->rank_base = c->rp.base; out->rank_bits = c->rp.rank_bits;
    out->hash_fallback = (int32_t) c->rp.hash_fallback; out->kvalue = (int32_t) c->rp.k;
}

// genome layout on the host: genome_sequences (library.cpp:244,268) as CSR
static void build_genome_layout(pdl_ctx *c) {
    const uint32_t N = c->N;
    uint32_t G = 0;
    for (uint32_t i = 0; i < N; i++) G = std::max(G, c->h_genome_of[i] + 1);      // library.cpp:242
    if ((uint64_t) G > (uint64_t) N) PDL_FAIL(PDL_ERR_ARGUMENT, "genome ids are not dense: max id %u with %u genes", G - 1, N);
    c->G = G;
    c->h_genome_row_off.assign((size_t) G + 1, 0);
    for (uint32_t i = 0; i < N; i++) c->h_genome_row_off[c->h_genome_of[i] + 1]++;
    for (uint32_t g = 0; g < G; g++) c->h_genome_row_off[g + 1] += c->h_genome_row_off[g];
    c->h_genome_rows.resize(N);
    std::vector<uint32_t> cur(c->h_genome_row_off.begin(), c->h_genome_row_off.end() - 1);
    for (uint32_t i = 0; i < N; i++) c->h_genome_rows[cur[c->h_genome_of[i]]++] = i;
}

// genome layout + what depends on the genome count (shard check)
static void layout_and_shard(pdl_ctx *c) {
    build_genome_layout(c);
    c->dict_shard.clear();
    if (c->shard_set) {
        for (uint32_t g : c->shard)
            if (g >= c->G) PDL_FAIL(PDL_ERR_ARGUMENT, "genome shard: id %u out of range (%u genomes)", g, c->G);
        c->dict_shard = c->shard;            // posting-range lists are built for these genomes' genes only
    }
}

}  // extern "C"   (C++ helpers of the other translation units)
void pdl_input_arrived(pdl_ctx *c) {
    PDL_HIP(hipEventSynchronize(c->ev_gen));
    const uint64_t *ends = reinterpret_cast<const uint64_t *>(c->gen_pin + ((c->N + 1) & ~1u));
    if (ends[0] != 0 || ends[1] != c->R)
        PDL_FAIL(PDL_ERR_ARGUMENT, "offsets[0] = %llu, offsets[n] = %llu do not span the %llu residues", (unsigned long long) ends[0],
                 (unsigned long long) ends[1], (unsigned long long) c->R);
}
void pdl_finish_layout(pdl_ctx *c) {
    c->h_genome_of.assign(c->gen_pin, c->gen_pin + c->N);
    c->layout_deferred = false;
    layout_and_shard(c);
}
void pdl_set_create_error(const std::string &msg) { g_create_error = msg; }

int pdl_preprocess_common(pdl_ctx *c, uint32_t n, uint64_t n_res, int k, int only_complexity, pdl_cost *out_cost) {
    PDL_GUARD_BEGIN
    PDL_HIP(hipSetDevice(c->device));
    c->preprocessed = false; c->scored = false; c->tasks_ready = false; c->reshard_pending = false;      // the genome shard, if one was set, stays in force
    c->N = n; c->R = n_res;
    c->U = c->Ushared = c->NG = c->P = c->M = 0;
    if (k <= 0) PDL_FAIL(PDL_ERR_KVALUE, "K value must be greater than 0.");
    if (n == 0) PDL_FAIL(PDL_ERR_EMPTY, "empty dataset");
    if (c->dist) { c->shard.clear(); c->shard_set = false; c->dist = false; c->dist_stage = 0; }    // (a multi-GPU build's own deal does not carry over)
    if (!c->layout_deferred) layout_and_shard(c);        // (device input: done behind the first kernels, pdl_finish_layout)
    pdl_run_preprocess(c, k, only_complexity != 0);
    c->preprocessed = true;
    fill_cost(c, out_cost);
    return PDL_OK;
    PDL_GUARD_END(c)
}
static int preprocess_common(pdl_ctx *c, uint32_t n, uint64_t n_res, int k, int only_complexity, pdl_cost *out_cost) {
    return pdl_preprocess_common(c, n, n_res, k, only_complexity, out_cost);
}

extern "C" {

// device-resident input: genome ids come to the host (task layout), and the two ends of the offsets are checked against
// n_res (a wrong n_res would make the ranking kernels read past the residues; ascending order is checked by K-len)
static void adopt_device_input(pdl_ctx *c, const uint8_t *d_residues, const uint64_t *d_offsets, const uint32_t *d_genome_of,
                               uint32_t n, uint64_t n_res) {
    c->d_res = d_residues; c->d_off = d_offsets; c->d_gen = d_genome_of;
    c->h_genome_of.resize(n);
    uint64_t ends[2] = {0, n_res};
    if (n) {
        PDL_HIP(hipMemcpyAsync(c->h_genome_of.data(), d_genome_of, (size_t) n * 4, hipMemcpyDeviceToHost, c->stream));
        PDL_HIP(hipMemcpyAsync(&ends[0], d_offsets, 8, hipMemcpyDeviceToHost, c->stream));
        PDL_HIP(hipMemcpyAsync(&ends[1], d_offsets + n, 8, hipMemcpyDeviceToHost, c->stream));
    }
    PDL_HIP(hipStreamSynchronize(c->stream));
    if (ends[0] != 0 || ends[1] != n_res)
        PDL_FAIL(PDL_ERR_ARGUMENT, "offsets[0] = %llu, offsets[n] = %llu do not span the %llu residues", (unsigned long long) ends[0],
                 (unsigned long long) ends[1], (unsigned long long) n_res);
}

int pdl_preprocess(pdl_ctx *c, const uint8_t *residues, const uint64_t *offsets, const uint32_t *genome_of,
                   uint32_t n, int k, int only_complexity, pdl_cost *out_cost) {
    if (!c) return PDL_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lk(c->mu);
    PDL_GUARD_BEGIN
    if (!offsets || !genome_of || (!residues && n && offsets[n] > 0)) PDL_FAIL(PDL_ERR_ARGUMENT, "null input pointer");
    PDL_HIP(hipSetDevice(c->device));
    const uint64_t n_res = n ? offsets[n] - offsets[0] : 0;
    if (n && offsets[0] != 0) PDL_FAIL(PDL_ERR_ARGUMENT, "offsets[0] must be 0");
    c->in_res.alloc(n_res + 16); c->in_off.alloc(((size_t) n + 1) * 8); c->in_gen.alloc((size_t) n * 4 + 4);
    if (n_res) PDL_HIP(hipMemcpyAsync(c->in_res.p, residues, n_res, hipMemcpyHostToDevice, c->stream));
    PDL_HIP(hipMemcpyAsync(c->in_off.p, offsets, ((size_t) n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    if (n) PDL_HIP(hipMemcpyAsync(c->in_gen.p, genome_of, (size_t) n * 4, hipMemcpyHostToDevice, c->stream));
    c->d_res = c->in_res.as<uint8_t>(); c->d_off = c->in_off.as<uint64_t>(); c->d_gen = c->in_gen.as<uint32_t>();
    c->h_genome_of.assign(genome_of, genome_of + n);
    PDL_HIP(hipStreamSynchronize(c->stream));
    PDL_GUARD_END(c)
    return preprocess_common(c, n, n ? offsets[n] : 0, k, only_complexity, out_cost);
}

int pdl_preprocess_device(pdl_ctx *c, const uint8_t *d_residues, const uint64_t *d_offsets, const uint32_t *d_genome_of,
                          uint32_t n, uint64_t n_res, int k, int only_complexity, pdl_cost *out_cost) {
    if (!c) return PDL_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lk(c->mu);
    PDL_GUARD_BEGIN
    if (!d_offsets || !d_genome_of || (!d_residues && n_res)) PDL_FAIL(PDL_ERR_ARGUMENT, "null input pointer");
    if (((uintptr_t) d_residues & 15) != 0) PDL_FAIL(PDL_ERR_ARGUMENT, "d_residues must be 16-byte aligned");
    PDL_HIP(hipSetDevice(c->device));
    if (n == 0 || k <= 0) {
        adopt_device_input(c, d_residues, d_offsets, d_genome_of, n, n_res);
    } else {
        // genome ids and the two ends of the offsets start their way to the host (pinned: a true asynchronous copy); the
        // first kernels are queued behind them and the host builds the genome layout while those run
        c->d_res = d_residues; c->d_off = d_offsets; c->d_gen = d_genome_of;
        const size_t words = (((size_t) n + 1) & ~(size_t) 1) + 4;
        if (c->gen_pin_words < words) {
            if (c->gen_pin) (void) hipHostFree(c->gen_pin);
            c->gen_pin = nullptr; c->gen_pin_words = 0;
            PDL_HIP(hipHostMalloc((void **) &c->gen_pin, (words + words / 4) * sizeof(uint32_t), hipHostMallocDefault));
            c->gen_pin_words = words + words / 4;
        }
        if (!c->ev_gen) PDL_HIP(hipEventCreateWithFlags(&c->ev_gen, hipEventDisableTiming));
        uint64_t *ends = reinterpret_cast<uint64_t *>(c->gen_pin + (((size_t) n + 1) & ~(size_t) 1));
        // (on the side stream, behind whatever the caller's stream holds so far: the build's first kernels do not queue up
        //  behind three copies)
        if (!c->copy_stream) {
            PDL_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
            PDL_HIP(hipEventCreateWithFlags(&c->ev_tasks, hipEventDisableTiming));
        }
        if (!c->ev_entry) PDL_HIP(hipEventCreateWithFlags(&c->ev_entry, hipEventDisableTiming));
        PDL_HIP(hipEventRecord(c->ev_entry, c->stream));
        PDL_HIP(hipStreamWaitEvent(c->copy_stream, c->ev_entry, 0));
        PDL_HIP(hipMemcpyAsync(ends, d_offsets, 8, hipMemcpyDeviceToHost, c->copy_stream));
        PDL_HIP(hipMemcpyAsync(ends + 1, d_offsets + n, 8, hipMemcpyDeviceToHost, c->copy_stream));
        PDL_HIP(hipMemcpyAsync(c->gen_pin, d_genome_of, (size_t) n * 4, hipMemcpyDeviceToHost, c->copy_stream));
        PDL_HIP(hipEventRecord(c->ev_gen, c->copy_stream));
        c->layout_deferred = true;
    }
    PDL_GUARD_END(c)
    const int rc = preprocess_common(c, n, n_res, k, only_complexity, out_cost);
    c->layout_deferred = false;
    return rc;
}

int pdl_genome_cost(const pdl_ctx *cc, uint32_t genome, uint64_t *out) {
    pdl_ctx *c = const_cast<pdl_ctx *>(cc);
    if (!c || !out) return PDL_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->preprocessed) return PDL_ERR_STATE;
    if (genome >= c->G) return PDL_ERR_ARGUMENT;
    PDL_GUARD_BEGIN
    if (!c->costs_ready) {                       // (packed ranges: the per-gene / per-genome lookups are made on first request)
        PDL_HIP(hipSetDevice(c->device));
        pdl_ensure_costs(c);
    }
    *out = c->h_genome_cost[genome];
    return PDL_OK;
    PDL_GUARD_END(c)
}

int pdl_sequence_costs(const pdl_ctx *cc, uint64_t *out_cost, uint32_t *out_kseq) {
    pdl_ctx *c = const_cast<pdl_ctx *>(cc);
    if (!c || !out_cost) return PDL_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->preprocessed) return PDL_ERR_STATE;
    PDL_GUARD_BEGIN
    if (c->dist && c->dist_sender) PDL_FAIL(PDL_ERR_STATE, "per-gene costs are not kept by a multi-GPU build whose range lists came from the senders (pdl_genome_cost works)");
    PDL_HIP(hipSetDevice(c->device));
    pdl_ensure_costs(c);
    PDL_HIP(hipMemcpyAsync(out_cost, c->cost.p, (size_t) c->N * 8, hipMemcpyDeviceToHost, c->stream));
    if (out_kseq) PDL_HIP(hipMemcpyAsync(out_kseq, c->kseq_len.p, (size_t) c->N * 4, hipMemcpyDeviceToHost, c->stream));
    PDL_HIP(hipStreamSynchronize(c->stream));
    return PDL_OK;
    PDL_GUARD_END(c)
}

int pdl_set_genome_shard(pdl_ctx *c, const uint32_t *genomes, uint32_t count) {
    if (!c || (!genomes && count)) return PDL_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->dist && c->dist_stage) { c->err = "genome shard: a multi-GPU build deals the genomes itself (pdl_dist_genome_owner)"; return PDL_ERR_STATE; }
    std::vector<uint32_t> s(genomes, genomes + count);
    std::sort(s.begin(), s.end());
    for (size_t i = 0; i < s.size(); i++) {
        if ((c->preprocessed && s[i] >= c->G) || (i && s[i] == s[i - 1])) { c->err = "genome shard: id out of range or repeated"; return PDL_ERR_ARGUMENT; }
    }
    if (c->preprocessed && c->upper_only && count != 0 && s.size() != c->G) {
        // without a shard the ranges hold only the genes above each row and every cell is produced once for both
        // rows; scoring a subset needs the full ranges of a shard-aware build
        c->err = "genome shard: set it before pdl_preprocess (the dictionary on the device was built for all genomes)";
        return PDL_ERR_STATE;
    }
    bool rebuild = false;
    if (c->preprocessed && !c->dict_shard.empty()) {
        // the dictionary on the device only has range lists for the genes of dict_shard: another shard gets its own before the next
        // scoring pass (the postings stay: two passes over them, pdl_run_reshard) — how a large set is scored a batch of genomes at a time
        if (count == 0) { c->err = "genome shard: the dictionary was built for a shard; run pdl_preprocess again to widen it to all genomes"; return PDL_ERR_STATE; }
        for (uint32_t g : s)
            if (!std::binary_search(c->dict_shard.begin(), c->dict_shard.end(), g)) rebuild = true;
        if (rebuild && (c->only_complexity || !c->head_bits.p)) { c->err = "genome shard: not a subset of the shard the dictionary was built for; run pdl_preprocess again"; return PDL_ERR_STATE; }
    }
    c->reshard_pending = c->reshard_pending || rebuild;
    c->shard = std::move(s);
    c->shard_set = count != 0;
    c->scored = false;
    c->tasks_ready = false;
    return PDL_OK;
}

static int score_all_locked(pdl_ctx *c) {
    PDL_GUARD_BEGIN
    if (!c->preprocessed) PDL_FAIL(PDL_ERR_STATE, "pdl_score_all before pdl_preprocess");
    if (c->only_complexity) PDL_FAIL(PDL_ERR_STATE, "the dictionary was built in complexity-only mode (no posting ranges)");
    if (c->scored) return PDL_OK;
    if (c->dist) PDL_FAIL(PDL_ERR_STATE, "multi-GPU context: score with pdl_dist_score_begin / pdl_dist_score_finish");
    PDL_HIP(hipSetDevice(c->device));
    c->mirror_valid = false; c->edges_valid = false;
    if (c->reshard_pending) { pdl_run_reshard(c); c->reshard_pending = false; }
    pdl_run_score_all(c);
    return PDL_OK;
    PDL_GUARD_END(c)
}

int pdl_score_all(pdl_ctx *c) {
    if (!c) return PDL_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lk(c->mu);
    return score_all_locked(c);
}

int pdl_scores_counts(pdl_ctx *c, uint32_t *out) {
    if (!c || !out) return PDL_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = score_all_locked(c);
    if (rc != PDL_OK) return rc;
    for (uint32_t g = 0; g < c->G; g++) {
        int32_t lg = c->h_local_genome[g];
        out[g] = lg < 0 ? 0u : (uint32_t) (c->h_cell_off[lg + 1] - c->h_cell_off[lg]);
    }
    return PDL_OK;
}

void pdl_free_scores(pdl_scores *s) {
    if (!s) return;
    free(s->scores); free(s->percs); free(s->tr_percs); free(s->row); free(s->column);
    free(s->first_seq_genome); free(s->second_seq_genome); free(s->max_genome_score);
    free(s->max_genome_score_col); free(s->scoresMaxMappings);
    memset(s, 0, sizeof(*s));
}

// Device -> host copy for one calling thread, without the context lock: a stream and two pinned bounce buffers per
// thread (results beyond the host mirror's limit: the Java pool calls computeScores from ThreadsNum threads at once,
// Pangenes.java:54-66; the device arrays are immutable between scoring and the next preprocess).
namespace {
struct ThreadCopier {
    static constexpr size_t CHUNK = (size_t) 4 << 20;
    int device = -1;
    hipStream_t stream = nullptr;
    uint8_t *buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    void open(int dev) {
        if (device == dev && stream) return;
        close();
        PDL_HIP(hipSetDevice(dev));
        PDL_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) {
            PDL_HIP(hipHostMalloc((void **) &buf[i], CHUNK, hipHostMallocDefault));
            PDL_HIP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
        device = dev;
    }
    void close() {
        for (int i = 0; i < 2; i++) { if (buf[i]) (void) hipHostFree(buf[i]); if (ev[i]) (void) hipEventDestroy(ev[i]); buf[i] = nullptr; ev[i] = nullptr; }
        if (stream) (void) hipStreamDestroy(stream);
        stream = nullptr; device = -1;
    }
    // dst (pageable) = src (device): chunk i+1 is in flight while chunk i is copied out of its bounce buffer
    void fetch(void *dst, const void *src, size_t bytes) {
        uint8_t *d = static_cast<uint8_t *>(dst);
        const uint8_t *sp = static_cast<const uint8_t *>(src);
        size_t issued = 0, drained = 0;
        int slot = 0;
        size_t len[2] = {0, 0};
        while (drained < bytes) {
            if (issued < bytes && (issued - drained) < 2 * CHUNK) {
                const size_t n = std::min(CHUNK, bytes - issued);
                PDL_HIP(hipMemcpyAsync(buf[slot], sp + issued, n, hipMemcpyDeviceToHost, stream));
                PDL_HIP(hipEventRecord(ev[slot], stream));
                len[slot] = n; issued += n; slot ^= 1;
                if (issued < bytes && issued - drained < 2 * CHUNK) continue;
            }
            const int dslot = (int) ((drained / CHUNK) & 1);
            PDL_HIP(hipEventSynchronize(ev[dslot]));
            memcpy(d + drained, buf[dslot], len[dslot]);
            drained += len[dslot];
        }
    }
    ~ThreadCopier() { close(); }
};
thread_local ThreadCopier t_copier;
}  // namespace

int pdl_compute_scores(pdl_ctx *c, uint32_t genome, pdl_scores *out) {
    if (!c || !out) return PDL_ERR_ARGUMENT;
    memset(out, 0, sizeof(*out));
    try {
    bool use_mirror = false;
    size_t b_cells = 0, b_ms = 0;
    {
        // The lock covers the scoring pass (first call) and the one-time fill of the host mirror; slicing a genome's block
        // out of either the mirror or the device arrays runs without it, so the pool's threads do not serialise.
        std::lock_guard<std::mutex> lk(c->mu);
        int rc = c->scored ? PDL_OK : score_all_locked(c);
        if (rc != PDL_OK) return rc;
        if (genome >= c->G) PDL_FAIL(PDL_ERR_ARGUMENT, "genome %u out of range (%u genomes)", genome, c->G);
        if (c->h_local_genome[genome] < 0) PDL_FAIL(PDL_ERR_ARGUMENT, "genome %u is not in this context's shard", genome);
        // Results up to PDL_MIRROR_LIMIT come to the host once, into pinned memory, with one copy per array; the per-genome
        // calls (the reference's thread pool makes G of them, Pangenes.java:54-66) then only slice that mirror.
        const uint64_t Zall = c->h_cell_off.back();
        const size_t S = c->shard.size();
        b_cells = (size_t) Zall * 4; b_ms = (size_t) c->n_task_rows * c->G * 4;
        const size_t b_cm = S * (size_t) c->N * 4;
        const size_t need = 5 * b_cells + b_ms + b_cm;
        constexpr size_t PDL_MIRROR_LIMIT = (size_t) 1 << 30;
        use_mirror = need <= PDL_MIRROR_LIMIT && c->opt_host_mirror;
        if (use_mirror && !c->mirror_valid) {
            PDL_HIP(hipSetDevice(c->device));
            hipStream_t st = c->stream;
            if (c->mirror_bytes < need) {
                if (c->mirror) { (void) hipHostFree(c->mirror); c->mirror = nullptr; c->mirror_bytes = 0; }
                PDL_HIP(hipHostMalloc((void **) &c->mirror, need + need / 4 + 64, hipHostMallocDefault));
                c->mirror_bytes = need + need / 4 + 64;
            }
            uint8_t *m = c->mirror;
            const void *src[5] = {c->c_score.p, c->c_perc.p, c->c_tr.p, c->c_row.p, c->c_col.p};
            for (int i = 0; i < 5; i++)
                if (b_cells) PDL_HIP(hipMemcpyAsync(m + (size_t) i * b_cells, src[i], b_cells, hipMemcpyDeviceToHost, st));
            if (b_ms) PDL_HIP(hipMemcpyAsync(m + 5 * b_cells, c->MS.p, b_ms, hipMemcpyDeviceToHost, st));
            if (b_cm) PDL_HIP(hipMemcpyAsync(m + 5 * b_cells + b_ms, c->CM.p, b_cm, hipMemcpyDeviceToHost, st));
            PDL_HIP(hipStreamSynchronize(st));
            c->mirror_valid = true;
        }
    }
    const int32_t lg = c->h_local_genome[genome];
    const uint32_t N = c->N, G = c->G;
    const uint64_t z0 = c->h_cell_off[lg], z1 = c->h_cell_off[lg + 1];
    const uint32_t z = (uint32_t) (z1 - z0);
    const uint32_t p0 = c->h_task_row_off[lg], rows = c->h_task_row_off[lg + 1] - p0;
    out->scoresCount = z; out->rows = rows; out->genomes = G; out->sequences = N;
    out->scores = xalloc<float>(z); out->percs = xalloc<float>(z); out->tr_percs = xalloc<float>(z);
    out->row = xalloc<int32_t>(z); out->column = xalloc<int32_t>(z);
    out->first_seq_genome = xalloc<int32_t>(z); out->second_seq_genome = xalloc<int32_t>(z);
    out->max_genome_score = xalloc<float>((size_t) rows * G);
    out->max_genome_score_col = xalloc<float>(N);
    out->scoresMaxMappings = xalloc<int32_t>(N);
    if (use_mirror) {
        const uint8_t *m = c->mirror;
        void *dst[5] = {out->scores, out->percs, out->tr_percs, out->row, out->column};
        for (int i = 0; i < 5; i++) if (z) memcpy(dst[i], m + (size_t) i * b_cells + (size_t) z0 * 4, (size_t) z * 4);
        if (rows) memcpy(out->max_genome_score, m + 5 * b_cells + (size_t) p0 * G * 4, (size_t) rows * G * 4);
        memcpy(out->max_genome_score_col, m + 5 * b_cells + b_ms + (size_t) lg * N * 4, (size_t) N * 4);
    } else {
        ThreadCopier &tc = t_copier;
        tc.open(c->device);
        if (z) {
            tc.fetch(out->scores, c->c_score.as<float>() + z0, (size_t) z * 4);
            tc.fetch(out->percs, c->c_perc.as<float>() + z0, (size_t) z * 4);
            tc.fetch(out->tr_percs, c->c_tr.as<float>() + z0, (size_t) z * 4);
            tc.fetch(out->row, c->c_row.as<int32_t>() + z0, (size_t) z * 4);
            tc.fetch(out->column, c->c_col.as<int32_t>() + z0, (size_t) z * 4);
        }
        if (rows) tc.fetch(out->max_genome_score, c->MS.as<float>() + (size_t) p0 * G, (size_t) rows * G * 4);
        tc.fetch(out->max_genome_score_col, c->CM.as<float>() + (size_t) lg * N, (size_t) N * 4);
    }
    // library.cpp:571-575: genome of row / column per cell;  :428-432: flat map
    for (uint32_t i = 0; i < z; i++) {
        out->first_seq_genome[i] = (int32_t) c->h_genome_of[out->row[i]];
        out->second_seq_genome[i] = (int32_t) c->h_genome_of[out->column[i]];
    }
    for (uint32_t i = 0; i < N; i++) out->scoresMaxMappings[i] = std::numeric_limits<int32_t>::max();
    for (uint32_t j = 0; j < rows; j++) out->scoresMaxMappings[c->h_genome_rows[c->h_genome_row_off[genome] + j]] = (int32_t) j;
    return PDL_OK;
    } catch (const pdl_error &e) { { std::lock_guard<std::mutex> lk(c->mu); c->err = e.msg; } pdl_free_scores(out); return e.code;
    } catch (const std::bad_alloc &) { { std::lock_guard<std::mutex> lk(c->mu); c->err = "host allocation failed"; } pdl_free_scores(out); return PDL_ERR_DEVICE; }
}

void pdl_free_edges(pdl_edges *e) {
    if (!e) return;
    free(e->src); free(e->dst); free(e->score);
    memset(e, 0, sizeof(*e));
}

int pdl_compute_edges(pdl_ctx *c, uint32_t genome, pdl_edges *out) {
    if (!c || !out) return PDL_ERR_ARGUMENT;
    memset(out, 0, sizeof(*out));
    try {
    {
        std::lock_guard<std::mutex> lk(c->mu);      // the scoring pass / the filter run once; slicing needs no lock
        int rc = c->scored ? PDL_OK : score_all_locked(c);
        if (rc != PDL_OK) return rc;
        if (genome >= c->G) PDL_FAIL(PDL_ERR_ARGUMENT, "genome %u out of range (%u genomes)", genome, c->G);
        if (c->h_local_genome[genome] < 0) PDL_FAIL(PDL_ERR_ARGUMENT, "genome %u is not in this context's shard", genome);
        if (!c->edges_valid) {
            PDL_HIP(hipSetDevice(c->device));
            pdl_run_bbh_all(c);
        }
    }
    const int32_t lg = c->h_local_genome[genome];
    const uint64_t a1 = c->h_edge1[lg], b1 = c->h_edge1[lg + 1], a2 = c->h_edge_off[lg], b2 = c->h_edge_off[lg + 1];
    const uint64_t n1 = c->n_edges1, n2 = c->n_edges - c->n_edges1;
    const uint32_t cnt = (uint32_t) ((b1 - a1) + (b2 - a2));
    out->count = cnt;
    out->src = xalloc<int32_t>(cnt); out->dst = xalloc<int32_t>(cnt); out->score = xalloc<float>(cnt);
    const uint8_t *m = c->edge_mirror;
    const size_t k1 = (size_t) (b1 - a1), k2 = (size_t) (b2 - a2);
    if (k1) {
        memcpy(out->src, m + a1 * 4, k1 * 4); memcpy(out->dst, m + n1 * 4 + a1 * 4, k1 * 4); memcpy(out->score, m + n1 * 8 + a1 * 4, k1 * 4);
    }
    if (k2) {
        const uint8_t *m2 = m + n1 * 12;
        memcpy(out->src + k1, m2 + a2 * 4, k2 * 4); memcpy(out->dst + k1, m2 + n2 * 4 + a2 * 4, k2 * 4); memcpy(out->score + k1, m2 + n2 * 8 + a2 * 4, k2 * 4);
    }
    return PDL_OK;
    } catch (const pdl_error &e) { { std::lock_guard<std::mutex> lk(c->mu); c->err = e.msg; } pdl_free_edges(out); return e.code;
    } catch (const std::bad_alloc &) { { std::lock_guard<std::mutex> lk(c->mu); c->err = "host allocation failed"; } pdl_free_edges(out); return PDL_ERR_DEVICE; }
}

int pdl_set_option(pdl_ctx *c, const char *name, int64_t value) {
    if (!c || !name) return PDL_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lk(c->mu);
    const std::string n(name);
    if (n == "join_tier1") {
        if (!(value == -1 || value == 0 || (value >= 9 && value <= 11) || value == 20 || value == 21)) { c->err = "join_tier1: -1, 0, 9, 10, 11, 20 or 21"; return PDL_ERR_ARGUMENT; }
        c->opt_tier1 = (int) value;
    } else if (n == "join_tier0") c->opt_tier0 = value < 0 ? -1 : (value != 0);
    else if (n == "join_tiny_tier2") c->opt_tiny_tier2 = value != 0;
    else if (n == "join_grid_pct") c->opt_grid_pct = (int) value;
    else if (n == "stage_timers") c->opt_stage_timers = value != 0;
    else if (n == "host_mirror") c->opt_host_mirror = value != 0;
    else if (n == "staging_cap") c->opt_staging_cap = value > 0 ? (uint64_t) value : 0;
    else if (n == "aside_test_reload") c->opt_aside_test_reload = value != 0;
    else if (n == "onepass_scan") c->opt_onepass_scan = value != 0;
    else if (n == "low_memory") c->opt_low_memory = value != 0;
    else { c->err = "unknown option " + n; return PDL_ERR_ARGUMENT; }
    c->scored = false;        // the next scoring call runs with the new setting
    return PDL_OK;
}

// ---- multi-GPU (see include/pandelos_amd.h) -------------------------------------------------------------------------
int pdl_dist_preprocess_begin(pdl_ctx *c, const uint8_t *d_residues, const uint64_t *d_offsets, const uint32_t *d_genome_of,
                              uint32_t n, uint64_t n_res, int k, uint32_t world, uint32_t rank, pdl_dist_slice *out) {
    if (!c || !out) return PDL_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lk(c->mu);
    PDL_GUARD_BEGIN
    if (!d_offsets || !d_genome_of || (!d_residues && n_res)) PDL_FAIL(PDL_ERR_ARGUMENT, "null input pointer");
    if (((uintptr_t) d_residues & 15) != 0) PDL_FAIL(PDL_ERR_ARGUMENT, "d_residues must be 16-byte aligned");
    if (world == 0 || rank >= world || world > 64) PDL_FAIL(PDL_ERR_ARGUMENT, "rank %u of %u (at most 64 ranks)", rank, world);
    PDL_HIP(hipSetDevice(c->device));
    c->preprocessed = false; c->scored = false; c->tasks_ready = false; c->mirror_valid = false;
    c->dist = false; c->dist_stage = 0;
    adopt_device_input(c, d_residues, d_offsets, d_genome_of, n, n_res);
    c->N = n; c->R = n_res;
    c->U = c->Ushared = c->NG = c->P = c->M = 0;
    if (k <= 0) PDL_FAIL(PDL_ERR_KVALUE, "K value must be greater than 0.");
    if (n == 0) PDL_FAIL(PDL_ERR_EMPTY, "empty dataset");
    build_genome_layout(c);
    c->world = world; c->rank = rank;
    c->shard.clear(); c->shard_set = false; c->dict_shard.clear();
    pdl_run_dist_begin(c, k);
    out->d_postings = c->post.p; out->records = c->U_slice; out->kmers = c->M_slice;
    out->genome_weights = c->h_run_weights.data(); out->genomes = c->G;
    out->genome_costs = c->h_run_costs.data();
    return PDL_OK;
    PDL_GUARD_END(c)
}

int pdl_dist_preprocess_finish(pdl_ctx *c, void *d_postings_all, uint64_t total_records, const uint64_t *genome_weights, pdl_cost *out_cost) {
    if (!c) return PDL_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lk(c->mu);
    PDL_GUARD_BEGIN
    if (!c->dist || c->dist_stage != 1) PDL_FAIL(PDL_ERR_STATE, "pdl_dist_preprocess_finish without pdl_dist_preprocess_begin");
    if (c->dist_sender) PDL_FAIL(PDL_ERR_STATE, "pdl_dist_preprocess_ranges built this run's range tuples (and took its group-head bits out): finish with pdl_dist_preprocess_finish_ranges");
    if (!d_postings_all || ((uintptr_t) d_postings_all & 7) != 0) PDL_FAIL(PDL_ERR_ARGUMENT, "the gathered dictionary must be an 8-byte aligned device array");
    if (total_records < c->U_slice) PDL_FAIL(PDL_ERR_ARGUMENT, "the gathered dictionary (%llu records) is smaller than this rank's run (%llu)",
                                             (unsigned long long) total_records, (unsigned long long) c->U_slice);
    PDL_HIP(hipSetDevice(c->device));
    c->post_ext = static_cast<uint2 *>(d_postings_all);
    pdl_run_dist_finish(c, total_records, genome_weights);
    c->preprocessed = true;
    fill_cost(c, out_cost);
    return PDL_OK;
    PDL_GUARD_END(c)
}

int pdl_dist_preprocess_ranges(pdl_ctx *c, const uint64_t *run_records, const uint64_t *genome_weights, const uint64_t *genome_costs, pdl_dist_ranges *out) {
    if (!c || !out || !run_records || !genome_weights || !genome_costs) return PDL_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lk(c->mu);
    PDL_GUARD_BEGIN
    if (!c->dist || c->dist_stage != 1) PDL_FAIL(PDL_ERR_STATE, "pdl_dist_preprocess_ranges without pdl_dist_preprocess_begin");
    PDL_HIP(hipSetDevice(c->device));
    memset(out, 0, sizeof(*out));
    out->available = pdl_run_dist_ranges(c, run_records, genome_weights, genome_costs) ? 1 : 0;
    out->d_keys = c->dist_out_keys; out->d_ranges = reinterpret_cast<const uint64_t *>(c->dist_out_ranges);
    out->counts = c->h_tuple_counts.data(); out->total = c->dist_out_total;
    out->shared_records = c->dist_run_counters[0]; out->groups = c->dist_run_counters[1]; out->repeat_sample = c->dist_run_counters[2];
    return PDL_OK;
    PDL_GUARD_END(c)
}

int pdl_dist_preprocess_finish_ranges(pdl_ctx *c, void *d_postings_all, uint64_t total_records, uint32_t *d_keys, uint64_t *d_ranges, uint64_t n_tuples,
                                      const uint64_t *counter_sums, pdl_cost *out_cost) {
    if (!c || !counter_sums || ((!d_keys || !d_ranges) && n_tuples)) return PDL_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lk(c->mu);
    PDL_GUARD_BEGIN
    if (!c->dist || c->dist_stage != 1 || !c->dist_sender) PDL_FAIL(PDL_ERR_STATE, "pdl_dist_preprocess_finish_ranges without a pdl_dist_preprocess_ranges that said \"available\"");
    if (!d_postings_all || ((uintptr_t) d_postings_all & 7) != 0) PDL_FAIL(PDL_ERR_ARGUMENT, "the gathered dictionary must be an 8-byte aligned device array");
    if (n_tuples && ((((uintptr_t) d_keys) & 3) != 0 || (((uintptr_t) d_ranges) & 7) != 0)) PDL_FAIL(PDL_ERR_ARGUMENT, "the received tuples must be aligned device arrays");
    PDL_HIP(hipSetDevice(c->device));
    c->post_ext = static_cast<uint2 *>(d_postings_all);
    pdl_run_dist_finish_ranges(c, total_records, d_keys, reinterpret_cast<unsigned long long *>(d_ranges), n_tuples, counter_sums);
    c->preprocessed = true;
    fill_cost(c, out_cost);
    return PDL_OK;
    PDL_GUARD_END(c)
}

int pdl_dist_genome_owner(const pdl_ctx *c, uint32_t *out) {
    if (!c || !out) return PDL_ERR_ARGUMENT;
    if (!c->dist || c->dist_stage < 2) return PDL_ERR_STATE;
    memcpy(out, c->h_owner.data(), (size_t) c->G * 4);
    return PDL_OK;
}

int pdl_dist_score_begin(pdl_ctx *c, pdl_dist_outbox *out) {
    if (!c || !out) return PDL_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lk(c->mu);
    PDL_GUARD_BEGIN
    if (!c->dist || c->dist_stage < 2) PDL_FAIL(PDL_ERR_STATE, "pdl_dist_score_begin before pdl_dist_preprocess_finish");
    PDL_HIP(hipSetDevice(c->device));
    c->scored = false; c->mirror_valid = false; c->edges_valid = false; c->dist_stage = 2;
    pdl_run_dist_score_begin(c);
    out->d_cells = c->outbox.as<pdl_dist_cell>();
    out->counts = c->h_outbox_counts.data();
    out->total = 0;
    for (uint64_t v : c->h_outbox_counts) out->total += v;
    return PDL_OK;
    PDL_GUARD_END(c)
}

int pdl_dist_score_finish(pdl_ctx *c, const pdl_dist_cell *d_inbox, uint64_t n_inbox) {
    if (!c || (!d_inbox && n_inbox)) return PDL_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lk(c->mu);
    PDL_GUARD_BEGIN
    if (!c->dist || c->dist_stage != 3) PDL_FAIL(PDL_ERR_STATE, "pdl_dist_score_finish without pdl_dist_score_begin");
    if (n_inbox >= 0xffffffffull) PDL_FAIL(PDL_ERR_UNSUPPORTED, "more than 2^32 received cells");
    PDL_HIP(hipSetDevice(c->device));
    pdl_run_dist_score_finish(c, d_inbox, n_inbox);
    return PDL_OK;
    PDL_GUARD_END(c)
}

int pdl_copy_device(pdl_ctx *c, void *d_dst, const void *d_src, uint64_t bytes) {
    if (!c || ((!d_dst || !d_src) && bytes)) return PDL_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lk(c->mu);
    PDL_GUARD_BEGIN
    PDL_HIP(hipSetDevice(c->device));
    if (bytes) PDL_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, c->stream));
    PDL_HIP(hipStreamSynchronize(c->stream));
    return PDL_OK;
    PDL_GUARD_END(c)
}

int pdl_get_dictionary(pdl_ctx *c, uint64_t *ranks, uint32_t *seqs, uint32_t *counts) {
    if (!c) return PDL_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lk(c->mu);
    PDL_GUARD_BEGIN
    if (!c->preprocessed) PDL_FAIL(PDL_ERR_STATE, "pdl_get_dictionary before pdl_preprocess");
    if (c->dist) PDL_FAIL(PDL_ERR_STATE, "pdl_get_dictionary: a multi-GPU context keeps ranks only for its own interval");
    if (!c->keys_b.p || !c->recpos.p) PDL_FAIL(PDL_ERR_STATE, "pdl_get_dictionary: the sorted k-mer stream was released (option low_memory)");
    PDL_HIP(hipSetDevice(c->device));
    const uint64_t U = c->U, M = c->M;
    std::vector<uint32_t> recpos(U);
    std::vector<uint2> post(U);
    PDL_HIP(hipMemcpyAsync(recpos.data(), c->recpos.p, U * 4, hipMemcpyDeviceToHost, c->stream));
    PDL_HIP(hipMemcpyAsync(post.data(), c->post.p, U * 8, hipMemcpyDeviceToHost, c->stream));
    std::vector<uint8_t> keys(M * (c->key64 ? 8 : 4));
    PDL_HIP(hipMemcpyAsync(keys.data(), c->keys_b.p, keys.size(), hipMemcpyDeviceToHost, c->stream));
    PDL_HIP(hipStreamSynchronize(c->stream));
    for (uint64_t u = 0; u < U; u++) {
        if (ranks) ranks[u] = c->key64 ? reinterpret_cast<uint64_t *>(keys.data())[recpos[u]]
                                       : (uint64_t) reinterpret_cast<uint32_t *>(keys.data())[recpos[u]];
        if (seqs) seqs[u] = post[u].x;
        if (counts) counts[u] = post[u].y & 0x7fffffffu;       // (complexity-only mode leaves the group-head bit in place)
    }
    return PDL_OK;
    PDL_GUARD_END(c)
}

int pdl_get_rank_table(const pdl_ctx *c, uint8_t out[256], uint64_t *out_lm) {
    if (!c || !out) return PDL_ERR_ARGUMENT;
    if (!c->preprocessed) return PDL_ERR_STATE;
    memcpy(out, c->rp.rank_values, 256);
    if (out_lm) *out_lm = c->rp.last_multiplier;
    return PDL_OK;
}

int pdl_get_timings(pdl_ctx *c, pdl_timings *out) {
    if (!c || !out) return PDL_ERR_ARGUMENT;
    std::lock_guard<std::mutex> lk(c->mu);
    *out = c->tm;
    return PDL_OK;
}

}  // extern "C"
