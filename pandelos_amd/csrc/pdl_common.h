// pdl_common.h — shared declarations of the HIP implementation behind include/pandelos_amd.h.
// gfx950 only: wave = 64 lanes, 160 KiB LDS per CU, no other target is considered.
#pragma once

#include <atomic>
#include <chrono>

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <mutex>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pandelos_amd.h"

#define PDL_WAVE 64

// The workgroup barrier of every kernel here: all LDS operations of the wave have COMPLETED before it arrives.
// hipcc (ROCm 7.2, gfx950) leaves the `s_waitcnt lgkmcnt(0)` out in front of an `s_barrier` when the wave needs no LDS result
// any more — e.g. behind a run of no-return LDS atomics or stores — relying on the LDS pipe serving the CU's waves in order.
// On MI355X that does not hold across waves: with two or three 53-KB workgroups per CU the 2048-slot join tier read staged
// ranges and table slots of the row BEFORE (the wave in front of the barrier still had writes queued), a few wrong cells per
// pass on the last genomes of a 16-genome set, 0 with this wait in place.  The wait costs nothing
// when nothing is outstanding.
#ifdef __HIPCC__
__device__ __forceinline__ void pdl_sync() {
#ifndef PDL_PLAIN_SYNCTHREADS        // (defined only by the negative control of tests/test_isa.py: the barrier as hipcc emits it by itself)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    __syncthreads();
}
#endif

// layout of the control block (pdl_ctx::scalars, u64 words): scalars | residue histogram | per-genome cost
constexpr size_t PDL_CTL_HIST = 16, PDL_CTL_GCOST = 16 + 256;

// ---- error plumbing -------------------------------------------------------------------------
struct pdl_error {
    int code;
    std::string msg;
};

#define PDL_HIP(expr)                                                                           \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            throw pdl_error{PDL_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)}; \
        }                                                                                       \
    } while (0)

#define PDL_FAIL(code_, ...)                                  \
    do {                                                      \
        char _b[512];                                         \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                \
        throw pdl_error{(code_), std::string(_b)};            \
    } while (0)

// ---- device buffer ----------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    void alloc(size_t n) {            // grows only; contents undefined
        if (n <= bytes && p) return;
        release();
        size_t want = n ? n : 16;
        PDL_HIP(hipMalloc(&p, want));
        bytes = want;
    }
    void release() {
        if (p) (void) hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
    ~DevBuf() { release(); }
    void grow_keep(size_t n, hipStream_t st) {   // grows and keeps the contents (a device copy on `st`, waited for)
        if (n <= bytes && p) return;
        void *q = nullptr;
        PDL_HIP(hipMalloc(&q, n));
        if (p && bytes) {
            PDL_HIP(hipMemcpyAsync(q, p, bytes, hipMemcpyDeviceToDevice, st));
            PDL_HIP(hipStreamSynchronize(st));
            (void) hipFree(p);
        }
        p = q;
        bytes = n;
    }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};

// a typed window into someone else's device allocation
struct DevView {
    void *p = nullptr;
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// ---- alphabet / rank parameters (library.cpp:56-64, 88-132) -----------------------------------
struct RankParams {
    uint8_t rank_values[256];
    uint64_t last_multiplier;
    uint32_t base;          // B (kept as the reference's unsigned char value)
    uint32_t k;
    uint32_t rank_bits;     // bits of a rank as rank_init sees them: what the sort by rank goes over (the reference's rank_byte_order bytes)
    uint32_t hash_fallback;
    uint32_t key_bits;      // bits a rank can really occupy: rank_bits — or 64 where B^k wrapped past 2^64 UNNOTICED by rank_init's overflow test
                            // (library.cpp:104-111 looks one multiplication ahead with wrapped products: 22 letters, k = 15), the ranks being the polynomial mod 2^64
};

struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    bool used = false;
};

// ---- the context --------------------------------------------------------------------------------
struct pdl_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint32_t flags = 0;
    std::string err;
    std::mutex mu;

    // inputs (device)
    const uint8_t *d_res = nullptr;
    const uint64_t *d_off = nullptr;
    const uint32_t *d_gen = nullptr;
    DevBuf in_res, in_off, in_gen;

    // device-resident input: genome ids (and the two ends of the offsets) come to the host through this pinned buffer while the
    // first kernels already run; the genome layout is built behind them (pdl_finish_layout)
    uint32_t *gen_pin = nullptr; size_t gen_pin_words = 0;
    hipEvent_t ev_gen = nullptr;
    bool layout_deferred = false;

    bool preprocessed = false;
    bool only_complexity = false;
    uint32_t N = 0, G = 0;
    uint64_t R = 0, M = 0, U = 0, Ushared = 0, NG = 0, P = 0, sum_kseq = 0, max_kseq = 0, min_kseq = 0;
    uint64_t Urepeat = 0;          // dictionary records with a count >= 2 (the k-mer repeats inside the gene): the "heavy" lookups of the join come from these
    RankParams rp{};

    // K-ingest (pdl_ingest.hip): the input a .faa file was parsed into, and the two pinned staging buffers it travelled through
    DevBuf ing_res, ing_off, ing_gen;
    uint8_t *ing_pin[2] = {nullptr, nullptr};    // [0]: the pinned staging buffer of the ingest (the whole file's residues)
    size_t ing_pin_bytes = 0;
    hipEvent_t ing_ev[2] = {nullptr, nullptr};
    hipStream_t ing_stream = nullptr;
    std::vector<uint64_t> ing_h_off;
    std::vector<uint32_t> ing_h_gen;
    std::vector<std::string> ing_genome_names;
    uint64_t ing_R = 0;
    bool ingested = false;

    // per sequence
    DevBuf kseq_len;      // u32 [N]
    DevBuf kmer_off;      // u64 [N+1]
    DevBuf cost;          // u64 [N]   total_visited
    std::vector<uint64_t> h_genome_cost;

    // sort buffers / dictionary
    DevBuf keys_a, keys_b, vals_a, vals_b, sort_tmp;
    bool key64 = false;
    DevBuf recpos;        // u32 [U+1] position of each record's first occurrence in the sorted stream
    DevBuf post;          // uint2 [U] {seq, count}  — the dictionary postings, rank-group major (bit 31 of count: opens a rank-group, until K-ranges removes it)
    DevBuf ranges;        // uint4 [U'] {group start, group length, own count, 0}, gene major
    const uint2 *ranges8 = nullptr;   // packed 8-byte ranges (inside `scratch`, the gene sort's output) when the dataset allows: see GroupTileArgs::pay8
    DevBuf head_bits;     // u64 [U / 64] "opens a rank-group" per record, kept by K-ranges for the per-gene costs made on demand
    bool costs_ready = true;          // cost[] / h_genome_cost hold the per-gene / per-genome lookups (packed ranges: made on first request)
    DevBuf seq_off;       // u32 [N+1] range list of each gene
    bool upper_only = false;  // ranges hold only the columns above the row: the join mirrors every cell
    DevBuf scan_tmp;      // block sums of the scans
    DevBuf lb_tile, lb_chunk, lb_ctr;     // decoupled look-back (pdl_scan.h): status words of tiles / chunks, ticket + arrival counters
    uint32_t lb_epoch = 0;
    bool opt_low_memory = false;          // batches of genomes on a large set: build buffers released after the dictionary, small HBM tables for tier 3
    bool reshard_pending = false;         // pdl_set_genome_shard named genomes the range lists on the device do not cover: they are built before the next scoring pass
    bool opt_onepass_scan = false;        // scans in one launch (decoupled look-back) instead of three: measured 3-10 % slower per scan on MI355X, kept as an option
    DevBuf scratch;       // transient buffers of the range build
    DevBuf scalars;       // control block, u64: totals [16] | residue histogram [256] | per-genome cost [G] (PDL_CTL_*)

    // genome / task layout (host and device)
    std::vector<uint32_t> h_genome_of;
    std::vector<uint32_t> h_genome_row_off;   // [G+1] into h_genome_rows
    std::vector<uint32_t> h_genome_rows;      // gene ids grouped by genome, ascending inside a genome
    std::vector<uint32_t> shard;              // genomes scored by this context (ascending)
    bool shard_set = false;
    std::vector<uint32_t> dict_shard;         // genomes whose genes have range lists on the device (empty = all)
    DevBuf seq_in_shard;                      // u8 [N] gene belongs to dict_shard
    DevBuf own_iv;                            // uint2 [intervals] the same as sorted gene-id intervals, when genomes are consecutive id ranges
    std::vector<uint2> h_own_iv;
    std::vector<uint8_t> h_seq_in_shard;      // its host copy (kept alive for the asynchronous upload)

    // scoring results (device)
    bool scored = false;
    uint32_t n_task_rows = 0;
    std::vector<int32_t> h_local_genome;      // [G] genome -> index in shard or -1
    std::vector<uint32_t> h_task_row_off;     // [shard+1] task position of each shard genome's first row
    std::vector<uint64_t> h_cell_off;         // [shard+1] first cell of each shard genome
    // task layout on the device: one allocation, one upload from a pinned staging buffer (pdl_prepare_tasks)
    DevBuf task_blob;
    uint32_t *task_pin = nullptr; size_t task_pin_words = 0;
    hipStream_t copy_stream = nullptr; hipEvent_t ev_tasks = nullptr, ev_entry = nullptr;     // uploads of the task layout, beside the build's stream
    DevView task_rows;    // u32 [n_task_rows] gene id of each task position
    DevView task_lg;      // u32 [n_task_rows] shard-local genome index
    DevBuf row_desc;      // uint4 [n_task_rows] {task position, gene, first range, ranges} in processing order
    DevBuf MS;            // f32 [n_task_rows][G]      max_genome_score rows
    DevBuf CM;            // f32 [shard][N]            max_genome_score_col per genome task
    DevBuf row_base, row_cnt, fin_off;   // u32 [n_task_rows(+1)]
    DevBuf st_score, st_perc, st_tr, st_col, st_first;   // staging cells (unordered inside a row)
    uint64_t st_cap = 0;
    DevBuf c_score, c_perc, c_tr, c_row, c_col;          // final cells, task order + emission order
    uint64_t Z = 0;
    DevBuf join_ctr;      // u32 [8] cursors/counters of the join
    DevBuf overflow_rows; // u32 [n_task_rows]
    DevBuf gene_info;             // uint4 [N] {k-mers, genome, task position, shard-local genome}: one load per candidate column in finalize
    DevBuf join_defer;            // filter tiers of the join: per workgroup, the first sightings put aside
    DevBuf glb_table;     // HBM tables of the overflow pass
    bool glb_clean = false;   // all-zero (k_join_hbm leaves them that way)
    DevBuf row_desc2;     // descriptors of the rows handed from tier 1 to tier 2
    DevBuf mirror_cnt, mirror_ref;   // mirror mode (see pdl_join.hip); mirror_ref: the mirrored cells (MCell), per row
    DevView taskpos_of;                      // u32 [N] task position of every gene (0xffffffff: not a row of this context)
    std::vector<uint32_t> h_fin;
    bool tasks_ready = false;  // task layout uploaded for the current shard
    DevBuf scratch2;      // small transient device scratch (interval histogram, per-genome lookups)
    DevBuf task_off;      // u32 [shard+1] task offsets | gathered cell offsets + 8 counters + cell total
    int cus = 0;
    uint32_t occ_tier1[5] = {0, 0, 0, 0, 0};
    uint32_t occ_tier0 = 0, occ_tier0b = 0;

    // K-bbh (pdl_bbh.hip): network edges of every genome task, on the host after the first pdl_compute_edges
    bool edges_valid = false;
    DevBuf bbh_kind, bbh_tab, e_src, e_dst, e_score;
    uint8_t *edge_mirror = nullptr;           // pinned: src1 | dst1 | score1 (phase 1) | src2 | dst2 | score2 (phase 2)
    size_t edge_mirror_bytes = 0;
    uint64_t n_edges = 0, n_edges1 = 0;
    std::vector<uint32_t> h_bbh_at;
    std::vector<uint64_t> h_edge1, h_edge_off;   // [shard+1] first phase-1 / phase-2 edge of every genome task

    // tuning / test switches (pdl_set_option)
    int opt_tier1 = -1;           // -1: by genome count
    int opt_tier0 = -1;           // the partition tier in front of tier 1: -1 by row length, 0 off, 1 on
    bool opt_tiny_tier2 = false;
    bool opt_stage_timers = true; // per-stage HIP events (pdl_timings' stage fields); the totals and the join's are always taken
    int opt_grid_pct = 0;         // > 0: tier-1 grid as a percentage of what fits the chip (experiments)
    bool opt_host_mirror = true;
    uint64_t opt_staging_cap = 0; // 0: estimate
    bool opt_aside_test_reload = false;   // test switch: the next pass with 8-byte put-aside entries counts as one that saw a reload

    // multi-GPU (pdl_dist_*): this context is rank `rank` of `world`; the postings live in caller-owned memory
    bool dist = false;
    uint32_t world = 1, rank = 0;
    int dist_stage = 0;           // 0 none | 1 slice built | 2 dictionary adopted | 3 join done, outbox listed | 4 scored
    uint2 *post_ext = nullptr;    // the all-gathered dictionary (caller's buffer), used instead of `post`
    uint64_t U_slice = 0, M_slice = 0;
    std::vector<uint32_t> h_owner;            // [G] rank of every genome
    std::vector<uint64_t> h_upper_cost;       // [G] lookups above the diagonal per genome (what a rank's join walks)
    std::vector<uint64_t> h_run_weights;      // [G] the same inside this rank's run of the dictionary (summed over ranks: the deal's weights)
    std::vector<uint64_t> h_run_costs;        // [G] lookups as the reference counts them, inside this run (summed over ranks: "Genome g cost")
    // sender-built range lists (pdl_dist_preprocess_ranges / _finish_ranges): every rank makes the range tuples of ITS run and
    // files them by the rank that owns the gene; the owners only sort what they receive
    bool dist_sender = false;                 // the range lists of this build came from the senders (per-gene costs are not kept then)
    int dist_tail = -1;                       // last rank with a non-empty interval (known to every rank: the cuts depend on the input only)
    uint64_t run_base = 0, dist_total = 0;    // first record of this run in the gathered dictionary; records of all runs
    DevBuf seq_owner;                         // u8 [N] rank that owns the gene
    DevBuf tuple_off;                         // u32 [world + 1] where each destination's tuples start in the outbox
    std::vector<uint64_t> h_tuple_counts;     // [world]
    uint32_t *dist_out_keys = nullptr;        // the outbox (inside `scratch`): (owner << 24 | gene), packed range
    unsigned long long *dist_out_ranges = nullptr;
    uint64_t dist_out_total = 0, dist_run_counters[3] = {0, 0, 0};     // shared records, groups, repeat statistic of this run
    DevBuf owner_of_genome;                   // u32 [G]
    DevView local_genome;                     // u32 [G] index in the shard, 0xffffffff for other ranks' genomes
    DevBuf outbox;                            // pdl_dist_cell [remote mirrored cells], grouped by destination rank
    DevBuf outbox_tab;                        // u32 [workgroups][world] counts, then offsets
    std::vector<uint64_t> h_outbox_counts;    // [world]
    uint64_t st_local = 0;                    // staging slots below this hold the cells of this rank's own rows
    uint64_t n_inbox = 0;
    uint32_t order_grid1 = 0;

    pdl_timings tm{};
    EventPair ev[16];
    uint8_t *pin = nullptr;       // pinned host scratch for the small device->host reads (true async DMA, no staging copy)
    uint8_t *pin_dev = nullptr;   // the same buffer as the device addresses it (hipHostGetDevicePointer)
    size_t pin_bytes = 0;
    uint32_t pin_epoch = 0;       // last value a k_pin_read raised the flag at the tail of `pin` to
    // host mirror of the whole scoring result (pinned): filled by ONE set of device->host copies at the first
    // pdl_compute_scores after a scoring pass, so the per-genome calls are host memcpys (results up to PDL_MIRROR_LIMIT)
    uint8_t *mirror = nullptr;
    size_t mirror_bytes = 0;
    bool mirror_valid = false;
};

// Small device->host reads through the pinned scratch: queue with add(), one sync(), then read the returned pointers.
// Small device -> host reads (counters, control blocks) WITHOUT a DMA copy and without hipStreamSynchronize: on this
// platform a copy of a few hundred bytes is ~30 us of copy-engine latency and the wake-up from a stream wait another
// 20-30 — and a step of the 64-genome set has three such reads.  Instead one tiny kernel, stream-ordered behind the work
// it reads, stores the words straight into pinned host memory (PCIe posted writes), fences at system scope and raises an
// epoch flag in the same buffer; the host spins on the flag (a few us).  A read that is large, oddly sized or does not
// fit the pinned buffer, and a wait that outlasts the spin budget (the stream has long kernels ahead), take the old road.
struct PinReadArgs {
    const uint32_t *src[8];
    uint32_t dst_word[8], words[8];
    uint32_t n;
    uint32_t *pin;             // pinned host buffer as the device sees it
    uint32_t flag_word, epoch;
};
// The arrival checksum weights every word by its position in the WHOLE pinned buffer (odd weights: a single late word always
// shows) — weights that restarted in every segment let two late words at the same index of different segments cancel.
__host__ __device__ inline uint32_t pin_weight(uint32_t dst_word) { return 2u * dst_word + 1u; }
// Host side of the protocol, on plain memory (exported for the CPU tests as pdl_pin_arrived): the flag shows this read's epoch
// AND the words in place add up to the checksum the kernel left beside the flag.
inline bool pin_arrived(const volatile uint32_t *pin, const uint32_t *dst_word, const uint32_t *words, uint32_t n, uint32_t flag_word, uint32_t epoch) {
    if (pin[flag_word] != epoch) return false;
    std::atomic_thread_fence(std::memory_order_acquire);
    uint32_t sum = 0;
    for (uint32_t s = 0; s < n; s++)
        for (uint32_t i = 0; i < words[s]; i++) sum += pin[dst_word[s] + i] * pin_weight(dst_word[s] + i);
    return sum == pin[flag_word + 1];
}
#ifdef __HIPCC__
static __global__ __launch_bounds__(256) void k_pin_read(PinReadArgs a) {
    __shared__ uint32_t s_sum;
    if (threadIdx.x == 0) s_sum = 0;
    pdl_sync();
    uint32_t sum = 0;
    for (uint32_t s = 0; s < a.n; s++)
        for (uint32_t i = threadIdx.x; i < a.words[s]; i += 256) { const uint32_t v = a.src[s][i]; a.pin[a.dst_word[s] + i] = v; sum += v * pin_weight(a.dst_word[s] + i); }
    atomicAdd(&s_sum, sum);
    __threadfence_system();
    pdl_sync();
    // The flag says "all of it has been sent", the checksum beside it lets the host see that all of it has ARRIVED: the
    // words travel from several waves over several paths, and the flag may overtake the last of them.
    if (threadIdx.x == 0) {
        *(volatile uint32_t *) (a.pin + a.flag_word + 1) = s_sum;
        __threadfence_system();
        *(volatile uint32_t *) (a.pin + a.flag_word) = a.epoch;
        __threadfence_system();
    }
}
#endif
constexpr size_t PDL_PIN_FLAG_BYTES = 64;                    // the tail of the pinned buffer holds the flag
struct PinRead {
    pdl_ctx *c;
    size_t used = 0;
    PinReadArgs k{};
    bool by_copy = false;                                    // at least one read went through hipMemcpyAsync
    std::vector<std::pair<void *, std::pair<const void *, size_t>>> spill;   // reads that did not fit: done pageable
    explicit PinRead(pdl_ctx *ctx) : c(ctx) {}
    template <class T> const T *add(const void *d_src, size_t count) {
        const size_t bytes = count * sizeof(T);
        const size_t at = (used + 15) & ~(size_t) 15;
        const size_t room = c->pin_bytes > PDL_PIN_FLAG_BYTES ? c->pin_bytes - PDL_PIN_FLAG_BYTES : 0;
        if (c->pin && at + bytes <= room) {
            used = at + bytes;
            if (k.n < 8 && bytes % 4 == 0 && bytes <= (64u << 10) && (reinterpret_cast<uintptr_t>(d_src) & 3) == 0) {
                k.src[k.n] = static_cast<const uint32_t *>(d_src); k.dst_word[k.n] = (uint32_t) (at / 4); k.words[k.n] = (uint32_t) (bytes / 4);
                k.n++;
            } else {
                by_copy = true;
                PDL_HIP(hipMemcpyAsync(c->pin + at, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
            }
            return reinterpret_cast<const T *>(c->pin + at);
        }
        void *h = malloc(bytes ? bytes : 1);
        spill.push_back({h, {d_src, bytes}});
        by_copy = true;
        PDL_HIP(hipMemcpyAsync(h, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
        return reinterpret_cast<const T *>(h);
    }
    bool issued = false;
    void issue() {                                           // queue the read now; the host may do other work before wait()
        if (issued) return;
        issued = true;
        if (!k.n) return;
        k.pin = reinterpret_cast<uint32_t *>(c->pin_dev ? c->pin_dev : c->pin);     // (the buffer as the device addresses it)
        k.flag_word = (uint32_t) ((c->pin_bytes - PDL_PIN_FLAG_BYTES) / 4);
        k.epoch = ++c->pin_epoch;
        hipLaunchKernelGGL(k_pin_read, dim3(1), dim3(256), 0, c->stream, k);
        PDL_HIP(hipGetLastError());
    }
    void sync() {
        issue();
        if (k.n) {
            if (!by_copy) {
                const volatile uint32_t *words = reinterpret_cast<const volatile uint32_t *>(c->pin);
                auto arrived = [&]() -> bool { return pin_arrived(words, k.dst_word, k.words, k.n, k.flag_word, k.epoch); };    // flag up and every word in place
                const auto t0 = std::chrono::steady_clock::now();
                bool ok = false;
                for (uint32_t spins = 0; !(ok = arrived()); spins++) {
                    if ((spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
                    __builtin_ia32_pause();
                }
                if (ok) { std::atomic_thread_fence(std::memory_order_acquire); return; }
            }
        }
        PDL_HIP(hipStreamSynchronize(c->stream));
    }
    ~PinRead() { for (auto &e : spill) free(e.first); }
};

// stage entry points (pdl_dict.hip / pdl_join.hip)
void pdl_run_preprocess(pdl_ctx *c, int kvalue, bool only_complexity);
void pdl_run_reshard(pdl_ctx *c);       // the range lists again, for the genomes of c->shard, on the dictionary that is there
void pdl_run_dist_begin(pdl_ctx *c, int kvalue);
void pdl_run_dist_finish(pdl_ctx *c, uint64_t total_records, const uint64_t *genome_weights);
bool pdl_run_dist_ranges(pdl_ctx *c, const uint64_t *run_records, const uint64_t *genome_weights, const uint64_t *genome_costs);   // false: not available for this build (use pdl_run_dist_finish)
void pdl_run_dist_finish_ranges(pdl_ctx *c, uint64_t total_records, uint32_t *d_keys, unsigned long long *d_ranges, uint64_t n_tuples, const uint64_t *counter_sums);
void pdl_run_score_all(pdl_ctx *c);
void pdl_run_dist_score_begin(pdl_ctx *c);
void pdl_run_dist_score_finish(pdl_ctx *c, const pdl_dist_cell *d_inbox, uint64_t n_inbox);
void pdl_prepare_tasks(pdl_ctx *c);
void pdl_run_bbh_all(pdl_ctx *c);
void pdl_ensure_costs(pdl_ctx *c);
void pdl_input_arrived(pdl_ctx *c);      // deferred device input: waits for the genome ids / offset ends and checks them
void pdl_finish_layout(pdl_ctx *c);      // ... then builds the genome layout on the host
inline uint2 *pdl_postings(const pdl_ctx *c) { return c->post_ext ? c->post_ext : c->post.as<uint2>(); }

// event helpers
enum { EV_HIST, EV_RANK, EV_SORT1, EV_DICT, EV_SORT2, EV_RANGES, EV_JOIN, EV_JOIN_OVF, EV_ORDER, EV_PRE_TOTAL, EV_SCORE_TOTAL,
       EV_DIST_BEGIN, EV_DIST_FINISH, EV_DIST_SCORE_FINISH, EV_DIST_RANGES, EV_COUNT };

// An event record is a marker packet between two dispatches (a few us of idle stream each); the per-stage pairs can be
// switched off ("stage_timers" 0) when only the totals and the join's launch time are wanted (bench.py's timed loop).
inline bool ev_is_stage(int i) { return i == EV_HIST || i == EV_RANK || i == EV_SORT1 || i == EV_DICT || i == EV_SORT2 || i == EV_RANGES || i == EV_ORDER || i == EV_JOIN_OVF; }
inline void ev_begin(pdl_ctx *c, int i) {
    c->ev[i].used = false;
    if (!c->opt_stage_timers && ev_is_stage(i)) return;
    if (!c->ev[i].a) { PDL_HIP(hipEventCreate(&c->ev[i].a)); PDL_HIP(hipEventCreate(&c->ev[i].b)); }
    PDL_HIP(hipEventRecord(c->ev[i].a, c->stream));
}
inline void ev_end(pdl_ctx *c, int i) {
    if (!c->opt_stage_timers && ev_is_stage(i)) return;
    PDL_HIP(hipEventRecord(c->ev[i].b, c->stream));
    c->ev[i].used = true;
}
inline float ev_ms(pdl_ctx *c, int i) {
    if (!c->ev[i].used) return 0.f;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev[i].a, c->ev[i].b) != hipSuccess) return 0.f;
    return ms;
}

static inline uint32_t bit_length64(uint64_t v) {
    uint32_t b = 0;
    while (v) { b++; v >>= 1; }
    return b;
}
