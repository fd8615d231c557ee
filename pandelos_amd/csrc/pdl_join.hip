// pdl_join.hip — the scoring stage on the device: computeScores (ig/native/library.cpp:409-527) for
// every gene (row) of every genome of the shard in one pass.
//
//   K-join      k_join_lds   sparse all-vs-all multiset-Jaccard join of one row against the dictionary,
//                            accumulators in an LDS hash table (the reference's dense N-length arrays +
//                            colour stamps, library.cpp:421-426,467-477), fused finalize
//                            (library.cpp:493-517) and per-(row, genome) / per-column maxima
//   K-join-hbm  k_join_hbm   same row program with direct-addressed tables in HBM, for the rows whose
//                            candidate set does not fit the LDS table
//   K-order     k_order_rows puts every row's cells in the reference's emission order
//                            (first-touch order, library.cpp:456-482,493 — SURVEY.md §8a row 9a)
//
// Row program (both kernels).  A row gene r owns a list of posting ranges, one per record of r that
// sits in a rank-group of >= 2 records: {group start, group length, own count}.  For every posting
// {c, cnt_c} of every range with own count cnt_r (library.cpp:461-479):
//       inter[c] += min(cnt_c, cnt_r);  perc_cnt[c] += cnt_r;  tr_cnt[c] += cnt_c
// The three sums are packed in one 64-bit word (21 bits each: every sum is bounded by twice the
// k-mer count of a gene, and genes with >= 2^20 k-mers are refused up front), so one lookup is one
// LDS read + one 64-bit LDS atomic add.  `first` keeps the smallest group start that touched c:
// the reference emits a row's cells by (column chunk of 2048, first range that touched the column,
// column), and group starts are monotone in the row's range order.
#include "pdl_common.h"
#include "pdl_scan.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

constexpr int HBM_THREADS = 512;                      // workgroup of the HBM-table kernel
constexpr uint32_t EMPTY_KEY = 0xffffffffu;
constexpr uint64_t FIELD_MASK = (1ull << 21) - 1;
constexpr uint32_t CELL_CHUNK = 16384;                 // staging cells a workgroup reserves at a time (>= any row's candidates in LDS)
// LDS-table geometry: 8192 slots x 16 B = 128 KiB (+ lists and the staged ranges = 154 KiB), one
// 1024-thread workgroup per CU (16 waves).  PDL_JOIN_TABLE_BITS=9 selects a deliberately tiny table
// (512 slots, 64 threads) so that tests can push small inputs through the HBM-table kernel.

struct JoinArgs {
    const uint2 *post;
    const uint4 *ranges;
    const uint32_t *seq_off;
    const uint32_t *kseq_len;
    const uint32_t *genome_of;
    const uint32_t *task_rows;     // gene id of task position p
    const uint32_t *task_lg;       // shard-local genome of task position p
    const uint32_t *work;          // task positions to process (k_join_hbm: the overflow list)
    const uint4 *desc;             // k_join_lds: {task position, gene, first range, ranges} per work item
    uint32_t n_work;
    uint32_t N, G, k;
    uint32_t min_kseq;             // smallest non-zero kseq_length of the dataset (finalize pre-filter)
    uint32_t canonical;            // PDL_FLAG_CANONICAL_ORDER: no first-touch tracking
    float *MS;                     // [n_task_rows][G]
    float *CM;                     // [shard][N]
    uint32_t *row_base, *row_cnt;  // [n_task_rows]
    float *st_score, *st_perc, *st_tr;
    uint32_t *st_col, *st_first;
    unsigned long long st_cap;
    uint32_t *work_cursor;         // persistent-workgroup row dispenser
    unsigned long long *cell_cursor;
    uint32_t *overflow_count;
    uint32_t *error_count;         // internal consistency violations (must stay 0)
    uint32_t *overflow_rows;
    // HBM tables (k_join_hbm only): per workgroup acc u64[N], first u32[N], touched u32[N], emit u32[N]
    unsigned long long *hbm_acc;
    uint32_t *hbm_u32;
};

// finalize one candidate (library.cpp:494-502); returns score (0 when not emitted)
__device__ __forceinline__ float finalize_cell(unsigned long long acc, uint32_t my_kcnt, uint32_t other_kcnt, float threshold,
                                               float &perc, float &tr_perc) {
    const int inter = (int) (acc & FIELD_MASK);
    const int pc = (int) ((acc >> 21) & FIELD_MASK);
    const int tc = (int) (acc >> 42);
    const int union_size = (int) my_kcnt + (int) other_kcnt - inter;
    perc = (float) pc / (float) (int) my_kcnt;
    tr_perc = (float) tc / (float) (int) other_kcnt;
    const bool score_valid = perc >= threshold || tr_perc >= threshold;
    return (float) inter / (float) union_size * (score_valid ? 1.0f : 0.0f);
}

// ------------------------------------------------------------------------------------------------
// K-join (LDS table).  Persistent workgroups pull rows from a global cursor.
//
// Per row:
//   stage     the row's ranges are copied to LDS (coalesced 16-byte loads) with the exclusive prefix of
//             their lengths: the row's lookups become one flat index space [0, L)
//   lookups   lane t handles flat indices t, t+T, ...; four at a time: four binary searches in the LDS
//             prefix (independent, interleaved by the compiler), then four 8-byte posting loads in flight,
//             then per posting one 8-byte LDS read (key, first) and one 64-bit LDS atomic add
//   finalize  touched slots only (the reference's colored_cells); candidates that cannot pass the
//             validity threshold are dropped before their column's k-mer count is fetched
//   emit      cells go to a per-workgroup chunk of the staging area; the emitted-slot list overwrites the
//             touched list in place (a round's slots are in registers before anything is overwritten)
// The table is cleared once per workgroup; afterwards every slot is reset by whoever consumes it.
// ------------------------------------------------------------------------------------------------
template <int HT_BITS_, int T_>
struct JoinCfg {
    static constexpr uint32_t HT = 1u << HT_BITS_;
    static constexpr uint32_t LIMIT = HT / 4 * 3;          // rows with more candidates go to the HBM table
    static constexpr uint32_t TOUCH_CAP = LIMIT + T_;      // < HT: the probe loop always finds a free slot
    static constexpr uint32_t RB = T_;                     // ranges staged per batch
    static_assert(TOUCH_CAP < HT, "table must never fill up");
};

template <int HT_BITS_, int T_>
__global__ __launch_bounds__(T_) void k_join_lds(JoinArgs a) {
    using Cfg = JoinCfg<HT_BITS_, T_>;
    constexpr uint32_t HT = Cfg::HT, LIMIT = Cfg::LIMIT, TOUCH_CAP = Cfg::TOUCH_CAP, RB = Cfg::RB;
    constexpr int T = T_;
    __shared__ unsigned long long s_acc[HT];
    __shared__ uint2 s_kf[HT];                       // {column id, 0xffffffff - smallest group start that touched it}
    __shared__ uint16_t s_touched[TOUCH_CAP];
    __shared__ uint32_t s_gs[RB], s_mc[RB], s_cum[RB + 1];
    __shared__ uint32_t s_wave[17];
    __shared__ uint32_t s_ntouched, s_nemit, s_overflow, s_next;
    __shared__ uint4 s_desc;
    __shared__ unsigned long long s_base, s_chunk_next, s_chunk_end;
    static_assert(TOUCH_CAP <= CELL_CHUNK, "a chunk must hold any row");

    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < HT; i += T) { s_kf[i] = make_uint2(EMPTY_KEY, 0u); s_acc[i] = 0; }
    if (tid == 0) {
        s_ntouched = 0; s_nemit = 0; s_overflow = 0; s_chunk_next = 0; s_chunk_end = 0;
        const uint32_t w0 = atomicAdd(a.work_cursor, 1u);
        s_next = w0;
        if (w0 < a.n_work) s_desc = a.desc[w0];
    }
    const float threshold = 1.0f / (2.0f * (float) a.k);
    const float f_min_kseq = (float) (int) a.min_kseq;
    const bool track_first = a.canonical == 0;

    auto accumulate = [&](uint32_t c, uint32_t cc, uint32_t mc, uint32_t finv) {
        uint32_t slot = (c * 2654435761u) >> (32 - HT_BITS_);
        uint32_t seen_first;
        for (;;) {
            const uint2 kf = s_kf[slot];
            seen_first = kf.y;
            if (kf.x == c) break;
            if (kf.x == EMPTY_KEY) {
                const uint32_t old = atomicCAS(&s_kf[slot].x, EMPTY_KEY, c);
                if (old == EMPTY_KEY) {
                    const uint32_t idx = atomicAdd(&s_ntouched, 1u);
                    if (idx < TOUCH_CAP) s_touched[idx] = (uint16_t) slot;
                    if (idx >= LIMIT) s_overflow = 1;
                    break;
                }
                if (old == c) break;
            }
            slot = (slot + 1) & (HT - 1);
        }
        if (track_first && seen_first < finv) atomicMax(&s_kf[slot].y, finv);
        const unsigned long long add = (unsigned long long) min(cc, mc) | ((unsigned long long) mc << 21) | ((unsigned long long) cc << 42);
        atomicAdd(&s_acc[slot], add);
    };

    for (;;) {
        __syncthreads();
        const uint32_t wi = s_next;
        const uint4 d = s_desc;
        __syncthreads();
        if (wi >= a.n_work) break;                       // uniform: every wave leaves here
        // the next row's ticket and descriptor are fetched now and parked in registers of lane 0 until the
        // end of this row, so the row after this one starts without a dependent global load
        uint32_t next_reg = 0;
        uint4 next_desc = make_uint4(0, 0, 0, 0);
        if (tid == 0) {
            next_reg = atomicAdd(a.work_cursor, 1u);
            if (next_reg < a.n_work) next_desc = a.desc[next_reg];
        }
        const uint32_t p = d.x, r = d.y, e0 = d.z, nr = d.w;
        if (nr == 0) {                                   // gene shares no k-mer group: no candidates
            if (tid == 0) { a.row_base[p] = 0; a.row_cnt[p] = 0; s_next = next_reg; s_desc = next_desc; }
            continue;
        }
        // ---- accumulate (library.cpp:461-479) ------------------------------------------------------
        for (uint32_t b0 = 0; b0 < nr; b0 += RB) {
            const uint32_t nb = min(RB, nr - b0);
            uint32_t len = 0;
            if (tid < nb) {
                const uint4 rg = a.ranges[e0 + b0 + tid];     // {group start, length, own count}
                s_gs[tid] = rg.x; s_mc[tid] = rg.z; len = rg.y;
            }
            uint32_t total;
            const uint32_t ex = block_exclusive_scan_u32(len, s_wave, total);
            s_cum[tid] = tid < nb ? ex : 0xffffffffu;
            if (tid == 0) s_cum[RB] = 0xffffffffu;
            __syncthreads();
            if (tid == 0) s_cum[nb] = total;                 // > every flat index
            __syncthreads();
            for (uint32_t f0 = tid; f0 < total; f0 += 4 * T) {
                if (*(volatile uint32_t *) &s_overflow) break;
                uint32_t rr[4], ff[4];
                uint2 po[4];
                bool live[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    ff[u] = f0 + u * T;
                    live[u] = ff[u] < total;
                    uint32_t pos = 0;                        // largest pos with s_cum[pos] <= f
#pragma unroll
                    for (uint32_t step = RB / 2; step >= 1; step >>= 1)
                        if (s_cum[pos + step] <= ff[u]) pos += step;
                    rr[u] = pos;
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (live[u]) po[u] = a.post[s_gs[rr[u]] + (ff[u] - s_cum[rr[u]])];
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (live[u] && !*(volatile uint32_t *) &s_overflow)   // <= 1 insertion per lane after the flag: TOUCH_CAP < HT holds
                        accumulate(po[u].x, po[u].y, s_mc[rr[u]], 0xffffffffu - s_gs[rr[u]]);
            }
            __syncthreads();
        }
        const uint32_t ntouched = min(s_ntouched, TOUCH_CAP);
        if (s_overflow) {
            // candidate set too large for LDS: hand the row to the HBM kernel, wipe the table
            if (tid == 0) a.overflow_rows[atomicAdd(a.overflow_count, 1u)] = p;
            __syncthreads();
            for (uint32_t i = tid; i < HT; i += T) { s_kf[i] = make_uint2(EMPTY_KEY, 0u); s_acc[i] = 0; }
            if (tid == 0) { s_ntouched = 0; s_nemit = 0; s_overflow = 0; s_next = next_reg; s_desc = next_desc; }
            continue;
        }
        // ---- finalize + emit (library.cpp:485-517) ------------------------------------------------
        // The staging chunk of this workgroup must have room for every touched candidate, so cells can be
        // written where they are decided; a fresh chunk is taken otherwise (its tail stays unused).
        if (tid == 0) {
            unsigned long long nx = s_chunk_next;
            if (nx + ntouched > s_chunk_end) {
                nx = atomicAdd(a.cell_cursor, (unsigned long long) CELL_CHUNK);
                s_chunk_end = nx + CELL_CHUNK;
                s_chunk_next = nx;
            }
            s_base = nx;
        }
        const uint32_t my_kcnt = a.kseq_len[r];
        const float f_my = (float) (int) my_kcnt;
        float *ms_row = a.MS + (size_t) p * a.G;
        float *cm_row = a.CM + (size_t) a.task_lg[p] * a.N;
        __syncthreads();
        const unsigned long long base = s_base;
        const bool fits = base + ntouched <= a.st_cap;
        for (uint32_t t = tid; t < ntouched; t += T) {
            const uint32_t slot = s_touched[t];
            const uint2 kf = s_kf[slot];
            const uint32_t c = kf.x;
            const unsigned long long acc = s_acc[slot];
            s_kf[slot] = make_uint2(EMPTY_KEY, 0u); s_acc[slot] = 0;        // slot consumed
            if (c >= a.N) { atomicAdd(a.error_count, 1u); continue; }       // a listed slot must hold a column id
            if (c == r) continue;                         // identity cell is zeroed (library.cpp:485-487)
            // score_valid needs perc >= thr or tr_perc >= thr (library.cpp:497-500).  tr_perc = tc / K_c and
            // K_c >= min_kseq, and IEEE division is monotone in the divisor, so tc / min_kseq < thr rules the
            // second test out without fetching K_c.
            const float perc0 = (float) (int) ((acc >> 21) & FIELD_MASK) / f_my;
            const float tr_ub = (float) (int) (acc >> 42) / f_min_kseq;
            if (!(perc0 >= threshold || tr_ub >= threshold)) continue;
            float perc, tr;
            const float score = finalize_cell(acc, my_kcnt, a.kseq_len[c], threshold, perc, tr);
            if (score > 0.0f) {
                const uint32_t i = atomicAdd(&s_nemit, 1u);
                if (fits) {
                    const unsigned long long o = base + i;
                    a.st_score[o] = score; a.st_perc[o] = perc; a.st_tr[o] = tr;
                    a.st_col[o] = c; a.st_first[o] = 0xffffffffu - kf.y;
                    // scores are positive floats: their bit patterns order like the values
                    atomicMax(reinterpret_cast<uint32_t *>(ms_row + a.genome_of[c]), __float_as_uint(score));
                    atomicMax(reinterpret_cast<uint32_t *>(cm_row + c), __float_as_uint(score));
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            const uint32_t nemit = s_nemit;
            a.row_base[p] = (uint32_t) base;
            a.row_cnt[p] = nemit;
            s_chunk_next = base + nemit;
            s_ntouched = 0; s_nemit = 0; s_next = next_reg; s_desc = next_desc;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K-join (HBM tables): direct-addressed by column id, private to the workgroup; every access to the
// tables is an L2-level atomic or an agent-scope (L1-bypassing) load/store, so the workgroup sees
// its own updates without fences.  Tables are all-zero between rows.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ld_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(HBM_THREADS) void k_join_hbm(JoinArgs a) {
    constexpr int JOIN_THREADS = HBM_THREADS;
    __shared__ uint32_t s_ntouched, s_nemit, s_work;
    __shared__ unsigned long long s_base;
    const uint32_t tid = threadIdx.x;
    unsigned long long *t_acc = a.hbm_acc + (size_t) blockIdx.x * a.N;
    uint32_t *t_first = a.hbm_u32 + (size_t) blockIdx.x * 3 * a.N;
    uint32_t *t_touched = t_first + a.N;
    uint32_t *t_emit = t_touched + a.N;
    if (tid == 0) { s_ntouched = 0; s_nemit = 0; }
    const float threshold = 1.0f / (2.0f * (float) a.k);
    __syncthreads();
    for (;;) {
        if (tid == 0) s_work = atomicAdd(a.work_cursor, 1u);
        __syncthreads();
        const uint32_t wi = s_work;
        if (wi >= a.n_work) break;
        const uint32_t p = a.work[wi];
        const uint32_t r = a.task_rows[p];
        const uint32_t e0 = a.seq_off[r], e1 = a.seq_off[r + 1];
        // one wave per range; the whole workgroup strides over the long ones implicitly via wave count
        const uint32_t wave = tid / PDL_WAVE, lane = tid % PDL_WAVE, nwaves = JOIN_THREADS / PDL_WAVE;
        for (uint32_t e = e0 + wave; e < e1; e += nwaves) {
            const uint4 rg = a.ranges[e];
            const uint32_t finv = 0xffffffffu - rg.x;
            for (uint32_t q = lane; q < rg.y; q += PDL_WAVE) {
                const uint2 po = a.post[rg.x + q];
                const uint32_t c = po.x;
                if (ld_agent(&t_first[c]) < finv) {
                    const uint32_t old = atomicMax(&t_first[c], finv);
                    if (old == 0) {                                                // first toucher lists the column
                        const uint32_t idx = atomicAdd(&s_ntouched, 1u);
                        if (idx < a.N) t_touched[idx] = c; else atomicAdd(a.error_count, 1u);
                    }
                }
                const unsigned long long add = (unsigned long long) min(po.y, rg.z) | ((unsigned long long) rg.z << 21) | ((unsigned long long) po.y << 42);
                atomicAdd(&t_acc[c], add);
            }
        }
        __threadfence();      // the touched list was written with plain stores by other waves of this workgroup
        __syncthreads();
        const uint32_t ntouched = min(s_ntouched, a.N);
        const uint32_t my_kcnt = a.kseq_len[r];
        for (uint32_t t = tid; t < ntouched; t += JOIN_THREADS) {
            const uint32_t c = ld_agent(&t_touched[t]);
            if (c == r || c >= a.N) continue;
            float perc, tr;
            const float score = finalize_cell(ld_agent(&t_acc[c]), my_kcnt, a.kseq_len[c], threshold, perc, tr);
            if (score > 0.0f) {
                const uint32_t idx = atomicAdd(&s_nemit, 1u);
                if (idx < a.N) st_agent(&t_emit[idx], c);
            }
        }
        __syncthreads();
        const uint32_t nemit = min(s_nemit, a.N);
        if (tid == 0) {
            const unsigned long long base = atomicAdd(a.cell_cursor, (unsigned long long) nemit);
            s_base = base;
            a.row_base[p] = (uint32_t) base;
            a.row_cnt[p] = nemit;
        }
        __syncthreads();
        const unsigned long long base = s_base;
        if (base + nemit <= a.st_cap) {
            float *ms_row = a.MS + (size_t) p * a.G;
            float *cm_row = a.CM + (size_t) a.task_lg[p] * a.N;
            for (uint32_t i = tid; i < nemit; i += JOIN_THREADS) {
                const uint32_t c = ld_agent(&t_emit[i]);
                float perc, tr;
                const float score = finalize_cell(ld_agent(&t_acc[c]), my_kcnt, a.kseq_len[c], threshold, perc, tr);
                const unsigned long long o = base + i;
                a.st_score[o] = score; a.st_perc[o] = perc; a.st_tr[o] = tr;
                a.st_col[o] = c; a.st_first[o] = 0xffffffffu - ld_agent(&t_first[c]);
                atomicMax(reinterpret_cast<uint32_t *>(ms_row + a.genome_of[c]), __float_as_uint(score));
                atomicMax(reinterpret_cast<uint32_t *>(cm_row + c), __float_as_uint(score));
            }
        }
        __syncthreads();
        for (uint32_t t = tid; t < ntouched; t += JOIN_THREADS) {
            const uint32_t c = ld_agent(&t_touched[t]);
            if (c < a.N) { st_agent(&t_acc[c], 0ull); st_agent(&t_first[c], 0u); }
        }
        if (tid == 0) { s_ntouched = 0; s_nemit = 0; }
        __threadfence();      // the zeroing stores must have landed before the next row's atomics
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// K-order: cells of task position p move from the staging area (arrival order) to
// fin_off[p] + rank, rank = position in the reference's emission order.  The reference visits columns
// chunk by chunk of 2048 ids ("<= max_allowed_sequence": chunk 0 is ids 0..2048, library.cpp:456-464),
// inside a chunk range by range, inside a range by ascending gene; a cell is emitted where its column
// is first touched.  Key = (chunk(col), first group start, col); canonical mode: key = col.
// Rank by counting over LDS tiles of the row's keys.
// ------------------------------------------------------------------------------------------------
constexpr int ORDER_THREADS = 256;
constexpr int ORDER_TILE = 2048;

struct OrderArgs {
    const uint32_t *row_base, *row_cnt, *fin_off, *task_rows;
    const float *st_score, *st_perc, *st_tr;
    const uint32_t *st_col, *st_first;
    float *c_score, *c_perc, *c_tr;
    int32_t *c_row, *c_col;
    uint32_t n_rows;
    uint32_t canonical;
};

__device__ __forceinline__ unsigned long long order_key_hi(uint32_t col, uint32_t first, uint32_t canonical) {
    if (canonical) return 0ull;
    const uint32_t chunk = col == 0 ? 0u : (col - 1) >> 11;
    return ((unsigned long long) chunk << 32) | first;
}

__global__ __launch_bounds__(ORDER_THREADS) void k_order_rows(OrderArgs a) {
    __shared__ unsigned long long s_hi[ORDER_TILE];
    __shared__ uint32_t s_col[ORDER_TILE];
    const uint32_t p = blockIdx.x;
    const uint32_t cnt = a.row_cnt[p];
    if (cnt == 0) return;
    const uint32_t base = a.row_base[p];
    const uint32_t out0 = a.fin_off[p];
    const uint32_t row = a.task_rows[p];
    for (uint32_t i0 = 0; i0 < cnt; i0 += ORDER_THREADS) {
        const uint32_t i = i0 + threadIdx.x;
        const bool live = i < cnt;
        uint32_t col = 0, first = 0;
        unsigned long long hi = 0;
        if (live) { col = a.st_col[base + i]; first = a.st_first[base + i]; hi = order_key_hi(col, first, a.canonical); }
        uint32_t rank = 0;
        for (uint32_t j0 = 0; j0 < cnt; j0 += ORDER_TILE) {
            const uint32_t tn = min((uint32_t) ORDER_TILE, cnt - j0);
            __syncthreads();
            for (uint32_t j = threadIdx.x; j < tn; j += ORDER_THREADS) {
                const uint32_t cj = a.st_col[base + j0 + j];
                s_col[j] = cj;
                s_hi[j] = order_key_hi(cj, a.st_first[base + j0 + j], a.canonical);
            }
            __syncthreads();
            if (live) {
                for (uint32_t j = 0; j < tn; j++) {
                    const unsigned long long hj = s_hi[j];
                    rank += (hj < hi) || (hj == hi && s_col[j] < col);
                }
            }
        }
        if (live) {
            const uint32_t o = out0 + rank;
            a.c_score[o] = a.st_score[base + i];
            a.c_perc[o] = a.st_perc[base + i];
            a.c_tr[o] = a.st_tr[base + i];
            a.c_row[o] = (int32_t) row;
            a.c_col[o] = (int32_t) col;
        }
    }
}

// work-item descriptors of the LDS join, in processing order (currently task order)
__global__ __launch_bounds__(256) void k_row_desc(const uint32_t *__restrict__ task_rows, const uint32_t *__restrict__ seq_off,
                                                  uint32_t n_rows, uint4 *__restrict__ desc) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_rows) return;
    const uint32_t r = task_rows[p];
    const uint32_t e0 = seq_off[r];
    desc[p] = make_uint4(p, r, e0, seq_off[r + 1] - e0);
}

struct RowCntFlag {
    const uint32_t *row_cnt;
    __device__ uint32_t operator()(uint64_t p) const { return row_cnt[p]; }
};
struct FinOffApply {
    uint32_t *fin_off;
    __device__ void operator()(uint64_t p, uint32_t, uint32_t prefix) const { fin_off[p] = prefix; }
};
__global__ void k_gather_u32(const uint32_t *src, const uint32_t *idx, uint32_t n, uint32_t *dst) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// ------------------------------------------------------------------------------------------------
void pdl_run_score_all(pdl_ctx *c) {
    hipStream_t st = c->stream;
    const uint32_t N = c->N, G = c->G;
    if (c->max_kseq >= (1ull << 20))
        PDL_FAIL(PDL_ERR_UNSUPPORTED, "a gene with %llu k-mers exceeds the 2^20 limit of the packed accumulators", (unsigned long long) c->max_kseq);
    ev_begin(c, EV_SCORE_TOTAL);

    // ---- task layout: shard genomes ascending, rows of a genome ascending (library.cpp:244) ----
    if (!c->shard_set) { c->shard.resize(G); for (uint32_t g = 0; g < G; g++) c->shard[g] = g; }
    const uint32_t S = (uint32_t) c->shard.size();
    c->h_local_genome.assign(G, -1);
    c->h_task_row_off.assign(S + 1, 0);
    std::vector<uint32_t> h_rows, h_lg;
    for (uint32_t i = 0; i < S; i++) {
        const uint32_t g = c->shard[i];
        c->h_local_genome[g] = (int32_t) i;
        c->h_task_row_off[i] = (uint32_t) h_rows.size();
        for (uint32_t j = c->h_genome_row_off[g]; j < c->h_genome_row_off[g + 1]; j++) { h_rows.push_back(c->h_genome_rows[j]); h_lg.push_back(i); }
    }
    c->h_task_row_off[S] = (uint32_t) h_rows.size();
    const uint32_t n_rows = (uint32_t) h_rows.size();
    c->n_task_rows = n_rows;
    c->h_cell_off.assign(S + 1, 0);
    c->Z = 0;
    c->tm.emitted_cells = 0; c->tm.scored_rows = n_rows; c->tm.overflow_rows = 0; c->tm.join_launches = 0;
    c->tm.scored_lookups = 0;
    for (uint32_t i = 0; i < S; i++) c->tm.scored_lookups += c->h_genome_cost[c->shard[i]];
    if (n_rows == 0) { c->scored = true; ev_end(c, EV_SCORE_TOTAL); return; }

    c->task_rows.alloc((size_t) n_rows * 4); c->task_lg.alloc((size_t) n_rows * 4);
    PDL_HIP(hipMemcpyAsync(c->task_rows.p, h_rows.data(), (size_t) n_rows * 4, hipMemcpyHostToDevice, st));
    PDL_HIP(hipMemcpyAsync(c->task_lg.p, h_lg.data(), (size_t) n_rows * 4, hipMemcpyHostToDevice, st));
    PDL_HIP(hipStreamSynchronize(st));   // h_rows / h_lg go out of use below

    c->MS.alloc((size_t) n_rows * G * sizeof(float));
    c->CM.alloc((size_t) S * N * sizeof(float));
    c->row_base.alloc((size_t) n_rows * 4); c->row_cnt.alloc((size_t) n_rows * 4); c->fin_off.alloc(((size_t) n_rows + 1) * 4);
    c->join_ctr.alloc(64);
    c->row_desc.alloc((size_t) n_rows * sizeof(uint4));
    hipLaunchKernelGGL(k_row_desc, dim3((n_rows + 255) / 256), dim3(256), 0, st, c->task_rows.as<uint32_t>(), c->seq_off.as<uint32_t>(),
                       n_rows, c->row_desc.as<uint4>());
    c->overflow_rows.alloc((size_t) n_rows * 4);

    int cus = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) cus = prop.multiProcessorCount; }

    // staging capacity: emitted cells are, in practice, the homologous pairs (about one per genome and row);
    // if the guess is short the pass is repeated once with the exact size.
    // (every workgroup reserves staging in chunks of CELL_CHUNK cells, so allow one partly used chunk per
    // workgroup and per oversized row on top of the cell estimate)
    int geometry = 0;
    if (const char *e = getenv("PDL_JOIN_TABLE_BITS")) geometry = atoi(e) == 9 ? 1 : 0;
    const uint32_t grid = std::min<uint32_t>(n_rows, (uint32_t) cus * (geometry == 1 ? 4 : 1));
    const unsigned long long slack = 2ull * grid * CELL_CHUNK + 4ull * cus * CELL_CHUNK;
    unsigned long long cap = std::max<unsigned long long>(1ull << 20, (unsigned long long) n_rows * (G + 16ull));
    cap = std::min<unsigned long long>(cap, std::max<unsigned long long>(c->P, 1ull)) + slack;
    for (int attempt = 0; attempt < 2; attempt++) {
        if (cap >= 0xffffffffull) PDL_FAIL(PDL_ERR_UNSUPPORTED, "more than 2^32 emitted cells on one device: shard the genomes over more devices");
        c->st_cap = cap;
        c->st_score.alloc(cap * 4); c->st_perc.alloc(cap * 4); c->st_tr.alloc(cap * 4); c->st_col.alloc(cap * 4); c->st_first.alloc(cap * 4);
        PDL_HIP(hipMemsetAsync(c->MS.p, 0, (size_t) n_rows * G * sizeof(float), st));
        PDL_HIP(hipMemsetAsync(c->CM.p, 0, (size_t) S * N * sizeof(float), st));
        PDL_HIP(hipMemsetAsync(c->join_ctr.p, 0, 64, st));

        JoinArgs a{};
        a.post = c->post.as<uint2>(); a.ranges = c->ranges.as<uint4>(); a.seq_off = c->seq_off.as<uint32_t>();
        a.kseq_len = c->kseq_len.as<uint32_t>(); a.genome_of = c->d_gen;
        a.task_rows = c->task_rows.as<uint32_t>(); a.task_lg = c->task_lg.as<uint32_t>();
        a.work = nullptr; a.desc = c->row_desc.as<uint4>(); a.n_work = n_rows; a.N = N; a.G = G; a.k = c->rp.k;
        a.min_kseq = (uint32_t) std::max<uint64_t>(c->min_kseq, 1); a.canonical = (c->flags & PDL_FLAG_CANONICAL_ORDER) ? 1u : 0u;
        a.MS = c->MS.as<float>(); a.CM = c->CM.as<float>();
        a.row_base = c->row_base.as<uint32_t>(); a.row_cnt = c->row_cnt.as<uint32_t>();
        a.st_score = c->st_score.as<float>(); a.st_perc = c->st_perc.as<float>(); a.st_tr = c->st_tr.as<float>();
        a.st_col = c->st_col.as<uint32_t>(); a.st_first = c->st_first.as<uint32_t>(); a.st_cap = cap;
        uint32_t *ctr32 = c->join_ctr.as<uint32_t>();
        a.work_cursor = ctr32 + 0; a.overflow_count = ctr32 + 1; a.error_count = ctr32 + 6;
        a.cell_cursor = reinterpret_cast<unsigned long long *>(ctr32 + 4);
        a.overflow_rows = c->overflow_rows.as<uint32_t>();

        ev_begin(c, EV_JOIN);
        if (geometry == 1) hipLaunchKernelGGL((k_join_lds<9, 64>), dim3(grid), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_join_lds<13, 1024>), dim3(grid), dim3(1024), 0, st, a);
        PDL_HIP(hipGetLastError());
        ev_end(c, EV_JOIN);
        c->tm.join_launches++;

        uint32_t h_ctr[8];
        PDL_HIP(hipMemcpyAsync(h_ctr, c->join_ctr.p, sizeof(h_ctr), hipMemcpyDeviceToHost, st));
        PDL_HIP(hipStreamSynchronize(st));
        const uint32_t n_ovf = h_ctr[1];
        c->tm.overflow_rows = n_ovf;
        c->ev[EV_JOIN_OVF].used = false;
        if (n_ovf) {
            // rows whose candidate set exceeded the LDS table: HBM tables, one set per workgroup
            const uint32_t wgs = std::min<uint32_t>(n_ovf, (uint32_t) cus);
            c->glb_table.alloc((size_t) wgs * N * (sizeof(uint64_t) + 3 * sizeof(uint32_t)));
            PDL_HIP(hipMemsetAsync(c->glb_table.p, 0, (size_t) wgs * N * (sizeof(uint64_t) + 3 * sizeof(uint32_t)), st));
            a.hbm_acc = c->glb_table.as<unsigned long long>();
            a.hbm_u32 = reinterpret_cast<uint32_t *>(a.hbm_acc + (size_t) wgs * N);
            a.work = c->overflow_rows.as<uint32_t>(); a.n_work = n_ovf;
            a.work_cursor = ctr32 + 2;
            ev_begin(c, EV_JOIN_OVF);
            hipLaunchKernelGGL(k_join_hbm, dim3(wgs), dim3(HBM_THREADS), 0, st, a);
            PDL_HIP(hipGetLastError());
            ev_end(c, EV_JOIN_OVF);
            c->tm.join_launches++;
            PDL_HIP(hipMemcpyAsync(h_ctr, c->join_ctr.p, sizeof(h_ctr), hipMemcpyDeviceToHost, st));
            PDL_HIP(hipStreamSynchronize(st));
        }
        if (h_ctr[6]) PDL_FAIL(PDL_ERR_DEVICE, "join: %u internal consistency violations", h_ctr[6]);
        unsigned long long z;                    // staging cells reserved (>= cells emitted: chunk tails are unused)
        memcpy(&z, &h_ctr[4], sizeof(z));
        if (z <= cap) break;
        if (attempt == 1) PDL_FAIL(PDL_ERR_DEVICE, "staging overflow persisted (%llu cells > %llu)", z, cap);
        cap = z + slack;      // what was asked for plus chunk slack, second and last attempt
    }
    // ---- order ------------------------------------------------------------------------------------
    ev_begin(c, EV_ORDER);
    uint64_t *d_scal = c->scalars.as<uint64_t>();
    scan_and_apply(c, n_rows, RowCntFlag{c->row_cnt.as<uint32_t>()}, FinOffApply{c->fin_off.as<uint32_t>()}, d_scal + 6);
    uint64_t zsum = 0;
    PDL_HIP(hipMemcpyAsync(&zsum, d_scal + 6, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    PDL_HIP(hipStreamSynchronize(st));
    c->Z = zsum;
    c->tm.emitted_cells = c->Z;
    const uint32_t z32 = (uint32_t) c->Z;
    PDL_HIP(hipMemcpyAsync(c->fin_off.as<uint32_t>() + n_rows, &z32, 4, hipMemcpyHostToDevice, st));
    const size_t zc = c->Z ? (size_t) c->Z : 1;
    c->c_score.alloc(zc * 4); c->c_perc.alloc(zc * 4); c->c_tr.alloc(zc * 4); c->c_row.alloc(zc * 4); c->c_col.alloc(zc * 4);
    OrderArgs o{};
    o.row_base = c->row_base.as<uint32_t>(); o.row_cnt = c->row_cnt.as<uint32_t>(); o.fin_off = c->fin_off.as<uint32_t>();
    o.task_rows = c->task_rows.as<uint32_t>();
    o.st_score = c->st_score.as<float>(); o.st_perc = c->st_perc.as<float>(); o.st_tr = c->st_tr.as<float>();
    o.st_col = c->st_col.as<uint32_t>(); o.st_first = c->st_first.as<uint32_t>();
    o.c_score = c->c_score.as<float>(); o.c_perc = c->c_perc.as<float>(); o.c_tr = c->c_tr.as<float>();
    o.c_row = c->c_row.as<int32_t>(); o.c_col = c->c_col.as<int32_t>();
    o.n_rows = n_rows; o.canonical = (c->flags & PDL_FLAG_CANONICAL_ORDER) ? 1u : 0u;
    hipLaunchKernelGGL(k_order_rows, dim3(n_rows), dim3(ORDER_THREADS), 0, st, o);
    PDL_HIP(hipGetLastError());
    ev_end(c, EV_ORDER);

    // first cell of every shard genome = fin_off at its first task row
    {
        std::vector<uint32_t> h_fin(S + 1);
        c->scratch.alloc((size_t) (S + 1) * 8);
        uint32_t *d_idx = c->scratch.as<uint32_t>();
        uint32_t *d_out = d_idx + (S + 1);
        PDL_HIP(hipMemcpyAsync(d_idx, c->h_task_row_off.data(), (size_t) (S + 1) * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_gather_u32, dim3((S + 1 + 255) / 256), dim3(256), 0, st, c->fin_off.as<uint32_t>(), d_idx, S + 1, d_out);
        PDL_HIP(hipMemcpyAsync(h_fin.data(), d_out, (size_t) (S + 1) * 4, hipMemcpyDeviceToHost, st));
        ev_end(c, EV_SCORE_TOTAL);
        PDL_HIP(hipStreamSynchronize(st));
        for (uint32_t i = 0; i <= S; i++) c->h_cell_off[i] = h_fin[i];
    }
    c->tm.join_ms = ev_ms(c, EV_JOIN);
    c->tm.join_overflow_ms = ev_ms(c, EV_JOIN_OVF);
    c->tm.order_ms = ev_ms(c, EV_ORDER);
    c->tm.score_total_ms = ev_ms(c, EV_SCORE_TOTAL);
    c->scored = true;
}
