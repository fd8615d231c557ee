// pdl_dict.hip — the dictionary stage on the device: everything preprocessSequences does
// (ig/native/library.cpp:189-371), one kernel (or kernel group) per reference function:
//
//   K-hist    k_hist              alphabet (which letters occur)             library.cpp:216-228
//   (host)    rank_init_host      rank table, B^(k-1), overflow -> hashing   library.cpp:88-132
//   K-len     k_kseq_len          kseq_lengths + k-mer stream offsets        library.cpp:250-262
//   K-rank    k_rank / k_rank_hash   per-gene k-mer ranks                    library.cpp:75-86,134-150
//   K-sort    pdl_sort_pairs      stable LSD radix sort by rank              library.cpp:172-187,270-278
//   K-rle     RecHead/RecScatter scan (records built in the apply)   dedup -> (rank,gene,count)   library.cpp:280-287
//   K-groups  k_fold_last_record, k_group_waves (group extents from head bits, per tile)  library.cpp:297-335
//   K-ranges  k_group_waves (range tuples), sort by gene, k_gather_ranges (+ per-gene cost), k_seq_offsets   library.cpp:312-327
//   K-cost    k_genome_cost       per-genome and total lookups               library.cpp:337-350,535-538
//
// HBM layout after this stage (what the join reads):
//   post   uint2[U]   {gene, count}       rank-group major, ascending gene inside a group
//   ranges uint4[U']  {first posting, postings, own count, group size}   gene major (U' = records in groups >= 2);
//                     without a shard a gene's range holds only the postings after its own record (genes above it)
//   seq_off u32[N+1]  range list of each gene;   kseq_len u32[N];   cost u64[N]
#include "pdl_common.h"
#include "pdl_scan.h"
#include "pdl_sort.h"

#include <algorithm>
#include <cstring>

#define RABIN_MODULO 18446744073709551557ULL   /* 2^64 - 59, library.cpp:19 */


__global__ __launch_bounds__(256) void k_zero_u64(uint64_t *p, size_t n) {
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) p[i] = 0;
}
// ------------------------------------------------------------------------------------------------
// K-hist: which byte values occur among the residues (library.cpp:216-228).  The reference counts every letter, but the
// counters are only ever asked "> 0?" (rank_init, library.cpp:96-99): the alphabet is a 256-entry presence table.  That
// needs no atomic anywhere: a lane stores a 1 into the LDS word of each byte it sees (lanes that meet the same letter store
// the same value to the same address; ~20 distinct letters fall in as many banks), and a workgroup stores a 1 into the
// global counter of every letter it has seen (same value from every workgroup).  16-byte coalesced loads, four in flight
// per lane: the pass runs at the speed the residues stream in (the counting version — LDS atomics on ~20 hot addresses —
// took 41 us for 17.6 MB).
// ------------------------------------------------------------------------------------------------
constexpr int HIST_THREADS = 256;
__global__ __launch_bounds__(HIST_THREADS) void k_hist(const uint8_t *__restrict__ res, uint64_t n, unsigned long long *__restrict__ hist) {
    __shared__ uint32_t s_seen[256];
    s_seen[threadIdx.x] = 0;
    pdl_sync();
    const uint64_t n16 = n / 16;
    const uint4 *res16 = reinterpret_cast<const uint4 *>(res);
    const uint64_t stride = (uint64_t) gridDim.x * HIST_THREADS;
    auto mark = [&](const uint4 &v) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            s_seen[w[j] & 0xff] = 1u; s_seen[(w[j] >> 8) & 0xff] = 1u; s_seen[(w[j] >> 16) & 0xff] = 1u; s_seen[w[j] >> 24] = 1u;
        }
    };
    uint64_t i = (uint64_t) blockIdx.x * HIST_THREADS + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 v0 = res16[i], v1 = res16[i + stride], v2 = res16[i + 2 * stride], v3 = res16[i + 3 * stride];
        mark(v0); mark(v1); mark(v2); mark(v3);
    }
    for (; i < n16; i += stride) mark(res16[i]);
    if (blockIdx.x == 0)    // tail bytes
        for (uint64_t t = n16 * 16 + threadIdx.x; t < n; t += HIST_THREADS) s_seen[res[t]] = 1u;
    pdl_sync();
    if (s_seen[threadIdx.x]) hist[threadIdx.x] = 1ull;
}

// library.cpp:88-132, literally (64-bit wraparound included); adds rank_bits for the device sort.
static void rank_init_host(RankParams &rp, const uint64_t counters[256], int kvalue) {
    memset(&rp, 0, sizeof(rp));
    rp.k = (uint32_t) kvalue;
    int rank = 0;
    for (int i = 0; i < 256; i++)
        if (counters[i] > 0) rp.rank_values[i] = (uint8_t) rank++;
    const uint8_t rank_base = (uint8_t) rank;
    rp.base = rank_base;
    uint64_t last_multiplier = 1;
    bool has_overflow = false;
    int tmp_kvalue = kvalue - 1;
    while (tmp_kvalue--) {
        uint64_t ovflw_test = last_multiplier;
        last_multiplier *= rank_base;
        if (has_overflow) {
            last_multiplier %= RABIN_MODULO;
        } else if ((ovflw_test > last_multiplier) || (ovflw_test * rank_base > last_multiplier * rank_base)) {
            has_overflow = true;
            last_multiplier = ((ovflw_test % RABIN_MODULO) * rank_base) % RABIN_MODULO;
        }
    }
    rp.last_multiplier = last_multiplier;
    rp.hash_fallback = has_overflow ? 1u : 0u;
    rp.key_bits = 64;
    if (has_overflow) {
        rp.rank_bits = 64;
    } else {
        uint64_t rank_tmp = last_multiplier * rank_base;   // B^k, fits (the loop above looked one step ahead)
        rp.rank_bits = rank_tmp ? bit_length64(rank_tmp - 1) : 1;
        if (rp.rank_bits == 0) rp.rank_bits = 1;
        // did B^k fit?  (exact arithmetic: the reference's test compares wrapped products and lets some wraps through)
        unsigned __int128 exact = 1;
        bool fits = true;
        for (int i = 0; i < kvalue && fits; i++) { exact *= rank_base; fits = (exact >> 64) == 0; }
        if (fits) rp.key_bits = rp.rank_bits;
    }
}

// ------------------------------------------------------------------------------------------------
// K-len: kseq_lengths (library.cpp:250-262).  The exclusive scan of these is the offset of each
// gene in the k-mer stream (gene order = the order the reference emplace_back()s them, :255-258).
// ------------------------------------------------------------------------------------------------
struct KseqFlag {
    const uint64_t *off; uint32_t k;
    __device__ uint32_t operator()(uint64_t i) const {
        const uint64_t b = off[i], e = off[i + 1];
        const uint64_t len = e >= b ? e - b : 0;        // (descending offsets are reported by KseqApply)
        return len >= k ? (uint32_t) (len - k + 1) : 0u;
    }
};
struct KseqApply {
    uint32_t *kseq_len; uint64_t *kmer_off;   // kmer_off as u64 for the API; values < 2^32 (checked on the host)
    unsigned long long *cost;                 // total_visited starts at zero (spares a fill)
    const uint64_t *off; uint64_t n_res; unsigned long long *bad;    // *bad |= 1 when the offsets are not an ascending cover of [0, n_res]
    __device__ void operator()(uint64_t i, uint32_t f, uint32_t prefix) const {
        kseq_len[i] = f;
        kmer_off[i] = prefix;
        cost[i] = 0;
        if (off[i + 1] < off[i] || off[i + 1] > n_res) atomicOr(bad, 1ull);     // (never taken on valid input)
    }
};

// ------------------------------------------------------------------------------------------------
// K-rank (exact): one thread per k-mer of the stream.  Because B^k < 2^64 here, the reference's
// rolling update (library.cpp:75-79) equals the direct base-B polynomial of the k residues, so
// k-mers are independent.  A workgroup covers RANK_TILE consecutive stream slots; two lanes
// bracket the genes of the tile with a binary search, every lane then searches only that bracket.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t RANK_SPAN = 128;     // genes whose offsets a tile stages in LDS (more: global lookups)
constexpr int RANK_THREADS = 256;
constexpr int RANK_ITEMS = 4;
constexpr int RANK_TILE = RANK_THREADS * RANK_ITEMS;

__device__ __forceinline__ uint32_t upper_bound_u64(const uint64_t *a, uint32_t lo, uint32_t hi, uint64_t v) {
    // first index in [lo,hi) with a[idx] > v
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

struct __attribute__((packed, aligned(1))) UnalignedU64 { uint64_t v; };
// eight residues from byte position p (any alignment); near the end of the buffer byte by byte
__device__ __forceinline__ uint64_t load_residues8(const uint8_t *__restrict__ res, uint64_t p, uint64_t n_res) {
    if (p + 8 <= n_res) return reinterpret_cast<const UnalignedU64 *>(res + p)->v;
    uint64_t w = 0;
    for (uint32_t b = 0; b < 8 && p + b < n_res; b++) w |= (uint64_t) res[p + b] << (8 * b);
    return w;
}

// Interval histogram of the multi-GPU build: the rank space is cut where the top DIST_BIN_BITS bits of a rank change.
constexpr uint32_t DIST_BIN_BITS = 12, DIST_BINS = 1u << DIST_BIN_BITS;

// MODE 0: keys[q] = rank, vals[q] = gene for every slot q of the k-mer stream (one workgroup per tile).
// MODE 1: the same, and counts the ranks by their top bits into bins[DIST_BINS] (persistent workgroups over the tiles,
//         LDS histogram, one global atomic per non-empty bin and workgroup): the k-mer count of every rank interval,
//         from which pdl_dist_preprocess_begin derives the same cuts on every GPU.
template <class KeyT, int MODE>
__global__ __launch_bounds__(RANK_THREADS) void k_rank(const uint8_t *__restrict__ res, const uint64_t *__restrict__ off,
                                                       const uint64_t *__restrict__ kmer_off, uint32_t n_seq, uint64_t m, uint64_t n_res,
                                                       RankParams rp, KeyT *__restrict__ keys, uint32_t *__restrict__ vals,
                                                       uint32_t bin_shift, uint32_t *__restrict__ bins) {
    __shared__ uint8_t s_rv[256];
    __shared__ uint32_t s_lo, s_hi;
    __shared__ uint64_t s_koff[RANK_SPAN + 1], s_off[RANK_SPAN];    // k-mer and residue offsets of the genes this tile touches
    __shared__ uint32_t s_bins[MODE == 1 ? DIST_BINS : 1];
    for (int i = threadIdx.x; i < 256; i += RANK_THREADS) s_rv[i] = rp.rank_values[i];
    if constexpr (MODE == 1) for (uint32_t i = threadIdx.x; i < DIST_BINS; i += RANK_THREADS) s_bins[i] = 0;
    const uint64_t tiles = (m + RANK_TILE - 1) / RANK_TILE;
  for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {      // MODE 0: grid = tiles, one trip
    pdl_sync();                                            // (the staged boundaries of the previous tile are done with)
    const uint64_t q0 = tile * RANK_TILE;
    const uint64_t q_last = min(q0 + RANK_TILE, m) - 1;
    if (threadIdx.x == 0) s_lo = upper_bound_u64(kmer_off, 0, n_seq + 1, q0) - 1;
    if (threadIdx.x == 64) s_hi = upper_bound_u64(kmer_off, 0, n_seq + 1, q_last) - 1;
    pdl_sync();
    const uint32_t lo = s_lo, hi = s_hi;
    const uint32_t span = hi - lo + 1;                          // genes under this tile (uniform)
    const bool staged = span <= RANK_SPAN;
    if (staged) {
        for (uint32_t i = threadIdx.x; i <= span; i += RANK_THREADS) s_koff[i] = kmer_off[lo + i];
        for (uint32_t i = threadIdx.x; i < span; i += RANK_THREADS) s_off[i] = off[lo + i];
    }
    pdl_sync();
    const uint32_t k = rp.k;
    const KeyT base = (KeyT) rp.base;                           // the polynomial fits KeyT (checked on the host): KeyT arithmetic
    // Phase 1: gene and residue position of the lane's four k-mers.  Phase 2: their residues, eight bytes per load, the
    // loads of the four k-mers in flight together (a byte loop with a wait per byte serialises 4 x k round trips).
    uint32_t sq[RANK_ITEMS];
    uint64_t pos[RANK_ITEMS];
#pragma unroll
    for (int j = 0; j < RANK_ITEMS; j++) {
        const uint64_t qj = q0 + (uint64_t) j * RANK_THREADS + threadIdx.x;
        const uint64_t q = qj < m ? qj : m - 1;
        if (staged) {                                           // (uniform) boundaries from LDS: no chain of global loads
            uint32_t a = 0;                                     // last i in [0, span) with s_koff[i] <= q
            if (span <= 8) {                                    // (uniform, the usual case) count the gene starts at or below q:
#pragma unroll                                                  //  independent broadcast reads instead of a dependent search
                for (uint32_t i = 1; i < 8; i++) a += (uint32_t) (i < span && s_koff[i < span ? i : 0] <= q);
            } else {
                uint32_t b = span;
                while (a < b) { const uint32_t mid = (a + b) >> 1; if (s_koff[mid + 1] <= q) a = mid + 1; else b = mid; }
            }
            sq[j] = lo + a;
            pos[j] = s_off[a] + (q - s_koff[a]);
        } else {
            sq[j] = upper_bound_u64(kmer_off, lo, hi + 1, q) - 1;   // kmer_off[s] <= q < kmer_off[s+1]
            pos[j] = off[sq[j]] + (q - kmer_off[sq[j]]);
        }
    }
    KeyT r[RANK_ITEMS];
#pragma unroll
    for (int j = 0; j < RANK_ITEMS; j++) r[j] = 0;
    for (uint32_t c0 = 0; c0 < k; c0 += 8) {
        uint64_t w[RANK_ITEMS];
        bool safe = true;
#pragma unroll
        for (int j = 0; j < RANK_ITEMS; j++) safe = safe && pos[j] + c0 + 8 <= n_res;
        if (__all(safe)) {                                      // (every tile but the last): four unaligned 8-byte loads, no branches between them
#pragma unroll
            for (int j = 0; j < RANK_ITEMS; j++) w[j] = reinterpret_cast<const UnalignedU64 *>(res + pos[j] + c0)->v;
        } else {
#pragma unroll
            for (int j = 0; j < RANK_ITEMS; j++) w[j] = load_residues8(res, pos[j] + c0, n_res);
        }
        const uint32_t nb = min(8u, k - c0);
        for (uint32_t b = 0; b < nb; b++) {
#pragma unroll
            for (int j = 0; j < RANK_ITEMS; j++) r[j] = r[j] * base + s_rv[(uint32_t) (w[j] >> (8 * b)) & 0xffu];
        }
    }
#pragma unroll
    for (int j = 0; j < RANK_ITEMS; j++) {
        const uint64_t q = q0 + (uint64_t) j * RANK_THREADS + threadIdx.x;
        if (q < m) {
            if constexpr (MODE == 1) atomicAdd(&s_bins[(uint32_t) (r[j] >> bin_shift)], 1u);
            keys[q] = r[j]; vals[q] = sq[j];
        }
    }
  }
    if constexpr (MODE == 1) {
        pdl_sync();
        for (uint32_t i = threadIdx.x; i < DIST_BINS; i += RANK_THREADS) { const uint32_t v = s_bins[i]; if (v) atomicAdd(&bins[i], v); }
    }
}

// K-rank (hash fallback, library.cpp:81-86): the 64-bit wrap of the first line makes the value
// depend on the whole prefix of the gene, so genes are ranked sequentially, one lane per gene.
__device__ __forceinline__ uint64_t update_rank_hash_dev(uint64_t current, uint64_t vnext, uint64_t vpop, uint64_t lm, uint64_t base) {
    uint64_t wrapped = current + RABIN_MODULO - vpop * lm;       // evaluated in 64 bits, wraps
    // (wrapped * base + vnext) mod (2^64 - 59) with a 128-bit intermediate
    uint64_t lo = wrapped * base;
    uint64_t hi = __umul64hi(wrapped, base);
    uint64_t lo2 = lo + vnext;
    hi += (lo2 < lo);
    // hi * 2^64 + lo2  ==  hi * 59 + lo2   (mod 2^64 - 59); hi < 256 so hi*59 is tiny
    uint64_t t = hi * 59ull;
    uint64_t r = lo2 + t;
    if (r < lo2) r += 59ull;                                     // one more 2^64 folded (cannot carry again)
    if (r >= RABIN_MODULO) r -= RABIN_MODULO;
    return r;
}

// MODE 1 also counts the ranks by their top DIST_BIN_BITS bits (see k_rank).
template <int MODE>
__global__ __launch_bounds__(256) void k_rank_hash(const uint8_t *__restrict__ res, const uint64_t *__restrict__ off,
                                                   const uint64_t *__restrict__ kmer_off, uint32_t n_seq, RankParams rp,
                                                   uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, uint32_t *__restrict__ bins) {
    __shared__ uint8_t s_rv[256];
    for (int i = threadIdx.x; i < 256; i += 256) s_rv[i] = rp.rank_values[i];
    pdl_sync();
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_seq) return;
    const uint64_t b = off[s], e = off[s + 1];
    const uint32_t k = rp.k;
    if (e - b < k) return;
    const uint8_t *p = res + b;
    const uint64_t len = e - b;
    uint64_t q = kmer_off[s];
    uint64_t rank = 0;
    const uint64_t v0 = s_rv[0];
    for (uint32_t i = 0; i < k; i++) rank = update_rank_hash_dev(rank, s_rv[p[i]], v0, rp.last_multiplier, rp.base);
    if constexpr (MODE == 1) atomicAdd(&bins[rank >> (64 - DIST_BIN_BITS)], 1u);
    keys[q] = rank; vals[q] = s; q++;
    for (uint64_t i = k; i < len; i++) {
        rank = update_rank_hash_dev(rank, s_rv[p[i]], s_rv[p[i - k]], rp.last_multiplier, rp.base);
        if constexpr (MODE == 1) atomicAdd(&bins[rank >> (64 - DIST_BIN_BITS)], 1u);
        keys[q] = rank; vals[q] = s; q++;
    }
}

// ------------------------------------------------------------------------------------------------
// K-rle: run-length dedup of the sorted stream (library.cpp:280-287).  A record head is a position
// whose (rank, gene) differs from its predecessor; recpos[u] = position of the u-th head, the run
// length to the next head is the k-mer's multiplicity in the gene.
// ------------------------------------------------------------------------------------------------
template <class KeyT> struct RecHead {
    const KeyT *keys; const uint32_t *vals;
    __device__ uint32_t operator()(uint64_t q) const {     // straight-line (no short-circuit): the loads of a thread's items overlap
        const uint64_t qp = q ? q - 1 : 0;
        const KeyT k0 = keys[q], k1 = keys[qp];
        const uint32_t v0 = vals[q], v1 = vals[qp];
        return (uint32_t) (q == 0) | (uint32_t) (k0 != k1) | (uint32_t) (v0 != v1);
    }
};
// The apply side also builds the record: post[u] = {gene, run length | HEAD_BIT when record u opens a rank-group}
// (its rank differs from the element just before it, which belongs to the previous record).  Runs are short (a k-mer
// repeated inside one gene), so the head walks its own run.
template <class KeyT> struct RecScatter {
    const KeyT *keys; const uint32_t *vals; uint64_t m;
    uint32_t *recpos; uint2 *post;
    struct Loaded { uint32_t val, run; uint8_t head; };
    __device__ Loaded load(uint64_t q, uint32_t) const {    // straight-line for the common run of one; the rare longer run loops
        const uint64_t qp = q ? q - 1 : 0, qn = q + 1 < m ? q + 1 : q;
        const KeyT key = keys[q], kprev = keys[qp], knext = keys[qn];
        const uint32_t val = vals[q], vnext = vals[qn];
        const bool head = q == 0 || kprev != key;
        uint64_t j = q + 1;
        if (j < m && knext == key && vnext == val) {
            j++;
            while (j < m && keys[j] == key && vals[j] == val) j++;
        }
        return Loaded{val, (uint32_t) (j - q), (uint8_t) (head ? 1 : 0)};
    }
    __device__ void store(uint64_t q, uint32_t f, uint32_t prefix, const Loaded &v) const {
        if (!f) return;
        recpos[prefix] = (uint32_t) q;
        post[prefix] = make_uint2(v.val, v.run | ((uint32_t) v.head << 31));
    }
};

// Interval selection of the multi-GPU build: the k-mers whose rank falls into this GPU's bins, in stream order.
template <class KeyT> struct SelFlag {
    const KeyT *keys; uint32_t shift, b_lo, b_hi;
    __device__ uint32_t operator()(uint64_t q) const { const uint32_t b = (uint32_t) (keys[q] >> shift); return (uint32_t) (b >= b_lo) & (uint32_t) (b < b_hi); }
};
template <class KeyT> struct SelApply {
    const KeyT *keys; const uint32_t *vals; KeyT *keys_out; uint32_t *vals_out;
    struct Loaded { KeyT key; uint32_t val; };
    __device__ Loaded load(uint64_t q, uint32_t f) const { return f ? Loaded{keys[q], vals[q]} : Loaded{0, 0u}; }      // (seven k-mers in eight belong to other ranks: their key and gene are not read again)
    __device__ void store(uint64_t, uint32_t f, uint32_t prefix, const Loaded &v) const { if (f) { keys_out[prefix] = v.key; vals_out[prefix] = v.val; } }
};

// ------------------------------------------------------------------------------------------------
// K-groups + K-ranges, fused (library.cpp:289-335).  The dictionary arrives as postings {gene, count} in (rank, gene)
// order with "opens a rank-group" in bit 31 of the count (set by K-rle; it is also the form the runs of a multi-GPU
// build travel in).  A group is the records from one head to the next; nothing is materialised about groups: every
// tile of 1024 records (one wave) rebuilds the extents it needs from the head bits (a 64-bit ballot per round, prev/next
// head by bit scans), looks beyond its borders only for the groups that cross them, and emits — for the records that
// get a posting range — the 16-byte tuple {first posting, postings, own count, group size} and the gene as sort key:
//
//   k_fold_last_record   the reference's scan closes the current group at the LAST record whatever its rank
//                        (library.cpp:300-306): a last record that opens a group of its own is folded into the preceding
//                        group, moved to its gene-order place there (:312-315) and the head bits are put right, so that
//                        from here on the bits alone say what the reference's groups are
//   k_group_waves<COUNT> ranges per tile (+ first/last head of every tile; + counters U', shared groups; + per-genome
//                        lookups; + per-gene costs in complexity-only mode)
//   (scan of the tile counts)
//   k_group_waves<WRITE> tuples and keys at tile offset + position inside the tile (record order), the group size of a
//                        group's last member added to its gene's cost (it has no range of its own), head bits removed
//
// Two passes over the postings replace the group scan (gid/goff), the shared-record compaction and the counter pass
// of the first version, i.e. about six passes over record-sized arrays.  The WRITE pass removes the head bits of its
// own tile only and asks the per-tile head positions of the COUNT pass about its neighbours, so no tile ever reads
// a bit another one may already have removed.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t HEAD_BIT = 0x80000000u;
constexpr uint32_t GT_NONE = 0xffffffffu;
constexpr uint32_t COST_LDS_GENOMES = 4096;

// last head at or before pos (record 0 always is one); one wave, 64 records per step
__device__ __forceinline__ uint32_t find_head_back(const uint2 *post, uint32_t pos, uint32_t lane) {
    for (;;) {
        const uint32_t base = pos + 1 >= PDL_WAVE ? pos + 1 - PDL_WAVE : 0;
        const uint32_t idx = base + lane;
        const bool f = idx <= pos && ((post[idx <= pos ? idx : pos].y >> 31) || idx == 0);
        const unsigned long long m = __ballot(f);
        if (m) return base + 63u - (uint32_t) __clzll((long long) m);
        pos = base - 1;                                   // (base > 0 here: index 0 always answers)
    }
}
// first head at or after pos, n when there is none
__device__ __forceinline__ uint32_t find_head_fwd(const uint2 *post, uint32_t pos, uint32_t n, uint32_t lane) {
    for (uint32_t base = pos; base < n; base += PDL_WAVE) {
        const uint32_t idx = base + lane;
        const bool f = idx < n && (post[idx < n ? idx : n - 1].y >> 31);
        const unsigned long long m = __ballot(f);
        if (m) return base + (uint32_t) __ffsll((long long) m) - 1u;
    }
    return n;
}

// One workgroup.  recpos (position of each record's first occurrence in the sorted stream, for pdl_get_dictionary) moves
// along when present.
__global__ __launch_bounds__(1024) void k_fold_last_record(uint2 *__restrict__ post, uint32_t *__restrict__ recpos, const uint64_t *d_u) {
    __shared__ uint32_t s_gs, s_p;
    const uint32_t u_count = (uint32_t) *d_u;
    if (u_count < 2) return;
    const uint32_t lastp = u_count - 1;
    uint2 last = post[lastp];
    if (!(last.y >> 31)) return;                         // (uniform) the last record belongs to its group anyway, in gene order
    last.y &= ~HEAD_BIT;                                 // it never opens a group (library.cpp:300-306)
    const uint32_t last_rp = recpos ? recpos[lastp] : 0u;
    if (threadIdx.x < PDL_WAVE) {
        const uint32_t gs = find_head_back(post, lastp - 1, threadIdx.x);       // the group it joins
        if (threadIdx.x == 0) {
            uint32_t lo = gs, hi = lastp;                // first index in [gs, lastp) whose gene is above the last record's
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (post[mid].x <= last.x) lo = mid + 1; else hi = mid;
            }
            s_gs = gs; s_p = lo;
        }
    }
    pdl_sync();
    const uint32_t p = s_p, gs = s_gs;
    if (p == lastp) { if (threadIdx.x == 0) post[lastp] = last; return; }       // already in place (uniform)
    for (uint32_t hi = lastp; hi > p; hi = hi > 1024 ? hi - 1024 : 0) {
        const bool live = hi >= 1 + threadIdx.x && hi - 1 - threadIdx.x >= p;
        const uint32_t i = hi - 1 - threadIdx.x;
        uint2 v = make_uint2(0, 0);
        uint32_t rp = 0;
        if (live) { v = post[i]; if (recpos) rp = recpos[i]; }
        pdl_sync();
        if (live) { post[i + 1] = v; if (recpos) recpos[i + 1] = rp; }
        pdl_sync();
        if (hi <= 1024) break;
    }
    if (threadIdx.x == 0) {
        if (p == gs) { last.y |= HEAD_BIT; post[gs + 1].y &= ~HEAD_BIT; }       // the moved record is the group's smallest gene: it is the head now
        post[p] = last;
        if (recpos) recpos[p] = last_rp;
    }
}

struct GroupTileArgs {
    uint2 *post;
    uint64_t n_bound; const uint64_t *d_n;      // record count: on the device (at most n_bound) or n_bound itself
    const uint8_t *in_shard;                    // MODE 0, 2: the genes that get range lists, one byte per gene ...
    const uint2 *own_iv; uint32_t n_own_iv;     // ... or (n_own_iv > 0) as sorted, disjoint gene-id intervals [x, y): searched in LDS,
                                                //     where a per-record byte gather would cost 64 addresses per instruction
    uint32_t *tile_sums;                        // [tiles] COUNT: ranges of the tile; k_tile_prefix: of the tiles before it in its block of 64
    uint32_t *chunk_sums;                       // [blocks of 64 tiles] ranges of a block; exclusive-scanned between the passes
    uint32_t n_blocks;
    uint32_t *th_first, *th_last;               // [tiles] first / last head of a tile (GT_NONE: none), COUNT -> WRITE
    uint32_t *key2; uint4 *tuples;              // WRITE: sort key (gene) and the 16-byte range tuple, or ...
    unsigned long long *pay8;                   // ... (non-null) the packed 8-byte range {first posting | (postings + (min(own count, 1023) << 22)) << 32},
                                                //     carried through the gene sort as its payload: no gather afterwards
    unsigned long long *head_bits;              // WRITE: [tiles * 16] the head bits it removes from the postings, kept for the lazy cost pass
    uint32_t pos_base;                          // WRITE, packed ranges: added to every first posting (a run of a multi-GPU build: its place in the gathered dictionary)
    unsigned long long *cost;                   // per-gene total_visited (library.cpp:327): last members (WRITE), all shared records (COUNT, RECORD_COSTS)
    unsigned long long *counters;               // COUNT: [0] += records in groups >= 2, [1] += such groups, [3] += records whose k-mer repeats inside its gene (range modes);  WRITE: [2] += lookups of the
                                                //        records that belong to this context (library.cpp:327 summed: "Total cost")
    const uint32_t *genome_of; uint32_t n_genomes;
    unsigned long long *g_full, *g_upper;       // COUNT, GENOMES: per genome, lookups as the reference counts them / above the diagonal
};

// One WAVE per tile of GW_TILE consecutive records, no LDS and no barrier on the data path: the head bits of a round of
// 64 records are one ballot (a scalar register pair); previous / next head of a record come from bit scans of its
// round's mask, from scalar scans over the rounds, and — for the groups that cross the tile's borders — from a look at
// the records around the tile (COUNT) or at the per-tile head positions the COUNT pass left (WRITE).
// The ranges of a tile go to (scanned total of the 64-tile blocks before) + (tiles before it in its block) + rank
// inside the tile, i.e. in record order.
// MODE 0: whole groups for the genes of a shard | 1: the postings above the record, every gene | 2: those, for the genes
// of a shard | 3: no ranges (counters / costs only).  PASS 0 = COUNT, 1 = WRITE.
constexpr int GW_ROUNDS = 16, GW_TILE = GW_ROUNDS * PDL_WAVE, GW_THREADS = 256, GW_WAVES = GW_THREADS / PDL_WAVE;
constexpr uint32_t GW_MAX_IV = 2048;                     // gene-id intervals of a shard held in LDS (16 KiB); more: the byte table
template <int PASS, int MODE, bool GENOMES, bool RECORD_COSTS>
__global__ __launch_bounds__(GW_THREADS) void k_group_waves(GroupTileArgs a) {
    __shared__ uint32_t s_red[2];
    extern __shared__ unsigned long long s_dyn[];        // GENOMES with <= COST_LDS_GENOMES genomes: full[G] | upper[G];  shard modes: intervals
    unsigned long long *s_full = s_dyn, *s_upper = s_dyn + a.n_genomes;
    uint2 *s_iv = reinterpret_cast<uint2 *>(s_dyn);      // (GENOMES and the shard modes never come together)
    const uint32_t n_iv = (MODE == 0 || MODE == 2) ? a.n_own_iv : 0u;
    if constexpr (MODE == 0 || MODE == 2) { for (uint32_t i = threadIdx.x; i < n_iv; i += GW_THREADS) s_iv[i] = a.own_iv[i]; }
    const uint32_t tid = threadIdx.x, lane = tid & (PDL_WAVE - 1);
    const uint32_t gw = blockIdx.x * GW_WAVES + tid / PDL_WAVE;                  // this wave's index = its chunk of tiles
    const uint32_t n = (uint32_t) scan_count(a.n_bound, a.d_n);
    const uint32_t tiles = (n + GW_TILE - 1) / GW_TILE;
    const bool lds_table = GENOMES && a.n_genomes <= COST_LDS_GENOMES;
    if constexpr (GENOMES) { if (lds_table) { for (uint32_t i = tid; i < 2 * a.n_genomes; i += GW_THREADS) s_dyn[i] = 0; } }
    if (tid < 2) s_red[tid] = 0;
    pdl_sync();
    const unsigned long long lt_mask = (1ull << lane) - 1ull, le_mask = (2ull << lane) - 1ull;
    uint32_t n_rec = 0, n_grp = 0;
    unsigned long long own_lookups = 0;
    // Tiles are dealt round-robin over the waves: the waves in flight read neighbouring tiles (a wave that owned a run of
    // consecutive tiles kept every wave on its own far-apart addresses, and the pass at a quarter of the streaming rate).
    for (uint32_t tile = gw; tile < tiles; tile += gridDim.x * GW_WAVES) {
        const uint32_t t0 = tile * GW_TILE, t1 = min(t0 + (uint32_t) GW_TILE, n);
        uint2 po[GW_ROUNDS];
#pragma unroll
        for (int j = 0; j < GW_ROUNDS; j++) {            // all loads first, branch-free
            const uint32_t u = t0 + j * PDL_WAVE + lane;
            po[j] = a.post[u < n ? u : n - 1];
        }
        // COUNT: the 64 records behind the tile say where the group that runs out of it ends.  WRITE: other waves may have
        // removed their tiles' head bits already, so the COUNT pass's per-tile heads answer: the nearest tile before / after
        // with a head (64 tiles per look, issued with the tile's own loads).
        constexpr bool may_peek = PASS == 0;
        const uint32_t pu = t1 + lane;
        const uint32_t peek = (may_peek && pu < n) ? a.post[pu].y >> 31 : 0u;
        uint32_t hb = GT_NONE, ha = GT_NONE;
        if constexpr (PASS == 1) {
            hb = lane < tile ? a.th_last[tile - 1 - lane] : GT_NONE;
            ha = tile + 1 + lane < tiles ? a.th_first[tile + 1 + lane] : GT_NONE;
        }
        // (shard modes) "this gene gets ranges" for all sixteen records at once: behind the ballots below each lookup would
        // wait for the one before it
        uint32_t ins = 0xffffffffu;
        if constexpr (MODE == 0 || MODE == 2) {
            ins = 0;
            if (n_iv) {                                  // (uniform) sixteen independent binary searches over the interval starts, in LDS
                uint32_t lo_i[GW_ROUNDS], hi_i[GW_ROUNDS];
#pragma unroll
                for (int j = 0; j < GW_ROUNDS; j++) { lo_i[j] = 0; hi_i[j] = n_iv; }          // last interval with start <= gene is lo_i - 1
                for (uint32_t span = n_iv; span > 0; span >>= 1) {
#pragma unroll
                    for (int j = 0; j < GW_ROUNDS; j++) {
                        const uint32_t mid = (lo_i[j] + hi_i[j]) >> 1;
                        const bool go = lo_i[j] < hi_i[j] && s_iv[mid < n_iv ? mid : n_iv - 1].x <= po[j].x;
                        if (lo_i[j] < hi_i[j]) { if (go) lo_i[j] = mid + 1; else hi_i[j] = mid; }
                    }
                }
#pragma unroll
                for (int j = 0; j < GW_ROUNDS; j++) ins |= (uint32_t) (lo_i[j] > 0 && po[j].x < s_iv[lo_i[j] > 0 ? lo_i[j] - 1 : 0].y) << j;
            } else {
                uint8_t inb[GW_ROUNDS];
#pragma unroll
                for (int j = 0; j < GW_ROUNDS; j++) inb[j] = a.in_shard[po[j].x];
#pragma unroll
                for (int j = 0; j < GW_ROUNDS; j++) ins |= (uint32_t) (inb[j] != 0) << j;
            }
        }
        unsigned long long m[GW_ROUNDS];
#pragma unroll
        for (int j = 0; j < GW_ROUNDS; j++) m[j] = __ballot(t0 + j * PDL_WAVE + lane < n && (po[j].y >> 31));
        // head of the group that runs into the tile
        uint32_t before = t0;
        if (!(m[0] & 1ull)) {                            // (uniform) the tile starts inside a group (t0 > 0: record 0 is a head)
            if constexpr (PASS == 0) before = find_head_back(a.post, t0 - 1, lane);
            else {
                unsigned long long hm = __ballot(hb != GT_NONE);
                if (hm) before = (uint32_t) __shfl((int) hb, __ffsll((long long) hm) - 1, PDL_WAVE);        // lane 0 = the tile just before
                else {
                    before = 0;
                    for (uint32_t hi = tile >= PDL_WAVE ? tile - PDL_WAVE : 0; hi > 0;) {                    // further back, 64 tiles per step
                        const uint32_t base = hi >= PDL_WAVE ? hi - PDL_WAVE : 0, idx = base + lane;
                        const uint32_t v = idx < hi ? a.th_last[idx] : GT_NONE;
                        hm = __ballot(v != GT_NONE);
                        if (hm) { before = (uint32_t) __shfl((int) v, 63 - __clzll((long long) hm), PDL_WAVE); break; }
                        hi = base;
                    }
                }
            }
        }
        // end of the group that runs out of the tile
        uint32_t after;
        if constexpr (PASS == 0) {
            const unsigned long long pm = __ballot(peek != 0);
            if (pm) after = t1 + (uint32_t) __ffsll((long long) pm) - 1u;
            else if (t1 + PDL_WAVE >= n) after = n;
            else after = find_head_fwd(a.post, t1 + PDL_WAVE, n, lane);
        } else {
            unsigned long long hm = __ballot(ha != GT_NONE);
            if (hm) after = (uint32_t) __shfl((int) ha, __ffsll((long long) hm) - 1, PDL_WAVE);
            else {
                after = n;
                for (uint32_t lo = tile + 1 + PDL_WAVE; lo < tiles; lo += PDL_WAVE) {
                    const uint32_t idx = lo + lane;
                    const uint32_t v = idx < tiles ? a.th_first[idx] : GT_NONE;
                    hm = __ballot(v != GT_NONE);
                    if (hm) { after = (uint32_t) __shfl((int) v, __ffsll((long long) hm) - 1, PDL_WAVE); break; }
                }
            }
        }
        before = (uint32_t) __builtin_amdgcn_readfirstlane((int) before);        // (uniform by construction: keep them in scalar registers)
        after = (uint32_t) __builtin_amdgcn_readfirstlane((int) after);
        // first head in the rounds after round j (scalar scan from the back)
        uint32_t nextr[GW_ROUNDS];
        uint32_t nx = after, first_in_tile = GT_NONE;
#pragma unroll
        for (int j = GW_ROUNDS - 1; j >= 0; j--) {
            nextr[j] = nx;
            if (m[j]) { nx = t0 + j * PDL_WAVE + (uint32_t) __ffsll((long long) m[j]) - 1u; first_in_tile = nx; }
        }
        uint32_t pr = before, cnt_tile = 0;              // (uniform) last head before the current round; ranges so far in the tile
        const uint32_t tile_prefix = PASS == 1 ? a.chunk_sums[tile / PDL_WAVE] + a.tile_sums[tile] : 0u;
#pragma unroll
        for (int j = 0; j < GW_ROUNDS; j++) {
            const uint32_t u = t0 + j * PDL_WAVE + lane;
            const unsigned long long at_or_below = m[j] & le_mask, above = m[j] & ~le_mask;
            const uint32_t gs = at_or_below ? t0 + j * PDL_WAVE + 63u - (uint32_t) __clzll((long long) at_or_below) : pr;
            const uint32_t ge = above ? t0 + j * PDL_WAVE + (uint32_t) __ffsll((long long) above) - 1u : nextr[j];
            const bool live = u < n;
            const bool shared = live && ge - gs >= 2;
            bool r = shared;
            if constexpr (MODE == 1 || MODE == 2) r = r && u + 1 < ge;           // the last member of a group has nothing above it
            if constexpr (MODE == 0 || MODE == 2) r = r && ((ins >> j) & 1u);
            if constexpr (MODE == 3) r = false;
            const unsigned long long rb = __ballot(r);
            if constexpr (PASS == 0) {
                n_rec += shared; n_grp += shared && u == gs;
                if constexpr (RECORD_COSTS) { if (shared) atomicAdd(&a.cost[po[j].x], (unsigned long long) (ge - gs)); }
                if constexpr (GENOMES) {
                    if (shared) {
                        const uint32_t gen = a.genome_of[po[j].x];
                        const unsigned long long full = ge - gs, up = ge - u - 1;
                        if (lds_table) { atomicAdd(&s_full[gen], full); if (up) atomicAdd(&s_upper[gen], up); }
                        else { atomicAdd(&a.g_full[gen], full); if (up) atomicAdd(&a.g_upper[gen], up); }
                    }
                }
            } else if (live) {
                const uint32_t cnt = po[j].y & ~HEAD_BIT;
                if (po[j].y >> 31) a.post[u].y = cnt;                              // the bit has done its job
                const bool mine = MODE == 1 || ((ins >> j) & 1u);
                if (shared && mine) own_lookups += ge - gs;
                if (r) {
                    const uint32_t at = tile_prefix + cnt_tile + (uint32_t) __popcll(rb & lt_mask);
                    const uint32_t start = MODE == 0 ? gs : u + 1;
                    a.key2[at] = po[j].x;
                    if (a.pay8) a.pay8[at] = (unsigned long long) (start + a.pos_base) | ((unsigned long long) ((ge - start) | (min(cnt, 1023u) << 22)) << 32);
                    else a.tuples[at] = make_uint4(start, ge - start, cnt, ge - gs);     // {first posting, postings, own count, group size}
                } else if (MODE == 1 || MODE == 2) {
                    if (!a.pay8 && ge - gs >= 2 && u + 1 == ge && mine)                  // (packed ranges: per-gene costs are made on demand)
                        atomicAdd(&a.cost[po[j].x], (unsigned long long) (ge - gs));
                }
            }
            if (PASS == 1 && a.head_bits && lane == 0) a.head_bits[(size_t) tile * GW_ROUNDS + j] = m[j];
            cnt_tile += (uint32_t) __popcll(rb);
            if (m[j]) pr = t0 + j * PDL_WAVE + 63u - (uint32_t) __clzll((long long) m[j]);
        }
        if constexpr (PASS == 0) {
            if (lane == 0) { a.tile_sums[tile] = cnt_tile; a.th_first[tile] = first_in_tile; a.th_last[tile] = first_in_tile != GT_NONE ? pr : GT_NONE; }
        }
    }
    if constexpr (PASS == 0) {
#pragma unroll
        for (int d = PDL_WAVE / 2; d > 0; d >>= 1) { n_rec += __shfl_xor(n_rec, d, PDL_WAVE); n_grp += __shfl_xor(n_grp, d, PDL_WAVE); }
        if (lane == 0) { atomicAdd(&s_red[0], n_rec); atomicAdd(&s_red[1], n_grp); }
        pdl_sync();
        if (tid < 2 && s_red[tid]) atomicAdd(&a.counters[tid], (unsigned long long) s_red[tid]);
        if constexpr (GENOMES) {
            if (lds_table) for (uint32_t i = tid; i < a.n_genomes; i += GW_THREADS) {
                if (s_full[i]) atomicAdd(&a.g_full[i], s_full[i]);
                if (s_upper[i]) atomicAdd(&a.g_upper[i], s_upper[i]);
            }
        }
    } else {
        __shared__ unsigned long long s_own;
        if (tid == 0) s_own = 0;
        pdl_sync();
#pragma unroll
        for (int d = PDL_WAVE / 2; d > 0; d >>= 1) own_lookups += __shfl_xor(own_lookups, d, PDL_WAVE);
        if (lane == 0 && own_lookups) atomicAdd(&s_own, own_lookups);
        pdl_sync();
        if (tid == 0 && s_own) atomicAdd(&a.counters[2], s_own);
    }
}

// Per-gene total_visited (library.cpp:327: every record of a group with >= 2 records adds the group size to its gene)
// made on demand from the head bits the WRITE pass kept — only pdl_sequence_costs / pdl_genome_cost ask for it once the
// ranges travel packed.  One thread per record; the extents come from word scans of the bit array.
__global__ __launch_bounds__(256) void k_gene_costs_lazy(const uint2 *__restrict__ post, const unsigned long long *__restrict__ head_bits,
                                                         uint32_t n, unsigned long long *__restrict__ cost) {
    const uint32_t u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n) return;
    // word layout: tile t, round j -> word t * 16 + j holds records t * 1024 + j * 64 .. +63  == record >> 6
    const uint32_t w = u >> 6, b = u & 63u, words = (n + 63) >> 6;
    uint32_t gs, ge;
    {
        unsigned long long m = head_bits[w] & ((2ull << b) - 1ull);
        uint32_t ww = w;
        while (!m && ww > 0) m = head_bits[--ww];
        gs = m ? ww * 64u + 63u - (uint32_t) __clzll((long long) m) : 0u;
    }
    {
        unsigned long long m = b == 63 ? 0ull : head_bits[w] & ~((2ull << b) - 1ull);
        uint32_t ww = w;
        while (!m && ww + 1 < words) m = head_bits[++ww];
        ge = m ? ww * 64u + (uint32_t) __ffsll((long long) m) - 1u : n;
        if (ge > n) ge = n;
    }
    if (ge - gs >= 2) atomicAdd(&cost[post[u].x], (unsigned long long) (ge - gs));
}

// COUNT for the range modes, per-thread code (the mask arithmetic of k_group_waves keeps a wave's uniform values in
// vector registers and ran at a quarter of the streaming rate): a record gets a range iff its successor does not open a
// group [and its gene belongs to the shard]; MODE 0: iff it is not alone in its group.  One wave per 1024-record tile.
// Also leaves the first / last head of every tile for the WRITE pass, and the counters U' / shared groups.
template <int MODE>
__global__ __launch_bounds__(GW_THREADS) void k_range_count(GroupTileArgs a) {
    extern __shared__ unsigned long long s_dyn[];
    __shared__ uint32_t s_red[2];
    uint2 *s_iv = reinterpret_cast<uint2 *>(s_dyn);
    const uint32_t n_iv = (MODE == 0 || MODE == 2) ? a.n_own_iv : 0u;
    if constexpr (MODE == 0 || MODE == 2) { for (uint32_t i = threadIdx.x; i < n_iv; i += GW_THREADS) s_iv[i] = a.own_iv[i]; }
    if (threadIdx.x < 2) s_red[threadIdx.x] = 0;
    pdl_sync();
    const uint32_t tid = threadIdx.x, lane = tid & (PDL_WAVE - 1);
    const uint32_t gw = blockIdx.x * GW_WAVES + tid / PDL_WAVE;
    const uint32_t n = (uint32_t) scan_count(a.n_bound, a.d_n);
    const uint32_t tiles = (n + GW_TILE - 1) / GW_TILE;
    uint32_t n_rec = 0, n_grp = 0, n_rep = 0;
    for (uint32_t tile = gw; tile < tiles; tile += gridDim.x * GW_WAVES) {
        const uint32_t t0 = tile * GW_TILE;
        uint2 po[GW_ROUNDS];
        uint32_t ynext[GW_ROUNDS];
#pragma unroll
        for (int j = 0; j < GW_ROUNDS; j++) {            // all loads first, branch-free
            const uint32_t u = t0 + j * PDL_WAVE + lane;
            po[j] = a.post[u < n ? u : n - 1];
            ynext[j] = a.post[u + 1 < n ? u + 1 : n - 1].y;
        }
        uint32_t ins = 0xffffffffu;
        if constexpr (MODE == 0 || MODE == 2) {
            ins = 0;
            if (n_iv) {                                  // (uniform) sixteen independent binary searches over the interval starts, in LDS
                uint32_t lo_i[GW_ROUNDS], hi_i[GW_ROUNDS];
#pragma unroll
                for (int j = 0; j < GW_ROUNDS; j++) { lo_i[j] = 0; hi_i[j] = n_iv; }
                for (uint32_t span = n_iv; span > 0; span >>= 1) {
#pragma unroll
                    for (int j = 0; j < GW_ROUNDS; j++) {
                        const uint32_t mid = (lo_i[j] + hi_i[j]) >> 1;
                        const bool go = lo_i[j] < hi_i[j] && s_iv[mid < n_iv ? mid : n_iv - 1].x <= po[j].x;
                        if (lo_i[j] < hi_i[j]) { if (go) lo_i[j] = mid + 1; else hi_i[j] = mid; }
                    }
                }
#pragma unroll
                for (int j = 0; j < GW_ROUNDS; j++) ins |= (uint32_t) (lo_i[j] > 0 && po[j].x < s_iv[lo_i[j] > 0 ? lo_i[j] - 1 : 0].y) << j;
            } else {
                uint8_t inb[GW_ROUNDS];
#pragma unroll
                for (int j = 0; j < GW_ROUNDS; j++) inb[j] = a.in_shard[po[j].x];
#pragma unroll
                for (int j = 0; j < GW_ROUNDS; j++) ins |= (uint32_t) (inb[j] != 0) << j;
            }
        }
        uint32_t cnt = 0, first_h = GT_NONE, last_h = 0, any_h = 0;
#pragma unroll
        for (int j = 0; j < GW_ROUNDS; j++) {
            const uint32_t u = t0 + j * PDL_WAVE + lane;
            const bool live = u < n;
            const bool head = live && (po[j].y >> 31);
            const bool next_head = u + 1 >= n || (ynext[j] >> 31);           // the successor opens a group, or is the end
            const bool shared = live && !(head && next_head);
            bool r = MODE == 0 ? shared : (live && !next_head);
            if constexpr (MODE == 0 || MODE == 2) r = r && ((ins >> j) & 1u);
            cnt += r;
            n_rec += shared; n_grp += head && !next_head;
            if (head) { first_h = min(first_h, u); last_h = max(last_h, u); any_h = 1; }
        }
        if ((tile & 7u) == 0) {                          // (uniform) records whose k-mer repeats inside its gene: a statistic, taken from every eighth tile
#pragma unroll
            for (int j = 0; j < GW_ROUNDS; j++) n_rep += 8u * (uint32_t) (t0 + j * PDL_WAVE + lane < n && (po[j].y & ~HEAD_BIT) >= 2u);
        }
#pragma unroll
        for (int d = PDL_WAVE / 2; d > 0; d >>= 1) {
            cnt += __shfl_xor(cnt, d, PDL_WAVE);
            first_h = min(first_h, (uint32_t) __shfl_xor((int) first_h, d, PDL_WAVE));
            last_h = max(last_h, (uint32_t) __shfl_xor((int) last_h, d, PDL_WAVE));
            any_h |= (uint32_t) __shfl_xor((int) any_h, d, PDL_WAVE);
        }
        if (lane == 0) { a.tile_sums[tile] = cnt; a.th_first[tile] = first_h; a.th_last[tile] = any_h ? last_h : GT_NONE; }
    }
#pragma unroll
    for (int d = PDL_WAVE / 2; d > 0; d >>= 1) { n_rec += __shfl_xor(n_rec, d, PDL_WAVE); n_grp += __shfl_xor(n_grp, d, PDL_WAVE); n_rep += __shfl_xor(n_rep, d, PDL_WAVE); }
    if (lane == 0) { atomicAdd(&s_red[0], n_rec); atomicAdd(&s_red[1], n_grp); if (n_rep) atomicAdd(&a.counters[3], (unsigned long long) n_rep); }
    pdl_sync();
    if (tid < 2 && s_red[tid]) atomicAdd(&a.counters[tid], (unsigned long long) s_red[tid]);
}

// ---- the range tuples sorted by gene without being written in record order first ------------------------------------------
// (upper ranges for every gene, packed: the single-GPU build.)  The first radix pass of the gene sort reads what the WRITE
// pass has just written; here the kernel that BUILDS the tuples is that pass: a workgroup takes PDL_RADIX_TILE = 4 tiles of
// records, k_range_count_hist has counted its tuples by the low byte of their gene (the pass's histogram; a record that gets
// no range is simply not there), the scan of those counts says where every (block, byte) run starts, and k_range_scatter
// makes the tuples as k_group_waves<WRITE> does and files them as k_rs_scatter does (ballot ranks, digit-sorted in LDS,
// coalesced runs out).  Saves the tuples' trip through HBM (12 B written + 16 B read per range) and three launches.
static_assert(GW_WAVES * GW_TILE == (int) PDL_RADIX_TILE && GW_THREADS == (int) PDL_RADIX_BINS, "a workgroup's four tiles are one tile of the radix pass");
__global__ __launch_bounds__(GW_THREADS) void k_range_count_hist(GroupTileArgs a, uint32_t n_tiles4, uint32_t *__restrict__ counts) {
    __shared__ uint32_t s_h[PDL_RADIX_BINS];
    __shared__ uint32_t s_red[2];
    const uint32_t tid = threadIdx.x, lane = tid & (PDL_WAVE - 1), wave = tid / PDL_WAVE;
    const uint32_t n = (uint32_t) scan_count(a.n_bound, a.d_n);
    if (tid < 2) s_red[tid] = 0;
    uint32_t n_rec = 0, n_grp = 0, n_rep = 0;
    for (uint32_t blk = blockIdx.x; blk < n_tiles4; blk += gridDim.x) {       // (uniform loop: barriers inside)
        s_h[tid] = 0;
        pdl_sync();
        const uint32_t tile = blk * GW_WAVES + wave, t0 = tile * GW_TILE;
        if (t0 < n) {                                     // (wave-uniform)
            uint2 po[GW_ROUNDS];
            uint32_t ynext[GW_ROUNDS];
#pragma unroll
            for (int j = 0; j < GW_ROUNDS; j++) {        // all loads first, branch-free
                const uint32_t u = t0 + j * PDL_WAVE + lane;
                po[j] = a.post[u < n ? u : n - 1];
                ynext[j] = a.post[u + 1 < n ? u + 1 : n - 1].y;
            }
            uint32_t cnt = 0, first_h = GT_NONE, last_h = 0, any_h = 0;
#pragma unroll
            for (int j = 0; j < GW_ROUNDS; j++) {
                const uint32_t u = t0 + j * PDL_WAVE + lane;
                const bool live = u < n;
                const bool head = live && (po[j].y >> 31);
                const bool next_head = u + 1 >= n || (ynext[j] >> 31);
                const bool r = live && !next_head;
                if (r) atomicAdd(&s_h[po[j].x & (PDL_RADIX_BINS - 1)], 1u);
                cnt += r;
                n_rec += live && !(head && next_head); n_grp += head && !next_head;
                if (head) { first_h = min(first_h, u); last_h = max(last_h, u); any_h = 1; }
            }
#pragma unroll
            for (int d = PDL_WAVE / 2; d > 0; d >>= 1) {
                cnt += __shfl_xor(cnt, d, PDL_WAVE);
                first_h = min(first_h, (uint32_t) __shfl_xor((int) first_h, d, PDL_WAVE));
                last_h = max(last_h, (uint32_t) __shfl_xor((int) last_h, d, PDL_WAVE));
                any_h |= (uint32_t) __shfl_xor((int) any_h, d, PDL_WAVE);
            }
            if (lane == 0) { a.tile_sums[tile] = cnt; a.th_first[tile] = first_h; a.th_last[tile] = any_h ? last_h : GT_NONE; }
            if ((tile & 7u) == 0) {                      // (uniform) records whose k-mer repeats inside its gene: a statistic, taken from every eighth tile
#pragma unroll
                for (int j = 0; j < GW_ROUNDS; j++) n_rep += 8u * (uint32_t) (t0 + j * PDL_WAVE + lane < n && (po[j].y & ~HEAD_BIT) >= 2u);
            }
        }
        pdl_sync();
        counts[(size_t) tid * n_tiles4 + blk] = s_h[tid];
    }
#pragma unroll
    for (int d = PDL_WAVE / 2; d > 0; d >>= 1) { n_rec += __shfl_xor(n_rec, d, PDL_WAVE); n_grp += __shfl_xor(n_grp, d, PDL_WAVE); n_rep += __shfl_xor(n_rep, d, PDL_WAVE); }
    if (lane == 0) { atomicAdd(&s_red[0], n_rec); atomicAdd(&s_red[1], n_grp); if (n_rep) atomicAdd(&a.counters[3], (unsigned long long) n_rep); }
    pdl_sync();
    if (tid < 2 && s_red[tid]) atomicAdd(&a.counters[tid], (unsigned long long) s_red[tid]);
}

__global__ __launch_bounds__(GW_THREADS) void k_range_scatter(GroupTileArgs a, uint32_t n_tiles4, const uint32_t *__restrict__ offs,
                                                              uint32_t *__restrict__ keys_out, unsigned long long *__restrict__ vals_out) {
    const uint32_t n = (uint32_t) scan_count(a.n_bound, a.d_n);
    if ((uint64_t) blockIdx.x * PDL_RADIX_TILE >= n) return;               // (uniform) block past the end
    __shared__ uint32_t s_key[PDL_RADIX_TILE];
    __shared__ unsigned long long s_val[PDL_RADIX_TILE];
    __shared__ uint16_t s_cnt[GW_WAVES][PDL_RADIX_BINS];   // per wave: running count of each byte value, then its base inside the block (16-bit: three workgroups per CU)
    __shared__ uint32_t s_tile_off[PDL_RADIX_BINS];
    __shared__ uint32_t s_goff[PDL_RADIX_BINS];
    __shared__ uint32_t s_wsum[17];
    __shared__ unsigned long long s_own;
    const uint32_t tid = threadIdx.x, lane = tid & (PDL_WAVE - 1), wave = tid / PDL_WAVE;
    const uint32_t tiles = (n + GW_TILE - 1) / GW_TILE;
    const uint32_t tile = blockIdx.x * GW_WAVES + wave, t0 = tile * GW_TILE, t1 = min(t0 + (uint32_t) GW_TILE, n);
    const bool active = tile < tiles;                    // (wave-uniform; the last block may hold fewer than four tiles)
    for (int w = 0; w < GW_WAVES; w++) s_cnt[w][tid] = 0;
    s_goff[tid] = offs[(size_t) tid * n_tiles4 + blockIdx.x];
    if (tid == 0) s_own = 0;
    pdl_sync();

    const unsigned long long lt_mask = (1ull << lane) - 1ull, le_mask = (2ull << lane) - 1ull;
    uint2 po[GW_ROUNDS];
    unsigned long long val[GW_ROUNDS];
    uint16_t rank[GW_ROUNDS];
    uint32_t rbits = 0;                                  // bit j: this lane's record of round j gets a range
    unsigned long long own_lookups = 0;
#pragma unroll
    for (int j = 0; j < GW_ROUNDS; j++) {                // all loads first, branch-free
        const uint32_t u = t0 + j * PDL_WAVE + lane;
        po[j] = a.post[u < n ? u : n - 1];
    }
    if (active) {
        // the nearest tiles before / after with a head, 64 tiles per look (the COUNT pass's per-tile heads: other
        // workgroups may have removed their head bits already)
        const uint32_t hb = lane < tile ? a.th_last[tile - 1 - lane] : GT_NONE;
        const uint32_t ha = tile + 1 + lane < tiles ? a.th_first[tile + 1 + lane] : GT_NONE;
        unsigned long long m[GW_ROUNDS];
#pragma unroll
        for (int j = 0; j < GW_ROUNDS; j++) m[j] = __ballot(t0 + j * PDL_WAVE + lane < n && (po[j].y >> 31));
        uint32_t before = t0;
        if (!(m[0] & 1ull)) {                            // (uniform) the tile starts inside a group
            unsigned long long hm = __ballot(hb != GT_NONE);
            if (hm) before = (uint32_t) __shfl((int) hb, __ffsll((long long) hm) - 1, PDL_WAVE);
            else {
                before = 0;
                for (uint32_t hi = tile >= PDL_WAVE ? tile - PDL_WAVE : 0; hi > 0;) {
                    const uint32_t base = hi >= PDL_WAVE ? hi - PDL_WAVE : 0, idx = base + lane;
                    const uint32_t v = idx < hi ? a.th_last[idx] : GT_NONE;
                    hm = __ballot(v != GT_NONE);
                    if (hm) { before = (uint32_t) __shfl((int) v, 63 - __clzll((long long) hm), PDL_WAVE); break; }
                    hi = base;
                }
            }
        }
        uint32_t after;
        {
            unsigned long long hm = __ballot(ha != GT_NONE);
            if (hm) after = (uint32_t) __shfl((int) ha, __ffsll((long long) hm) - 1, PDL_WAVE);
            else {
                after = n;
                for (uint32_t lo = tile + 1 + PDL_WAVE; lo < tiles; lo += PDL_WAVE) {
                    const uint32_t idx = lo + lane;
                    const uint32_t v = idx < tiles ? a.th_first[idx] : GT_NONE;
                    hm = __ballot(v != GT_NONE);
                    if (hm) { after = (uint32_t) __shfl((int) v, __ffsll((long long) hm) - 1, PDL_WAVE); break; }
                }
            }
        }
        before = (uint32_t) __builtin_amdgcn_readfirstlane((int) before);
        after = (uint32_t) __builtin_amdgcn_readfirstlane((int) after);
        uint32_t nextr[GW_ROUNDS];
        uint32_t nx = after;
#pragma unroll
        for (int j = GW_ROUNDS - 1; j >= 0; j--) {
            nextr[j] = nx;
            if (m[j]) nx = t0 + j * PDL_WAVE + (uint32_t) __ffsll((long long) m[j]) - 1u;
        }
        uint32_t pr = before;
#pragma unroll
        for (int j = 0; j < GW_ROUNDS; j++) {
            const uint32_t u = t0 + j * PDL_WAVE + lane;
            const unsigned long long at_or_below = m[j] & le_mask, above = m[j] & ~le_mask;
            const uint32_t gs = at_or_below ? t0 + j * PDL_WAVE + 63u - (uint32_t) __clzll((long long) at_or_below) : pr;
            const uint32_t ge = above ? t0 + j * PDL_WAVE + (uint32_t) __ffsll((long long) above) - 1u : nextr[j];
            const bool live = u < t1;
            const bool shared = live && ge - gs >= 2;
            const bool r = shared && u + 1 < ge;         // the last member of a group has nothing above it
            const uint32_t cnt = po[j].y & ~HEAD_BIT;
            if (live && (po[j].y >> 31)) a.post[u].y = cnt;                    // the bit has done its job
            if (shared) own_lookups += ge - gs;
            val[j] = (unsigned long long) (u + 1) | ((unsigned long long) ((ge - u - 1) | (min(cnt, 1023u) << 22)) << 32);
            rbits |= (uint32_t) r << j;
            if (lane == 0) a.head_bits[(size_t) tile * GW_ROUNDS + j] = m[j];
            if (m[j]) pr = t0 + j * PDL_WAVE + 63u - (uint32_t) __clzll((long long) m[j]);
        }
    }
    // ---- the radix pass on the low byte of the gene (k_rs_scatter's ranking; an element is a record with a range) ----------
#pragma unroll
    for (int j = 0; j < GW_ROUNDS; j++) {
        const bool valid = (rbits >> j) & 1u;
        const uint32_t d = po[j].x & (PDL_RADIX_BINS - 1);
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long mb = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? mb : ~mb;
        }
        const uint32_t seen = s_cnt[wave][d];                       // earlier rounds (own wave only: no race)
        rank[j] = (uint16_t) (seen + (uint32_t) __popcll(same & lt_mask));
        if (valid && (same & lt_mask) == 0) s_cnt[wave][d] = (uint16_t) (seen + (uint32_t) __popcll(same));      // lowest lane of the set
    }
    pdl_sync();
    uint32_t tot = 0;
    uint32_t wcnt[GW_WAVES];
#pragma unroll
    for (int w = 0; w < GW_WAVES; w++) { wcnt[w] = s_cnt[w][tid]; tot += wcnt[w]; }
    uint32_t tile_total;
    const uint32_t ex = block_exclusive_scan_u32(tot, s_wsum, tile_total);
    s_tile_off[tid] = ex;
    uint32_t run = ex;
#pragma unroll
    for (int w = 0; w < GW_WAVES; w++) { s_cnt[w][tid] = (uint16_t) run; run += wcnt[w]; }
    pdl_sync();
#pragma unroll
    for (int j = 0; j < GW_ROUNDS; j++) {
        if ((rbits >> j) & 1u) {
            const uint32_t lp = s_cnt[wave][po[j].x & (PDL_RADIX_BINS - 1)] + rank[j];
            s_key[lp] = po[j].x;
            s_val[lp] = val[j];
        }
    }
#pragma unroll
    for (int d = PDL_WAVE / 2; d > 0; d >>= 1) own_lookups += __shfl_xor(own_lookups, d, PDL_WAVE);
    if (lane == 0 && own_lookups) atomicAdd(&s_own, own_lookups);
    pdl_sync();
#pragma unroll
    for (int j = 0; j < GW_ROUNDS; j++) {
        const uint32_t e = j * GW_THREADS + tid;                     // coalesced over the digit-sorted block
        if (e < tile_total) {
            const uint32_t k = s_key[e];
            const uint32_t d = k & (PDL_RADIX_BINS - 1);
            const uint64_t dst = (uint64_t) s_goff[d] + (e - s_tile_off[d]);
            keys_out[dst] = k;
            vals_out[dst] = s_val[e];
        }
    }
    if (tid == 0 && s_own) atomicAdd(&a.counters[2], s_own);
}

// Between the passes: tile_sums[t] (ranges of tile t) becomes the count of the tiles before t inside its block of 64 tiles,
// chunk_sums[b] the block's total (scanned next); one wave per block.
__global__ __launch_bounds__(256) void k_tile_prefix(uint32_t *__restrict__ tile_sums, const uint64_t *d_n, uint64_t n_bound,
                                                     uint32_t *__restrict__ chunk_sums, uint32_t n_blocks) {
    const uint32_t b = blockIdx.x * 4 + threadIdx.x / PDL_WAVE, lane = threadIdx.x & (PDL_WAVE - 1);
    if (b >= n_blocks) return;
    const uint32_t tiles = (uint32_t) ((scan_count(n_bound, d_n) + GW_TILE - 1) / GW_TILE);
    const uint32_t t = b * PDL_WAVE + lane;
    const uint32_t v = t < tiles ? tile_sums[t] : 0u;
    const uint32_t inc = wave_inclusive_scan_u32(v);
    if (t < tiles) tile_sums[t] = inc - v;
    if (lane == PDL_WAVE - 1) chunk_sums[b] = inc;
}

// in_shard[gene] = the gene's genome belongs to this rank (multi-GPU: from the genome deal, without a host round trip)
__global__ __launch_bounds__(256) void k_genes_of_rank(const uint32_t *__restrict__ genome_of, const uint32_t *__restrict__ owner, uint32_t rank,
                                                       uint32_t n_seq, uint8_t *__restrict__ in_shard) {
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s < n_seq) in_shard[s] = owner[genome_of[s]] == rank ? 1 : 0;
}
// seq_owner[gene] = rank that owns the gene's genome
__global__ __launch_bounds__(256) void k_gene_owner(const uint32_t *__restrict__ genome_of, const uint32_t *__restrict__ owner, uint32_t n_seq,
                                                    uint8_t *__restrict__ seq_owner) {
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s < n_seq) seq_owner[s] = (uint8_t) owner[genome_of[s]];
}
// key = owner(gene) << 24 | gene: one radix pass on the top byte files the tuples by destination, in record order
__global__ __launch_bounds__(256) void k_owner_keys(uint32_t *__restrict__ key, const uint64_t *d_n, const uint8_t *__restrict__ seq_owner) {
    const uint32_t n = (uint32_t) *d_n;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t g = key[i];
        key[i] = g | ((uint32_t) seq_owner[g] << 24);
    }
}

// Also adds up total_visited (library.cpp:327) = the group sizes over a gene's ranges: the list is gene-sorted, so a
// wave holds one or two genes as a rule; one atomic per (wave, gene).
__global__ __launch_bounds__(256) void k_gather_ranges(const uint32_t *__restrict__ idx_sorted, const uint32_t *__restrict__ key_sorted,
                                                       const uint4 *__restrict__ tuples, const uint64_t *d_n, uint4 *__restrict__ ranges,
                                                       unsigned long long *__restrict__ cost) {
    const uint32_t n = (uint32_t) *d_n;                  // U' (grid sized for the bound M)
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256u >= n) return;                  // (uniform)
    const uint32_t lane = threadIdx.x & (PDL_WAVE - 1);
    const bool live = e < n;
    uint32_t g = 0xffffffffu;
    unsigned long long w = 0;
    if (live) {
        const uint4 t = tuples[idx_sorted[e]];
        ranges[e] = t;
        g = key_sorted[e];
        w = t.w;
    }
    unsigned long long todo = __ballot(live);
    while (todo) {
        const int leader = __ffsll((long long) todo) - 1;
        const uint32_t gl = __shfl(g, leader, PDL_WAVE);
        const bool in = live && g == gl;
        unsigned long long sum = in ? w : 0ull;
#pragma unroll
        for (int d = PDL_WAVE / 2; d > 0; d >>= 1) sum += __shfl_xor(sum, d, PDL_WAVE);
        if ((int) lane == leader) atomicAdd(&cost[gl], sum);
        todo &= ~__ballot(in);
    }
}
// seq_off[s] = first range of gene s in the gene-sorted list (lower bound), seq_off[N] = number of ranges
// (the sorted field of a key is (key >> shift) & mask: the tuples of a multi-GPU build carry the owner rank above the gene)
__global__ __launch_bounds__(256) void k_seq_offsets(const uint32_t *__restrict__ key_sorted, const uint64_t *d_n, uint32_t n_seq,
                                                     uint32_t *__restrict__ seq_off, uint32_t shift = 0, uint32_t mask = 0xffffffffu) {
    const uint32_t n = (uint32_t) *d_n;
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s > n_seq) return;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (((key_sorted[mid] >> shift) & mask) < s) lo = mid + 1; else hi = mid;
    }
    seq_off[s] = lo;
}

// K-cost: per-genome cost (library.cpp:535-538); the total is their sum (library.cpp:337-349).
// Genes of a genome are usually adjacent, so a wave first tries to add up as one.  The three dataset-wide values
// (sum / max / min of kseq_lengths) are kept per lane over a grid-stride loop and leave the workgroup as ONE atomic each:
// same-address device atomics serialise at ~10-50 ns apiece, a wave's worth per 64 genes took longer than the sums.
__global__ __launch_bounds__(256) void k_genome_cost(const unsigned long long *__restrict__ cost, const uint32_t *__restrict__ kseq_len,
                                                     const uint32_t *__restrict__ genome_of, uint32_t n_seq,
                                                     unsigned long long *__restrict__ genome_cost, unsigned long long *__restrict__ sum_kseq,
                                                     unsigned long long *__restrict__ max_kseq, unsigned long long *__restrict__ min_kseq) {
    __shared__ unsigned long long s_sum, s_max, s_min;
    if (threadIdx.x == 0) { s_sum = 0; s_max = 0; s_min = ~0ull; }
    pdl_sync();
    const uint32_t lane = threadIdx.x & (PDL_WAVE - 1);
    unsigned long long ksum = 0, kmax = 0, kmin = ~0ull;
    for (uint32_t s0 = blockIdx.x * 256; s0 < n_seq; s0 += gridDim.x * 256) {
        const uint32_t s = s0 + threadIdx.x;
        const bool live = s < n_seq;
        unsigned long long c = live ? cost[s] : 0ull;
        const unsigned long long kl = live ? (unsigned long long) kseq_len[s] : 0ull;
        const uint32_t g = live ? genome_of[s] : 0xffffffffu;
        ksum += kl; kmax = kl > kmax ? kl : kmax; if (kl && kl < kmin) kmin = kl;
        const uint32_t g0 = __shfl(g, 0, PDL_WAVE);
        if (__all(g == g0 || !live)) {
            unsigned long long csum = c;
#pragma unroll
            for (int d = PDL_WAVE / 2; d > 0; d >>= 1) csum += __shfl_down(csum, d, PDL_WAVE);
            if (lane == 0 && csum && g0 != 0xffffffffu) atomicAdd(&genome_cost[g0], csum);
        } else if (live && c) {
            atomicAdd(&genome_cost[g], c);
        }
    }
#pragma unroll
    for (int d = PDL_WAVE / 2; d > 0; d >>= 1) {
        ksum += __shfl_down(ksum, d, PDL_WAVE);
        unsigned long long o = __shfl_down(kmax, d, PDL_WAVE);
        kmax = o > kmax ? o : kmax;
        o = __shfl_down(kmin, d, PDL_WAVE);
        kmin = o < kmin ? o : kmin;
    }
    if (lane == 0) { atomicAdd(&s_sum, ksum); atomicMax(&s_max, kmax); atomicMin(&s_min, kmin); }
    pdl_sync();
    if (threadIdx.x == 0 && s_sum) { atomicAdd(sum_kseq, s_sum); atomicMax(max_kseq, s_max); atomicMax(min_kseq, ~s_min); }   // min kept as a max of complements: zero-initialised like the rest
}

// ------------------------------------------------------------------------------------------------
// Host side of the stages.  Control block words (c->scalars, u64): 0 U | 1 groups of any size | 2 ranges built |
// 3 bad-offsets flag | 4 sum kseq | 5 M | 6 emitted cells | 7 max kseq | 8 ~min kseq | 9 mirrored cells |
// 10 U' | 11 shared groups | 12 lookups of this context's genes | 13 records whose k-mer repeats inside its gene | 15 sort scratch | PDL_CTL_HIST.. histogram | PDL_CTL_GCOST.. per-genome cost
// ------------------------------------------------------------------------------------------------

// K-hist + K-len + the rank table; leaves M, the rank parameters and the key width in the context.
static void stage_alphabet_and_lengths(pdl_ctx *c, int kvalue, bool only_complexity) {
    hipStream_t st = c->stream;
    if (kvalue <= 0) PDL_FAIL(PDL_ERR_KVALUE, "K value must be greater than 0.");
    // control block: scalars[16] | residue histogram[256] | per-genome cost[G] (+ [G] lookups above the diagonal, multi-GPU)
    // — one allocation, one clearing fill, one device->host copy whenever the host looks
    // (device input whose genome ids are still on their way to the host: G is not known yet, at most N)
    const size_t ctl_words = PDL_CTL_GCOST + 2 * (size_t) (c->layout_deferred ? c->N : c->G);
    c->scalars.alloc(ctl_words * sizeof(uint64_t));
    uint64_t *d_scal = c->scalars.as<uint64_t>();
    hipLaunchKernelGGL(k_zero_u64, dim3((uint32_t) std::min<size_t>((ctl_words + 255) / 256, 1024)), dim3(256), 0, st, d_scal, ctl_words);     // (a kernel: a fill is a blit with ~10 us of barrier packets around it)

    ev_begin(c, EV_HIST);
    // K-len (independent of the histogram, reads the offsets only); its apply step also clears cost[], its total also lands in kmer_off[N]
    c->kseq_len.alloc((size_t) c->N * sizeof(uint32_t));
    c->kmer_off.alloc(((size_t) c->N + 1) * sizeof(uint64_t));
    c->cost.alloc((size_t) c->N * sizeof(uint64_t));
    scan_and_apply(c, c->N, KseqFlag{c->d_off, (uint32_t) kvalue},
                   KseqApply{c->kseq_len.as<uint32_t>(), c->kmer_off.as<uint64_t>(), c->cost.as<unsigned long long>(), c->d_off, c->R,
                             reinterpret_cast<unsigned long long *>(d_scal + 3)}, d_scal + 5, c->kmer_off.as<uint64_t>() + c->N);
    if (c->layout_deferred) pdl_input_arrived(c);        // offsets[0] = 0 and offsets[N] = R checked before anything indexes residues
    if (c->R) {
        uint32_t blocks = (uint32_t) std::min<uint64_t>((c->R / 64 + HIST_THREADS - 1) / HIST_THREADS + 1, 2048);
        hipLaunchKernelGGL(k_hist, dim3(blocks), dim3(HIST_THREADS), 0, st, c->d_res, c->R,
                           reinterpret_cast<unsigned long long *>(d_scal + PDL_CTL_HIST));
    }
    uint64_t counters[256];
    uint64_t M = 0, bad = 0;
    {
        PinRead rd(c);
        const uint64_t *pc = rd.add<uint64_t>(d_scal, PDL_CTL_GCOST);          // scalars + histogram in one read
        ev_end(c, EV_HIST);
        rd.sync();
        memcpy(counters, pc + PDL_CTL_HIST, sizeof(counters));
        M = pc[5]; bad = pc[3];
    }
    if (bad) PDL_FAIL(PDL_ERR_ARGUMENT, "offsets must ascend from 0 to the residue count (%llu)", (unsigned long long) c->R);
    rank_init_host(c->rp, counters, kvalue);
    c->M = M;
    c->only_complexity = only_complexity;
    if (M == 0) PDL_FAIL(PDL_ERR_EMPTY, "no gene is at least k=%d residues long: the dictionary is empty", kvalue);
    // the scan above sums in 32 bits: make sure it cannot have wrapped, and keep stream positions in u32
    if (c->R >= 0xfffffff0ull) PDL_FAIL(PDL_ERR_UNSUPPORTED, "more than 2^32 residues (%llu) need 64-bit stream positions", (unsigned long long) c->R);
    c->key64 = c->rp.rank_bits > 32;
}

// K-rank over the whole stream into (keys_a, vals_a); d_bins != nullptr: also the interval histogram (multi-GPU build)
template <class KeyT>
static void stage_rank(pdl_ctx *c, uint32_t *d_bins = nullptr, uint32_t bin_shift = 0) {
    hipStream_t st = c->stream;
    const uint64_t M = c->M;
    c->keys_a.alloc(M * sizeof(KeyT)); c->keys_b.alloc(M * sizeof(KeyT));
    c->vals_a.alloc(M * sizeof(uint32_t)); c->vals_b.alloc(M * sizeof(uint32_t));
    if (c->rp.hash_fallback) {
        if constexpr (sizeof(KeyT) == 8) {
            if (d_bins) hipLaunchKernelGGL(k_rank_hash<1>, dim3((c->N + 255) / 256), dim3(256), 0, st, c->d_res, c->d_off,
                                           c->kmer_off.as<uint64_t>(), c->N, c->rp, c->keys_a.as<uint64_t>(), c->vals_a.as<uint32_t>(), d_bins);
            else hipLaunchKernelGGL(k_rank_hash<0>, dim3((c->N + 255) / 256), dim3(256), 0, st, c->d_res, c->d_off,
                                    c->kmer_off.as<uint64_t>(), c->N, c->rp, c->keys_a.as<uint64_t>(), c->vals_a.as<uint32_t>(), (uint32_t *) nullptr);
        }
    } else {
        const uint64_t tiles = (M + RANK_TILE - 1) / RANK_TILE;
        if (d_bins) hipLaunchKernelGGL((k_rank<KeyT, 1>), dim3((uint32_t) std::min<uint64_t>(tiles, 2048)), dim3(RANK_THREADS), 0, st, c->d_res, c->d_off,
                                       c->kmer_off.as<uint64_t>(), c->N, M, c->R, c->rp, c->keys_a.as<KeyT>(), c->vals_a.as<uint32_t>(), bin_shift, d_bins);
        else hipLaunchKernelGGL((k_rank<KeyT, 0>), dim3((uint32_t) tiles), dim3(RANK_THREADS), 0, st, c->d_res, c->d_off,
                                c->kmer_off.as<uint64_t>(), c->N, M, c->R, c->rp, c->keys_a.as<KeyT>(), c->vals_a.as<uint32_t>(), 0u, (uint32_t *) nullptr);
    }
    PDL_HIP(hipGetLastError());
}

// K-sort + K-rle over the first m elements of (keys_in, vals_in): records into c->post / recpos.
// d_scal[0] receives the record count.
template <class KeyT>
static void stage_sort_and_dedup(pdl_ctx *c, KeyT *keys_in, KeyT *keys_out, uint32_t *vals_in, uint32_t *vals_out, uint64_t m) {
    uint64_t *d_scal = c->scalars.as<uint64_t>();
    ev_begin(c, EV_SORT1);
    pdl_sort_pairs<KeyT>(c, keys_in, keys_out, vals_in, vals_out, m, c->rp.rank_bits, false, nullptr, 0, c->rp.key_bits == c->rp.rank_bits);
    ev_end(c, EV_SORT1);
    // remember which physical buffers hold the sorted stream (pdl_get_dictionary reads them)
    if ((void *) keys_out != c->keys_b.p) { std::swap(c->keys_a.p, c->keys_b.p); std::swap(c->keys_a.bytes, c->keys_b.bytes); }
    if ((void *) vals_out != c->vals_b.p) { std::swap(c->vals_a.p, c->vals_b.p); std::swap(c->vals_a.bytes, c->vals_b.bytes); }
    const KeyT *skeys = c->keys_b.as<KeyT>();
    const uint32_t *svals = c->vals_b.as<uint32_t>();
    ev_begin(c, EV_DICT);                                       // (ended by the caller, behind K-groups where it runs them)
    c->recpos.alloc((m + 1) * sizeof(uint32_t));
    c->post.alloc(m * sizeof(uint2));                           // U <= m records (sized before U is known)
    scan_and_apply(c, m, RecHead<KeyT>{skeys, svals},
                   RecScatter<KeyT>{skeys, svals, m, c->recpos.as<uint32_t>(), c->post.as<uint2>()}, d_scal + 0);
}

// Launch helpers of k_group_waves.  group_tiles_plan sizes the grid and the scratch: tile_sums[tiles] | th_first[tiles] |
// th_last[tiles] | chunk_sums[blocks of 64 tiles].  Returns the number of workgroups.
static uint32_t group_tiles_plan(pdl_ctx *c, GroupTileArgs &a) {
    const uint64_t tiles = (a.n_bound + GW_TILE - 1) / GW_TILE;
    if (tiles > 0x7fffffffull) PDL_FAIL(PDL_ERR_UNSUPPORTED, "dictionary of %llu records exceeds the grid limit", (unsigned long long) a.n_bound);
    int cus = c->cus;
    if (cus <= 0) {
        cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) cus = prop.multiProcessorCount;
        c->cus = cus;
    }
    a.n_blocks = (uint32_t) ((tiles + PDL_WAVE - 1) / PDL_WAVE);
    const uint32_t grid = (uint32_t) std::max<uint64_t>(1, std::min<uint64_t>((tiles + GW_WAVES - 1) / GW_WAVES, (uint64_t) cus * 8));
    c->scan_tmp.alloc(((size_t) tiles * 3 + a.n_blocks + 1) * sizeof(uint32_t));
    a.tile_sums = c->scan_tmp.as<uint32_t>(); a.th_first = a.tile_sums + tiles; a.th_last = a.th_first + tiles; a.chunk_sums = a.th_last + tiles;
    return grid;
}
template <int PASS, int MODE, bool GENOMES, bool RECORD_COSTS>
static void launch_group_tiles(pdl_ctx *c, const GroupTileArgs &a, uint32_t grid) {
    size_t dyn = GENOMES && a.n_genomes <= COST_LDS_GENOMES ? 2 * (size_t) a.n_genomes * sizeof(uint64_t) : 0;
    if (MODE == 0 || MODE == 2) dyn = (size_t) a.n_own_iv * sizeof(uint2);
    hipLaunchKernelGGL((k_group_waves<PASS, MODE, GENOMES, RECORD_COSTS>), dim3(grid), dim3(GW_THREADS), dyn, c->stream, a);
    PDL_HIP(hipGetLastError());
}

// K-groups + K-ranges + K-cost over the dictionary (postings with head bits; `bound` records at most, the count is at
// d_scal[0]).  mode 0: whole groups for the shard's genes | 1: upper ranges, every gene | 2: upper ranges, the shard's genes.
static void stage_ranges_and_costs(pdl_ctx *c, uint64_t bound, int mode, bool only_complexity) {
    hipStream_t st = c->stream;
    c->costs_ready = true; c->ranges8 = nullptr;
    uint64_t *d_scal = c->scalars.as<uint64_t>();
    const uint64_t *d_u = d_scal + 0;
    uint2 *post = pdl_postings(c);
    GroupTileArgs ga{};
    ga.post = post; ga.n_bound = bound; ga.d_n = d_u;
    const uint32_t grid = group_tiles_plan(c, ga);
    ga.cost = c->cost.as<unsigned long long>();
    ga.counters = reinterpret_cast<unsigned long long *>(d_scal + 10);
    ga.genome_of = c->d_gen; ga.n_genomes = c->G;
    hipLaunchKernelGGL(k_fold_last_record, dim3(1), dim3(1024), 0, st, post, c->post_ext ? (uint32_t *) nullptr : c->recpos.as<uint32_t>(), d_u);
    if (only_complexity) {                   // (cost[] was zeroed by K-len's apply, the counters with the control block)
        launch_group_tiles<0, 3, false, true>(c, ga, grid);
    } else {
        ev_begin(c, EV_SORT2);
        // Ranges travel packed (8 bytes, carried through the gene sort as its payload: no gather afterwards) whenever a
        // range sits right behind its own record (the upper modes) and a posting count fits 22 bits; else as 16-byte
        // tuples fetched through the sorted positions.
        const bool packed = mode != 0 && c->N < (1u << 22);
        // scratch: packed   pay_a u64[cap] | pay_b u64[cap] | k2b u32[cap]
        //          tuples   tuples uint4[cap] | v2a u32[cap] | k2b u32[cap] | v2b u32[cap];   key2 lives in vals_a (free after sort 1) or behind them
        // cap = the bound (every record may get a range) when the whole build stays on the device without a look from the host
        // (mode 1); with a shard (modes 0, 2: 1/W or one batch of the genes) the COUNT pass's total is read first and the buffers
        // are sized for it — a 512-genome set would otherwise allocate 20-28 bytes for each of its 0.9 G records per shard
        const bool exact = mode != 1;
        uint64_t cap = bound;
        uint4 *tuples = nullptr;
        unsigned long long *pay_a = nullptr, *pay_b = nullptr;
        uint32_t *k2a = nullptr, *v2a = nullptr, *k2b = nullptr, *v2b = nullptr;
        auto carve = [&]() {
            c->scratch.alloc(cap * (packed ? 2 * sizeof(uint64_t) + sizeof(uint32_t) : sizeof(uint4) + 3 * sizeof(uint32_t)) + (exact ? cap * sizeof(uint32_t) : 0) + 64);
            tuples = c->scratch.as<uint4>();
            pay_a = c->scratch.as<unsigned long long>(); pay_b = pay_a + cap;
            v2a = packed ? nullptr : reinterpret_cast<uint32_t *>(tuples + cap);
            k2b = packed ? reinterpret_cast<uint32_t *>(pay_b + cap) : v2a + cap;
            v2b = packed ? nullptr : k2b + cap;
            if (exact) k2a = (packed ? k2b : v2b) + cap;                     // (the sort's first buffer may be gone: "low_memory")
            else { c->vals_a.alloc(cap * sizeof(uint32_t)); k2a = c->vals_a.as<uint32_t>(); }
        };
        if (!exact) carve();
        if (mode != 1) {                    // the shard as gene-id intervals, when every genome's genes are consecutive ids (the usual .faa)
            std::vector<uint2> &iv = c->h_own_iv;
            iv.clear();
            bool contiguous = true;
            for (uint32_t g : c->dict_shard) {           // (ascending genome ids; genomes in first-seen order: ascending gene ids too)
                const uint32_t b0 = c->h_genome_row_off[g], e0 = c->h_genome_row_off[g + 1];
                if (b0 == e0) continue;
                const uint32_t first = c->h_genome_rows[b0], last = c->h_genome_rows[e0 - 1];
                if (last - first + 1 != e0 - b0) { contiguous = false; break; }
                if (!iv.empty() && iv.back().y == first) iv.back().y = last + 1; else iv.push_back(make_uint2(first, last + 1));
            }
            if (contiguous) std::sort(iv.begin(), iv.end(), [](const uint2 &p, const uint2 &q) { return p.x < q.x; });
            for (size_t i = 1; contiguous && i < iv.size(); i++) if (iv[i].x < iv[i - 1].y) contiguous = false;      // (cannot happen: genes belong to one genome)
            if (contiguous && !iv.empty() && iv.size() <= GW_MAX_IV) {
                c->own_iv.alloc(iv.size() * sizeof(uint2));
                PDL_HIP(hipMemcpyAsync(c->own_iv.p, iv.data(), iv.size() * sizeof(uint2), hipMemcpyHostToDevice, st));
                ga.own_iv = c->own_iv.as<uint2>(); ga.n_own_iv = (uint32_t) iv.size();
            }
        }
        if (mode != 1 && c->dist) {         // multi-GPU: the deal is on the device already (pdl_run_dist_finish)
            ga.in_shard = c->seq_in_shard.as<uint8_t>();
        } else if (mode != 1) {             // only the genes this context scores need range lists
            std::vector<uint8_t> &h = c->h_seq_in_shard;   // lives in the context: the copy below needs no synchronisation
            h.assign((size_t) c->N, 0);
            std::vector<uint8_t> gsel((size_t) c->G, 0);
            for (uint32_t g : c->dict_shard) gsel[g] = 1;
            for (uint32_t i = 0; i < c->N; i++) h[i] = gsel[c->h_genome_of[i]];
            c->seq_in_shard.alloc(c->N);
            PDL_HIP(hipMemcpyAsync(c->seq_in_shard.p, h.data(), c->N, hipMemcpyHostToDevice, st));
            ga.in_shard = c->seq_in_shard.as<uint8_t>();
        }
        // the head bits the WRITE pass takes out of the postings are kept: for the per-gene costs made on demand (packed ranges), and
        // to put them back when the ranges are built again for another shard of genomes (pdl_run_reshard)
        c->head_bits.alloc(((bound + GW_TILE - 1) / GW_TILE) * GW_ROUNDS * sizeof(uint64_t));
        ga.head_bits = c->head_bits.as<unsigned long long>();
        const uint64_t *d_us = d_scal + 2;       // ranges built = the total of the tile counts
        // single-GPU build with packed ranges: the tuples are filed by the low byte of their gene by the kernel that makes
        // them (the first pass of the gene sort without a trip through HBM in between)
        const bool fused = mode == 1 && packed;
        if (fused) {
            const uint32_t n_tiles4 = (uint32_t) ((bound + PDL_RADIX_TILE - 1) / PDL_RADIX_TILE);
            const size_t table = (size_t) PDL_RADIX_BINS * n_tiles4, tiles = (bound + GW_TILE - 1) / GW_TILE;
            // (the scan below uses scan_tmp: the per-tile heads move next to the radix tables)
            c->sort_tmp.alloc((2 * table + 3 * tiles + 1) * sizeof(uint32_t));
            uint32_t *counts = c->sort_tmp.as<uint32_t>(), *offs = counts + table;
            ga.tile_sums = offs + table; ga.th_first = ga.tile_sums + tiles; ga.th_last = ga.th_first + tiles; ga.chunk_sums = nullptr;
            const uint32_t grid4 = std::max<uint32_t>(1, std::min<uint32_t>(n_tiles4, (uint32_t) c->cus * 8));
            hipLaunchKernelGGL(k_range_count_hist, dim3(grid4), dim3(GW_THREADS), 0, st, ga, n_tiles4, counts);
            pdl_radix_offsets(c, counts, offs, n_tiles4, d_scal + 2);
            hipLaunchKernelGGL(k_range_scatter, dim3(n_tiles4), dim3(GW_THREADS), 0, st, ga, n_tiles4, offs, k2b, pay_b);
            PDL_HIP(hipGetLastError());
        } else {
        {
            const size_t dyn = (size_t) ga.n_own_iv * sizeof(uint2);
            if (mode == 1) hipLaunchKernelGGL(k_range_count<1>, dim3(grid), dim3(GW_THREADS), 0, st, ga);
            else if (mode == 2) hipLaunchKernelGGL(k_range_count<2>, dim3(grid), dim3(GW_THREADS), dyn, st, ga);
            else hipLaunchKernelGGL(k_range_count<0>, dim3(grid), dim3(GW_THREADS), dyn, st, ga);
            PDL_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL(k_tile_prefix, dim3((ga.n_blocks + 3) / 4), dim3(256), 0, st, ga.tile_sums, ga.d_n, ga.n_bound, ga.chunk_sums, ga.n_blocks);
        hipLaunchKernelGGL(k_scan_tile_scan, dim3(1), dim3(1024), 0, st, ga.chunk_sums, ga.n_blocks, d_scal + 2, (uint64_t *) nullptr);
        if (exact) {                         // the shard's range count, then buffers of that size
            if (!c->tasks_ready) pdl_prepare_tasks(c);       // host work + small uploads while the device counts
            PinRead rd(c);
            const uint64_t *pn = rd.add<uint64_t>(d_us, 1);
            rd.sync();
            cap = std::max<uint64_t>(pn[0], 1);
            carve();
        }
        ga.key2 = k2a; ga.tuples = tuples; ga.pay8 = packed ? pay_a : nullptr;
        if (mode == 1) launch_group_tiles<1, 1, false, false>(c, ga, grid);
        else if (mode == 2) launch_group_tiles<1, 2, false, false>(c, ga, grid);
        else launch_group_tiles<1, 0, false, false>(c, ga, grid);
        }
        c->upper_only = mode != 0;
        const uint32_t seq_bits = std::max<uint32_t>(1, bit_length64(c->N ? c->N - 1 : 0));
        // One GPU: the range count stays on the device, the kernels behind it are sized for the bound.  Multi-GPU: a rank
        // builds ranges for 1/world of the records, so it reads the count (one synchronisation) and sizes them exactly.
        uint64_t n_sort = bound;
        const uint64_t *d_sort_n = d_us;
        if (exact) { n_sort = cap; d_sort_n = nullptr; }
        c->seq_off.alloc(((size_t) c->N + 1) * sizeof(uint32_t));
        if (packed) {
            if (fused) {                     // pass 1 is done: (k2b, pay_b) hold its output, the remaining passes go on from there
                std::swap(k2a, k2b); std::swap(pay_a, pay_b);
                pdl_sort_pairs<uint32_t, unsigned long long>(c, k2a, k2b, pay_a, pay_b, n_sort, seq_bits, false, d_sort_n, 8, true);
            } else
            pdl_sort_pairs<uint32_t, unsigned long long>(c, k2a, k2b, pay_a, pay_b, n_sort, seq_bits, false, d_sort_n, 0, true);   // sorted pairs now in (k2b, pay_b)
            ev_end(c, EV_SORT2);
            ev_begin(c, EV_RANGES);
            c->ranges8 = reinterpret_cast<const uint2 *>(pay_b);       // gene major: the join reads them where the sort left them
            c->costs_ready = false;
        } else {
            pdl_sort_pairs<uint32_t>(c, k2a, k2b, v2a, v2b, n_sort, seq_bits, true, d_sort_n, 0, true);     // values = tuple positions; sorted pairs now in (k2b, v2b)
            ev_end(c, EV_SORT2);
            ev_begin(c, EV_RANGES);
            c->ranges8 = nullptr;
            c->ranges.alloc(std::max<uint64_t>(n_sort, 1) * sizeof(uint4));
            const uint32_t gblocks = (uint32_t) std::max<uint64_t>((n_sort + 255) / 256, 1);
            hipLaunchKernelGGL(k_gather_ranges, dim3(gblocks), dim3(256), 0, st, v2b, k2b, tuples, d_us, c->ranges.as<uint4>(), c->cost.as<unsigned long long>());
        }
        hipLaunchKernelGGL(k_seq_offsets, dim3((c->N + 1 + 255) / 256), dim3(256), 0, st, k2b, d_us, c->N, c->seq_off.as<uint32_t>());
        PDL_HIP(hipGetLastError());
        ev_end(c, EV_RANGES);
    }

    // K-cost
    unsigned long long *d_gcost = reinterpret_cast<unsigned long long *>(d_scal + PDL_CTL_GCOST);
    hipLaunchKernelGGL(k_genome_cost, dim3(std::min<uint32_t>((c->N + 255) / 256, 128)), dim3(256), 0, st, c->cost.as<unsigned long long>(),
                       c->kseq_len.as<uint32_t>(), c->d_gen, c->N, d_gcost,
                       reinterpret_cast<unsigned long long *>(d_scal + 4), reinterpret_cast<unsigned long long *>(d_scal + 7),
                       reinterpret_cast<unsigned long long *>(d_scal + 8));
    PDL_HIP(hipGetLastError());

    uint64_t tail[12] = {0}, tail_own = 0, tail_rep = 0;
    {
        PinRead rd(c);                       // one copy: the whole control block
        const uint64_t *pt = rd.add<uint64_t>(d_scal, PDL_CTL_GCOST + (size_t) c->G);
        const uint32_t *lbe = lookback_error_word(c, rd);
        rd.sync();
        lookback_check(c, lbe);
        c->h_genome_cost.assign(pt + PDL_CTL_GCOST, pt + PDL_CTL_GCOST + c->G);     // (a shard, one rank of several: its own genomes only)
        memcpy(tail, pt, sizeof(tail));
        tail_own = pt[12]; tail_rep = pt[13];
    }
    c->U = tail[0];
    c->Ushared = tail[10];
    c->Urepeat = tail_rep;
    c->NG = tail[11];
    c->sum_kseq = tail[4];
    c->max_kseq = tail[7];
    c->min_kseq = tail[8] == 0 ? 1 : ~tail[8];
    c->P = 0;
    if (c->costs_ready) for (uint64_t v : c->h_genome_cost) c->P += v;
    else c->P = tail_own;                     // packed ranges: "Total cost" straight from the WRITE pass; per-gene / per-genome costs on demand
}

// ---- the ranges again, for another shard of genomes, on the dictionary that is there ---------------------------------------------
// (scoring a large set a batch of genomes at a time: the postings — rank, sort, dedup, most of the build — are made once; a batch
// costs the two passes over the postings that form its genes' range lists.)  The WRITE pass took the group-head bits out of the
// postings; they are put back from the copy it kept, the counters of the range stage start at zero again.
__global__ __launch_bounds__(256) void k_restore_heads(uint2 *__restrict__ post, const unsigned long long *__restrict__ head_bits, uint32_t n) {
    const uint32_t u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n) return;
    if ((head_bits[u >> 6] >> (u & 63u)) & 1ull) post[u].y |= HEAD_BIT;          // (word layout: record >> 6, see k_gene_costs_lazy)
}

void pdl_run_reshard(pdl_ctx *c) {
    hipStream_t st = c->stream;
    if (c->dist || c->post_ext) PDL_FAIL(PDL_ERR_STATE, "genome shard: a multi-GPU context deals the genomes itself");
    if (c->dict_shard.empty() || c->upper_only) PDL_FAIL(PDL_ERR_STATE, "genome shard: the dictionary was built for all genomes; set the first shard before pdl_preprocess");
    uint64_t *d_scal = c->scalars.as<uint64_t>();
    const uint32_t n = (uint32_t) c->U;
    ev_begin(c, EV_PRE_TOTAL);
    hipLaunchKernelGGL(k_restore_heads, dim3((n + 255) / 256), dim3(256), 0, st, c->post.as<uint2>(), c->head_bits.as<unsigned long long>(), n);
    PDL_HIP(hipMemsetAsync(d_scal + 2, 0, sizeof(uint64_t), st));
    PDL_HIP(hipMemsetAsync(d_scal + 10, 0, 6 * sizeof(uint64_t), st));
    PDL_HIP(hipMemsetAsync(d_scal + PDL_CTL_GCOST, 0, 2 * (size_t) c->G * sizeof(uint64_t), st));
    PDL_HIP(hipMemsetAsync(c->cost.p, 0, (size_t) c->N * sizeof(uint64_t), st));
    // (k_genome_cost adds the k-mer statistics of all genes up once more: they are taken from the first build)
    const uint64_t sum_kseq = c->sum_kseq, max_kseq = c->max_kseq, min_kseq = c->min_kseq;
    PDL_HIP(hipMemsetAsync(d_scal + 4, 0, sizeof(uint64_t), st));
    PDL_HIP(hipMemsetAsync(d_scal + 7, 0, 2 * sizeof(uint64_t), st));
    c->dict_shard = c->shard;
    c->tasks_ready = false;
    ev_begin(c, EV_DICT); ev_end(c, EV_DICT);
    stage_ranges_and_costs(c, c->U, 0, false);
    c->sum_kseq = sum_kseq; c->max_kseq = max_kseq; c->min_kseq = min_kseq;
    ev_end(c, EV_PRE_TOTAL);
    PDL_HIP(hipStreamSynchronize(st));
    c->tm.sort_seq_ms = ev_ms(c, EV_SORT2);
    c->tm.ranges_ms = ev_ms(c, EV_RANGES);
    c->tm.reshard_ms = ev_ms(c, EV_PRE_TOTAL);
}

// cost[] (per gene) and h_genome_cost (per genome), when the build left them for later (packed ranges)
void pdl_ensure_costs(pdl_ctx *c) {
    if (c->costs_ready) return;
    hipStream_t st = c->stream;
    uint64_t *d_scal = c->scalars.as<uint64_t>();
    PDL_HIP(hipMemsetAsync(c->cost.p, 0, (size_t) c->N * sizeof(uint64_t), st));
    PDL_HIP(hipMemsetAsync(d_scal + PDL_CTL_GCOST, 0, (size_t) c->G * sizeof(uint64_t), st));
    const uint32_t n = (uint32_t) c->U;
    hipLaunchKernelGGL(k_gene_costs_lazy, dim3((n + 255) / 256), dim3(256), 0, st, pdl_postings(c), c->head_bits.as<unsigned long long>(), n,
                       c->cost.as<unsigned long long>());
    // per genome (the kseq statistics it also adds up go to scratch words)
    hipLaunchKernelGGL(k_genome_cost, dim3(std::min<uint32_t>((c->N + 255) / 256, 128)), dim3(256), 0, st, c->cost.as<unsigned long long>(),
                       c->kseq_len.as<uint32_t>(), c->d_gen, c->N, reinterpret_cast<unsigned long long *>(d_scal + PDL_CTL_GCOST),
                       reinterpret_cast<unsigned long long *>(d_scal + 13), reinterpret_cast<unsigned long long *>(d_scal + 14),
                       reinterpret_cast<unsigned long long *>(d_scal + 15));
    PDL_HIP(hipGetLastError());
    PinRead rd(c);
    const uint64_t *pg = rd.add<uint64_t>(d_scal + PDL_CTL_GCOST, c->G);
    rd.sync();
    c->h_genome_cost.assign(pg, pg + c->G);
    c->costs_ready = true;
}

template <class KeyT>
static void dictionary_pipeline(pdl_ctx *c, bool only_complexity) {
    const uint64_t M = c->M;
    ev_begin(c, EV_RANK);
    stage_rank<KeyT>(c);
    ev_end(c, EV_RANK);
    KeyT *keys_in = c->keys_a.as<KeyT>(), *keys_out = c->keys_b.as<KeyT>();
    uint32_t *vals_in = c->vals_a.as<uint32_t>(), *vals_out = c->vals_b.as<uint32_t>();
    stage_sort_and_dedup<KeyT>(c, keys_in, keys_out, vals_in, vals_out, M);
    ev_end(c, EV_DICT);
    if (c->layout_deferred) pdl_finish_layout(c);   // device input: the genome layout is host work too, and nothing before this point needed it
    if (!only_complexity) pdl_prepare_tasks(c);     // host work + small uploads while the device sorts
    // U (records) and the range count stay on the device until the end of the build: everything below is sized and
    // launched for the bound M and reads the counts there — no host round trip in the middle of the pipeline
    stage_ranges_and_costs(c, M, c->dict_shard.empty() ? 1 : 0, only_complexity);
    if (c->opt_low_memory) {                 // what only the build needed goes back (the sorted k-mer stream with it: pdl_get_dictionary is not available then)
        PDL_HIP(hipStreamSynchronize(c->stream));
        c->keys_a.release(); c->keys_b.release(); c->vals_a.release(); c->vals_b.release(); c->recpos.release(); c->sort_tmp.release();
        if (c->ranges8 == nullptr) {}        // (packed ranges live in `scratch`: it stays)
    }
}

void pdl_run_preprocess(pdl_ctx *c, int kvalue, bool only_complexity) {
    hipStream_t st = c->stream;
    ev_begin(c, EV_PRE_TOTAL);
    c->dist = false; c->dist_stage = 0; c->post_ext = nullptr; c->dist_sender = false;
    stage_alphabet_and_lengths(c, kvalue, only_complexity);
    if (c->key64) dictionary_pipeline<uint64_t>(c, only_complexity);
    else dictionary_pipeline<uint32_t>(c, only_complexity);
    ev_end(c, EV_PRE_TOTAL);
    PDL_HIP(hipStreamSynchronize(st));
    c->tm.hist_ms = ev_ms(c, EV_HIST);
    c->tm.rank_ms = ev_ms(c, EV_RANK);
    c->tm.sort_rank_ms = ev_ms(c, EV_SORT1);
    c->tm.dict_ms = ev_ms(c, EV_DICT);
    c->tm.sort_seq_ms = only_complexity ? 0.f : ev_ms(c, EV_SORT2);
    c->tm.ranges_ms = only_complexity ? 0.f : ev_ms(c, EV_RANGES);
    c->tm.preprocess_total_ms = ev_ms(c, EV_PRE_TOTAL);
    c->tm.dist_begin_ms = c->tm.dist_finish_ms = 0.f;
}

// ------------------------------------------------------------------------------------------------
// Multi-GPU build (include/pandelos_amd.h, pdl_dist_*).  Every GPU ranks all k-mers (the input is shared), keeps the
// ones of its rank interval, and sorts + dedups those: its run of the dictionary.  The caller all-gathers the runs; the
// concatenation in rank order is the dictionary of library.cpp:270-287, so no merge is needed.
// ------------------------------------------------------------------------------------------------
template <class KeyT>
static void dist_slice_pipeline(pdl_ctx *c) {
    hipStream_t st = c->stream;
    const uint64_t M = c->M;
    uint64_t *d_scal = c->scalars.as<uint64_t>();
    const uint32_t W = c->world, me = c->rank;
    // 1. every k-mer's rank, and as a by-product the k-mers per bin of the rank space (top DIST_BIN_BITS bits of a rank)
    ev_begin(c, EV_RANK);
    // (the bins go over the bits a rank can occupy; where ranks wrapped unnoticed AND the reference's sort goes over fewer bytes than
    // they fill, its dictionary is ordered by the low bytes only: intervals of the high bits cannot reproduce that across GPUs)
    if (c->rp.key_bits != c->rp.rank_bits && c->rp.rank_bits <= 56)
        PDL_FAIL(PDL_ERR_UNSUPPORTED, "k-mer ranks wrap past 2^64 unnoticed by the reference's overflow test and its sort covers %u bits of them: one GPU reproduces that order, several do not", c->rp.rank_bits);
    const uint32_t shift = c->rp.key_bits > DIST_BIN_BITS ? c->rp.key_bits - DIST_BIN_BITS : 0;
    c->scratch2.alloc(std::max<size_t>(c->scratch2.bytes, DIST_BINS * sizeof(uint32_t)));
    uint32_t *d_bins = c->scratch2.as<uint32_t>();
    PDL_HIP(hipMemsetAsync(d_bins, 0, DIST_BINS * sizeof(uint32_t), st));
    stage_rank<KeyT>(c, d_bins, shift);
    ev_end(c, EV_RANK);
    std::vector<uint64_t> pre(DIST_BINS + 1, 0);
    {
        PinRead rd(c);
        const uint32_t *pb = rd.add<uint32_t>(d_bins, DIST_BINS);
        rd.sync();
        for (uint32_t b = 0; b < DIST_BINS; b++) pre[b + 1] = pre[b] + pb[b];
    }
    if (pre[DIST_BINS] != M) PDL_FAIL(PDL_ERR_DEVICE, "interval histogram counts %llu k-mers, the stream holds %llu", (unsigned long long) pre[DIST_BINS], (unsigned long long) M);
    // cut w = first bin whose exclusive prefix reaches w * M / W: the same on every rank (function of the input only)
    auto cut = [&](uint32_t w) -> uint32_t {
        if (w == 0) return 0;
        if (w >= W) return DIST_BINS;
        const uint64_t target = (uint64_t) ((unsigned __int128) M * w / W);
        return (uint32_t) (std::lower_bound(pre.begin(), pre.end() - 1, target) - pre.begin());
    };
    const uint32_t b_lo = cut(me), b_hi = cut(me + 1);
    const uint64_t m_own = pre[b_hi] - pre[b_lo];
    c->M_slice = m_own;
    c->dist_tail = -1;                      // the last rank whose interval holds k-mers: it holds the dictionary's last record
    for (uint32_t w = 0; w < W; w++) if (pre[cut(w + 1)] > pre[cut(w)]) c->dist_tail = (int) w;
    // 3. this rank's k-mers, in stream order (the sort below is stable: equal ranks keep ascending gene order)
    KeyT *sel_k = c->keys_b.as<KeyT>();
    uint32_t *sel_v = c->vals_b.as<uint32_t>();
    scan_and_apply(c, M, SelFlag<KeyT>{c->keys_a.as<KeyT>(), shift, b_lo, b_hi},
                   SelApply<KeyT>{c->keys_a.as<KeyT>(), c->vals_a.as<uint32_t>(), sel_k, sel_v}, d_scal + 15);
    // 4. sort + dedup 1/W of the stream; the group-head flag rides in bit 31 of the count
    c->U_slice = 0;
    if (m_own) {
        KeyT *keys_in = sel_k, *keys_out = c->keys_a.as<KeyT>();
        uint32_t *vals_in = sel_v, *vals_out = c->vals_a.as<uint32_t>();
        stage_sort_and_dedup<KeyT>(c, keys_in, keys_out, vals_in, vals_out, m_own);
        // The reference folds the dictionary's LAST record into the group before it (library.cpp:300-306).  That record is in
        // the run of the last rank with k-mers, and so is the group it joins whenever that run has two records or more (groups
        // never straddle runs): the fold is made here, on the run, and travels with it (it is idempotent: the finish that
        // looks at the gathered dictionary finds nothing left to do; a last run of ONE record is left to that finish).
        if ((int) me == c->dist_tail)
            hipLaunchKernelGGL(k_fold_last_record, dim3(1), dim3(1024), 0, st, c->post.as<uint2>(), c->recpos.as<uint32_t>(), d_scal + 0);
        // every genome's lookups inside this run (groups never straddle runs), as the reference counts them and above the
        // diagonal: summed over the ranks the former are "Genome g cost" (exact when the fold above was made), the latter the
        // weights of the genome deal
        GroupTileArgs ga{};
        ga.post = c->post.as<uint2>(); ga.n_bound = m_own; ga.d_n = d_scal + 0;
        const uint32_t grid = group_tiles_plan(c, ga);
        ga.counters = reinterpret_cast<unsigned long long *>(d_scal + 12);      // (scratch words: the real counters come from the finish)
        ga.genome_of = c->d_gen; ga.n_genomes = c->G;
        ga.g_full = reinterpret_cast<unsigned long long *>(d_scal + PDL_CTL_GCOST);             // scratch here, cleared again by the finish
        ga.g_upper = reinterpret_cast<unsigned long long *>(d_scal + PDL_CTL_GCOST) + c->G;
        launch_group_tiles<0, 3, true, false>(c, ga, grid);
        ev_end(c, EV_DICT);
    } else {
        c->post.alloc(16);
    }
}

void pdl_run_dist_begin(pdl_ctx *c, int kvalue) {
    hipStream_t st = c->stream;
    ev_begin(c, EV_PRE_TOTAL);
    ev_begin(c, EV_DIST_BEGIN);
    c->dist = true; c->dist_stage = 0; c->post_ext = nullptr;
    stage_alphabet_and_lengths(c, kvalue, false);
    if (c->key64) dist_slice_pipeline<uint64_t>(c); else dist_slice_pipeline<uint32_t>(c);
    ev_end(c, EV_DIST_BEGIN);
    c->h_run_weights.assign(c->G, 0);
    c->h_run_costs.assign(c->G, 0);
    c->dist_sender = false;
    if (c->M_slice) {
        PinRead rd(c);
        const uint64_t *pu = rd.add<uint64_t>(c->scalars.as<uint64_t>(), 1);
        const uint64_t *pw = rd.add<uint64_t>(c->scalars.as<uint64_t>() + PDL_CTL_GCOST + c->G, c->G);
        const uint64_t *pf = rd.add<uint64_t>(c->scalars.as<uint64_t>() + PDL_CTL_GCOST, c->G);
        rd.sync();
        c->U_slice = pu[0];
        c->h_run_weights.assign(pw, pw + c->G);
        c->h_run_costs.assign(pf, pf + c->G);
    } else {
        PDL_HIP(hipStreamSynchronize(st));
    }
    c->tm.hist_ms = ev_ms(c, EV_HIST);
    c->tm.rank_ms = ev_ms(c, EV_RANK);
    c->tm.sort_rank_ms = c->M_slice ? ev_ms(c, EV_SORT1) : 0.f;
    c->tm.dict_ms = c->M_slice ? ev_ms(c, EV_DICT) : 0.f;
    c->tm.dist_begin_ms = ev_ms(c, EV_DIST_BEGIN);
    c->dist_stage = 1;
}

// longest-processing-time assignment, deterministic (ties: lower genome id first, then lower rank)
static void lpt_owner(const std::vector<uint64_t> &w, uint32_t world, std::vector<uint32_t> &owner) {
    const uint32_t G = (uint32_t) w.size();
    std::vector<uint32_t> order(G);
    for (uint32_t g = 0; g < G; g++) order[g] = g;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return w[a] > w[b]; });
    std::vector<uint64_t> load(world, 0);
    owner.assign(G, 0);
    for (uint32_t g : order) {
        uint32_t best = 0;
        for (uint32_t r = 1; r < world; r++) if (load[r] < load[best]) best = r;
        owner[g] = best;
        load[best] += w[g] + 1;          // (+1: genomes without any lookup still spread over the ranks)
    }
}

void pdl_run_dist_finish(pdl_ctx *c, uint64_t total, const uint64_t *weights) {
    hipStream_t st = c->stream;
    uint64_t *d_scal = c->scalars.as<uint64_t>();
    ev_begin(c, EV_DIST_FINISH);
    if (total == 0) PDL_FAIL(PDL_ERR_EMPTY, "empty dictionary");
    if (total >= 0xfffff000ull) PDL_FAIL(PDL_ERR_UNSUPPORTED, "a dictionary of %llu records needs 64-bit record indices", (unsigned long long) total);
    // the record count of the whole dictionary goes where the kernels expect it (d_scal[0]); the other counters and the
    // per-genome words restart
    uint64_t *h_u = reinterpret_cast<uint64_t *>(c->pin);       // (pinned scratch; rewritten only by the next PinRead, which comes after a sync)
    if (!h_u) PDL_FAIL(PDL_ERR_DEVICE, "pinned scratch missing");
    h_u[0] = total;
    PDL_HIP(hipMemcpyAsync(d_scal + 0, h_u, sizeof(uint64_t), hipMemcpyHostToDevice, st));
    // (words 3-8 belong to K-len and stay: bad-offsets flag, kseq sums, M)
    PDL_HIP(hipMemsetAsync(d_scal + 1, 0, 2 * sizeof(uint64_t), st));
    PDL_HIP(hipMemsetAsync(d_scal + 9, 0, 7 * sizeof(uint64_t), st));
    PDL_HIP(hipMemsetAsync(d_scal + PDL_CTL_GCOST, 0, 2 * (size_t) c->G * sizeof(uint64_t), st));
    // genomes -> ranks by longest-processing-time on each genome's lookups above the diagonal: the caller's sum of the
    // runs' weights, or — without one — an exact pass over the gathered dictionary first
    c->h_upper_cost.assign(c->G, 0);
    if (weights) {
        c->h_upper_cost.assign(weights, weights + c->G);
    } else {
        c->scratch2.alloc(std::max<size_t>(c->scratch2.bytes, ((size_t) c->G + 2) * sizeof(uint64_t)));
        PDL_HIP(hipMemsetAsync(c->scratch2.p, 0, ((size_t) c->G + 2) * sizeof(uint64_t), st));
        GroupTileArgs ga{};
        ga.post = c->post_ext; ga.n_bound = total; ga.d_n = nullptr;
        const uint32_t grid = group_tiles_plan(c, ga);
        ga.counters = c->scratch2.as<unsigned long long>() + c->G;
        ga.genome_of = c->d_gen; ga.n_genomes = c->G;
        ga.g_upper = c->scratch2.as<unsigned long long>();
        ga.g_full = reinterpret_cast<unsigned long long *>(d_scal + PDL_CTL_GCOST);       // (scratch: the K-cost words are cleared below)
        // (the fold of the last record is not applied yet: as with the callers' weights, these numbers only balance)
        launch_group_tiles<0, 3, true, false>(c, ga, grid);
        {
            PinRead rd(c);
            const uint64_t *pu = rd.add<uint64_t>(ga.g_upper, c->G);
            rd.sync();
            c->h_upper_cost.assign(pu, pu + c->G);
        }
        PDL_HIP(hipMemsetAsync(d_scal + PDL_CTL_GCOST, 0, 2 * (size_t) c->G * sizeof(uint64_t), st));
    }
    lpt_owner(c->h_upper_cost, c->world, c->h_owner);
    c->shard.clear();
    for (uint32_t g = 0; g < c->G; g++) if (c->h_owner[g] == c->rank) c->shard.push_back(g);
    c->shard_set = true;
    c->dict_shard = c->shard;
    c->tasks_ready = false;                 // (the task layout is prepared inside, behind the launches of the two passes)
    c->owner_of_genome.alloc((size_t) c->G * 4);
    PDL_HIP(hipMemcpyAsync(c->owner_of_genome.p, c->h_owner.data(), (size_t) c->G * 4, hipMemcpyHostToDevice, st));
    c->seq_in_shard.alloc(c->N);
    hipLaunchKernelGGL(k_genes_of_rank, dim3((c->N + 255) / 256), dim3(256), 0, st, c->d_gen, c->owner_of_genome.as<uint32_t>(), c->rank, c->N,
                       c->seq_in_shard.as<uint8_t>());
    stage_ranges_and_costs(c, total, 2, false);
    ev_end(c, EV_DIST_FINISH);
    ev_end(c, EV_PRE_TOTAL);
    PDL_HIP(hipStreamSynchronize(st));
    c->tm.sort_seq_ms = ev_ms(c, EV_SORT2);
    c->tm.ranges_ms = ev_ms(c, EV_RANGES);
    c->tm.dist_finish_ms = ev_ms(c, EV_DIST_FINISH);
    c->tm.preprocess_total_ms = c->tm.dist_begin_ms + c->tm.dist_finish_ms;
    c->dist_stage = 2;
}

// ---- sender-built range lists ----------------------------------------------------------------------------------------------------
// pdl_run_dist_finish makes every rank walk the WHOLE gathered dictionary twice (COUNT, WRITE) to find the ranges of its own
// genes: work that does not shrink with the number of ranks.  Groups never straddle runs, so everything a range tuple says —
// the gene, where its postings start, how many there are, the gene's own count — is known to the rank that holds the run, up to
// the run's place in the gathered array, which the record counts give.  Here a rank makes the tuples of ALL genes in its run
// (1/world of the records, no "is this gene mine" per record), files them by the rank that owns the gene (one stable radix
// pass on the owner byte: record order survives, so a gene's ranges still arrive in (rank, gene) order when the sources are
// concatenated in rank order), and the owners sort what they receive by gene.
static void dist_deal_genomes(pdl_ctx *c, const uint64_t *weights) {
    hipStream_t st = c->stream;
    c->h_upper_cost.assign(weights, weights + c->G);
    lpt_owner(c->h_upper_cost, c->world, c->h_owner);
    c->shard.clear();
    for (uint32_t g = 0; g < c->G; g++) if (c->h_owner[g] == c->rank) c->shard.push_back(g);
    c->shard_set = true;
    c->dict_shard = c->shard;
    c->tasks_ready = false;
    c->owner_of_genome.alloc((size_t) c->G * 4);
    PDL_HIP(hipMemcpyAsync(c->owner_of_genome.p, c->h_owner.data(), (size_t) c->G * 4, hipMemcpyHostToDevice, st));
}

bool pdl_run_dist_ranges(pdl_ctx *c, const uint64_t *run_records, const uint64_t *weights, const uint64_t *costs) {
    hipStream_t st = c->stream;
    uint64_t *d_scal = c->scalars.as<uint64_t>();
    const uint32_t W = c->world, me = c->rank;
    uint64_t total = 0, base = 0;
    int tail = -1;
    for (uint32_t r = 0; r < W; r++) { if (r < me) base += run_records[r]; total += run_records[r]; if (run_records[r]) tail = (int) r; }
    if (run_records[me] != c->U_slice) PDL_FAIL(PDL_ERR_ARGUMENT, "run_records[%u] = %llu, this rank's run holds %llu records", me, (unsigned long long) run_records[me], (unsigned long long) c->U_slice);
    if (total == 0) PDL_FAIL(PDL_ERR_EMPTY, "empty dictionary");
    if (total >= 0xfffff000ull) PDL_FAIL(PDL_ERR_UNSUPPORTED, "a dictionary of %llu records needs 64-bit record indices", (unsigned long long) total);
    c->dist_total = total; c->run_base = base;
    c->h_tuple_counts.assign(W, 0);
    c->dist_out_keys = nullptr; c->dist_out_ranges = nullptr; c->dist_out_total = 0;
    c->dist_run_counters[0] = c->dist_run_counters[1] = c->dist_run_counters[2] = 0;
    // Not this way (every rank decides the same: all of it is a function of the input and of the record counts): packed ranges
    // need gene ids of 22 bits, the owner a byte; a last run of ONE record has not been folded (the group it joins is in
    // another rank's run).
    c->dist_sender = c->N < (1u << 22) && W <= 256 && tail == c->dist_tail && (run_records[tail] >= 2 || total == 1);
    if (!c->dist_sender) return false;
    ev_begin(c, EV_DIST_RANGES);
    dist_deal_genomes(c, weights);
    c->h_genome_cost.assign(costs, costs + c->G);       // exact: the fold was made before the runs were counted
    c->seq_owner.alloc(c->N);
    hipLaunchKernelGGL(k_gene_owner, dim3((c->N + 255) / 256), dim3(256), 0, st, c->d_gen, c->owner_of_genome.as<uint32_t>(), c->N, c->seq_owner.as<uint8_t>());
    const uint64_t n_run = c->U_slice;
    if (n_run) {
        PDL_HIP(hipMemsetAsync(d_scal + 2, 0, sizeof(uint64_t), st));
        PDL_HIP(hipMemsetAsync(d_scal + 10, 0, 4 * sizeof(uint64_t), st));
        GroupTileArgs ga{};
        ga.post = c->post.as<uint2>(); ga.n_bound = n_run; ga.d_n = d_scal + 0;       // (the run's record count is still where K-rle left it)
        const uint32_t grid = group_tiles_plan(c, ga);
        ga.cost = c->cost.as<unsigned long long>();
        ga.counters = reinterpret_cast<unsigned long long *>(d_scal + 10);
        ga.genome_of = c->d_gen; ga.n_genomes = c->G;
        hipLaunchKernelGGL(k_range_count<1>, dim3(grid), dim3(GW_THREADS), 0, st, ga);
        hipLaunchKernelGGL(k_tile_prefix, dim3((ga.n_blocks + 3) / 4), dim3(256), 0, st, ga.tile_sums, ga.d_n, ga.n_bound, ga.chunk_sums, ga.n_blocks);
        hipLaunchKernelGGL(k_scan_tile_scan, dim3(1), dim3(1024), 0, st, ga.chunk_sums, ga.n_blocks, d_scal + 2, (uint64_t *) nullptr);
        PDL_HIP(hipGetLastError());
        uint64_t n_t = 0;
        {
            PinRead rd(c);
            const uint64_t *pn = rd.add<uint64_t>(d_scal + 2, 1);
            rd.sync();
            n_t = pn[0];
        }
        const uint64_t cap = std::max<uint64_t>(n_t, 1);
        c->scratch.alloc(cap * (2 * sizeof(uint64_t) + 2 * sizeof(uint32_t)) + 64);
        unsigned long long *pay_a = c->scratch.as<unsigned long long>(), *pay_b = pay_a + cap;
        uint32_t *k2a = reinterpret_cast<uint32_t *>(pay_b + cap), *k2b = k2a + cap;
        c->head_bits.alloc(((n_run + GW_TILE - 1) / GW_TILE) * GW_ROUNDS * sizeof(uint64_t));
        ga.head_bits = c->head_bits.as<unsigned long long>();
        ga.key2 = k2a; ga.tuples = nullptr; ga.pay8 = pay_a; ga.pos_base = (uint32_t) base;
        launch_group_tiles<1, 1, false, false>(c, ga, grid);
        if (n_t) {
            hipLaunchKernelGGL(k_owner_keys, dim3((uint32_t) std::min<uint64_t>((n_t + 255) / 256, (uint64_t) c->cus * 16)), dim3(256), 0, st, k2a, d_scal + 2, c->seq_owner.as<uint8_t>());
            pdl_sort_pairs<uint32_t, unsigned long long>(c, k2a, k2b, pay_a, pay_b, n_t, 32, false, nullptr, 24, true);      // -> (k2b, pay_b), by destination
            c->tuple_off.alloc(((size_t) W + 1) * sizeof(uint32_t));
            hipLaunchKernelGGL(k_seq_offsets, dim3((W + 1 + 255) / 256), dim3(256), 0, st, k2b, d_scal + 2, W, c->tuple_off.as<uint32_t>(), 24u, 0xffu);
            PDL_HIP(hipGetLastError());
            c->dist_out_keys = k2b; c->dist_out_ranges = pay_b; c->dist_out_total = n_t;
        }
        PinRead rd(c);
        const uint64_t *pc = rd.add<uint64_t>(d_scal + 10, 4);
        const uint32_t *po = n_t ? rd.add<uint32_t>(c->tuple_off.as<uint32_t>(), W + 1) : nullptr;
        rd.sync();
        c->dist_run_counters[0] = pc[0]; c->dist_run_counters[1] = pc[1]; c->dist_run_counters[2] = pc[3];
        if (po) {
            for (uint32_t r = 0; r < W; r++) c->h_tuple_counts[r] = po[r + 1] - po[r];
            if (po[W] != n_t) PDL_FAIL(PDL_ERR_DEVICE, "range tuples: %llu made, %u filed", (unsigned long long) n_t, po[W]);
        }
    }
    ev_end(c, EV_DIST_RANGES);
    PDL_HIP(hipStreamSynchronize(st));
    c->tm.dist_ranges_ms = ev_ms(c, EV_DIST_RANGES);
    return true;
}

// the owner's side: the gathered dictionary is adopted, the received tuples (source-rank major) are sorted by gene
void pdl_run_dist_finish_ranges(pdl_ctx *c, uint64_t total, uint32_t *d_keys, unsigned long long *d_ranges, uint64_t n_in, const uint64_t *sums) {
    hipStream_t st = c->stream;
    uint64_t *d_scal = c->scalars.as<uint64_t>();
    ev_begin(c, EV_DIST_FINISH);
    if (total != c->dist_total) PDL_FAIL(PDL_ERR_ARGUMENT, "the gathered dictionary holds %llu records, the runs add up to %llu", (unsigned long long) total, (unsigned long long) c->dist_total);
    if (n_in >= 0xfffff000ull) PDL_FAIL(PDL_ERR_UNSUPPORTED, "more than 2^32 range tuples for one rank");
    uint64_t *h_u = reinterpret_cast<uint64_t *>(c->pin);       // (pinned scratch; rewritten only by the next PinRead, which comes after a sync)
    if (!h_u) PDL_FAIL(PDL_ERR_DEVICE, "pinned scratch missing");
    h_u[0] = total; h_u[1] = 0; h_u[2] = n_in;
    PDL_HIP(hipMemcpyAsync(d_scal + 0, h_u, 3 * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    PDL_HIP(hipMemsetAsync(d_scal + 4, 0, sizeof(uint64_t), st));          // (k_genome_cost adds the k-mer statistics up: sum, max, ~min)
    PDL_HIP(hipMemsetAsync(d_scal + 7, 0, 2 * sizeof(uint64_t), st));
    PDL_HIP(hipMemsetAsync(d_scal + 9, 0, 7 * sizeof(uint64_t), st));
    PDL_HIP(hipMemsetAsync(d_scal + PDL_CTL_GCOST, 0, 2 * (size_t) c->G * sizeof(uint64_t), st));
    PDL_HIP(hipMemsetAsync(c->cost.p, 0, (size_t) c->N * sizeof(uint64_t), st));
    ev_begin(c, EV_SORT2);
    const uint32_t seq_bits = std::max<uint32_t>(1, bit_length64(c->N ? c->N - 1 : 0));
    c->seq_off.alloc(((size_t) c->N + 1) * sizeof(uint32_t));
    const uint64_t cap = std::max<uint64_t>(n_in, 1);
    // (the outbox in `scratch` has been delivered: its memory takes the sort's second pair)
    c->scratch.alloc(cap * (sizeof(uint64_t) + sizeof(uint32_t)) + 64);
    unsigned long long *pay_a = d_ranges, *pay_b = c->scratch.as<unsigned long long>();
    uint32_t *k2a = d_keys, *k2b = reinterpret_cast<uint32_t *>(pay_b + cap);
    c->dist_out_keys = nullptr; c->dist_out_ranges = nullptr; c->dist_out_total = 0;
    if (n_in) pdl_sort_pairs<uint32_t, unsigned long long>(c, k2a, k2b, pay_a, pay_b, n_in, seq_bits, false, nullptr, 0, true);      // the gene bits only (gene ids have 22 bits at most: the owner byte lies above every digit)
    else { k2b = k2a; pay_b = pay_a; }
    ev_end(c, EV_SORT2);
    ev_begin(c, EV_RANGES);
    c->ranges8 = reinterpret_cast<const uint2 *>(n_in ? pay_b : c->scratch.as<unsigned long long>());
    hipLaunchKernelGGL(k_seq_offsets, dim3((c->N + 1 + 255) / 256), dim3(256), 0, st, n_in ? k2b : reinterpret_cast<uint32_t *>(c->scratch.p), d_scal + 2, c->N,
                       c->seq_off.as<uint32_t>(), 0u, 0xffffffu);
    PDL_HIP(hipGetLastError());
    ev_end(c, EV_RANGES);
    c->upper_only = true;
    // the k-mer statistics of the genes (the per-gene costs are all zero here: what it adds up per genome is not used)
    hipLaunchKernelGGL(k_genome_cost, dim3(std::min<uint32_t>((c->N + 255) / 256, 128)), dim3(256), 0, st, c->cost.as<unsigned long long>(),
                       c->kseq_len.as<uint32_t>(), c->d_gen, c->N, reinterpret_cast<unsigned long long *>(d_scal + PDL_CTL_GCOST),
                       reinterpret_cast<unsigned long long *>(d_scal + 4), reinterpret_cast<unsigned long long *>(d_scal + 7),
                       reinterpret_cast<unsigned long long *>(d_scal + 8));
    PDL_HIP(hipGetLastError());
    if (!c->tasks_ready) pdl_prepare_tasks(c);           // host work + small uploads while the device sorts
    {
        PinRead rd(c);
        const uint64_t *pt = rd.add<uint64_t>(d_scal, 9);
        const uint32_t *lbe = lookback_error_word(c, rd);
        rd.sync();
        lookback_check(c, lbe);
        c->sum_kseq = pt[4]; c->max_kseq = pt[7]; c->min_kseq = pt[8] == 0 ? 1 : ~pt[8];
    }
    c->U = total; c->Ushared = sums[0]; c->NG = sums[1]; c->Urepeat = sums[2];
    c->P = 0;
    for (uint32_t g : c->shard) c->P += c->h_genome_cost[g];
    c->costs_ready = true;                   // per genome, from the runs' counts; per gene: not kept by this build (pdl_sequence_costs says so)
    ev_end(c, EV_DIST_FINISH);
    ev_end(c, EV_PRE_TOTAL);
    PDL_HIP(hipStreamSynchronize(st));
    c->tm.sort_seq_ms = ev_ms(c, EV_SORT2);
    c->tm.ranges_ms = ev_ms(c, EV_RANGES);
    c->tm.dist_finish_ms = ev_ms(c, EV_DIST_FINISH);
    c->tm.preprocess_total_ms = c->tm.dist_begin_ms + c->tm.dist_ranges_ms + c->tm.dist_finish_ms;
    c->dist_stage = 2;
}
