// pdl_sort.hip — K-sort: stable LSD radix sort of (key, u32 value) pairs, hand-written for gfx950.
//
// Counterpart of counting_sort_ext (ig/native/library.cpp:172-187): one stable counting pass per 8-bit
// digit.  The reference copies the whole 16-byte record vector per pass and runs single-threaded; here a
// pass is three launches over 4096-element tiles:
//
//   k_rs_hist     per-tile digit histogram in LDS (wave-coalesced key reads)      -> counts[digit][tile]
//   (scan)        exclusive scan of counts in digit-major order (pdl_scan.h)      -> offs[digit][tile]
//   k_rs_scatter  stable in-tile ranking without atomics on the data path:
//                   a tile is 4 waves x 16 rounds x 64 lanes, ordered wave-major, so every global read is a
//                   coalesced 64-element run; in a round the lanes holding the same digit find each other
//                   with 8 wave ballots (one per digit bit), rank = popcount of equal lanes below + the
//                   wave's running count of that digit (private LDS counters, no contention);
//                 elements are then placed digit-sorted in LDS and written out as coalesced runs to
//                   offs[digit][tile] + position inside the tile's digit run.
//
// HBM traffic per pass and element: key read twice (histogram, scatter), value read once, both written
// once: 3 * sizeof(key) + 8 bytes.  Stability makes the composition of passes an LSD sort and keeps the
// gene order of equal ranks (the k-mer stream is produced in gene order), which is what replaces the
// reference's seq-byte passes (library.cpp:270-274).
#include "pdl_sort.h"
#include "pdl_scan.h"

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / PDL_WAVE;
constexpr int RS_ROUNDS = 16;
constexpr int RS_TILE = RS_THREADS * RS_ROUNDS;           // 4096 elements
constexpr int RS_WAVE_SPAN = PDL_WAVE * RS_ROUNDS;        // elements of one wave inside a tile
constexpr int RS_BINS = 256;
static_assert(RS_THREADS == RS_BINS, "one thread per digit in the tile-level scans");

template <class KeyT>
__device__ __forceinline__ uint32_t rs_digit(KeyT k, uint32_t shift) { return (uint32_t) (k >> shift) & 0xffu; }

template <class KeyT>
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const KeyT *__restrict__ keys, uint64_t n_bound, const uint64_t *d_n, uint32_t shift,
                                                        uint32_t n_tiles, uint32_t *__restrict__ counts) {
    __shared__ uint32_t s_h[RS_BINS];
    const uint64_t n = scan_count(n_bound, d_n);         // the count may live on the device (grid sized for the bound)
    const uint64_t base = (uint64_t) blockIdx.x * RS_TILE;
    if (base >= n) {                                     // (uniform) tile past the end
        counts[(size_t) threadIdx.x * n_tiles + blockIdx.x] = 0;
        return;
    }
    s_h[threadIdx.x] = 0;
    pdl_sync();
    KeyT key[RS_ROUNDS];
#pragma unroll
    for (int j = 0; j < RS_ROUNDS; j++) {            // all sixteen loads in flight before the first LDS atomic
        const uint64_t i = base + (uint64_t) j * RS_THREADS + threadIdx.x;
        key[j] = keys[i < n ? i : n - 1];
    }
#pragma unroll
    for (int j = 0; j < RS_ROUNDS; j++) {
        const uint64_t i = base + (uint64_t) j * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&s_h[rs_digit(key[j], shift)], 1u);
    }
    pdl_sync();
    counts[(size_t) threadIdx.x * n_tiles + blockIdx.x] = s_h[threadIdx.x];
}

struct RsCountFlag {
    const uint32_t *counts;
    __device__ uint32_t operator()(uint64_t i) const { return counts[i]; }
};
struct RsOffsApply {
    uint32_t *offs;
    __device__ void operator()(uint64_t i, uint32_t, uint32_t prefix) const { offs[i] = prefix; }
};

template <class KeyT, class ValT>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const KeyT *__restrict__ keys_in, const ValT *vals_in,
                                                           KeyT *__restrict__ keys_out, ValT *__restrict__ vals_out, uint64_t n_bound,
                                                           const uint64_t *d_n, uint32_t shift, uint32_t n_tiles, const uint32_t *__restrict__ offs,
                                                           uint32_t nbits) {       // significant bits of this pass's digit (the last pass of a sort may have fewer than 8)
    const uint64_t n = scan_count(n_bound, d_n);
    if ((uint64_t) blockIdx.x * (RS_THREADS * RS_ROUNDS) >= n) return;       // (uniform) tile past the end
    __shared__ KeyT s_key[RS_TILE];
    __shared__ ValT s_val[RS_TILE];
    __shared__ uint32_t s_cnt[RS_WAVES][RS_BINS];     // per wave: running count of each digit, then its base inside the tile
    __shared__ uint32_t s_tile_off[RS_BINS];          // start of each digit's run inside the tile
    __shared__ uint32_t s_goff[RS_BINS];              // global start of this tile's run of each digit
    __shared__ uint32_t s_wsum[17];

    const uint32_t tid = threadIdx.x, lane = tid & (PDL_WAVE - 1), wave = tid / PDL_WAVE;
    const uint64_t tile_base = (uint64_t) blockIdx.x * RS_TILE;
    const uint64_t wave_base = tile_base + (uint64_t) wave * RS_WAVE_SPAN;
    for (int w = 0; w < RS_WAVES; w++) s_cnt[w][tid] = 0;
    s_goff[tid] = offs[(size_t) tid * n_tiles + blockIdx.x];
    pdl_sync();

    KeyT key[RS_ROUNDS];
    ValT val[RS_ROUNDS];
    uint16_t rank[RS_ROUNDS];          // position among the wave's elements of the same digit
    // every load of the tile first, branch-free (clamped index): thirty-two loads in flight per lane.  Behind a branch or
    // an LDS store the compiler keeps program order and would wait for each round's load before ranking it.
#pragma unroll
    for (int j = 0; j < RS_ROUNDS; j++) {
        const uint64_t i = wave_base + (uint64_t) j * PDL_WAVE + lane;
        key[j] = keys_in[i < n ? i : n - 1];
    }
    if (vals_in) {                                   // uniform
#pragma unroll
        for (int j = 0; j < RS_ROUNDS; j++) {
            const uint64_t i = wave_base + (uint64_t) j * PDL_WAVE + lane;
            val[j] = vals_in[i < n ? i : n - 1];
        }
    } else {                                         // the values are the positions 0, 1, 2, ...
#pragma unroll
        for (int j = 0; j < RS_ROUNDS; j++) val[j] = (ValT) (wave_base + (uint64_t) j * PDL_WAVE + lane);
    }
#pragma unroll
    for (int j = 0; j < RS_ROUNDS; j++) {
        const uint64_t i = wave_base + (uint64_t) j * PDL_WAVE + lane;
        const bool valid = i < n;
        const uint32_t d = rs_digit(key[j], shift);
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            if ((uint32_t) b >= nbits) break;                       // (uniform) the bits above are zero in every key
            const unsigned long long m = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? m : ~m;
        }
        // `same` = valid lanes of this wave holding digit d in this round
        const uint32_t before = s_cnt[wave][d];                     // earlier rounds (own wave only: no race)
        const uint32_t below_me = __builtin_amdgcn_mbcnt_hi((uint32_t) (same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) same, 0u));     // lanes of the set below this one
        rank[j] = (uint16_t) (before + below_me);
        if (valid && below_me == 0) s_cnt[wave][d] = before + (uint32_t) __popcll(same);   // lowest lane of the set
    }
    pdl_sync();
    // tile-level layout: digit runs in digit order, inside a run wave 0's elements first
    uint32_t tot = 0;
    uint32_t wcnt[RS_WAVES];
#pragma unroll
    for (int w = 0; w < RS_WAVES; w++) { wcnt[w] = s_cnt[w][tid]; tot += wcnt[w]; }
    uint32_t tile_total;
    const uint32_t ex = block_exclusive_scan_u32(tot, s_wsum, tile_total);
    s_tile_off[tid] = ex;
    uint32_t run = ex;
#pragma unroll
    for (int w = 0; w < RS_WAVES; w++) { s_cnt[w][tid] = run; run += wcnt[w]; }
    pdl_sync();
#pragma unroll
    for (int j = 0; j < RS_ROUNDS; j++) {
        const uint64_t i = wave_base + (uint64_t) j * PDL_WAVE + lane;
        if (i < n) {
            const uint32_t lp = s_cnt[wave][rs_digit(key[j], shift)] + rank[j];
            s_key[lp] = key[j];
            s_val[lp] = val[j];
        }
    }
    pdl_sync();
#pragma unroll
    for (int j = 0; j < RS_ROUNDS; j++) {
        const uint32_t e = j * RS_THREADS + tid;                      // coalesced over the digit-sorted tile
        if (e < tile_total) {
            const KeyT k = s_key[e];
            const uint32_t d = rs_digit(k, shift);
            const uint64_t dst = (uint64_t) s_goff[d] + (e - s_tile_off[d]);
            keys_out[dst] = k;
            vals_out[dst] = s_val[e];
        }
    }
}

template <class KeyT, class ValT>
void pdl_sort_pairs(pdl_ctx *c, KeyT *&keys_in, KeyT *&keys_out, ValT *&vals_in, ValT *&vals_out,
                    uint64_t n, uint32_t end_bit, bool iota_values, const uint64_t *d_n, uint32_t begin_bit, bool keys_below_end_bit) {
    if (n == 0) return;
    if (end_bit == 0) end_bit = 1;
    if (end_bit > sizeof(KeyT) * 8) end_bit = sizeof(KeyT) * 8;
    if (n >= 0xfffff000ull) PDL_FAIL(PDL_ERR_UNSUPPORTED, "sort of %llu pairs needs 64-bit positions", (unsigned long long) n);
    const uint32_t n_tiles = (uint32_t) ((n + RS_TILE - 1) / RS_TILE);
    const size_t table = (size_t) RS_BINS * n_tiles;
    c->sort_tmp.alloc(2 * table * sizeof(uint32_t));
    uint32_t *counts = c->sort_tmp.as<uint32_t>(), *offs = counts + table;
    uint64_t *d_total = c->scalars.as<uint64_t>() + 15;
    const uint32_t passes = (end_bit + 7) / 8;
    for (uint32_t p = begin_bit / 8; p < passes; p++) {
        const uint32_t shift = p * 8;
        hipLaunchKernelGGL((k_rs_hist<KeyT>), dim3(n_tiles), dim3(RS_THREADS), 0, c->stream, keys_in, n, d_n, shift, n_tiles, counts);
        scan_and_apply(c, table, RsCountFlag{counts}, RsOffsApply{offs}, d_total);
        hipLaunchKernelGGL((k_rs_scatter<KeyT, ValT>), dim3(n_tiles), dim3(RS_THREADS), 0, c->stream, keys_in,
                           (p == begin_bit / 8 && iota_values) ? (const ValT *) nullptr : vals_in, keys_out, vals_out, n, d_n, shift, n_tiles, offs,
                           keys_below_end_bit ? std::min<uint32_t>(8, end_bit - shift) : 8u);
        PDL_HIP(hipGetLastError());
        std::swap(keys_in, keys_out);
        std::swap(vals_in, vals_out);
    }
    // the sorted pairs are in the buffers the last pass wrote = (keys_in, vals_in) after the swap; hand them
    // back as the "out" pair
    std::swap(keys_in, keys_out);
    std::swap(vals_in, vals_out);
}

void pdl_radix_offsets(pdl_ctx *c, const uint32_t *counts, uint32_t *offs, uint32_t n_tiles, uint64_t *d_total) {
    static_assert(PDL_RADIX_TILE == RS_TILE && PDL_RADIX_BINS == RS_BINS, "the caller's kernels use the tile shape of the sort");
    scan_and_apply(c, (size_t) RS_BINS * n_tiles, RsCountFlag{counts}, RsOffsApply{offs}, d_total);
}

template void pdl_sort_pairs<uint32_t, uint32_t>(pdl_ctx *, uint32_t *&, uint32_t *&, uint32_t *&, uint32_t *&, uint64_t, uint32_t, bool, const uint64_t *, uint32_t, bool);
template void pdl_sort_pairs<uint64_t, uint32_t>(pdl_ctx *, uint64_t *&, uint64_t *&, uint32_t *&, uint32_t *&, uint64_t, uint32_t, bool, const uint64_t *, uint32_t, bool);
template void pdl_sort_pairs<uint32_t, unsigned long long>(pdl_ctx *, uint32_t *&, uint32_t *&, unsigned long long *&, unsigned long long *&, uint64_t, uint32_t, bool, const uint64_t *, uint32_t, bool);
