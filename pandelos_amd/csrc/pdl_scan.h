// pdl_scan.h — device-wide "flag → exclusive prefix → apply" in three launches.
//
// Used for every compaction of the dictionary build (record heads, group heads, shared records,
// per-row cell offsets).  The flag is recomputed from its inputs in both passes instead of being
// materialised, so the only extra HBM traffic is one u32 per 2048-element tile.
//
//   pass 1  k_scan_tile_sums   tile sums of flag(i)                 -> tile_sums[t]
//   pass 2  k_scan_tile_scan   exclusive scan of the tile sums (one workgroup), total -> *d_total
//   pass 3  k_scan_apply       per tile: flags staged in LDS, workgroup exclusive scan, per-item prefixes
//                              back through LDS so that apply(i, flag, exclusive_prefix) runs lane-coalesced
//
// FlagF :  __device__ uint32_t operator()(uint64_t i) const      (any small count, not only 0/1)
// ApplyF:  __device__ void operator()(uint64_t i, uint32_t flag, uint32_t exclusive_prefix) const
#pragma once

#include "pdl_common.h"

#include <type_traits>

// an apply functor with a nested type Loaded follows the two-phase protocol (see k_scan_apply)
template <class T, class = void> struct scan_two_phase : std::false_type {};
template <class T> struct scan_two_phase<T, std::void_t<typename T::Loaded>> : std::true_type {};

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;   // 2048

__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v) {
    const int lane = threadIdx.x & (PDL_WAVE - 1);
#pragma unroll
    for (int d = 1; d < PDL_WAVE; d <<= 1) {
        uint32_t o = __shfl_up(v, d, PDL_WAVE);
        if (lane >= d) v += o;
    }
    return v;
}

// exclusive prefix of `v` over the workgroup (blockDim.x multiple of 64, <= 1024); returns the
// exclusive prefix for this thread and the workgroup total through `total`.
__device__ __forceinline__ uint32_t block_exclusive_scan_u32(uint32_t v, uint32_t *s_wave /* [17] */, uint32_t &total) {
    const int lane = threadIdx.x & (PDL_WAVE - 1);
    const int wave = threadIdx.x / PDL_WAVE;
    const int nw = blockDim.x / PDL_WAVE;
    uint32_t inc = wave_inclusive_scan_u32(v);
    if (lane == PDL_WAVE - 1) s_wave[wave] = inc;
    pdl_sync();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int w = 0; w < nw; w++) { uint32_t t = s_wave[w]; s_wave[w] = run; run += t; }
        s_wave[16] = run;
    }
    pdl_sync();
    uint32_t res = inc - v + s_wave[wave];
    total = s_wave[16];
    pdl_sync();
    return res;
}

// n = the element count; when d_n is set the count lives on the device (*d_n, at most n) and the grid was sized for the
// bound n: the host does not have to read a count back before it can launch the kernels that depend on it.
__device__ __forceinline__ uint64_t scan_count(uint64_t n, const uint64_t *d_n) {
    if (!d_n) return n;
    const uint64_t v = *d_n;
    return v < n ? v : n;
}

template <class FlagF>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tile_sums(FlagF flag, uint64_t n_bound, const uint64_t *d_n, uint32_t *tile_sums) {
    __shared__ uint32_t s_wave[17];
    const uint64_t n = scan_count(n_bound, d_n);
    if ((uint64_t) blockIdx.x * SCAN_TILE >= n) {        // (uniform) tile past the end: contributes nothing
        if (threadIdx.x == 0) tile_sums[blockIdx.x] = 0;
        return;
    }
    const uint64_t base = (uint64_t) blockIdx.x * SCAN_TILE;
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {           // branch-free (index clamped, result masked): the loads of all items overlap
        const uint64_t i = base + (uint64_t) j * SCAN_THREADS + threadIdx.x;
        const uint32_t f = flag(i < n ? i : n - 1);
        sum += i < n ? f : 0u;
    }
    uint32_t total;
    (void) block_exclusive_scan_u32(sum, s_wave, total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// one workgroup of 1024 threads; tile_sums becomes its own exclusive scan
static __global__ __launch_bounds__(1024) void k_scan_tile_scan(uint32_t *tile_sums, uint32_t n_tiles, uint64_t *d_total, uint64_t *d_total2) {
    __shared__ uint32_t s_wave[17];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    pdl_sync();
    for (uint32_t base = 0; base < n_tiles; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < n_tiles ? tile_sums[i] : 0;
        uint32_t total;
        uint32_t ex = block_exclusive_scan_u32(v, s_wave, total);
        uint32_t carry = s_carry;
        if (i < n_tiles) tile_sums[i] = ex + carry;
        pdl_sync();
        if (threadIdx.x == 0) s_carry = carry + total;
        pdl_sync();
    }
    if (threadIdx.x == 0) {
        *d_total = s_carry;
        if (d_total2) {                      // second copy of the total, as two words: its address may be only 4-byte aligned
            reinterpret_cast<uint32_t *>(d_total2)[0] = s_carry;
            reinterpret_cast<uint32_t *>(d_total2)[1] = 0u;
        }
    }
}

// INLINE_PREFIX (short scans, <= SCAN_INLINE_TILES tiles): tile_sums holds the raw sums and every workgroup adds up the
// ones before its own tile itself (a few hundred L2-resident words) — the one-workgroup launch between the two passes,
// which is mostly launch latency, is gone; workgroup 0 writes the grand total.
constexpr uint32_t SCAN_INLINE_TILES = 1024;
template <class FlagF, class ApplyF, bool INLINE_PREFIX>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(FlagF flag, ApplyF apply, uint64_t n_bound, const uint64_t *d_n, const uint32_t *tile_sums,
                                                            uint64_t *d_total, uint64_t *d_total2) {
    __shared__ uint32_t s_flags[SCAN_TILE];
    __shared__ uint32_t s_pref[SCAN_TILE];
    __shared__ uint32_t s_wave[17];
    const uint64_t n = scan_count(n_bound, d_n);
    const uint64_t base = (uint64_t) blockIdx.x * SCAN_TILE;
    uint32_t tile_prefix = 0;
    if constexpr (INLINE_PREFIX) {
        const uint32_t upto = blockIdx.x == 0 ? gridDim.x : blockIdx.x;      // workgroup 0 adds up everything: the total
        uint32_t part = 0;
        for (uint32_t i = threadIdx.x; i < upto; i += SCAN_THREADS) part += tile_sums[i];
        uint32_t all;
        (void) block_exclusive_scan_u32(part, s_wave, all);
        if (blockIdx.x == 0) {
            if (threadIdx.x == 0) {
                *d_total = all;
                if (d_total2) { reinterpret_cast<uint32_t *>(d_total2)[0] = all; reinterpret_cast<uint32_t *>(d_total2)[1] = 0u; }
            }
        } else {
            tile_prefix = all;
        }
    } else {
        tile_prefix = tile_sums[blockIdx.x];
    }
    if (base >= n) return;                               // (uniform)
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {           // coalesced flag evaluation
        const uint32_t li = j * SCAN_THREADS + threadIdx.x;
        const uint64_t i = base + li;
        const uint32_t f = flag(i < n ? i : n - 1);
        s_flags[li] = i < n ? f : 0u;
    }
    pdl_sync();
    uint32_t f[SCAN_ITEMS];
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) { f[j] = s_flags[threadIdx.x * SCAN_ITEMS + j]; sum += f[j]; }
    uint32_t total;
    uint32_t prefix = block_exclusive_scan_u32(sum, s_wave, total) + tile_prefix;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) { s_pref[threadIdx.x * SCAN_ITEMS + j] = prefix; prefix += f[j]; }
    pdl_sync();
    if constexpr (scan_two_phase<ApplyF>::value) {
        // the functor splits into load(i, flag) -> Loaded and store(i, flag, prefix, Loaded): every load of the tile's
        // items is issued before the first store, so the items' dependent load chains overlap (behind a store the
        // compiler must assume aliasing and would run the items one after the other)
        typename ApplyF::Loaded v[SCAN_ITEMS];
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) {
            const uint32_t li = j * SCAN_THREADS + threadIdx.x;
            const uint64_t i = base + li;
            v[j] = apply.load(i < n ? i : n - 1, s_flags[li]);
        }
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) {
            const uint32_t li = j * SCAN_THREADS + threadIdx.x;
            const uint64_t i = base + li;
            if (i < n) apply.store(i, s_flags[li], s_pref[li], v[j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) {       // apply in the same coalesced order the flags were read in
            uint32_t li = j * SCAN_THREADS + threadIdx.x;
            uint64_t i = base + li;
            if (i < n) apply(i, s_flags[li], s_pref[li]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// ONE launch: flags, prefix and apply in a single pass over the data (the three-launch form above evaluates every flag twice
// and pays two or three dispatch boundaries per scan; a build makes about ten scans).  Workgroup b takes tile b — the hardware
// dispatcher hands out workgroups in ascending order, so every tile before it belongs to a workgroup that has been started (a
// ticket counter would guarantee it, and costs ~100 us for 8 500 tiles: one word serves ~90 atomics per microsecond) — adds up
// its flags, and learns the sum of all tiles before it by DECOUPLED LOOK-BACK on two levels:
//   tile_agg[t]     the tile's own sum, published as soon as it is known
//   chunk_agg[c]    the sum of a chunk of 64 tiles, published by the chunk's last tile (which reads the 63 before it anyway)
//   chunk_incl[c]   the sum through chunk c, published by the chunk's last tile once it knows its own prefix, and by the first
//                   tile of chunk c + 1
// A tile reads the (at most 63) tiles before it in its chunk with one wave-wide load, then walks back over the chunks 64 at a
// time — an INCL where one shows, AGGs up to there — until it meets an INCL or chunk 0.  With ~2000 tiles in flight when the
// launch starts a one-level look-back walks back over all of them (tens of microseconds on a launch that lasts about as
// long); this way it is two or three round trips.
// Every word is {launch epoch : 32 | value : 32}, written ONCE per launch with one agent-scope atomic store and read with
// agent-scope atomic loads: self-contained (epoch and value arrive together), the hand-off form MI355X_MICROARCH.md lists as
// valid between workgroups; words of earlier launches carry another epoch, so nothing is cleared between launches.  Nobody
// waits before its own tile_agg is out, and chunk_agg depends on tile_aggs only: no wait can close a cycle.  Polls are bounded
// all the same (and that is what stands behind the dispatch-order assumption): a word that never shows up raises the error
// counter — the host fails the call, option "onepass_scan" 0 brings the three-launch form back — instead of hanging the device.
// ------------------------------------------------------------------------------------------------------------------------
constexpr uint32_t LB_CHUNK = PDL_WAVE;
constexpr uint32_t LB_MAX_POLLS = 1u << 22;
constexpr uint32_t LB_CTR_WORDS = 16;              // [2] polls that gave up
struct LookbackArgs {
    unsigned long long *tile_agg, *chunk_agg, *chunk_incl;
    uint32_t *ctr;
    uint32_t epoch;
};
__device__ __forceinline__ unsigned long long lb_pack(uint32_t epoch, uint32_t value) { return (unsigned long long) epoch << 32 | value; }
__device__ __forceinline__ void lb_publish(unsigned long long *p, uint32_t epoch, uint32_t value) {
    __hip_atomic_store(p, lb_pack(epoch, value), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the value at *p once the word belongs to this launch (bounded poll)
__device__ __forceinline__ uint32_t lb_wait(const unsigned long long *p, uint32_t epoch, uint32_t *err) {
    for (uint32_t polls = 0;; polls++) {
        const unsigned long long w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t) (w >> 32) == epoch) return (uint32_t) w;
        if (polls >= LB_MAX_POLLS) { atomicAdd(err, 1u); return 0u; }
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ uint32_t lb_wave_sum(uint32_t v) {
#pragma unroll
    for (int d = PDL_WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, PDL_WAVE);
    return v;
}
// Sum of the values of all tiles before `tile`; publishes this tile's value and what falls to it of the chunk level.
// Called by ONE full wave (all 64 lanes active, same arguments); returns the same value in every lane.
__device__ __forceinline__ uint32_t lb_tile_prefix(const LookbackArgs &lb, uint32_t tile, uint32_t tiles, uint32_t total) {
    const uint32_t lane = threadIdx.x & (PDL_WAVE - 1);
    const uint32_t c = tile / LB_CHUNK, i = tile % LB_CHUNK;
    uint32_t *err = lb.ctr + 2;
    if (lane == 0) lb_publish(&lb.tile_agg[tile], lb.epoch, total);
    const uint32_t in_chunk = min(LB_CHUNK, tiles - c * LB_CHUNK);
    const bool completes = i == in_chunk - 1;                // (uniform) the chunk's last tile publishes the chunk's sum
    // the tiles before this one in its chunk: one wave-wide load
    const uint32_t v = lane < i ? lb_wait(&lb.tile_agg[c * LB_CHUNK + lane], lb.epoch, err) : 0u;
    const uint32_t within = lb_wave_sum(v);
    const uint32_t chunk_sum = within + total;
    if (completes && lane == 0) lb_publish(&lb.chunk_agg[c], lb.epoch, chunk_sum);
    // the chunks before this one, nearest first, 64 per step
    uint32_t before = 0;
    for (uint32_t base = c; base > 0;) {
        const bool live = lane < base;
        const uint32_t cc = live ? base - 1 - lane : 0u;
        const unsigned long long wi = live ? __hip_atomic_load(&lb.chunk_incl[cc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        const bool has_incl = live && (uint32_t) (wi >> 32) == lb.epoch;
        const unsigned long long mi = __ballot(has_incl);
        const uint32_t first = mi ? (uint32_t) __ffsll((long long) mi) - 1u : PDL_WAVE;        // nearest chunk that shows its INCL
        uint32_t part = 0;
        if (live && lane < first) part = lb_wait(&lb.chunk_agg[cc], lb.epoch, err);          // the chunks nearer than that: their own sums
        if (lane == first) part = (uint32_t) wi;
        before += lb_wave_sum(part);
        if (mi || base <= PDL_WAVE) break;
        base -= PDL_WAVE;
    }
    if (lane == 0) {
        if (i == 0 && c > 0) lb_publish(&lb.chunk_incl[c - 1], lb.epoch, before);
        if (completes) lb_publish(&lb.chunk_incl[c], lb.epoch, before + chunk_sum);
    }
    return before + within;
}
template <class FlagF, class ApplyF>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_onepass(FlagF flag, ApplyF apply, uint64_t n_bound, const uint64_t *d_n, LookbackArgs lb,
                                                               uint32_t n_chunks_bound, uint64_t *d_total, uint64_t *d_total2) {
    __shared__ uint32_t s_flags[SCAN_TILE];
    __shared__ uint32_t s_pref[SCAN_TILE];
    __shared__ uint32_t s_wave[17];
    __shared__ uint32_t s_prefix;
    const uint64_t n = scan_count(n_bound, d_n);
    const uint32_t tiles = (uint32_t) ((n + SCAN_TILE - 1) / SCAN_TILE);
    const uint32_t tile = blockIdx.x;
    if (tile < tiles) {                                   // (uniform)
        const uint64_t base = (uint64_t) tile * SCAN_TILE;
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) {           // coalesced flag evaluation
            const uint32_t li = j * SCAN_THREADS + threadIdx.x;
            const uint64_t i = base + li;
            const uint32_t f = flag(i < n ? i : n - 1);
            s_flags[li] = i < n ? f : 0u;
        }
        pdl_sync();
        uint32_t f[SCAN_ITEMS];
        uint32_t sum = 0;
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) { f[j] = s_flags[threadIdx.x * SCAN_ITEMS + j]; sum += f[j]; }
        uint32_t total;
        uint32_t prefix = block_exclusive_scan_u32(sum, s_wave, total);
        if (threadIdx.x < PDL_WAVE) {                    // (the first wave, whole)
            const uint32_t before = lb_tile_prefix(lb, tile, tiles, total);
            if (threadIdx.x == 0) {
                s_prefix = before;
                if (tile == tiles - 1) {
                    *d_total = (uint64_t) before + total;
                    if (d_total2) { reinterpret_cast<uint32_t *>(d_total2)[0] = before + total; reinterpret_cast<uint32_t *>(d_total2)[1] = 0u; }
                }
            }
        }
        pdl_sync();
        prefix += s_prefix;
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) { s_pref[threadIdx.x * SCAN_ITEMS + j] = prefix; prefix += f[j]; }
        pdl_sync();
        if constexpr (scan_two_phase<ApplyF>::value) {
            typename ApplyF::Loaded v[SCAN_ITEMS];
#pragma unroll
            for (int j = 0; j < SCAN_ITEMS; j++) {
                const uint32_t li = j * SCAN_THREADS + threadIdx.x;
                const uint64_t i = base + li;
                v[j] = apply.load(i < n ? i : n - 1, s_flags[li]);
            }
#pragma unroll
            for (int j = 0; j < SCAN_ITEMS; j++) {
                const uint32_t li = j * SCAN_THREADS + threadIdx.x;
                const uint64_t i = base + li;
                if (i < n) apply.store(i, s_flags[li], s_pref[li], v[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < SCAN_ITEMS; j++) {
                uint32_t li = j * SCAN_THREADS + threadIdx.x;
                uint64_t i = base + li;
                if (i < n) apply(i, s_flags[li], s_pref[li]);
            }
        }
    } else if (tiles == 0 && tile == 0 && threadIdx.x == 0) {          // (a count of zero on the device: the total is zero)
        *d_total = 0;
        if (d_total2) { reinterpret_cast<uint32_t *>(d_total2)[0] = 0u; reinterpret_cast<uint32_t *>(d_total2)[1] = 0u; }
    }
}

// The look-back state of a context: grown (and cleared once) as needed, a new epoch per launch.
inline LookbackArgs lookback_for(pdl_ctx *c, uint64_t tiles, size_t words_per_tile = 1) {
    const uint64_t chunks = (tiles + LB_CHUNK - 1) / LB_CHUNK;
    const size_t need_tile = (size_t) tiles * words_per_tile * 8, need_chunk = (size_t) chunks * words_per_tile * 8 * 2, need_ctr = LB_CTR_WORDS * 4;
    auto grow = [&](DevBuf &b, size_t bytes) {
        if (b.bytes >= bytes && b.p) return;
        b.alloc(bytes + bytes / 4 + 256);
        PDL_HIP(hipMemsetAsync(b.p, 0, b.bytes, c->stream));       // (epoch 0 never is a launch's)
    };
    grow(c->lb_tile, need_tile); grow(c->lb_chunk, need_chunk); grow(c->lb_ctr, need_ctr);
    LookbackArgs lb{};
    lb.tile_agg = c->lb_tile.as<unsigned long long>();
    lb.chunk_agg = c->lb_chunk.as<unsigned long long>();
    lb.chunk_incl = lb.chunk_agg + (size_t) chunks * words_per_tile;
    lb.ctr = c->lb_ctr.as<uint32_t>();
    lb.epoch = ++c->lb_epoch;
    if (c->lb_epoch == 0xffffffffu) {                   // (after 4 x 10^9 launches: start over on cleared words)
        PDL_HIP(hipMemsetAsync(c->lb_tile.p, 0, c->lb_tile.bytes, c->stream)); PDL_HIP(hipMemsetAsync(c->lb_chunk.p, 0, c->lb_chunk.bytes, c->stream));
        c->lb_epoch = 0; lb.epoch = ++c->lb_epoch;
    }
    return lb;
}

// The host's look at the look-back error counter, beside whatever else it reads back at the end of a stage:
//   const uint32_t *e = lookback_error_word(c, rd); rd.sync(); lookback_check(c, e);
inline const uint32_t *lookback_error_word(pdl_ctx *c, PinRead &rd) { return c->lb_ctr.p ? rd.add<uint32_t>(c->lb_ctr.as<uint32_t>() + 2, 1) : nullptr; }
inline void lookback_check(pdl_ctx *c, const uint32_t *word) {
    if (!word || *word == 0) return;
    const uint32_t n = *word;
    (void) hipMemsetAsync(c->lb_ctr.as<uint32_t>() + 2, 0, 4, c->stream);
    PDL_FAIL(PDL_ERR_DEVICE, "decoupled look-back: %u polls gave up (a tile's sum never arrived); set option onepass_scan 0", n);
}

// Host wrapper.  d_total receives the grand total (u64).  scan_tmp is grown as needed.
template <class FlagF, class ApplyF>
inline void scan_and_apply(pdl_ctx *c, uint64_t n, FlagF flag, ApplyF apply, uint64_t *d_total, uint64_t *d_total2 = nullptr,
                           const uint64_t *d_n = nullptr /* count on the device, <= n */) {
    if (n == 0) {
        PDL_HIP(hipMemsetAsync(d_total, 0, sizeof(uint64_t), c->stream));
        if (d_total2) PDL_HIP(hipMemsetAsync(d_total2, 0, sizeof(uint64_t), c->stream));
        return;
    }
    const uint64_t tiles64 = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (tiles64 > 0x7fffffffull) PDL_FAIL(PDL_ERR_UNSUPPORTED, "scan of %llu elements exceeds the grid limit", (unsigned long long) n);
    const uint32_t tiles = (uint32_t) tiles64;
    if (c->opt_onepass_scan) {
        const LookbackArgs lb = lookback_for(c, tiles);
        hipLaunchKernelGGL((k_scan_onepass<FlagF, ApplyF>), dim3(tiles), dim3(SCAN_THREADS), 0, c->stream, flag, apply, n, d_n, lb,
                           (uint32_t) ((tiles + LB_CHUNK - 1) / LB_CHUNK), d_total, d_total2);
        PDL_HIP(hipGetLastError());
        return;
    }
    c->scan_tmp.alloc((size_t) tiles * sizeof(uint32_t));
    uint32_t *ts = c->scan_tmp.as<uint32_t>();
    hipLaunchKernelGGL((k_scan_tile_sums<FlagF>), dim3(tiles), dim3(SCAN_THREADS), 0, c->stream, flag, n, d_n, ts);
    if (tiles <= SCAN_INLINE_TILES) {
        hipLaunchKernelGGL((k_scan_apply<FlagF, ApplyF, true>), dim3(tiles), dim3(SCAN_THREADS), 0, c->stream, flag, apply, n, d_n, ts, d_total, d_total2);
    } else {
        hipLaunchKernelGGL(k_scan_tile_scan, dim3(1), dim3(1024), 0, c->stream, ts, tiles, d_total, d_total2);
        hipLaunchKernelGGL((k_scan_apply<FlagF, ApplyF, false>), dim3(tiles), dim3(SCAN_THREADS), 0, c->stream, flag, apply, n, d_n, ts, d_total, d_total2);
    }
    PDL_HIP(hipGetLastError());
}
