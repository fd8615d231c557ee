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
