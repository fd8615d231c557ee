// pdl_sort.h — stable LSD radix sort of (key, value) pairs on a bit range of the key.
//
// Counterpart of counting_sort_ext (ig/native/library.cpp:172-187) and its call sites (:270-278):
// the reference sorts 16-byte records by the bytes of seq then the bytes of rank; here the k-mer
// stream is produced in gene order, so ONE stable sort on the rank bits alone yields the same
// (rank, gene) order, and a second stable sort of the deduplicated records on the gene bits yields
// the per-gene lists.
#pragma once

#include "pdl_common.h"

// Sorts n pairs by key bits [begin_bit, end_bit) (begin_bit a multiple of 8: the passes below it were made by
// the caller, see pdl_radix_offsets).  Input in (*keys_in, *vals_in); on return the sorted
// pairs are in (*keys_out, *vals_out) — the function may swap the roles of the buffers, the
// pointers passed by reference are updated accordingly (no pass to make: the input pair is handed back as the output).
template <class KeyT, class ValT = uint32_t>       // ValT: uint32_t, or unsigned long long (an 8-byte payload carried along)
void pdl_sort_pairs(pdl_ctx *c, KeyT *&keys_in, KeyT *&keys_out, ValT *&vals_in, ValT *&vals_out,
                    uint64_t n, uint32_t end_bit, bool iota_values = false,   // iota_values: the values are 0, 1, 2, ... (vals_in is not read)
                    const uint64_t *d_n = nullptr,                            // d_n: the count lives on the device (*d_n <= n, grids sized for n)
                    uint32_t begin_bit = 0,
                    bool keys_below_end_bit = false);                         // every key < 2^end_bit (or: nothing but zeros between end_bit and the end of the last digit): a pass
                                                                              // then matches the significant bits of its digit only; else all eight (ranks that wrapped, see RankParams)

// A radix pass whose elements are made by the caller's own kernel (K-ranges: the range tuples are scattered by the low gene
// byte by the kernel that builds them): tiles of PDL_RADIX_TILE source elements, counts[digit * n_tiles + tile] filled by the
// caller -> offs[digit * n_tiles + tile] = where the tile's run of that digit starts; *d_total = number of elements.
constexpr uint32_t PDL_RADIX_TILE = 4096, PDL_RADIX_BINS = 256;
void pdl_radix_offsets(pdl_ctx *c, const uint32_t *counts, uint32_t *offs, uint32_t n_tiles, uint64_t *d_total);

extern template void pdl_sort_pairs<uint32_t, uint32_t>(pdl_ctx *, uint32_t *&, uint32_t *&, uint32_t *&, uint32_t *&, uint64_t, uint32_t, bool, const uint64_t *, uint32_t, bool);
extern template void pdl_sort_pairs<uint64_t, uint32_t>(pdl_ctx *, uint64_t *&, uint64_t *&, uint32_t *&, uint32_t *&, uint64_t, uint32_t, bool, const uint64_t *, uint32_t, bool);
extern template void pdl_sort_pairs<uint32_t, unsigned long long>(pdl_ctx *, uint32_t *&, uint32_t *&, unsigned long long *&, unsigned long long *&, uint64_t, uint32_t, bool, const uint64_t *, uint32_t, bool);
