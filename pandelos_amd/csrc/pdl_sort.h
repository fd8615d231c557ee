// pdl_sort.h — stable LSD radix sort of (key, value) pairs on a bit range of the key.
//
// Counterpart of counting_sort_ext (ig/native/library.cpp:172-187) and its call sites (:270-278):
// the reference sorts 16-byte records by the bytes of seq then the bytes of rank; here the k-mer
// stream is produced in gene order, so ONE stable sort on the rank bits alone yields the same
// (rank, gene) order, and a second stable sort of the deduplicated records on the gene bits yields
// the per-gene lists.
#pragma once

#include "pdl_common.h"

// Sorts n pairs by key bits [0, end_bit).  Input in (*keys_in, *vals_in); on return the sorted
// pairs are in (*keys_out, *vals_out) — the function may swap the roles of the buffers, the
// pointers passed by reference are updated accordingly.
template <class KeyT, class ValT = uint32_t>       // ValT: uint32_t, or unsigned long long (an 8-byte payload carried along)
void pdl_sort_pairs(pdl_ctx *c, KeyT *&keys_in, KeyT *&keys_out, ValT *&vals_in, ValT *&vals_out,
                    uint64_t n, uint32_t end_bit, bool iota_values = false,   // iota_values: the values are 0, 1, 2, ... (vals_in is not read)
                    const uint64_t *d_n = nullptr);                           // d_n: the count lives on the device (*d_n <= n, grids sized for n)

extern template void pdl_sort_pairs<uint32_t, uint32_t>(pdl_ctx *, uint32_t *&, uint32_t *&, uint32_t *&, uint32_t *&, uint64_t, uint32_t, bool, const uint64_t *);
extern template void pdl_sort_pairs<uint64_t, uint32_t>(pdl_ctx *, uint64_t *&, uint64_t *&, uint32_t *&, uint32_t *&, uint64_t, uint32_t, bool, const uint64_t *);
extern template void pdl_sort_pairs<uint32_t, unsigned long long>(pdl_ctx *, uint32_t *&, uint32_t *&, unsigned long long *&, unsigned long long *&, uint64_t, uint32_t, bool, const uint64_t *);
