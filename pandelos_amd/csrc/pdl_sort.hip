// pdl_sort.hip — bring-up implementation of pdl_sort_pairs on rocPRIM's device radix sort.
// (Round-1 status: library primitive; the hand-written gfx950 onesweep replaces it, see DESIGN.md.)
#include "pdl_sort.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

template <class KeyT>
void pdl_sort_pairs(pdl_ctx *c, KeyT *&keys_in, KeyT *&keys_out, uint32_t *&vals_in, uint32_t *&vals_out,
                    uint64_t n, uint32_t end_bit) {
    if (n == 0) return;
    if (end_bit == 0) end_bit = 1;
    if (end_bit > sizeof(KeyT) * 8) end_bit = sizeof(KeyT) * 8;
    size_t tmp_bytes = 0;
    PDL_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t) n, 0u, end_bit, c->stream));
    c->sort_tmp.alloc(tmp_bytes);
    PDL_HIP(rocprim::radix_sort_pairs(c->sort_tmp.p, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t) n, 0u, end_bit, c->stream));
}

template void pdl_sort_pairs<uint32_t>(pdl_ctx *, uint32_t *&, uint32_t *&, uint32_t *&, uint32_t *&, uint64_t, uint32_t);
template void pdl_sort_pairs<uint64_t>(pdl_ctx *, uint64_t *&, uint64_t *&, uint32_t *&, uint32_t *&, uint64_t, uint32_t);
