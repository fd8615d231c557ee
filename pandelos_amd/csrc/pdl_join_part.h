// pdl_join_part.h — K-join, tier 0: the first tier of the join for datasets of SHORT rows.
// (Included by pdl_join.hip behind JoinArgs and the finalize helpers; one translation unit.)
//
// Same row program as every tier (library.cpp:461-517, see pdl_join.hip).  The filter tiers (k_join_lds<.., true>) decide
// lookup by lookup: bitmap test-and-set; bit clear = the column's first sighting, PUT ASIDE in a list in HBM; bit set = the
// column gets a table slot; when the walk is over the list is read back and the first sightings of kept columns are added.
// ~200 vector instructions per 64 lookups on a kernel whose vector pipe is the busy unit (SQ_ACTIVE_INST_VALU ~ 3/4 of its
// time), 8 bytes of HBM writes per first sighting, and ~14 us of fixed latency per row (dispense, stage, walk, put-aside
// pass, finalize, ten barriers) that a row of 1 650 lookups — the 64-genome benchmark set — cannot amortise: 0.26 of the
// HBM roofline there against 0.50 on rows of 6 500 lookups.
//
// Here a workgroup takes SEVERAL consecutive rows per cycle (up to 4 rows, 960 ranges, 4 096 lookups) and keeps every
// lookup of the cycle in a REGISTER (one 32-bit key {h(column) : 22 | staged range : 10}, at most 16 per lane) until all of
// them have been seen:
//
//   stage     the ranges of all rows of the cycle, one flat lookup space (as the other tiers do for one row)
//   walk      every lookup marks its column in two 32-Kbit bitmaps: "seen" and — when the bit was already set — "seen twice"
//   (barrier: the bitmaps are complete)
//   sift      a lookup whose "seen twice" bit is clear is the ONLY lookup of its column in this cycle.  With both counts 1 it
//             can never be emitted when no gene involved has <= 2k k-mers (library.cpp:497-500; such rows are not taken
//             here): it is dropped — the "seen twice" rule of the filter tiers, but decided AFTER all sightings are in: the
//             first sighting is judged like the second, nothing is put aside, nothing goes to HBM.
//   add       the others go to a 1024-slot table keyed by (row of the cycle, column).  A slot is ONE word, the smallest key
//             that reached it (= the column and the first range that touched it: the emission-order key), and one counter:
//             sightings with both counts 1 in the low half — each adds (1, 1, 1) to the three sums — and the rare ones with
//             a count >= 2 in the high half, whose counts wait in a short side list.
//   finalize  touched slots only, exactly as in the other tiers.
//
// h is a bijection of the 22-bit gene id (odd multiplier mod 2^22), so the column comes back out of the key.  LDS 31 KB:
// five workgroups per CU.  The kernel is a template of its thread count and is launched twice: 256 threads (a cycle of 4 096
// lookups) over every row, then 512 threads (8 192 lookups, one row per draw) over the rows that alone exceed the first form's
// cycle — on the 64-genome set 923 rows of 4 100-7 200 lookups, which as rows of the filter tier (one per workgroup) made that
// launch last as long as its longest row.  Rows neither form takes (more than 960 ranges or 8 192 lookups, a gene of <= 2k
// k-mers, too many columns or heavy lookups) go to the filter tier through the usual device-side list; a tier writes the
// descriptors of the rows it hands on, so the next one starts without a pass that makes them.
#pragma once

constexpr uint32_t PT_T = 256;                           // threads of the kernel's first form (the second, for the rows that exceed its cycle, has PT_T2)
constexpr uint32_t PT_T2 = 512, PT_WGS2 = 3;             // (85 registers: three workgroups of eight waves per CU)
constexpr uint32_t PT_RB = 960;                          // ranges staged per cycle (10-bit range index in a key)
constexpr uint32_t PT_ROWS = 4;                          // rows per cycle
#ifndef PT_WG_PER_CU
#define PT_WG_PER_CU 5
#endif
constexpr uint32_t PT_NCH = 4, PT_ITERS = PT_WG_PER_CU >= 5 ? 4 : 6;      // chunks of 64 lookups in flight per wave, steps per wave
constexpr uint32_t PT_KPT = PT_NCH * PT_ITERS;           // keys a lane holds between the walk and the sift
constexpr uint32_t PT_HEAVY_CAP = 64;                    // lookups with a count >= 2 per cycle
constexpr uint32_t PT_BM_BITS = PT_WG_PER_CU >= 5 ? 10 : 11, PT_BM_WORDS = 1u << PT_BM_BITS;      // each bitmap: 32 Kbit (64 with four workgroups per CU)
constexpr uint32_t PT_BM_SHIFT = 22 - PT_BM_BITS;        // word = h >> PT_BM_SHIFT, bit = the five bits below
constexpr uint32_t PT_HT_BITS = 10, PT_HT = 1u << PT_HT_BITS;
constexpr uint32_t PT_BATCH = 8;                         // rows a workgroup draws from the dispenser at a time
// h(c) = c * M mod 2^22 with M = 0x9E3779B1 mod 2^22: only the low 22 bits of the multiplier matter, so the product is that of two
// 24-bit values and v_mul_u32_u24 — a full-rate instruction, where v_mul_lo_u32 takes four times as long — gives the same hash.
// (The multiplier matters: 0x9E3779 in its place slowed the whole join by a quarter — homologs' gene ids are regularly spaced,
// and their bits in the two bitmaps collided.)
constexpr uint32_t PT_HASH_MUL = 0x9E3779B1u & 0x3fffffu;
constexpr uint32_t pt_inverse(uint32_t m) { uint32_t x = m; for (int i = 0; i < 5; i++) x *= 2u - m * x; return x; }
constexpr uint32_t PT_HASH_INV = pt_inverse(PT_HASH_MUL);
static_assert((uint32_t) (PT_HASH_MUL * PT_HASH_INV) == 1u, "multiplicative inverse mod 2^32 (hence mod 2^22)");
static_assert(PT_RB <= 1022 && PT_RB % PDL_WAVE == 0 && (3 * PT_HT) / 4 + PT_T2 <= CELL_CHUNK, "10-bit range index");

// low 32 bits of the product of two 24-bit values, in ONE full-rate instruction (the compiler sees through __umul24 once the result is
// masked — the low bits of a product do not depend on the high bits of its factors — and falls back to the quarter-rate v_mul_lo_u32)
__device__ __forceinline__ uint32_t pt_mul_u24(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

struct PartRow { uint32_t p, r, kcnt, genome, lg, pc_min, pad0, pad1; };

// TT threads: 256 (five workgroups per CU, a cycle of 4096 lookups) — every row goes through this form first — or 512 (a cycle of
// 8192 lookups, for the rows the first form hands on because they alone exceed its cycle: on the 64-genome set 923 rows of
// 4100-7200 lookups, which the filter tier used to take one per workgroup, a launch as long as its longest row)
template <uint32_t TT, uint32_t WGS>
__global__ __launch_bounds__(TT, WGS) void k_join_part(JoinArgs a) {
    constexpr uint32_t PT_T = TT, PT_NW = PT_T / PDL_WAVE;
    constexpr uint32_t PT_RPT = (PT_RB + PT_T - 1) / PT_T;              // ranges per thread in the staging loops
    constexpr uint32_t PT_LMAX = PT_KPT * PT_T;                         // lookups per cycle
    constexpr uint32_t PT_WLIST = (PT_BM_WORDS + PT_RB + 66) / PT_NW;   // surviving keys a wave can list
    constexpr uint32_t PT_TOUCH_CAP = (3 * PT_HT) / 4 + PT_T;           // every row's part of the table takes keys until it is three quarters full (+ one per thread in flight)
    static_assert(PT_WLIST >= PT_NCH * PDL_WAVE, "a wave's list holds at least one round of survivors");
    __shared__ uint32_t s_big[2 * PT_BM_WORDS + PT_RB + 66];   // "seen twice" | "seen" | prefix of the staged ranges' lengths; once the walk is
                                                               // over the last two are one stretch: every wave's list of surviving keys
    __shared__ uint32_t s_tkey[PT_HT];                   // table: smallest key + 1 that reached the slot (0: empty)
    __shared__ uint32_t s_tn[PT_HT];                     //        sightings (each adds (1, 1, 1) to the three sums; what a count >= 2 adds beyond that waits in s_heavy)
    __shared__ uint16_t s_touched[PT_TOUCH_CAP];
    __shared__ uint2 s_gm[PT_RB + 1];                    // staged ranges as they are in HBM: {first posting, postings | min(own count, 1023) << 22}
    __shared__ uint2 s_heavy[PT_HEAVY_CAP];              // {key, count of the posting} of the lookups with a count >= 2
    __shared__ uint4 s_bdesc[PT_BATCH], s_binfo[PT_BATCH];
    __shared__ PartRow s_row[PT_ROWS];
    __shared__ uint32_t s_wave[PT_NW];
    __shared__ uint2 s_lb[PT_ROWS + 1];                  // {wave, prefix inside that wave} of the first staged range of every slot
    __shared__ uint2 s_wstart[PT_NW];
    __shared__ uint32_t s_nheavy, s_ntouched, s_overflow, s_solo, s_tslot[PT_ROWS], s_nemit[PT_ROWS], s_w0, s_bn, s_bpos;
    __shared__ unsigned long long s_base, s_chunk_next, s_chunk_end;

    const uint32_t tid = threadIdx.x, lane = tid & (PDL_WAVE - 1), wave = tid / PDL_WAVE;
    const uint32_t n_work = a.n_work_ptr ? *a.n_work_ptr : a.n_work;
    if (n_work == 0) return;
    const uint32_t batch = min(max(a.work_batch, 1u), PT_BATCH);       // rows a workgroup draws at a time (one, for the few long rows of the second form)
    uint32_t *s_bm2 = s_big, *s_bm1 = s_big + PT_BM_WORDS, *s_cum = s_big + 2 * PT_BM_WORDS;
    for (uint32_t i = tid; i < PT_HT; i += PT_T) { s_tkey[i] = 0; s_tn[i] = 0; }      // cleared once; afterwards every slot is reset by whoever consumes it
    if (tid == 0) { s_nheavy = 0; s_ntouched = 0; s_overflow = 0; s_solo = 0; s_bn = 0; s_bpos = 0; s_chunk_next = 0; s_chunk_end = 0; }
    if (tid < PT_ROWS) { s_tslot[tid] = 0; s_nemit[tid] = 0; }
    const float threshold = 1.0f / (2.0f * (float) a.k);
    const uint32_t tc_min = min_numerator(threshold, (float) (int) a.min_kseq);
    const bool track_first = a.canonical == 0;
    const uint32_t two_k = 2 * a.k;
#ifdef PDL_JOIN_PHASES
    // diagnostic build: where a workgroup's time goes, by the clock of its first thread (100-MHz ticks): 0 draw + pick the cycle's
    // rows | 1 stage | 2 prefixes, segment starts | 3 walk | 4 sift + add | 5 staging room, finalize, emit; and what it worked on
    unsigned long long pph[6] = {0, 0, 0, 0, 0, 0}, pt_prev = wall_clock64();
    unsigned long long pt_cycles = 0, pt_over = 0, pt_surv = 0, pt_probe = 0, pt_look = 0, pt_rows = 0;       // (thread 0 / wave 0 count; survivors of wave 0, probe steps of thread 0 only)
#define PT_MARK(i) do { if (tid == 0) { const unsigned long long t_now = wall_clock64(); pph[i] += t_now - pt_prev; pt_prev = t_now; } } while (0)
#else
#define PT_MARK(i) do { } while (0)
#endif

    for (;;) {
        pdl_sync();
        PT_MARK(5);
        // ---- rows: a batch of consecutive work items at a time -------------------------------------------------------
        if (s_bpos >= s_bn) {                                // (uniform) batch used up: draw the next one
            if (tid == 0) s_w0 = atomicAdd(a.work_cursor, batch);
            pdl_sync();
            const uint32_t w0 = s_w0;
            if (w0 >= n_work) break;                         // (uniform) every wave leaves here
            if (tid < batch && w0 + tid < n_work) {
                const uint4 d = a.desc[w0 + tid];            // {task position, gene, first range, ranges}
                s_bdesc[tid] = d;
                s_binfo[tid] = a.gene_info[d.y];             // {k-mers, genome, task position, shard-local genome}
            }
            if (tid == 0) { s_bn = min(batch, n_work - w0); s_bpos = 0; }
            pdl_sync();
        }
        const uint32_t bn = (uint32_t) __builtin_amdgcn_readfirstlane((int) s_bn);
        uint32_t j = (uint32_t) __builtin_amdgcn_readfirstlane((int) s_bpos);
        // rows this tier leaves alone, at the head of the batch: no range at all (no candidate), or too many ranges / a gene
        // of <= 2k k-mers (its single sightings count: the filter tier knows how)
        auto uni = [](uint32_t v) -> uint32_t { return (uint32_t) __builtin_amdgcn_readfirstlane((int) v); };     // (the same in every lane: keep it in a scalar register)
        for (; j < bn; j++) {
            const uint4 d = s_bdesc[j];
            const uint32_t dw = uni(d.w), kc = uni(s_binfo[j].x);
            if (dw == 0) { if (tid == 0) { a.row_base[d.x] = 0; a.row_cnt[d.x] = 0; } continue; }
            if (dw > PT_RB || kc <= two_k) { if (tid == 0) { const uint32_t oi = atomicAdd(a.overflow_count, 1u); a.overflow_rows[oi] = d.x; a.overflow_desc[oi] = d; } continue; }
            break;
        }
        if (j >= bn) { pdl_sync(); if (tid == 0) s_bpos = j; continue; }        // (uniform)
        PT_MARK(0);
        // the cycle's rows: consecutive ordinary rows while their ranges fit
        const bool solo = uni(s_solo) != 0;                  // after a cycle whose rows did not fit the table together: one row at a time
        uint32_t ns = 0, nb = 0, rbv[PT_ROWS + 1], e0v[PT_ROWS], jv[PT_ROWS];
        {
            bool stop = false;
#pragma unroll
            for (uint32_t s = 0; s < PT_ROWS; s++) {
                rbv[s] = nb; e0v[s] = 0; jv[s] = j;
                if (!stop && j < bn && (s == 0 || !solo)) {
                    const uint4 d = s_bdesc[j];
                    const uint32_t dw = uni(d.w), kc = uni(s_binfo[j].x);
                    if (dw == 0 || dw > PT_RB || kc <= two_k || nb + dw > PT_RB) stop = true;
                    else { e0v[s] = uni(d.z); nb += dw; ns = s + 1; j++; }
                }
            }
            rbv[PT_ROWS] = nb;
#pragma unroll
            for (uint32_t s = 0; s < PT_ROWS; s++) if (s >= ns) rbv[s] = nb;       // (slots not taken are empty)
        }
        // ---- stage: the ranges of all rows of the cycle, their lengths prefixed ----------------------------------------
        uint32_t len[PT_RPT], sum = 0;
        {
            uint2 rg[PT_RPT];
#pragma unroll
            for (uint32_t q = 0; q < PT_RPT; q++) {          // the loads first, branch-free: they overlap
                const uint32_t i = min(tid * PT_RPT + q, nb - 1);
                const uint32_t s = (uint32_t) (i >= rbv[1]) + (uint32_t) (i >= rbv[2]) + (uint32_t) (i >= rbv[3]);
                const uint32_t e0 = s == 0 ? e0v[0] : s == 1 ? e0v[1] : s == 2 ? e0v[2] : e0v[3];
                const uint32_t rb = s == 0 ? rbv[0] : s == 1 ? rbv[1] : s == 2 ? rbv[2] : rbv[3];
                rg[q] = a.ranges8[e0 + (i - rb)];
            }
#pragma unroll
            for (uint32_t i = 0; i < 2 * PT_BM_WORDS / 4 / PT_T; i++) reinterpret_cast<uint4 *>(s_big)[i * PT_T + tid] = make_uint4(0, 0, 0, 0);      // both bitmaps
#pragma unroll
            for (uint32_t q = 0; q < PT_RPT; q++) {
                const uint32_t i = tid * PT_RPT + q;
                len[q] = 0;
                if (i < nb) { s_gm[i] = rg[q]; len[q] = rg[q].y & 0x3fffffu; }
                sum += len[q];
            }
        }
        const uint32_t inc = wave_inclusive_scan_u32(sum);
        if (lane == PDL_WAVE - 1) s_wave[wave] = inc;
        {   // the thread that holds a slot's first range says where the slot starts inside its wave
            uint32_t ex_w = inc - sum;
#pragma unroll
            for (uint32_t q = 0; q < PT_RPT; q++) {
                const uint32_t i = tid * PT_RPT + q;
#pragma unroll
                for (uint32_t s = 1; s < PT_ROWS; s++) if (s < ns && i == rbv[s]) s_lb[s] = make_uint2(wave, ex_w);
                ex_w += len[q];
            }
        }
        pdl_sync();
        PT_MARK(1);
        uint32_t woff[PT_NW + 1];
        woff[0] = 0;
#pragma unroll
        for (uint32_t w = 0; w < PT_NW; w++) woff[w + 1] = woff[w] + uni(s_wave[w]);
        // lookups before each slot; the cycle keeps the leading rows that fit PT_LMAX together
        uint32_t lbv[PT_ROWS + 1];
        lbv[0] = 0;
#pragma unroll
        for (uint32_t s = 1; s < PT_ROWS; s++) {
            lbv[s] = woff[PT_NW];
            if (s < ns) { const uint2 v = s_lb[s]; const uint32_t vx = uni(v.x); uint32_t wo = 0;
#pragma unroll
                for (uint32_t w = 0; w < PT_NW; w++) wo = vx == w ? woff[w] : wo;
                lbv[s] = wo + uni(v.y); }
        }
        lbv[PT_ROWS] = woff[PT_NW];
        uint32_t ns_keep = 0;
#pragma unroll
        for (uint32_t s = 0; s < PT_ROWS; s++) if (s < ns && lbv[s + 1] <= PT_LMAX) ns_keep = s + 1;       // (prefixes ascend: the largest s that fits)
        ns_keep = (uint32_t) __builtin_amdgcn_readfirstlane((int) ns_keep);
        if (ns_keep == 0) {                                  // (uniform) the head row alone has more lookups than a cycle takes
            pdl_sync();
            if (tid == 0) { const uint4 d = s_bdesc[jv[0]]; const uint32_t oi = atomicAdd(a.overflow_count, 1u); a.overflow_rows[oi] = d.x; a.overflow_desc[oi] = d; s_bpos = jv[0] + 1; }
            continue;
        }
        const uint32_t total = (uint32_t) __builtin_amdgcn_readfirstlane((int) (ns_keep == 1 ? lbv[1] : ns_keep == 2 ? lbv[2] : ns_keep == 3 ? lbv[3] : lbv[4]));
        const uint32_t nbk = ns_keep == 1 ? rbv[1] : ns_keep == 2 ? rbv[2] : ns_keep == 3 ? rbv[3] : rbv[4];
        const uint32_t next_pos = (ns_keep == 1 ? jv[0] : ns_keep == 2 ? jv[1] : ns_keep == 3 ? jv[2] : jv[3]) + 1;
        const uint32_t rb1 = ns_keep > 1 ? rbv[1] : nbk, rb2 = ns_keep > 2 ? rbv[2] : nbk, rb3 = ns_keep > 3 ? rbv[3] : nbk;
        auto slot_of = [&](uint32_t r) -> uint32_t { return (uint32_t) (r >= rb1) + (uint32_t) (r >= rb2) + (uint32_t) (r >= rb3); };
        // the table: every row of the cycle has a part of it to itself, so a probe compares the column alone and the slot says
        // which row it belongs to
        const uint32_t sub_bits = ns_keep == 1 ? PT_HT_BITS : ns_keep == 2 ? PT_HT_BITS - 1 : PT_HT_BITS - 2;
        const uint32_t sub_mask = (1u << sub_bits) - 1u, sub_limit = (3u << sub_bits) >> 2;
        auto add_one = [&](uint32_t kk) {
            const uint32_t k1 = kk + 1u, hk = kk >> 10;
            const uint32_t sl = slot_of(kk & 1023u);
            const uint32_t tbase = sl << sub_bits;
            uint32_t ts = hk >> (22 - sub_bits);
            bool placed = false;
            for (uint32_t step = 0; step <= sub_mask; step++) {          // (bounded: a full part of the table ends the search)
#ifdef PDL_JOIN_PHASES
                if (tid == 0) pt_probe++;
#endif
                const uint32_t w = s_tkey[tbase + ts];
                if (w != 0u && ((w - 1u) >> 10) == hk) { placed = true; break; }             // this column of this row sits here
                if (w == 0u) {
                    if (*(volatile uint32_t *) &s_tslot[sl] >= sub_limit) break;             // three quarters full: give up
                    const uint32_t old = atomicCAS(&s_tkey[tbase + ts], 0u, k1);
                    if (old == 0u) {
                        const uint32_t idx = atomicAdd(&s_ntouched, 1u);
                        if (idx < PT_TOUCH_CAP) s_touched[idx] = (uint16_t) (tbase + ts);
                        atomicAdd(&s_tslot[sl], 1u);
                        placed = true;
                        break;
                    }
                    if (((old - 1u) >> 10) == hk) { placed = true; break; }                  // someone else was faster, for the same column
                }
                ts = (ts + 1) & sub_mask;
            }
            if (placed) {
                atomicMin(&s_tkey[tbase + ts], k1);                    // the smallest key = the first range of the row that touched the column
                atomicAdd(&s_tn[tbase + ts], 1u);
            } else s_overflow = 1;
        };
        const uint32_t chunks = (total + PDL_WAVE - 1) / PDL_WAVE;
        const uint32_t cpw = (chunks + PT_NW - 1) / PT_NW;   // <= PT_KPT
        const uint32_t seg = cpw * PDL_WAVE;
        {
            uint32_t ex = inc - sum;
#pragma unroll
            for (uint32_t w = 0; w < PT_NW; w++) ex += wave == w ? woff[w] : 0u;
#pragma unroll
            for (uint32_t q = 0; q < PT_RPT; q++) {
                const uint32_t i = tid * PT_RPT + q;
                if (i < PT_RB) s_cum[i] = i < nbk ? ex : (i == nbk ? total : 0xffffffffu);
                if (len[q] && i < nbk) {                     // the range that holds a wave segment's first lookup registers itself
                    uint32_t w_lo = 0;
#pragma unroll
                    for (uint32_t w = 0; w < PT_NW; w++) w_lo += (uint32_t) (w * seg < ex);
                    for (uint32_t w = w_lo; w < PT_NW && w * seg < ex + len[q]; w++) s_wstart[w] = make_uint2(i, ex);
                }
                ex += len[q];
            }
            for (uint32_t i = PT_RB + tid; i < PT_RB + 66; i += PT_T) s_cum[i] = i == nbk ? total : 0xffffffffu;
            if (tid < PT_ROWS && tid < ns_keep) {            // what finalize wants to know about the rows
                const uint32_t jj = tid == 0 ? jv[0] : tid == 1 ? jv[1] : tid == 2 ? jv[2] : jv[3];
                const uint4 d = s_bdesc[jj], inf = s_binfo[jj];
                s_row[tid] = PartRow{d.x, d.y, inf.x, inf.y, inf.w, min_numerator(threshold, (float) (int) inf.x), 0u, 0u};
            }
        }
        pdl_sync();
        PT_MARK(2);
        // ---- walk: one key per lookup, kept in a register; the column marked "seen" / "seen twice" ----------------------------
        // (lane -> range mapping as in k_join_lds: one coalesced read of the next 64 range starts per step, boundaries by
        // ds_permute + ballot + popcount)
        uint32_t key[PT_KPT];
#pragma unroll
        for (uint32_t i = 0; i < PT_KPT; i++) key[i] = 0xffffffffu;      // (every key starts "done with": nothing is carried from the cycle before)
        const uint32_t ch0 = wave * cpw;
        {
            uint32_t ch = ch0;
            const uint32_t ch_end = min(chunks, ch + cpw);
            const uint2 ws = s_wstart[wave];
            uint32_t rs = (uint32_t) __builtin_amdgcn_readfirstlane((int) ws.x);
#pragma unroll
            for (uint32_t it = 0; it < PT_ITERS; it++) {
                if (ch < ch_end) {                           // (wave-uniform)
                    uint2 po[PT_NCH];
                    uint32_t adr[PT_NCH];
                    bool live[PT_NCH];
                    uint32_t ownhv = 0;                          // bit u: the row's own count of that k-mer is >= 2
                    const uint32_t nu = min(PT_NCH, ch_end - ch);
                    const uint32_t f_lo = ch * PDL_WAVE, f_hi = f_lo + nu * PDL_WAVE;
                    const uint32_t vw = s_cum[rs + 1 + lane];                    // starts of the following ranges (all > f_lo)
                    if ((uint32_t) __builtin_amdgcn_readlane((int) vw, PDL_WAVE - 1) >= f_hi) {
#pragma unroll
                        for (uint32_t u = 0; u < PT_NCH; u++) {
                            live[u] = false; adr[u] = 0; key[it * PT_NCH + u] = 0;
                            if (u < nu) {
                                const uint32_t lo = f_lo + u * PDL_WAVE, f = lo + lane;
                                const uint32_t cu = (uint32_t) __popcll(__ballot(vw <= lo));
                                const bool inside = vw > lo && vw < lo + PDL_WAVE;
                                const int recv = __builtin_amdgcn_ds_permute((int) ((inside ? vw - lo : 0u) << 2), inside ? 1 : 0);
                                const unsigned long long m = __ballot(recv != 0);
                                // ranges that start at or before this lane's lookup: those up to the chunk's first one, the starts below the lane (mbcnt), its own
                                const uint32_t r = rs + cu + __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u)) + (uint32_t) (recv != 0);
                                live[u] = f < total;
                                const uint32_t rl = live[u] ? r : 0u;
                                const uint2 gmu = s_gm[rl];
                                const uint32_t cum_r = s_cum[rl];                // where range rl starts in the row's lookups: one LDS read instead of a scan of the mask
                                key[it * PT_NCH + u] = rl;
                                adr[u] = gmu.x + (f - cum_r);

                                ownhv |= (uint32_t) ((gmu.y >> 22) >= 2u) << u;
                            }
                        }
                        const uint32_t ce = (uint32_t) __popcll(__ballot(vw <= f_hi));
                        rs += ce;
                    } else {
#pragma unroll
                        for (uint32_t u = 0; u < PT_NCH; u++) {
                            live[u] = false; adr[u] = 0; key[it * PT_NCH + u] = 0;
                            if (u < nu) {
                                const uint32_t f0 = (ch + u) * PDL_WAVE, f = f0 + lane;
                                const uint32_t v = s_cum[rs + 1 + lane];
                                const bool inside = v < f0 + PDL_WAVE;
                                const int recv = __builtin_amdgcn_ds_permute((int) ((inside ? v - f0 : 0u) << 2), inside ? 1 : 0);
                                const unsigned long long m = __ballot(recv != 0);
                                const uint32_t w = (uint32_t) __popcll(__ballot(inside));
                                const uint32_t r = rs + __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u)) + (uint32_t) (recv != 0);
                                live[u] = f < total;
                                const uint32_t rl = live[u] ? r : 0u;
                                const uint2 gmu = s_gm[rl];
                                const uint32_t cum_r = s_cum[rl];
                                key[it * PT_NCH + u] = rl;
                                adr[u] = gmu.x + (f - cum_r);
                                ownhv |= (uint32_t) ((gmu.y >> 22) >= 2u) << u;
                                const uint32_t nextb = w < PDL_WAVE ? (uint32_t) __builtin_amdgcn_readlane((int) v, w) : s_cum[rs + 1 + PDL_WAVE];
                                rs += w + (uint32_t) (nextb == f0 + PDL_WAVE);          // (a range that starts with the next chunk is that chunk's first)
                            }
                        }
                    }
#pragma unroll
                    for (uint32_t u = 0; u < PT_NCH; u++) po[u] = a.post[live[u] ? adr[u] : 0u];       // dead lanes read posting 0: no exec juggling
                    uint32_t seen[PT_NCH], bit[PT_NCH], wd[PT_NCH];
#pragma unroll
                    for (uint32_t u = 0; u < PT_NCH; u++) {          // four bitmap atomics in flight
                        const uint32_t h = pt_mul_u24(po[u].x, PT_HASH_MUL) & 0x3fffffu;      // (gene ids have 22 bits here)
                        key[it * PT_NCH + u] |= h << 10;
                        wd[u] = h >> PT_BM_SHIFT; bit[u] = 1u << ((h >> (PT_BM_SHIFT - 5)) & 31u);        // (the column alone: a column two rows of the cycle meet once each survives the sift and is told apart later)
                        seen[u] = live[u] ? atomicOr(&s_bm1[wd[u]], bit[u]) : 0u;
                        // a count >= 2 on either side (rare; an own count of 1023 stands for "1023 or more"): the lookup survives the sift by
                        // itself (both bits set); it is entered as an ordinary sighting, and what its counts add beyond (1, 1, 1) is listed
                        if (live[u] && (po[u].y >= 2u || ((ownhv >> u) & 1u))) {
                            seen[u] = bit[u];
                            const uint32_t i = atomicAdd(&s_nheavy, 1u);
                            if (i < PT_HEAVY_CAP) s_heavy[i] = make_uint2(key[it * PT_NCH + u], po[u].y);
                        }
                    }
#pragma unroll
                    for (uint32_t u = 0; u < PT_NCH; u++) if (seen[u] & bit[u]) atomicOr(&s_bm2[wd[u]], bit[u]);
                }
                ch += PT_NCH;
            }
        }
        pdl_sync();
        PT_MARK(3);
        // ---- sift + add ------------------------------------------------------------------------------------------------------------------
        // sift: the lookups that are not alone on their bit are compacted into the wave's list (bm1 and the prefix array are done
        // with); add: the list goes to the table, two keys per lane at a time
        uint32_t *wlist = s_bm1 + wave * PT_WLIST;
        uint32_t nsv = 0;                                    // (wave-uniform) keys in the list
        const uint32_t n_heavy = uni(s_nheavy);
        if (n_heavy > PT_HEAVY_CAP) s_overflow = 1;
        auto drain = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the wave reads back the list it has just written
            for (uint32_t i0 = 0; i0 < nsv; i0 += 2 * PDL_WAVE) {
                const uint32_t ia = i0 + lane, ib = ia + PDL_WAVE;
                const uint32_t ka = wlist[ia < nsv ? ia : 0u], kb = wlist[ib < nsv ? ib : 0u];          // two independent chains per lane
                if (ia < nsv) add_one(ka);
                if (ib < nsv) add_one(kb);
            }
            nsv = 0;
        };
        {
            const uint32_t nkeys = ch0 < chunks ? min(chunks - ch0, cpw) : 0u;       // keys per lane of this wave (the last chunk may be partial)
#pragma unroll
            for (uint32_t g = 0; g < PT_ITERS; g++) {
                if (g * PT_NCH >= nkeys) break;              // (uniform)
                if (nsv + PT_NCH * PDL_WAVE > PT_WLIST) drain();         // (uniform, rare) no room for another four rounds of survivors: the list goes to the table first
                uint32_t twice[PT_NCH];
#pragma unroll
                for (uint32_t u = 0; u < PT_NCH; u++) twice[u] = s_bm2[(key[g * PT_NCH + u] >> (10 + PT_BM_SHIFT)) & (PT_BM_WORDS - 1)];      // four reads in flight
#pragma unroll
                for (uint32_t u = 0; u < PT_NCH; u++) {
                    const uint32_t i = g * PT_NCH + u;
                    if (i >= nkeys) break;                   // (uniform)
                    const bool keep = (ch0 + i) * PDL_WAVE + lane < total && ((twice[u] >> ((key[i] >> (10 + PT_BM_SHIFT - 5)) & 31u)) & 1u);
                    const unsigned long long mk = __ballot(keep);
                    if (keep) wlist[nsv + (uint32_t) __popcll(mk & ((1ull << lane) - 1ull))] = key[i];
                    nsv += (uint32_t) __popcll(mk);
#ifdef PDL_JOIN_PHASES
                    if (tid == 0) pt_surv += (uint32_t) __popcll(mk);
#endif
                }
            }
            drain();
        }
        pdl_sync();
        PT_MARK(4);
#ifdef PDL_JOIN_PHASES
        if (tid == 0) { pt_cycles++; pt_look += total; pt_rows += ns_keep; if (uni(s_overflow)) pt_over++; }
#endif
        if (uni(s_overflow)) {     // (uniform, rare) a row's part of the table is full: the table is wiped; several rows are tried again one by one, a single one goes to the filter tier
            pdl_sync();
            for (uint32_t i = tid; i < PT_HT; i += PT_T) { s_tkey[i] = 0; s_tn[i] = 0; }
            const bool hand_on = ns_keep == 1;
            if (tid == 0 && hand_on) { const uint4 d = s_bdesc[jv[0]]; const uint32_t oi = atomicAdd(a.overflow_count, 1u); a.overflow_rows[oi] = d.x; a.overflow_desc[oi] = d; }
            if (tid < PT_ROWS) s_tslot[tid] = 0;
            if (tid == 0) {
                s_nheavy = 0; s_ntouched = 0; s_overflow = 0;
                s_solo = hand_on ? (solo ? uni(s_solo) - 1u : 0u) : ns_keep;
                s_bpos = hand_on ? next_pos : jv[0];
            }
            continue;
        }
        // ---- staging room: a row's cells are contiguous; the columns it touched bound them -----------------------------------------
        const uint32_t ntouched = min(uni(s_ntouched), PT_TOUCH_CAP);
        if (tid == 0) {
            unsigned long long nx = s_chunk_next;
            if (nx + ntouched > s_chunk_end) {
                nx = atomicAdd(a.cell_cursor, (unsigned long long) CELL_CHUNK);
                s_chunk_end = nx + CELL_CHUNK;
            }
            s_base = nx;
        }
        pdl_sync();
        const unsigned long long base = s_base;
        const bool fits = base + ntouched <= a.st_cap;
        const uint32_t c0 = uni(s_tslot[0]), c1 = uni(s_tslot[1]), c2 = uni(s_tslot[2]);
        const uint32_t sb1 = c0, sb2 = c0 + c1, sb3 = c0 + c1 + c2;
        // ---- finalize + emit (library.cpp:485-517): touched slots only -----------------------------------------------------------
        for (uint32_t t = tid; t < ntouched; t += PT_T) {
            const uint32_t ts = s_touched[t];
            const uint32_t w = s_tkey[ts] - 1u, light = s_tn[ts];
            s_tkey[ts] = 0u; s_tn[ts] = 0u;                  // slot consumed
            const uint32_t hk = w >> 10, minr = w & 1023u, sl = ts >> sub_bits;
            // sums: every sighting is (1, 1, 1); the ones with a count >= 2 bring what their counts add beyond that from the side list
            uint32_t s_min = 0, s_own = 0, s_cc = 0;
            for (uint32_t e = 0; e < n_heavy; e++) {         // (uniform; no such lookup in most cycles)
                const uint2 hv = s_heavy[e];
                if ((hv.x >> 10) == hk && slot_of(hv.x & 1023u) == sl) {
                    const uint2 g = s_gm[hv.x & 1023u];
                    uint32_t own = g.y >> 22;
                    if (own == 1023u) own = a.post[g.x - 1].y;
                    s_min += min(hv.y, own) - 1u; s_own += own - 1u; s_cc += hv.y - 1u;
                }
            }
            const uint32_t inter = light + s_min, pcn = light + s_own, tcn = light + s_cc;
            const PartRow row = s_row[sl];
            const uint32_t c = (hk * PT_HASH_INV) & 0x3fffffu;      // (the inverse mod 2^22 is all a 22-bit product needs)
            if (c >= a.N) { atomicAdd(a.error_count, 1u); continue; }
            if (c == row.r) continue;                        // identity cell is zeroed (library.cpp:485-487)
            if (pcn < row.pc_min && tcn < tc_min) continue;  // cannot be valid (see k_join_lds): among them every column sighted once
            const uint4 ci4 = a.gene_info[c];
            float perc, tr;
            const float score = finalize_counts((int) inter, (int) pcn, (int) tcn, row.kcnt, ci4.x, threshold, perc, tr);
            if (score > 0.0f) {
                const uint32_t i = atomicAdd(&s_nemit[sl], 1u);
                if (fits) {
                    const unsigned long long o = base + (sl == 0 ? 0u : sl == 1 ? sb1 : sl == 2 ? sb2 : sb3) + i;
                    const uint2 g = s_gm[minr];
                    a.st_score[o] = score; a.st_perc[o] = perc; a.st_tr[o] = tr;
                    a.st_col[o] = c; a.st_first[o] = track_first ? g.x + (g.y & 0x3fffffu) : 0xffffffffu;
                    const uint32_t gc = ci4.y;
                    atomicMax(reinterpret_cast<uint32_t *>(a.MS + (size_t) row.p * a.G + gc), __float_as_uint(score));
                    atomicMax(reinterpret_cast<uint32_t *>(a.CM + (size_t) row.lg * a.N + c), __float_as_uint(score));
                    if (a.mirror) {                          // cell (c, r): see k_join_lds
                        const uint32_t pc = ci4.z;
                        if (pc != 0xffffffffu) {
                            atomicAdd(&a.mirror_cnt[pc], 1u);
                            atomicMax(reinterpret_cast<uint32_t *>(a.MS + (size_t) pc * a.G + row.genome), __float_as_uint(score));
                            atomicMax(reinterpret_cast<uint32_t *>(a.CM + (size_t) ci4.w * a.N + row.r), __float_as_uint(score));
                        }
                    }
                }
            }
        }
        pdl_sync();
        if (tid < ns_keep) {
            const uint32_t sb = tid == 0 ? 0u : tid == 1 ? sb1 : tid == 2 ? sb2 : sb3;
            a.row_base[s_row[tid].p] = fits ? (uint32_t) (base + sb) : 0u;
            a.row_cnt[s_row[tid].p] = fits ? s_nemit[tid] : 0u;      // staging ran out: the rows hold nothing, the host repeats the pass
        }
        if (tid == 0) {
            // the last row's region ends with its cells; the rows before it keep their touched columns' worth (address space only)
            const uint32_t last = ns_keep - 1;
            const uint32_t lb = last == 0 ? 0u : last == 1 ? sb1 : last == 2 ? sb2 : sb3;
            s_chunk_next = base + lb + s_nemit[last];
            s_bpos = next_pos; s_nheavy = 0; s_ntouched = 0;
            if (solo) s_solo = uni(s_solo) - 1u;
        }
        pdl_sync();
        if (tid < PT_ROWS) { s_tslot[tid] = 0; s_nemit[tid] = 0; }
    }
#ifdef PDL_JOIN_PHASES
    if (tid == 0 && a.phase) {
        for (int i = 0; i < 6; i++) atomicAdd(&a.phase[i], pph[i]);
        atomicAdd(&a.phase[6], pt_cycles); atomicAdd(&a.phase[7], pt_over); atomicAdd(&a.phase[8], pt_surv); atomicAdd(&a.phase[9], pt_probe);
        atomicAdd(&a.phase[10], pt_look); atomicAdd(&a.phase[11], pt_rows);
    }
#endif
#undef PT_MARK
}
