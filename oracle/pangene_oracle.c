/*
 * pangene_oracle.c — CPU ORACLE (test infrastructure; NOT part of the shipped product).
 * See pangene_oracle.h for status and rules of use.
 *
 * Every function below restates one piece of /root/reference/ig/native/library.cpp and says
 * which lines.  It keeps the reference's data flow (16-byte records, LSD byte radix sort with
 * a full copy per pass, in-place dedup, rank-group scan with its last-record quirk, per-row
 * dense accumulators with colour stamps and the 2048-column chunk loop) so that it can also
 * serve as the "port" CPU baseline; only containers differ (flat arrays instead of
 * std::vector<std::vector<...>>).
 */
#include "pangene_oracle.h"

#include <stdlib.h>
#include <string.h>

#define RABIN_MODULO ((uint64_t) 18446744073709551557ULL) /* library.cpp:19 */

/* posting range of one (sequence, shared rank-group): library.cpp:41-45 (indices, not pointers) */
typedef struct {
    uint64_t start, current, end;
} po_range;

struct po_ctx {
    /* pair_info, library.cpp:56-73 */
    uint64_t rank_counters[256];
    uint8_t rank_values[256];
    uint8_t rank_base;
    uint8_t rank_byte_order;
    uint32_t kvalue;
    int hash_fallback;
    uint64_t last_multiplier;

    uint32_t sequences_count;
    uint32_t genomes_count;
    po_kmer *kmers;
    uint64_t kmers_size;
    uint64_t kmer_occurrences;
    uint32_t *kseq_lengths;
    uint32_t *seq_gen_mapping;
    uint32_t *genome_seq_off;    /* genome_sequences as CSR: [G+1] */
    uint32_t *genome_seq;        /* [N] ascending ids per genome */
    uint64_t *range_off;         /* kmers_ranges as CSR: [N+1] */
    po_range *ranges;
    uint64_t *total_visited;     /* computation_costs[].total_visited */
};

po_ctx *po_create(void) { return (po_ctx *) calloc(1, sizeof(po_ctx)); }

static void po_reset(po_ctx *c) {
    free(c->kmers); free(c->kseq_lengths); free(c->seq_gen_mapping); free(c->genome_seq_off);
    free(c->genome_seq); free(c->range_off); free(c->ranges); free(c->total_visited);
    memset(c, 0, sizeof(*c));
    c->last_multiplier = 1;
}
void po_destroy(po_ctx *c) { if (c) { po_reset(c); free(c); } }

/* library.cpp:75-79 */
static inline uint64_t update_rank(const po_ctx *info, uint64_t current, uint8_t next, uint8_t pop) {
    current -= (uint64_t) info->rank_values[pop] * info->last_multiplier;
    return current * (uint64_t) info->rank_base + (uint64_t) info->rank_values[next];
}

/* library.cpp:81-86 — the first line is evaluated in 64 bits (wraps) and only then widened */
static inline uint64_t update_rank_hash(const po_ctx *info, uint64_t current, uint8_t next, uint8_t pop) {
    uint64_t wrapped = current + RABIN_MODULO - (uint64_t) info->rank_values[pop] * info->last_multiplier;
    unsigned __int128 ncurrent = wrapped;
    return (uint64_t) ((ncurrent * (unsigned __int128) info->rank_base +
                        (unsigned __int128) info->rank_values[next]) % RABIN_MODULO);
}

/* library.cpp:88-132 */
static int rank_init(po_ctx *info, int kvalue) {
    if (kvalue <= 0) return -1;
    info->kvalue = (uint32_t) kvalue;
    int rank = 0;
    for (int i = 0; i < 256; i++)
        if (info->rank_counters[i] > 0) info->rank_values[i] = (uint8_t) rank++;
    info->rank_base = (uint8_t) rank;
    info->last_multiplier = 1;

    int has_overflow = 0;
    int tmp_kvalue = kvalue - 1;
    while (tmp_kvalue--) {
        uint64_t ovflw_test = info->last_multiplier;
        info->last_multiplier *= info->rank_base;
        if (has_overflow) {
            info->last_multiplier %= RABIN_MODULO;
        } else if ((ovflw_test > info->last_multiplier) ||
                   (ovflw_test * info->rank_base > info->last_multiplier * info->rank_base)) {
            has_overflow = 1;
            info->hash_fallback = 1;
            info->last_multiplier = ((ovflw_test % RABIN_MODULO) * info->rank_base) % RABIN_MODULO;
        }
    }
    if (!has_overflow) {
        uint64_t rank_tmp = info->last_multiplier * info->rank_base;
        info->rank_byte_order = 0;
        while (rank_tmp) { info->rank_byte_order++; rank_tmp /= 256; }
    } else {
        info->rank_byte_order = 8;
    }
    return 0;
}

/* library.cpp:134-150 */
static void do_ranking(const po_ctx *info, const uint8_t *chars, uint32_t len, uint64_t *result) {
    uint64_t rank = 0;
    const uint32_t k = info->kvalue;
    if (info->hash_fallback) {
        for (uint32_t i = 0; i < k; i++) rank = update_rank_hash(info, rank, chars[i], 0);
        result[0] = rank;
        for (uint32_t i = k; i < len; i++) {
            rank = update_rank_hash(info, rank, chars[i], chars[i - k]);
            result[i - k + 1] = rank;
        }
    } else {
        for (uint32_t i = 0; i < k; i++) rank = update_rank(info, rank, chars[i], 0);
        result[0] = rank;
        for (uint32_t i = k; i < len; i++) {
            rank = update_rank(info, rank, chars[i], chars[i - k]);
            result[i - k + 1] = rank;
        }
    }
}

int po_rank_gene(const po_ctx *c, const uint8_t *chars, uint32_t len, uint64_t *out) {
    if (len < c->kvalue) return 0;
    do_ranking(c, chars, len, out);
    return (int) (len - c->kvalue + 1);
}

/* library.cpp:168-187: stable counting sort on one byte of rank (by_rank) or seq, full copy per pass */
static void counting_sort_ext(po_kmer *vec, po_kmer *tmp, uint64_t n, unsigned byte_ref, int by_rank) {
    uint64_t counts[256 + 1];
    memset(counts, 0, sizeof(counts));
    memcpy(tmp, vec, n * sizeof(po_kmer));
    const unsigned sh = byte_ref * 8u;
    for (uint64_t i = 0; i < n; i++) {
        uint64_t key = by_rank ? vec[i].rank : (uint64_t) vec[i].seq;
        counts[((key >> sh) & 0xFFu) + 1]++;
    }
    for (int i = 1; i < 256; i++) counts[i] += counts[i - 1];
    for (uint64_t i = 0; i < n; i++) {
        uint64_t key = by_rank ? tmp[i].rank : (uint64_t) tmp[i].seq;
        vec[counts[(key >> sh) & 0xFFu]++] = tmp[i];
    }
}

static int cmp_seq(const void *a, const void *b) {
    const po_kmer *x = (const po_kmer *) a, *y = (const po_kmer *) b;
    if (x->seq != y->seq) return x->seq < y->seq ? -1 : 1;
    /* ties only exist inside the folded last group (library.cpp:300-306); their order is
       immaterial to every sum and to first-touch order, make it deterministic */
    if (x->rank != y->rank) return x->rank < y->rank ? -1 : 1;
    return 0;
}

/* library.cpp:189-371 */
int po_preprocess(po_ctx *info, const uint8_t *residues, const uint64_t *offsets,
                  const uint32_t *genome_of, uint32_t seq_count, int kvalue, int only_complexity) {
    po_reset(info);                                                    /* :192 */
    info->sequences_count = seq_count;
    info->seq_gen_mapping = (uint32_t *) calloc(seq_count ? seq_count : 1, sizeof(uint32_t));
    info->kseq_lengths = (uint32_t *) calloc(seq_count ? seq_count : 1, sizeof(uint32_t));

    /* alphabet histogram :216-228 */
    for (uint32_t i = 0; i < seq_count; i++)
        for (uint64_t j = offsets[i]; j < offsets[i + 1]; j++) info->rank_counters[residues[j]]++;

    if (rank_init(info, kvalue) != 0) return -1;                        /* :230 */

    /* ranking pass :234-265 */
    uint32_t genomes_count = 0;
    uint64_t m = 0;
    for (uint32_t i = 0; i < seq_count; i++) {
        int64_t kl = (int64_t) (offsets[i + 1] - offsets[i]) - (int64_t) info->kvalue + 1;
        if (kl > 0) m += (uint64_t) kl;
    }
    info->kmer_occurrences = m;
    po_kmer *kmers = (po_kmer *) malloc((m ? m : 1) * sizeof(po_kmer));
    uint64_t *tmp_ranks = NULL;
    uint64_t tmp_cap = 0;
    uint64_t w = 0;
    uint32_t *genome_counts = (uint32_t *) calloc((size_t) seq_count + 1, sizeof(uint32_t));
    for (uint32_t i = 0; i < seq_count; i++) {
        uint32_t gen_id = genome_of[i];
        info->seq_gen_mapping[i] = gen_id;
        if (gen_id + 1 > genomes_count) genomes_count = gen_id + 1;
        if (gen_id < seq_count) genome_counts[gen_id]++;
        uint64_t len = offsets[i + 1] - offsets[i];
        int64_t kseq_len = (int64_t) len - (int64_t) info->kvalue + 1;
        if (kseq_len > 0) {
            info->kseq_lengths[i] = (uint32_t) kseq_len;
            if ((uint64_t) kseq_len > tmp_cap) {
                tmp_cap = (uint64_t) kseq_len * 2;
                tmp_ranks = (uint64_t *) realloc(tmp_ranks, tmp_cap * sizeof(uint64_t));
            }
            do_ranking(info, residues + offsets[i], (uint32_t) len, tmp_ranks);
            for (int64_t j = 0; j < kseq_len; j++) {
                kmers[w].rank = tmp_ranks[j]; kmers[w].seq = i; kmers[w].count = 1; w++;
            }
        } else {
            info->kseq_lengths[i] = 0;
        }
    }
    free(tmp_ranks);
    info->genomes_count = genomes_count;                                /* :267 */
    info->kmers = kmers;
    info->kmers_size = m;

    /* genome_sequences :244,268 as CSR */
    info->genome_seq_off = (uint32_t *) calloc((size_t) genomes_count + 1, sizeof(uint32_t));
    info->genome_seq = (uint32_t *) malloc((seq_count ? seq_count : 1) * sizeof(uint32_t));
    for (uint32_t g = 0; g < genomes_count; g++) info->genome_seq_off[g + 1] = info->genome_seq_off[g] + genome_counts[g];
    {
        uint32_t *cur = (uint32_t *) malloc(((size_t) genomes_count + 1) * sizeof(uint32_t));
        memcpy(cur, info->genome_seq_off, ((size_t) genomes_count + 1) * sizeof(uint32_t));
        for (uint32_t i = 0; i < seq_count; i++) info->genome_seq[cur[genome_of[i]]++] = i;
        free(cur);
    }
    free(genome_counts);

    info->total_visited = (uint64_t *) calloc(seq_count ? seq_count : 1, sizeof(uint64_t));
    info->range_off = (uint64_t *) calloc((size_t) seq_count + 1, sizeof(uint64_t));
    if (m == 0) return -2;   /* reference dereferences kmers.begin() of an empty vector, :297 */

    /* LSD radix sort :270-278 */
    po_kmer *tmp = (po_kmer *) malloc(m * sizeof(po_kmer));
    {
        uint32_t max_sortval = 1;
        for (int b = 0; max_sortval < seq_count; b++) {
            counting_sort_ext(kmers, tmp, m, (unsigned) b, 0);
            max_sortval *= 256;
        }
    }
    for (int b = 0; b < info->rank_byte_order; b++) counting_sort_ext(kmers, tmp, m, (unsigned) b, 1);
    free(tmp);

    /* dedup :280-287 */
    uint64_t uniqi = 0;
    for (uint64_t i = 1; i < m; i++) {
        if (kmers[uniqi].seq == kmers[i].seq && kmers[uniqi].rank == kmers[i].rank) {
            kmers[uniqi].count += 1;
        } else {
            kmers[++uniqi] = kmers[i];
            kmers[uniqi].count = 1;
        }
    }
    const uint64_t u = uniqi + 1;
    info->kmers_size = u;

    /* group scan :297-335.  The scan itself runs once (it sorts groups in place, so it cannot be
       replayed); the shared groups it finds are remembered and the per-sequence range lists are
       then filled group by group, i.e. in the order the reference push_back()s them */
    uint64_t *grp = NULL;           /* (begin,end) pairs of groups with > 1 record */
    uint64_t n_grp = 0, cap_grp = 0;
    {
        uint64_t current_rank = kmers[0].rank;
        uint64_t current_rank_start = 0;
        for (uint64_t i = 0; i < u; i++) {
            int last = (i == u - 1);
            if (current_rank != kmers[i].rank || last) {
                uint64_t prev_rank_begin = current_rank_start;
                uint64_t prev_rank_end = i + (last ? 1 : 0);   /* :306 - last record folded in */
                if (prev_rank_end - prev_rank_begin > 1) {
                    if (!only_complexity)                      /* :310-316 */
                        qsort(kmers + prev_rank_begin, prev_rank_end - prev_rank_begin, sizeof(po_kmer), cmp_seq);
                    for (uint64_t j = prev_rank_begin; j < prev_rank_end; j++) {
                        info->total_visited[kmers[j].seq] += prev_rank_end - prev_rank_begin;  /* :327 */
                        if (!only_complexity) info->range_off[kmers[j].seq + 1]++;
                    }
                    if (!only_complexity) {
                        if (n_grp == cap_grp) {
                            cap_grp = cap_grp ? cap_grp * 2 : 1024;
                            grp = (uint64_t *) realloc(grp, cap_grp * 2 * sizeof(uint64_t));
                        }
                        grp[2 * n_grp] = prev_rank_begin; grp[2 * n_grp + 1] = prev_rank_end; n_grp++;
                    }
                }
                current_rank_start = i;
                current_rank = kmers[i].rank;
            }
        }
    }
    if (!only_complexity) {
        for (uint32_t s = 0; s < seq_count; s++) info->range_off[s + 1] += info->range_off[s];
        info->ranges = (po_range *) malloc((info->range_off[seq_count] ? info->range_off[seq_count] : 1) * sizeof(po_range));
        uint64_t *fill = (uint64_t *) malloc(((size_t) seq_count + 1) * sizeof(uint64_t));
        memcpy(fill, info->range_off, ((size_t) seq_count + 1) * sizeof(uint64_t));
        for (uint64_t gi = 0; gi < n_grp; gi++) {                                   /* :318-326 */
            for (uint64_t j = grp[2 * gi]; j < grp[2 * gi + 1]; j++) {
                po_range r = { grp[2 * gi], j, grp[2 * gi + 1] };
                info->ranges[fill[kmers[j].seq]++] = r;
            }
        }
        free(fill);
    }
    free(grp);
    return 0;
}

/* library.cpp:409-527 + SoA split :542-603 */
int po_compute_scores(const po_ctx *info, uint32_t genome, po_scores *out) {
    memset(out, 0, sizeof(*out));
    if (genome >= info->genomes_count) return -1;
    const uint32_t n = info->sequences_count, G = info->genomes_count;
    const uint32_t *sequences = info->genome_seq + info->genome_seq_off[genome];
    const uint32_t row_seqs_count = info->genome_seq_off[genome + 1] - info->genome_seq_off[genome];

    float *max_scores = (float *) calloc((size_t) row_seqs_count * G + 1, sizeof(float));   /* :417 */
    float *col_max_scores = (float *) calloc((size_t) n + 1, sizeof(float));                /* :418 */
    int *row_intersection_size = (int *) calloc((size_t) n + 1, sizeof(int));               /* :421 */
    int *row_connection_perc_cnt = (int *) calloc((size_t) n + 1, sizeof(int));
    int *row_transposed_perc_cnt = (int *) calloc((size_t) n + 1, sizeof(int));
    unsigned *reset_colors = (unsigned *) calloc((size_t) n + 1, sizeof(unsigned));
    int *colored_cells = (int *) malloc(((size_t) n + 1) * sizeof(int));
    int32_t *flat_map = (int32_t *) malloc(((size_t) n + 1) * sizeof(int32_t));           /* :428 */
    for (uint32_t i = 0; i < n; i++) flat_map[i] = INT32_MAX;
    for (uint32_t i = 0; i < row_seqs_count; i++) flat_map[sequences[i]] = (int32_t) i;

    uint64_t cap = 1024, z = 0;
    float *sc = (float *) malloc(cap * sizeof(float)), *pc = (float *) malloc(cap * sizeof(float)),
          *tr = (float *) malloc(cap * sizeof(float));
    int32_t *xr = (int32_t *) malloc(cap * sizeof(int32_t)), *yc = (int32_t *) malloc(cap * sizeof(int32_t));

    uint64_t max_ranges = 0;
    for (uint32_t i = 0; i < row_seqs_count; i++) {
        uint64_t c = info->range_off[sequences[i] + 1] - info->range_off[sequences[i]];
        if (c > max_ranges) max_ranges = c;
    }
    uint64_t *pointers = (uint64_t *) malloc((max_ranges ? max_ranges : 1) * sizeof(uint64_t));

    unsigned color = 0;
    const po_kmer *kmers = info->kmers;
    for (uint32_t ri = 0; ri < row_seqs_count; ri++) {
        const uint32_t row = sequences[ri];
        color++;
        uint64_t n_colored = 0;
        const po_range *rr = info->ranges + info->range_off[row];
        const uint64_t nr = info->range_off[row + 1] - info->range_off[row];
        for (uint64_t i = 0; i < nr; i++) pointers[i] = rr[i].start;            /* :448-451 */

        const int sequences_step = 2048;                                           /* :454 */
        for (int64_t max_allowed_sequence = sequences_step;
             max_allowed_sequence < (int64_t) n + sequences_step;
             max_allowed_sequence += sequences_step) {                             /* :456-458 */
            for (uint64_t index = 0; index < nr; index++) {
                uint64_t it = pointers[index];
                const unsigned my_cnt = kmers[rr[index].current].count;
                while (it < rr[index].end && (int64_t) kmers[it].seq <= max_allowed_sequence) {   /* :464 */
                    const uint32_t s = kmers[it].seq;
                    if (reset_colors[s] != color) {                                /* :467-473 */
                        reset_colors[s] = color;
                        colored_cells[n_colored++] = (int) s;
                        row_intersection_size[s] = 0;
                        row_connection_perc_cnt[s] = 0;
                        row_transposed_perc_cnt[s] = 0;
                    }
                    row_intersection_size[s] += (int) (kmers[it].count < my_cnt ? kmers[it].count : my_cnt);
                    row_connection_perc_cnt[s] += (int) my_cnt;
                    row_transposed_perc_cnt[s] += (int) kmers[it].count;
                    it++;
                }
                pointers[index] = it;
            }
        }
        row_intersection_size[row] = 0;                                            /* :485-487 */
        row_connection_perc_cnt[row] = 0;
        row_transposed_perc_cnt[row] = 0;

        for (uint64_t ci = 0; ci < n_colored; ci++) {                              /* :493-517 */
            const uint32_t col_index = (uint32_t) colored_cells[ci];
            int my_kcnt = (int) info->kseq_lengths[row];
            int other_kcnt = (int) info->kseq_lengths[col_index];
            int union_size = my_kcnt + other_kcnt - row_intersection_size[col_index];
            float perc = (float) row_connection_perc_cnt[col_index] / (float) my_kcnt;
            float tr_perc = (float) row_transposed_perc_cnt[col_index] / (float) other_kcnt;
            float threshold = 1.0f / (2.0f * (float) info->kvalue);
            int score_valid = perc >= threshold || tr_perc >= threshold;
            float score = (float) row_intersection_size[col_index] / (float) union_size *
                          (score_valid ? 1.0f : 0.0f);
            if (score > 0.0f) {
                if (z == cap) {
                    cap *= 2;
                    sc = (float *) realloc(sc, cap * sizeof(float)); pc = (float *) realloc(pc, cap * sizeof(float));
                    tr = (float *) realloc(tr, cap * sizeof(float));
                    xr = (int32_t *) realloc(xr, cap * sizeof(int32_t)); yc = (int32_t *) realloc(yc, cap * sizeof(int32_t));
                }
                sc[z] = score; pc[z] = perc; tr[z] = tr_perc; xr[z] = (int32_t) row; yc[z] = (int32_t) col_index; z++;
                float *cur = &max_scores[(size_t) ri * G + info->seq_gen_mapping[col_index]];   /* :405-407,513 */
                if (score > *cur) *cur = score;
                if (score > col_max_scores[col_index]) col_max_scores[col_index] = score;
            }
        }
    }
    free(pointers); free(row_intersection_size); free(row_connection_perc_cnt);
    free(row_transposed_perc_cnt); free(reset_colors); free(colored_cells);

    out->count = (uint32_t) z; out->rows = row_seqs_count; out->genomes = G; out->sequences = n;
    out->scores = sc; out->percs = pc; out->tr_percs = tr; out->row = xr; out->column = yc;
    out->first_seq_genome = (int32_t *) malloc((z ? z : 1) * sizeof(int32_t));    /* :571-575 */
    out->second_seq_genome = (int32_t *) malloc((z ? z : 1) * sizeof(int32_t));
    for (uint64_t i = 0; i < z; i++) {
        out->first_seq_genome[i] = (int32_t) info->seq_gen_mapping[xr[i]];
        out->second_seq_genome[i] = (int32_t) info->seq_gen_mapping[yc[i]];
    }
    out->max_genome_score = max_scores;
    out->max_genome_score_col = col_max_scores;
    out->scoresMaxMappings = flat_map;
    return 0;
}

void po_free_scores(po_scores *s) {
    free(s->scores); free(s->percs); free(s->tr_percs); free(s->row); free(s->column);
    free(s->first_seq_genome); free(s->second_seq_genome); free(s->max_genome_score);
    free(s->max_genome_score_col); free(s->scoresMaxMappings);
    memset(s, 0, sizeof(*s));
}

uint32_t po_sequences(const po_ctx *c) { return c->sequences_count; }
uint32_t po_genomes(const po_ctx *c) { return c->genomes_count; }
uint32_t po_rank_base(const po_ctx *c) { return c->rank_base; }
uint32_t po_rank_byte_order(const po_ctx *c) { return c->rank_byte_order; }
int po_hash_fallback(const po_ctx *c) { return c->hash_fallback; }
uint64_t po_last_multiplier(const po_ctx *c) { return c->last_multiplier; }
const uint8_t *po_rank_values(const po_ctx *c) { return c->rank_values; }
uint64_t po_dict_size(const po_ctx *c) { return c->kmers_size; }
const po_kmer *po_dict(const po_ctx *c) { return c->kmers; }
uint64_t po_kmer_occurrences(const po_ctx *c) { return c->kmer_occurrences; }
const uint32_t *po_kseq_lengths(const po_ctx *c) { return c->kseq_lengths; }
const uint64_t *po_total_visited(const po_ctx *c) { return c->total_visited; }
uint64_t po_total_cost(const po_ctx *c) {
    uint64_t t = 0;
    for (uint32_t i = 0; i < c->sequences_count; i++) t += c->total_visited[i];
    return t;
}
uint64_t po_genome_cost(const po_ctx *c, uint32_t g) {
    uint64_t t = 0;
    if (g >= c->genomes_count) return 0;
    for (uint32_t i = c->genome_seq_off[g]; i < c->genome_seq_off[g + 1]; i++) t += c->total_visited[c->genome_seq[i]];
    return t;
}
