/*
 * pangene_oracle.h — CPU ORACLE (test infrastructure; NOT part of the shipped product).
 *
 * Plain-C restatement of the reference's native hot path, ig/native/library.cpp, function by
 * function.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * Parity status: PINNED — checked cell-for-cell (float32 bit patterns, emission order, cost
 * counters) against the reference's own library.cpp compiled in place (oracle/_ref, see
 * oracle/Makefile) by tests/test_oracle_vs_reference.py, and against the fixtures that build
 * produced under tests/golden/.
 */
#ifndef PANGENE_ORACLE_H
#define PANGENE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct po_ctx po_ctx;

/* One dictionary record, library.cpp:21-39 (kmer_rank) */
typedef struct {
    uint64_t rank;
    uint32_t seq;
    uint32_t count;
} po_kmer;

/* Flat mirror of ig/infoasys/cli/pangenes/Scores.java:4-34 as filled by library.cpp:542-603 */
typedef struct {
    uint32_t count;               /* scoresCount */
    uint32_t rows;                /* rows of max_genome_score */
    uint32_t genomes;             /* columns of max_genome_score */
    uint32_t sequences;           /* length of the two N-sized arrays */
    float *scores, *percs, *tr_percs;
    int32_t *row, *column, *first_seq_genome, *second_seq_genome;
    float *max_genome_score;      /* [rows][genomes] row-major */
    float *max_genome_score_col;  /* [sequences] */
    int32_t *scoresMaxMappings;   /* [sequences] */
} po_scores;

po_ctx *po_create(void);
void po_destroy(po_ctx *);

/*
 * preprocessSequences (library.cpp:189-371).  Sequences are handed over flattened:
 * residues[offsets[i] .. offsets[i+1]) is gene i (one byte per UTF-16 unit; the reference
 * ignores units >= 256 in the histogram and indexes out of bounds on them later, so the
 * domain is Latin-1 bytes), genome_of[i] its dense genome id (PangeneIData.java:56-62).
 * Returns 0, or -1 for k <= 0 (the reference prints and exit(1)s, library.cpp:90-93),
 * -2 for an empty dataset (undefined behaviour in the reference, library.cpp:297).
 */
int po_preprocess(po_ctx *, const uint8_t *residues, const uint64_t *offsets,
                  const uint32_t *genome_of, uint32_t n_seqs, int k, int only_complexity);

/* computeScores (library.cpp:409-527) + the SoA split of library.cpp:542-603 */
int po_compute_scores(const po_ctx *, uint32_t genome, po_scores *out);
void po_free_scores(po_scores *);

/* Introspection for kernel-level parity tests */
uint32_t po_sequences(const po_ctx *);
uint32_t po_genomes(const po_ctx *);
uint32_t po_rank_base(const po_ctx *);           /* alphabet size B, library.cpp:100 */
uint32_t po_rank_byte_order(const po_ctx *);     /* library.cpp:121-131 */
int po_hash_fallback(const po_ctx *);            /* library.cpp:115 */
uint64_t po_last_multiplier(const po_ctx *);     /* library.cpp:101-119 */
const uint8_t *po_rank_values(const po_ctx *);   /* [256], library.cpp:96-99 */
uint64_t po_dict_size(const po_ctx *);           /* U */
const po_kmer *po_dict(const po_ctx *);          /* U records after dedup (+ Q1 re-sort) */
uint64_t po_kmer_occurrences(const po_ctx *);    /* M */
const uint32_t *po_kseq_lengths(const po_ctx *); /* [N], library.cpp:250-262 */
const uint64_t *po_total_visited(const po_ctx *);/* [N], library.cpp:327 */
uint64_t po_total_cost(const po_ctx *);          /* "Total cost", library.cpp:337-349 */
uint64_t po_genome_cost(const po_ctx *, uint32_t genome); /* library.cpp:535-538 */

/* All ranks of one gene in order (do_ranking, library.cpp:134-150); out has len-k+1 slots */
int po_rank_gene(const po_ctx *, const uint8_t *chars, uint32_t len, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif
