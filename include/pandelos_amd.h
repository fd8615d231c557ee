/*
 * pandelos_amd.h — C ABI of the MI355X-native PanDelos hot path
 * (k-mer dictionary matching + all-vs-all gene similarity scoring).
 *
 * This is the drop-in boundary.  In the reference the same boundary is the JNI pair
 *     Java_infoasys_cli_pangenes_PangeneNative_preprocessSequences   ig/native/pangene_native.h:16-17
 *     Java_infoasys_cli_pangenes_PangeneNative_computeScores         ig/native/pangene_native.h:24-25
 * implemented by ig/native/library.cpp:189-371 and :529-604 and declared on the Java side at
 * ig/infoasys/cli/pangenes/PangeneNative.java:14-15.  libnative.so (this project's JNI shim,
 * include/pdl_jni_abi.h + pandelos_amd/csrc/jni_shim.cpp) exports those two symbols and forwards
 * to the functions below; any other host (the Python mirror in pandelos_amd/, the C++ driver, a cgo /
 * N-API / ctypes stub) binds the functions below directly.
 *
 * Plain C: pointers and sizes only.  All work runs on one HIP device (gfx950); there is no CPU
 * fallback — pdl_create fails when no device is usable.
 */
#ifndef PANDELOS_AMD_H
#define PANDELOS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDL_API __attribute__((visibility("default")))

typedef struct pdl_ctx pdl_ctx;

enum {
    PDL_OK = 0,
    PDL_ERR_KVALUE = -1,       /* k <= 0: the reference prints "K value must be greater than 0." and exit(1)s (library.cpp:90-93) */
    PDL_ERR_EMPTY = -2,        /* no k-mer at all: undefined behaviour in the reference (library.cpp:297) */
    PDL_ERR_ARGUMENT = -3,     /* NULL pointer, genome id out of range, ... */
    PDL_ERR_DEVICE = -4,       /* a HIP call failed; see pdl_last_error */
    PDL_ERR_STATE = -5,        /* compute before preprocess, etc. */
    PDL_ERR_UNSUPPORTED = -6   /* input outside the implemented domain (stated in the message) */
};

typedef struct {
    int32_t device;            /* HIP device ordinal; -1 = current device */
    void *stream;              /* hipStream_t to run on; NULL = a stream owned by the context */
    uint32_t flags;            /* PDL_FLAG_* */
    uint32_t reserved;
} pdl_config;

#define PDL_FLAG_CANONICAL_ORDER 1u  /* emit each row's cells by ascending column instead of the reference's
                                        first-touch order (library.cpp:456-482,493); cheaper, same cell set */

/* What preprocessSequences prints as its cost model (library.cpp:337-350) plus the sizes every
 * run must report for the roofline (R, M, U, P). */
typedef struct {
    uint64_t residues;          /* R */
    uint64_t kmer_occurrences;  /* M  = sum over genes of max(len-k+1, 0) */
    uint64_t dictionary_records;/* U  = unique (rank, gene) records (library.cpp:280-287) */
    uint64_t shared_records;    /* U' = records in rank-groups with >= 2 records (= posting ranges, library.cpp:318-326) */
    uint64_t groups;            /* rank-groups with >= 2 records (multi-GPU: counted over the whole dictionary as well) */
    uint64_t total_cost;        /* P  = "Total cost: P lookups" (library.cpp:327,349) */
    float linear_ratio;         /* P / sum kseq_lengths (library.cpp:350) */
    uint32_t sequences;         /* N */
    uint32_t genomes;           /* G */
    uint32_t rank_base;         /* alphabet size B (library.cpp:96-100) */
    uint32_t rank_bits;         /* bits needed for a rank (64 in hash mode) */
    int32_t hash_fallback;      /* 1 when B^k overflows 64 bits (library.cpp:103-119) */
    int32_t kvalue;
} pdl_cost;

/* Flat mirror of ig/infoasys/cli/pangenes/Scores.java:4-34 exactly as library.cpp:542-603 fills it.
 * All arrays are host memory owned by the library until pdl_free_scores. */
typedef struct {
    uint32_t scoresCount;
    uint32_t rows;                 /* genes of this genome = first dimension of max_genome_score */
    uint32_t genomes;              /* G = second dimension of max_genome_score */
    uint32_t sequences;            /* N = length of max_genome_score_col and scoresMaxMappings */
    float *scores;                 /* [scoresCount] */
    float *percs;                  /* [scoresCount] */
    float *tr_percs;               /* [scoresCount] */
    int32_t *row;                  /* [scoresCount] */
    int32_t *column;               /* [scoresCount] */
    int32_t *first_seq_genome;     /* [scoresCount] */
    int32_t *second_seq_genome;    /* [scoresCount] */
    float *max_genome_score;       /* [rows][genomes] row-major (Java float[][]) */
    float *max_genome_score_col;   /* [sequences] */
    int32_t *scoresMaxMappings;    /* [sequences]; INT32_MAX for genes of other genomes (library.cpp:428-432) */
} pdl_scores;

/* Device time of the stages of the last preprocess / score pass, from HIP events on the context's
 * stream (milliseconds), and the counters the roofline needs. */
typedef struct {
    float hist_ms, rank_ms, sort_rank_ms, dict_ms, sort_seq_ms, ranges_ms;  /* preprocess */
    float join_ms, join_overflow_ms, order_ms;                             /* scoring: the join's three tiers; of which the HBM-table tier; K-order */
    float preprocess_total_ms, score_total_ms;
    uint64_t emitted_cells;        /* Z over the genomes scored by this context */
    uint64_t scored_rows;          /* rows (genes) scored by this context */
    uint64_t scored_lookups;       /* P restricted to those rows */
    uint64_t overflow_rows;        /* rows that left the LDS tables for the HBM table (tier 3) */
    uint32_t join_launches;
    uint32_t tier2_rows;           /* rows the filtered small-table tier handed to the big LDS table */
    /* multi-GPU passes (zero otherwise) */
    float dist_begin_ms, dist_finish_ms;   /* pdl_dist_preprocess_begin / _finish, device time */
    float dist_score_begin_ms, dist_score_finish_ms;
    uint64_t walked_lookups;       /* postings the join actually read: each unordered pair of a group once when rows only meet the genes above them */
    uint64_t outbox_cells, inbox_cells;
    uint64_t aside_reloads;        /* join, filter tier: entries of a put-aside list that did not yet show what the wave had stored there
                                      when it first read them back (they are loaded again until they do) */
    uint32_t aside_repeats;        /* scoring passes thrown away and repeated with fully tagged (16-byte) list entries because a pass with
                                      the 10-bit tags saw a reload: no result is ever returned from such a pass */
    uint32_t tier1_rows;           /* rows that reached the filtered small-table tier (all of them unless the partition tier ran in front) */
    float reshard_ms;              /* the range lists built again for another genome shard on the existing dictionary (pdl_set_genome_shard after pdl_preprocess) */
    float dist_ranges_ms;          /* pdl_dist_preprocess_ranges, device time (0 when the owners build the range lists) */
} pdl_timings;

PDL_API pdl_ctx *pdl_create(const pdl_config *cfg /* may be NULL */);
PDL_API void pdl_destroy(pdl_ctx *);
/* Message of the last failure on this context (or of pdl_create when ctx == NULL); never NULL. */
PDL_API const char *pdl_last_error(const pdl_ctx *);

/* preprocessSequences (library.cpp:189-371).  Gene i is residues[offsets[i] .. offsets[i+1]), one byte
 * per character (the reference reads UTF-16 units and is only defined for units < 256, library.cpp:76,223);
 * genome_of[i] is its dense genome id in first-seen order (PangeneIData.java:56-62).  Resets the context
 * (library.cpp:192).  only_complexity != 0 stops after the cost model (PangeneNative.java:10-12).
 * The _host form takes host pointers and copies them to the device; the _device form takes device
 * pointers that must stay valid until the next preprocess or destroy. */
PDL_API int pdl_preprocess(pdl_ctx *, const uint8_t *residues, const uint64_t *offsets,
                           const uint32_t *genome_of, uint32_t n_sequences, int kvalue,
                           int only_complexity, pdl_cost *out_cost /* may be NULL */);
PDL_API int pdl_preprocess_device(pdl_ctx *, const uint8_t *d_residues, const uint64_t *d_offsets,
                                  const uint32_t *d_genome_of, uint32_t n_sequences, uint64_t n_residues,
                                  int kvalue, int only_complexity, pdl_cost *out_cost);

/* ---- ingest: .faa -> HBM (the Java side's PangeneIData.readFromFile, PangeneIData.java:30-75, and calculate_k.py:9-30) --
 * One pass over the mapped file: readLine's terminators (\n, \r, \r\n), String.trim, blank lines skipped, header / sequence
 * alternating, genome = the header's text before the first tab, ids dense in first-seen order; a header with fewer than three
 * tab-separated fields fails (the reader indexes cc[1], cc[2]).  Sequence bytes are copied once, into pinned staging buffers
 * that leave for the device while the parser goes on.  k_suggested is what calculate_k.py prints for the same file (raw line
 * parity, str.strip, entropy summed in first-seen letter order; 0 where the script would divide by zero). */
typedef struct {
    uint64_t file_bytes;
    uint64_t residues;             /* R */
    uint32_t sequences, genomes;   /* N, G */
    int32_t k_suggested;
    uint32_t reserved;
    double parse_ms;               /* wall time: open -> last byte on the device (pdl_scan_faa: -> parsed) */
    const uint64_t *offsets;       /* host [sequences + 1], owned by the context until its next ingest (pdl_scan_faa: NULL) */
    const uint32_t *genome_of;     /* host [sequences] */
    const uint8_t *d_residues;     /* device, owned by the context: the arguments of pdl_preprocess_device / pdl_dist_preprocess_begin */
    const uint64_t *d_offsets;
    const uint32_t *d_genome_of;
} pdl_ingest;
PDL_API int pdl_ingest_faa(pdl_ctx *, const char *path, pdl_ingest *out);
/* genome names in id order (PangeneIData.genomeNames); NULL when out of range */
PDL_API const char *pdl_ingest_genome_name(const pdl_ctx *, uint32_t genome);
/* preprocessSequences on the ingested input (no copy, no read-back of the genome ids) */
PDL_API int pdl_preprocess_ingested(pdl_ctx *, int kvalue, int only_complexity, pdl_cost *out_cost /* may be NULL */);
/* The same parser without a device or a context, into caller buffers (each may be NULL: count only).  Errors: pdl_last_error(NULL). */
PDL_API int pdl_scan_faa(const char *path, pdl_ingest *out, uint8_t *residues, uint64_t cap_residues, uint64_t *offsets /* [cap_sequences + 1] */,
                         uint32_t *genome_of, uint32_t cap_sequences);

/* "Genome g cost = ..." (library.cpp:535-538) */
PDL_API int pdl_genome_cost(const pdl_ctx *, uint32_t genome, uint64_t *out_cost);
/* Per-gene cost (computation_costs[].total_visited, library.cpp:327) and k-mer count (kseq_lengths, :250-262) */
PDL_API int pdl_sequence_costs(const pdl_ctx *, uint64_t *out_total_visited /* [N] */, uint32_t *out_kseq_lengths /* [N], may be NULL */);

/* Restrict the genomes this context scores (multi-GPU sharding: one context per GPU, each with the
 * full dictionary postings and a disjoint genome list).  Must be set BEFORE pdl_preprocess: the
 * posting-range lists, per-gene costs and pdl_cost.total_cost are then built for the shard's genes only
 * (the part of the dictionary build that is proportional to the rows scored).  Without a shard the
 * range lists hold only the genes above each row and every cell is produced once for both of its rows, so
 * a dictionary built for all genomes cannot score a subset (PDL_ERR_STATE).  A dictionary built for a shard can be
 * narrowed further — or pointed at ANOTHER shard: the postings stay, the range lists of the new shard's genes are built
 * before the next scoring pass (two passes over the postings; pdl_timings.reshard_ms).  That is how a set too large for one
 * device's memory is scored a batch of genomes at a time (the reference's unit of work, Pangenes.java:60-66, with the
 * reference's per-task scratch, library.cpp:417-428): shard = batch 0, pdl_preprocess, fetch the batch's Scores, shard =
 * batch 1, fetch, ... — only one batch's maxima, staging and cells are on the device at a time (option "low_memory" also
 * returns the build's transient buffers).  count == 0 clears the shard.  The shard stays in force across pdl_preprocess calls. */
PDL_API int pdl_set_genome_shard(pdl_ctx *, const uint32_t *genomes, uint32_t count);

/* computeScores for every genome of the shard in one device pass (library.cpp:409-527 for each
 * genome); results stay in HBM.  Idempotent until the next preprocess. */
PDL_API int pdl_score_all(pdl_ctx *);

/* computeScores + the marshalling of library.cpp:542-603 for one genome: runs pdl_score_all on first
 * use, then copies that genome's block to the host.  Re-entrant from several host threads, like the
 * reference (Pangenes.java:54-66). */
PDL_API int pdl_compute_scores(pdl_ctx *, uint32_t genome, pdl_scores *out);
PDL_API void pdl_free_scores(pdl_scores *);

/* The host stage behind computeScores, on the device: the bidirectional-best-hit filter of Pangenes.java:98-176 over one
 * genome task's cells.  Returns the edges that task adds to the network in the host's insertion order — phase 1 (inter-genome
 * best hits, (row, column) then (column, row) per cell), then phase 2 (kept intra-genome cells) — i.e. exactly the
 * addConnection calls of Pangenes.java:103-104,171-175; what is left to the host is the network container and its
 * text form (PangeneNet.java).  The first call filters every genome of the context and brings all edges to the host once. */
typedef struct {
    uint32_t count;
    int32_t *src, *dst;            /* [count] gene ids */
    float *score;                  /* [count] */
} pdl_edges;
PDL_API int pdl_compute_edges(pdl_ctx *, uint32_t genome, pdl_edges *out);
PDL_API void pdl_free_edges(pdl_edges *);

/* Number of emitted cells per genome after pdl_score_all ([G], 0 for genomes outside the shard) */
PDL_API int pdl_scores_counts(pdl_ctx *, uint32_t *out_counts);

/* Parity-test introspection: the dictionary in (rank, gene) order as library.cpp:270-287 leaves it
 * (before the group scan's re-sort of the folded last group).  Arrays sized by pdl_cost.dictionary_records. */
PDL_API int pdl_get_dictionary(pdl_ctx *, uint64_t *ranks, uint32_t *seqs, uint32_t *counts);
/* Alphabet rank table (library.cpp:96-99) and B^(k-1) (library.cpp:101-119) */
PDL_API int pdl_get_rank_table(const pdl_ctx *, uint8_t out_rank_values[256], uint64_t *out_last_multiplier);

PDL_API int pdl_get_timings(pdl_ctx *, pdl_timings *out);

/* Tuning / test switches of one context (no environment variable is read by the library).  Unknown names fail with
 * PDL_ERR_ARGUMENT.  Options: "join_tier1" 0|9|10|11|20|21 (table of the join's first tier; -1 = default: by genome count),
 * "join_tier0" -1|0|1 (the partition tier in front of tier 1 — several short rows per workgroup cycle, no hash table: -1 = by the
 * average row length, the default; it stays off when "join_tier1" was set by hand unless forced with 1),
 * "join_tiny_tier2" 0|1 (512-slot second tier, so that small test sets reach the HBM-table kernel), "host_mirror" 0|1
 * (pdl_compute_scores slices ONE pinned copy of the whole result (1, default) or copies each genome's block from the
 * device (0); results above 1 GiB always take the second way), "staging_cap" n (cells of staging the first scoring
 * attempt may use, 0 = estimate; a pass that overflows it is repeated once with the exact size), "join_grid_pct" n (first
 * tier of the join launched with n % of the workgroups the chip holds, 0 = all: an experiment knob — how the join scales
 * with rows in flight), "stage_timers" 0|1 (default 1: HIP events around every stage fill the stage
 * fields of pdl_timings; 0: only the totals and the join's launch time are taken — each event pair is two marker packets
 * between dispatches, a few microseconds of idle stream on a two-millisecond step), "low_memory" 0|1 (for genome batches on a large set: the buffers only the dictionary build needed are released after it —
 * pdl_get_dictionary is then not available — and tier 3's tables in HBM take 1 GB instead of 8), "onepass_scan" 0|1 (prefix scans
 * in one launch with decoupled look-back instead of three launches; measured slower on MI355X, default 0), "aside_test_reload" 0|1 (test switch: the next scoring pass
 * behaves as if an entry of a put-aside list had needed a second look, so the repeat with fully tagged entries runs). */
PDL_API int pdl_set_option(pdl_ctx *, const char *name, int64_t value);

/* ---- multi-GPU: one context per GPU, the caller moves bytes between them (RCCL over xGMI) -----------------------
 * Counterpart of the reference's per-genome tasks on a thread pool over ONE shared dictionary (Pangenes.java:54-66,
 * library.cpp:73): here every GPU holds the whole dictionary, built co-operatively, and scores a disjoint set of genomes.
 * All ranks call the same sequence with the SAME input (every host reads the .faa); nothing below communicates by
 * itself — the caller owns the collectives (pandelos_amd/distributed.py: torch.distributed):
 *
 *   pdl_dist_preprocess_begin   histogram, rank table, k-mer ranks (every rank, from the shared input); the rank space is
 *                               cut into `world` intervals of ~equal k-mer counts (same cuts on every rank: they depend on
 *                               the input only) and this rank sorts + dedups ITS interval (library.cpp:270-287 on 1/world
 *                               of the records).  Result: its run of the dictionary, pdl_dist_slice.
 *   -- caller: all-gather the record counts, then the runs themselves into ONE device array in rank order
 *      (the concatenation IS the dictionary in (rank, gene) order: no merge) --
 *   pdl_dist_preprocess_finish  adopts that array (must stay valid until the next preprocess), deals the genomes to ranks
 *                               (longest-processing-time on each genome's lookups above the diagonal: identical input,
 *                               identical deal on every rank) and builds the rank-groups and posting-range lists
 *                               (library.cpp:289-335) of this rank's genes.
 *   pdl_dist_score_begin        scores this rank's rows against the genes ABOVE them only; a cell whose column belongs to
 *                               another rank's genome is also that rank's cell (c, r) (same sums, perc/tr_perc swapped):
 *                               such cells are listed per destination rank, pdl_dist_outbox.
 *   -- caller: all-to-all the per-destination counts, then the 24-byte cells --
 *   pdl_dist_score_finish       files the received cells with the local ones, folds them into the per-(row, genome) and
 *                               per-column maxima and puts every row in the reference's emission order.
 * Afterwards pdl_compute_scores / pdl_scores_counts work for the genomes of this rank (pdl_dist_genome_owner).
 * pdl_genome_cost and pdl_cost.total_cost cover this rank's genomes only (the reference prints "Genome g cost" from the
 * task that scores g, library.cpp:535-538); "Total cost" is their sum over the ranks — a scalar all-reduce of the caller.
 * The counters over the dictionary (records, shared records, groups) are complete on every rank. */
typedef struct {
    const void *d_postings;   /* device, records x 8 bytes {gene u32, count u32 | group-head flag in bit 31} */
    uint64_t records;         /* unique (rank, gene) records of this rank's interval */
    uint64_t kmers;           /* k-mer occurrences of this rank's interval */
    const uint64_t *genome_weights; /* host, [genomes]: every genome's lookups above the diagonal inside this run; summed over
                                 the ranks they are the weights of the genome deal; valid until the next call on this context */
    uint32_t genomes;
    const uint64_t *genome_costs;   /* host, [genomes]: every genome's lookups inside this run as the reference counts them (library.cpp:327);
                                 summed over the ranks: "Genome g cost" (exact whenever pdl_dist_preprocess_ranges says "available") */
} pdl_dist_slice;

/* Range lists built by the senders (the default flow of pandelos_amd/distributed.py).  pdl_dist_preprocess_finish makes every rank
 * walk the whole gathered dictionary twice to find the ranges of its own genes — work that does not shrink with the number of
 * ranks.  Groups never straddle runs, so between begin and finish each rank can make the range tuples of ALL genes of its run
 * (library.cpp:289-335 on 1/world of the records) and file them by the rank that owns the gene:
 *
 *   pdl_dist_preprocess_begin
 *   -- caller: all-gather [records | genome_weights | genome_costs] --
 *   pdl_dist_preprocess_ranges         deals the genomes (as _finish would), builds + files the tuples of this run
 *   -- caller: all-gather [counts[world] | shared_records | groups | repeat_sample]; all-to-all the tuples (keys: 4 bytes, ranges:
 *      8 bytes, same split sizes); the runs into one device array as before (AFTER this call: it takes the group-head bits out of
 *      the run) --
 *   pdl_dist_preprocess_finish_ranges  adopts the dictionary, sorts the received tuples (source-rank major) by gene
 *
 * `available` = 0 (gene ids beyond 22 bits, more than 256 ranks, a last run of exactly one record — the same verdict on every rank:
 * it depends on the input and the record counts only): nothing was built, the caller goes on with pdl_dist_preprocess_finish. */
typedef struct {
    int available;
    const uint32_t *d_keys;         /* device, tuples grouped by destination rank, record order inside: (owner << 24 | gene) */
    const uint64_t *d_ranges;       /* device, {first posting in the GATHERED dictionary, postings | min(own count, 1023) << 22} */
    const uint64_t *counts;         /* host, [world]: tuples for each rank; valid until the next call on this context */
    uint64_t total;
    uint64_t shared_records, groups, repeat_sample;   /* counters over this run; the caller hands their sums over the ranks to _finish_ranges */
} pdl_dist_ranges;

typedef struct { float score, perc, tr_perc; uint32_t row, column, first_group; } pdl_dist_cell;   /* cell (row, column) as its row's rank computed it */

typedef struct {
    const pdl_dist_cell *d_cells;  /* device, grouped by destination rank in rank order */
    const uint64_t *counts;        /* host, [world]: cells for each rank (0 for this rank itself); valid until the next call on this context */
    uint64_t total;
} pdl_dist_outbox;

PDL_API int pdl_dist_preprocess_begin(pdl_ctx *, const uint8_t *d_residues, const uint64_t *d_offsets, const uint32_t *d_genome_of,
                                      uint32_t n_sequences, uint64_t n_residues, int kvalue, uint32_t world, uint32_t rank,
                                      pdl_dist_slice *out);
/* genome_weights: [G], the element-wise sum of every rank's pdl_dist_slice.genome_weights (one small all-reduce beside
 * the record counts) — the SAME vector on every rank; NULL: the library takes one more pass over the gathered
 * dictionary and computes them itself. */
PDL_API int pdl_dist_preprocess_finish(pdl_ctx *, void *d_postings_all, uint64_t total_records, const uint64_t *genome_weights /* may be NULL */,
                                       pdl_cost *out_cost /* may be NULL */);
/* run_records: [world] records of every rank's run; genome_weights / genome_costs: [G] element-wise sums over the ranks */
PDL_API int pdl_dist_preprocess_ranges(pdl_ctx *, const uint64_t *run_records, const uint64_t *genome_weights, const uint64_t *genome_costs,
                                       pdl_dist_ranges *out);
/* d_keys / d_ranges: the received tuples, source-rank major (what all_to_all leaves); they are sorted in place / into library
 * memory and must stay valid, like d_postings_all, until the next preprocess.  counter_sums: [3] shared_records, groups,
 * repeat_sample summed over the ranks.  pdl_genome_cost then answers for every genome; pdl_sequence_costs is not available. */
PDL_API int pdl_dist_preprocess_finish_ranges(pdl_ctx *, void *d_postings_all, uint64_t total_records, uint32_t *d_keys, uint64_t *d_ranges,
                                              uint64_t n_tuples, const uint64_t *counter_sums /* [3] */, pdl_cost *out_cost /* may be NULL */);
PDL_API int pdl_dist_genome_owner(const pdl_ctx *, uint32_t *out_owner /* [G] */);
PDL_API int pdl_dist_score_begin(pdl_ctx *, pdl_dist_outbox *out);
PDL_API int pdl_dist_score_finish(pdl_ctx *, const pdl_dist_cell *d_inbox, uint64_t n_inbox);
/* dst[0..bytes) = src[0..bytes), both on this context's device, queued on its stream and waited for (the runs and
 * outboxes above live in library-owned memory, the exchange buffers in the caller's). */
PDL_API int pdl_copy_device(pdl_ctx *, void *d_dst, const void *d_src, uint64_t bytes);

/* Test hooks of the small device->host read protocol (a kernel stores counters straight into pinned host memory, then a
 * checksum and an epoch flag; the host spins on both: pandelos_amd/csrc/pdl_common.h, PinRead), on plain host memory:
 * pdl_pin_checksum = what the kernel leaves beside the flag for a payload; pdl_pin_arrived = 1 when `pin` shows `epoch` at
 * flag_word and the words of the n segments (dst_word[s] .. + words[s]) add up to the checksum at flag_word + 1. */
PDL_API int pdl_pin_arrived(const uint32_t *pin, const uint32_t *dst_word, const uint32_t *words, uint32_t n, uint32_t flag_word, uint32_t epoch);
PDL_API uint32_t pdl_pin_checksum(const uint32_t *payload, const uint32_t *dst_word, const uint32_t *words, uint32_t n);

/* Library/build identification, e.g. "pandelos_amd 0.1 gfx950" */
PDL_API const char *pdl_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PANDELOS_AMD_H */
