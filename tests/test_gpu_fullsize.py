"""BASELINE.json configs[2] at full size (the 64-genome stand-in the bench runs) checked through properties that do not
need the oracle at that size: the reference's lookup count from the dictionary's own group sizes, order and
multiplicity invariants of the dictionary, symmetry of the score matrix, and the per-(row, genome) / per-column maxima
recomputed from the emitted cells."""
import numpy as np
import pytest

from pandelos_amd.calculate_k import calculate_k
from pandelos_amd.synth import CONFIGS, make_gene_set

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full():
    from pandelos_amd.pangene_native import PangeneNative
    gs = make_gene_set(**CONFIGS["mycoplasma64_standin"])
    k = calculate_k(gs.residues)
    nat = PangeneNative.from_arrays(k, gs.residues, gs.offsets, gs.genome_of)
    return gs, k, nat


def test_dictionary_invariants_and_lookup_count(full):
    gs, k, nat = full
    c = nat.cost
    lens = np.diff(gs.offsets.astype(np.int64))
    kseq = np.maximum(lens - k + 1, 0)
    assert c.kmer_occurrences == int(kseq.sum()) and c.residues == len(gs.residues)
    ranks, seqs, counts = nat.dictionary()
    assert int(counts.sum()) == c.kmer_occurrences                        # run lengths cover the whole k-mer stream
    assert len(ranks) == c.dictionary_records
    # (rank, gene) strictly ascending, except that the last record may have been folded into the preceding group
    body = slice(0, len(ranks) - 1)
    key = (ranks[body].astype(np.uint64) << np.uint64(32)) | seqs[body].astype(np.uint64)
    last_group_start = int(np.searchsorted(ranks[:-1], ranks[-2], side="left"))
    assert np.all(key[1:last_group_start] > key[:last_group_start - 1])
    # groups: equal ranks; the reference closes the last group at the last record whatever its rank (library.cpp:300-306)
    starts = np.flatnonzero(np.r_[True, ranks[1:-1] != ranks[:-2]])     # heads among records 0..U-2
    sizes = np.diff(np.r_[starts, len(ranks) - 1]).astype(np.int64)
    sizes[-1] += 1                                                       # the last record joins the last open group
    p = int((sizes[sizes >= 2] ** 2).sum())
    assert p == c.total_cost                                             # "Total cost: P lookups"
    cost, kl = nat.sequence_costs()
    assert int(cost.sum()) == c.total_cost and np.array_equal(kl.astype(np.int64), kseq)
    assert sum(nat.genome_cost(g) for g in range(c.genomes)) == c.total_cost


def test_scores_are_symmetric_and_maxima_follow_from_cells(full):
    gs, k, nat = full
    G = nat.cost.genomes
    blocks = [nat.generate_scores_part(g) for g in range(G)]
    row = np.concatenate([b.row for b in blocks]).astype(np.int64)
    col = np.concatenate([b.column for b in blocks]).astype(np.int64)
    sc = np.concatenate([b.scores for b in blocks])
    pc = np.concatenate([b.percs for b in blocks])
    tr = np.concatenate([b.tr_percs for b in blocks])
    assert len(row) == int(nat.scores_counts().sum())
    assert np.all(sc > 0) and np.all(sc <= 1) and np.all(row != col)
    thr = np.float32(1.0) / (np.float32(2.0) * np.float32(k))
    assert np.all((pc >= thr) | (tr >= thr))                              # score_valid (library.cpp:497-500)
    # (r, c) <-> (c, r): same score bits, perc and tr_perc swapped
    key = row * gs.genes + col
    o = np.argsort(key)
    mirror = np.searchsorted(key[o], col * gs.genes + row)
    assert np.all(key[o][mirror] == col * gs.genes + row)
    m = o[mirror]
    assert np.array_equal(sc.view(np.uint32), sc[m].view(np.uint32))
    assert np.array_equal(pc.view(np.uint32), tr[m].view(np.uint32)) and np.array_equal(tr.view(np.uint32), pc[m].view(np.uint32))
    # maxima (library.cpp:513-515) recomputed from the cells of every genome block
    for g in (0, 17, G - 1):
        b = blocks[g]
        ms = np.zeros_like(b.max_genome_score)
        np.maximum.at(ms, (b.scoresMaxMappings[b.row], b.second_seq_genome), b.scores)
        cm = np.zeros_like(b.max_genome_score_col)
        np.maximum.at(cm, b.column, b.scores)
        assert np.array_equal(ms.view(np.uint32), b.max_genome_score.view(np.uint32))
        assert np.array_equal(cm.view(np.uint32), b.max_genome_score_col.view(np.uint32))
        assert np.all(b.first_seq_genome == g) and np.array_equal(b.second_seq_genome, gs.genome_of[b.column].astype(np.int32))
        assert np.all(np.diff(b.row) >= 0)                                # rows ascending inside a block (library.cpp:437)


def test_planted_families_are_the_best_hits(full):
    """Sanity of the workload itself: a gene's best inter-genome hit is a copy of the same planted family."""
    gs, k, nat = full
    b = nat.generate_scores_part(3)
    best = {}
    for r, c, s in zip(b.row, b.column, b.scores):
        if gs.genome_of[c] != 3 and s > best.get(r, (0, -1))[0]:
            best[r] = (s, c)
    same = sum(gs.family_of[r] == gs.family_of[c] for r, (s, c) in best.items())
    assert same >= 0.99 * len(best)
