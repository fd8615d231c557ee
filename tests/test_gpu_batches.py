"""A set the whole-set pass cannot hold: 1024 genomes x 2000 genes x 200 aa (2.05 M genes, ~1.8 x 10^9 emitted cells).

With every cell and its mirror resident the pass needs more than 2^32 cell slots on the device and is refused
(PDL_ERR_UNSUPPORTED: 32-bit cell offsets).  A batch of genomes at a time — the reference's own granularity, one task per genome
with private scratch (Pangenes.java:60-66, library.cpp:417-428) — it runs on one MI355X: one dictionary, the range lists
rebuilt per batch (pdl_set_genome_shard on an existing dictionary).  No reference or oracle reaches this size: checked through
properties — every batch's cells obey (r, c) <-> (c, r) against the batch that holds the mirror, maxima follow from the cells,
the lookups of the batches add up to the dictionary's own "Total cost"."""
import numpy as np
import pytest

from pandelos_amd import _lib
from pandelos_amd.calculate_k import calculate_k
from pandelos_amd.synth import make_gene_set
from tests import helpers as H

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1500)]

SHAPE = dict(genomes=1024, genes_per_genome=2000, mean_len=200, sub_rate=0.08, seed=10241)


def test_1024_genomes_run_in_batches_where_the_whole_set_pass_is_refused():
    import torch
    from pandelos_amd.pangene_native import PangeneNative
    if torch.cuda.mem_get_info(0)[1] < 200 * 2 ** 30:
        pytest.skip("needs an MI355X-sized device")
    gs = make_gene_set(**SHAPE)
    k = calculate_k(gs.residues)
    first = np.searchsorted(gs.genome_of, np.arange(gs.genomes + 1))
    # the whole-set pass: refused, loudly
    nat = PangeneNative.from_arrays(k, gs.residues, gs.offsets, gs.genome_of)
    total_cost = nat.cost.total_cost
    with pytest.raises(_lib.PdlError) as e:
        nat.score_all()
    assert e.value.code == _lib.PDL_ERR_UNSUPPORTED and "2^32" in str(e.value)
    nat.close()
    # ... in batches of 128 genomes
    nat = PangeneNative.open()
    keep = {0: None, 1: None, 640: None, 1023: None}          # blocks kept for the cross-batch checks
    cells = lookups = 0
    for g, s in nat.scores_in_batches(k, gs.residues, gs.offsets, gs.genome_of, 128):
        cells += int(s.scoresCount)
        if g % 128 == 0:
            lookups += nat.timings()["scored_lookups"]
        if g in keep:
            keep[g] = s.as_dict()
        if g in (5, 900):                                    # maxima follow from the cells (library.cpp:513-515)
            b = s.as_dict()
            ms = np.zeros_like(b["max_genome_score"])
            np.maximum.at(ms, (b["scoresMaxMappings"][b["row"]], b["second_seq_genome"]), b["scores"])
            cm = np.zeros_like(b["max_genome_score_col"])
            np.maximum.at(cm, b["column"], b["scores"])
            assert np.array_equal(H.raw(ms), H.raw(b["max_genome_score"])) and np.array_equal(H.raw(cm), H.raw(b["max_genome_score_col"]))
    assert lookups == total_cost and cells > 10 ** 9
    # (r, c) in genome a's block <-> (c, r) in genome b's block, across batches: same score bits, perc and tr_perc swapped
    for a, b in ((0, 1023), (1, 640), (640, 1023), (0, 1)):
        x, y = keep[a], keep[b]
        mx, my = x["second_seq_genome"] == b, y["second_seq_genome"] == a
        ka = x["row"][mx].astype(np.int64) * gs.genes + x["column"][mx]
        kb = y["column"][my].astype(np.int64) * gs.genes + y["row"][my]
        oa, ob = np.argsort(ka), np.argsort(kb)
        assert len(ka) > 1000 and np.array_equal(ka[oa], kb[ob])
        assert np.array_equal(H.raw(x["scores"][mx][oa]), H.raw(y["scores"][my][ob]))
        assert np.array_equal(H.raw(x["percs"][mx][oa]), H.raw(y["tr_percs"][my][ob]))
    nat.close()
