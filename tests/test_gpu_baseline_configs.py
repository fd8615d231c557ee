"""BASELINE.json's configs at FULL size on the GPU, against the reference itself.

tests/golden/digests_baseline.json holds, per stand-in set of BASELINE.md §4, what the reference's library.cpp
(oracle/_ref, built from /root/reference in the build container) produced: k from the reference's calculate_k.py,
"Total cost", per-genome costs and SHA-256 of every Scores array of every genome (make_golden_baseline.py).
The HIP path must reproduce every digest.  For the canonical 64-genome set the .net the device pipeline writes must
equal the fixture byte for byte — the gene families after netclu_ng.py (.clus fixture beside it, made with the
reference's script) are then identical by construction.

configs[4] (512 x 5000 x 350) is not pinned: the reference needs ~60 GB for it in the build container (64 GB, no swap).
What pins its code path instead: manygenomes_384x400x160, a set of more than 320 genomes (the join tier of configs[4]) the
reference can process here."""
import gzip
import json

import numpy as np
import pytest

from pandelos_amd.calculate_k import calculate_k
from pandelos_amd.synth import make_gene_set
from tests import helpers as H

pytestmark = pytest.mark.gpu

BASE = json.loads((H.GOLDEN / "digests_baseline.json").read_text())


def _check(name, world=1):
    d = BASE[name]
    gs = make_gene_set(**d["shape"])
    assert gs.genes == d["sequences"] and gs.genomes == d["genomes"]
    assert calculate_k(gs.residues) == d["k"]                      # the reference's calculate_k.py said so
    if world == 1:
        from pandelos_amd.pangene_native import PangeneNative
        nat = PangeneNative.from_arrays(d["k"], gs.residues, gs.offsets, gs.genome_of)
        cost, get, costs = nat.cost, nat.generate_scores_part, nat.genome_cost
    else:
        from tests.test_gpu_dist import _local
        lr, cost = _local(world, gs.residues, gs.offsets, gs.genome_of, d["k"])
        lr.score_all()
        get, costs = lr.generate_scores_part, lr.genome_cost
        cost = type("Total", (), {"total_cost": lr.total_cost})
    assert cost.total_cost == d["total_cost"]
    if d["genome_cost"] is not None:
        assert [costs(g) for g in range(d["genomes"])] == d["genome_cost"]
    H.assert_scores_match_digest(lambda g: get(g).as_dict(), d, name)


@pytest.mark.parametrize("name", ["salmonella7_standin", "xanthomonas14_standin", "mycoplasma64_standin"])
def test_config_matches_the_reference_digests(name):
    _check(name)


@pytest.mark.skipif("synthetic_128x4000x300" not in BASE, reason="digests of configs[3] not generated")
@pytest.mark.timeout(900)
def test_config3_128x4000x300_matches_the_reference_digests():
    _check("synthetic_128x4000x300")


@pytest.mark.skipif("synthetic_128x4000x300" not in BASE, reason="digests of configs[3] not generated")
@pytest.mark.timeout(900)
def test_config3_under_the_2048_slot_tier_matches_the_reference_digests():
    """configs[3] (511 855 genes, 6.8 x 10^9 lookups, 55 M cells) pushed through the join tier that sets of more than 320
    genomes get: the kernel of configs[4] at a scale the reference pins."""
    from pandelos_amd.pangene_native import PangeneNative
    d = BASE["synthetic_128x4000x300"]
    gs = make_gene_set(**d["shape"])
    nat = PangeneNative.open()
    nat.set_option("join_tier1", 11)
    nat.preprocess(d["k"], gs.residues, gs.offsets, gs.genome_of)
    assert nat.cost.total_cost == d["total_cost"]
    H.assert_scores_match_digest(lambda g: nat.generate_scores_part(g).as_dict(), d, "configs[3] tier 11")
    assert nat.timings()["aside_reloads"] == 0


@pytest.mark.skipif("manygenomes_384x400x160" not in BASE, reason="digests of the 384-genome set not generated")
@pytest.mark.parametrize("world", [1, 4])
def test_more_than_320_genomes_match_the_reference_digests(world):
    """Not a BASELINE config: 384 genomes x 400 genes x 160 aa, a set the reference can process in the build container and that
    takes the join's 2048-slot filter tier by itself — the kernel configs[4] runs on, whose own results nothing can pin.
    One GPU and four ranks, every digest of the reference."""
    _check("manygenomes_384x400x160", world)


@pytest.mark.parametrize("world", [2, 8])
def test_canonical_set_sharded_matches_the_reference_digests(world):
    _check("mycoplasma64_standin", world)


def test_canonical_set_net_is_the_fixture(tmp_path):
    from pandelos_amd import pangenes as PH
    name = "mycoplasma64_standin"
    gs = make_gene_set(**BASE[name]["shape"])
    faa, net = tmp_path / "in.faa", tmp_path / "out.net"
    gs.write_faa(faa)
    assert PH.main(["-i", str(faa), "-k", str(BASE[name]["k"]), "-o", str(net)]) == 0
    want = gzip.open(H.GOLDEN / "net" / f"{name}.net.gz", "rb").read()
    assert net.read_bytes() == want


def test_repeated_scoring_reproduces_the_digest_every_time():
    """A regression test for a race that showed as a few extra cells in one run out of six: the join's list of lookups
    put aside (first sightings) was read back by other waves than the ones that wrote it, and — rarely, depending on how
    the kernel happened to be scheduled — a wave found the previous row's entries there.  The near-identical genomes of
    this set (a row's neighbour is the homolog of its homolog's neighbour) turn one stale entry into a wrong cell.
    It came back once more, with every wave reading only its own entries, when an unrelated edit changed the kernel's timing
    again (+1 to +3 cells in two passes out of four): the entries were stored `sc1`, which sends the line out of L2, and an
    `sc1` load behind them could reach memory first.  Now the stores are plain (the line stays in the XCD's L2), the loads
    `sc1`, and every entry names the row and the launch it was written for — one that does not is loaded again
    (pdl_timings.aside_reloads counts those: 0 in every run so far)."""
    from pandelos_amd.pangene_native import PangeneNative
    name = "salmonella7_standin"
    d = BASE[name]
    gs = make_gene_set(**d["shape"])
    for it in range(12):
        nat = PangeneNative.from_arrays(d["k"], gs.residues, gs.offsets, gs.genome_of)
        got = [int(nat.generate_scores_part(g).scoresCount) for g in range(d["genomes"])]
        assert got == d["scoresCount"], f"pass {it}: {[a - b for a, b in zip(got, d['scoresCount'])]}"
        if it % 4 == 0:
            H.assert_scores_match_digest(lambda g: nat.generate_scores_part(g).as_dict(), d, f"{name} pass {it}")
        tm = nat.timings()
        assert tm["aside_reloads"] == 0 and tm["aside_repeats"] == 0
        nat.close()
