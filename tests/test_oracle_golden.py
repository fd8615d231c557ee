"""The oracle (plain-C restatement, oracle/pangene_oracle.c) against the golden vectors that
the reference's own library.cpp produced (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H


@pytest.mark.parametrize("name", H.SMALL_CASES)
def test_oracle_matches_reference_fixture(name):
    res, off, gen, k, fx = H.load_small(name)
    o = ob.Oracle(res, off, gen, k)
    assert o.status == 0
    assert o.sequences == int(fx["sequences"]) and o.genomes == int(fx["genomes"])
    assert o.total_cost == int(fx["total_cost"])
    assert o.hash_fallback == bool(fx["hash_fallback"])
    assert [o.genome_cost(g) for g in range(o.genomes)] == [int(x) for x in fx["genome_cost"]]
    H.assert_scores_equal_fixture(o.scores, fx, o.genomes, name)


@pytest.mark.parametrize("name", sorted(H.DIGESTS))
def test_oracle_matches_reference_digest(name):
    res, off, gen, k, d = H.load_large(name)
    o = ob.Oracle(res, off, gen, k)
    assert o.sequences == d["sequences"] and o.genomes == d["genomes"]
    assert o.total_cost == d["total_cost"]
    assert [o.genome_cost(g) for g in range(o.genomes)] == d["genome_cost"]
    H.assert_scores_match_digest(o.scores, d, name)


def test_known_answers_from_survey():
    """Values quoted in SURVEY.md §8c (README sample, k=2)."""
    res, off, gen, k, _ = H.load_small("readme4_k2")
    o = ob.Oracle(res, off, gen, k)
    assert o.total_cost == 217 and o.genome_cost(0) == 49 and o.genome_cost(1) == 168
    s1 = o.scores(1)
    cells = {(int(r), int(c)): (float(s), float(p), float(t)) for r, c, s, p, t in
             zip(s1["row"], s1["column"], s1["scores"], s1["percs"], s1["tr_percs"])}
    assert cells[(2, 3)] == pytest.approx((0.849056602, 0.918367326, 0.938775539), abs=1e-8)
    assert cells[(3, 1)] == pytest.approx((0.100917429, 0.285714298, 0.169014081), abs=1e-8)


def test_complexity_only_mode_counts_the_same_cost():
    res, off, gen, k, fx = H.load_small("synth_5x60x80_k3")
    o = ob.Oracle(res, off, gen, k, only_complexity=True)
    assert o.total_cost == int(fx["total_cost"])


def test_k_must_be_positive():
    res, off, gen, _, _ = H.load_small("readme4_k2")
    assert ob.Oracle(res, off, gen, 0).status == -1
    assert ob.Oracle(res, off, gen, -3).status == -1


def test_rolling_rank_equals_horner_when_exact():
    res, off, gen, k, _ = H.load_small("synth_5x60x80_k13")
    o = ob.Oracle(res, off, gen, k)
    assert not o.hash_fallback
    rv, b = o.rank_values, o.rank_base
    gene = res[int(off[3]):int(off[4])].tobytes()
    ranks = o.rank_gene(gene)
    for i in (0, 1, len(ranks) - 1):
        h = 0
        for ch in gene[i:i + k]:
            h = (h * b + int(rv[ch])) % (1 << 64)
        assert int(ranks[i]) == h
