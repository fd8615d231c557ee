"""Pins the oracle against the reference's own library.cpp on fresh random inputs.
Needs oracle/_ref (built only where /root/reference exists) — skipped elsewhere."""
import numpy as np
import pytest

from oracle import binding as ob
from pandelos_amd.synth import make_gene_set
from tests import helpers as H

pytestmark = pytest.mark.skipif(not ob.have_reference(), reason="oracle/_ref not built (no /root/reference)")


@pytest.mark.parametrize("shape,k", [
    (dict(genomes=3, genes_per_genome=40, mean_len=60, sub_rate=0.15, seed=101), 2),
    (dict(genomes=6, genes_per_genome=80, mean_len=50, sub_rate=0.05, seed=102), 3),
    (dict(genomes=9, genes_per_genome=300, mean_len=120, sub_rate=0.2, seed=103), 4),
    (dict(genomes=4, genes_per_genome=50, mean_len=90, sub_rate=0.1, seed=104), 15),
    (dict(genomes=4, genes_per_genome=50, mean_len=90, sub_rate=0.1, seed=104), 20),
])
def test_oracle_equals_reference_on_random_sets(tmp_path, shape, k):
    gs = make_gene_set(**shape)
    faa = tmp_path / "in.faa"
    gs.write_faa(faa)
    info = ob.run_harness(ob.REF_SO, faa, k, dump=tmp_path / "ref.bin")
    ref = ob.read_dump(tmp_path / "ref.bin")
    o = ob.Oracle(gs.residues, gs.offsets, gs.genome_of, k)
    assert o.total_cost == info["total_cost"]
    assert o.hash_fallback == info["hash_fallback"]
    for g in range(ref["genomes"]):
        assert o.genome_cost(g) == info["genome_cost"][g]
        H.assert_scores_equal(o.scores(g), ref["per_genome"][g], f"genome {g}")


@pytest.mark.parametrize("letters,k", [(22, 15), (24, 14), (11, 19)])
def test_oracle_equals_reference_where_ranks_wrap_unnoticed(tmp_path, letters, k):
    """B^k > 2^64 but rank_init's overflow test lets it through (it compares wrapped products): no hashing, the ranks are the
    polynomial mod 2^64 and rank_byte_order comes from the wrapped B^k.  The oracle restates exactly that."""
    gs = H.wrapped_rank_set(letters, seed=letters * 100 + k)
    faa = tmp_path / "in.faa"
    gs.write_faa(faa)
    info = ob.run_harness(ob.REF_SO, faa, k, dump=tmp_path / "ref.bin")
    ref = ob.read_dump(tmp_path / "ref.bin")
    o = ob.Oracle(gs.residues, gs.offsets, gs.genome_of, k)
    assert letters ** k >= 1 << 64 and not info["hash_fallback"] and not o.hash_fallback
    assert o.total_cost == info["total_cost"] and info["total_cost"] > 0
    for g in range(ref["genomes"]):
        H.assert_scores_equal(o.scores(g), ref["per_genome"][g], f"genome {g}")


def test_reference_is_thread_safe_in_harness(tmp_path):
    gs = make_gene_set(genomes=6, genes_per_genome=80, mean_len=50, sub_rate=0.05, seed=102)
    faa = tmp_path / "in.faa"
    gs.write_faa(faa)
    ob.run_harness(ob.REF_SO, faa, 3, threads=1, dump=tmp_path / "a.bin")
    ob.run_harness(ob.REF_SO, faa, 3, threads=4, dump=tmp_path / "b.bin")
    assert (tmp_path / "a.bin").read_bytes() == (tmp_path / "b.bin").read_bytes()
