"""GPU parity, randomized: small gene sets of varied shape (alphabet size, k, gene lengths around k, duplicated and
empty genes, interleaved genome ids, repeats inside a gene) against the CPU oracle, bit for bit — default order,
canonical order and a genome shard.  Seeds are fixed: a failure names its seed."""
import os

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu
_OFF = int(os.environ.get("PDL_FUZZ_OFFSET", "0"))       # another stretch of the seed space (wide runs)


def _random_set(seed):
    rng = np.random.default_rng(seed)
    alpha = int(rng.choice([2, 3, 4, 8, 20, 23]))
    letters = rng.choice(np.arange(33, 127), alpha, replace=False).astype(np.uint8)
    if rng.random() < 0.2:
        letters[rng.integers(0, alpha)] = rng.integers(128, 256)         # a Latin-1 byte among them
    genomes = int(rng.integers(1, 9))
    n = int(rng.integers(1, 120))
    k = int(rng.choice([1, 2, 3, 4, 5, 7, 11, 15, 17]))                   # 15+ with 20 letters: hash fallback
    fam = [letters[rng.integers(0, alpha, int(rng.integers(max(1, k - 2), 6 * k + 40)))] for _ in range(int(rng.integers(1, 12)))]
    genes = []
    for _ in range(n):
        r = rng.random()
        if r < 0.05:
            g = np.zeros(0, np.uint8)                                     # empty gene
        elif r < 0.15:
            g = letters[rng.integers(0, alpha, int(rng.integers(1, k + 1)))]      # shorter than or equal to k
        else:
            g = fam[rng.integers(0, len(fam))].copy()
            m = rng.random(len(g)) < rng.choice([0.0, 0.02, 0.1, 0.3])
            g[m] = letters[rng.integers(0, alpha, int(m.sum()))]
            if rng.random() < 0.2:
                g = np.concatenate([g, g[: len(g) // 2]])                 # repeats inside the gene
            if rng.random() < 0.2:
                g = g[int(rng.integers(0, max(1, len(g) // 2))):]
        genes.append(g)
    gid_raw = rng.integers(0, genomes, n)
    # dense genome ids in first-seen order, as PangeneIData assigns them
    seen, gid = {}, []
    for x in gid_raw:
        gid.append(seen.setdefault(int(x), len(seen)))
    residues = np.concatenate(genes) if genes else np.zeros(0, np.uint8)
    offsets = np.zeros(n + 1, np.uint64)
    np.cumsum([len(g) for g in genes], out=offsets[1:])
    return residues.astype(np.uint8), offsets, np.asarray(gid, np.uint32), k


@pytest.mark.parametrize("seed", list(range(3000 + _OFF, 3000 + _OFF + int(os.environ.get("PDL_FUZZ_SEEDS", "240")))))   # widen with PDL_FUZZ_SEEDS=N
def test_random_small_sets_match_the_oracle(seed):
    from oracle import binding as ob
    from pandelos_amd import _lib
    from pandelos_amd.pangene_native import PangeneNative
    res, off, gen, k = _random_set(seed)
    if int((np.diff(off.astype(np.int64)) >= k).sum()) == 0:
        pytest.skip("no gene holds a k-mer (undefined in the reference)")
    ora = ob.Oracle(res, off, gen, k)
    nat = PangeneNative.from_arrays(k, res, off, gen)
    assert nat.cost.total_cost == ora.total_cost, f"seed {seed}"
    want = [ora.scores(g) for g in range(ora.genomes)]
    for g in range(ora.genomes):
        H.assert_scores_equal(nat.generate_scores_part(g).as_dict(), want[g], f"seed {seed} genome {g}")
    # canonical order: same cells, ascending column inside a row
    nat_c = PangeneNative.from_arrays(k, res, off, gen, flags=_lib.PDL_FLAG_CANONICAL_ORDER)
    for g in range(ora.genomes):
        got, w = nat_c.generate_scores_part(g).as_dict(), want[g]
        order = np.lexsort((w["column"], w["row"]))
        for f in ("scores", "percs", "tr_percs", "row", "column"):
            assert np.array_equal(H.raw(got[f]), H.raw(np.asarray(w[f])[order])), f"seed {seed} canonical genome {g} {f}"
    # a shard of every other genome, set before the dictionary build
    if ora.genomes >= 2:
        shard = list(range(0, ora.genomes, 2))
        nat_s = PangeneNative.open()
        nat_s.set_genome_shard(shard)
        nat_s.preprocess(k, res, off, gen)
        for g in shard:
            H.assert_scores_equal(nat_s.generate_scores_part(g).as_dict(), want[g], f"seed {seed} shard genome {g}")


@pytest.mark.parametrize("seed", list(range(7000 + _OFF, 7000 + _OFF + int(os.environ.get("PDL_STRESS_SETS", "6")))))   # widen with PDL_STRESS_SETS=N
def test_mid_size_sets_under_every_join_tier_match_the_oracle(seed):
    """The bugs that came and went with the timing (stale put-aside entries, barriers without an LDS wait) never showed on
    the tiny sets above: they need thousands of rows and several workgroups per CU.  Mid-size families-and-genomes sets,
    a first tier and a grid size drawn per seed, every set scored twice straight after its dictionary build."""
    from oracle import binding as ob
    from pandelos_amd.calculate_k import calculate_k
    from pandelos_amd.pangene_native import PangeneNative
    from pandelos_amd.synth import make_gene_set
    rng = np.random.default_rng(seed)
    gs = make_gene_set(genomes=int(rng.integers(6, 48)), genes_per_genome=int(rng.integers(150, 500)), mean_len=int(rng.integers(70, 220)),
                       sub_rate=float(rng.choice([0.02, 0.08, 0.2])), seed=seed, protein_like=bool(rng.random() < 0.3))
    k = max(3, calculate_k(gs.residues) - int(rng.integers(0, 2)))
    tier = int(rng.choice([10, 11, 11, 21, 9, 0]))
    pct = int(rng.choice([0, 0, 70, 40]))
    ora = ob.Oracle(gs.residues, gs.offsets, gs.genome_of, k)
    want = [ora.scores(g) for g in range(ora.genomes)]
    nat = PangeneNative.open()
    nat.set_option("join_tier1", tier)
    nat.set_option("join_grid_pct", pct)
    nat.set_option("join_tier0", int(rng.integers(0, 2)) if tier else 0)      # the partition tier in front of it, or not
    for it in range(2):
        nat.preprocess(k, gs.residues, gs.offsets, gs.genome_of)
        assert nat.cost.total_cost == ora.total_cost
        for g in range(ora.genomes):
            H.assert_scores_equal(nat.generate_scores_part(g).as_dict(), want[g], f"seed {seed} tier {tier} grid {pct}% pass {it} genome {g}")
        tm = nat.timings()
        assert tm["aside_reloads"] == 0 and tm["aside_repeats"] == 0, f"seed {seed} tier {tier}: the put-aside canary went off"
    nat.close()


@pytest.mark.parametrize("seed", list(range(9000 + _OFF, 9000 + _OFF + max(2, int(os.environ.get("PDL_STRESS_SETS", "6")) // 2))))
def test_mid_size_sets_in_genome_batches_match_the_oracle(seed):
    """Genome batches on one dictionary (pdl_set_genome_shard after the build: head bits put back, the batch's range lists made
    again; the build's buffers released under "low_memory" or kept), a batch size and a first tier drawn per seed."""
    from oracle import binding as ob
    from pandelos_amd.calculate_k import calculate_k
    from pandelos_amd.pangene_native import PangeneNative
    from pandelos_amd.synth import make_gene_set
    rng = np.random.default_rng(seed)
    gs = make_gene_set(genomes=int(rng.integers(3, 30)), genes_per_genome=int(rng.integers(60, 400)), mean_len=int(rng.integers(60, 200)),
                       sub_rate=float(rng.choice([0.02, 0.08, 0.2])), seed=seed, protein_like=bool(rng.random() < 0.3))
    k = max(3, calculate_k(gs.residues) - int(rng.integers(0, 2)))
    per_batch = int(rng.integers(1, gs.genomes + 1))
    ora = ob.Oracle(gs.residues, gs.offsets, gs.genome_of, k)
    nat = PangeneNative.open()
    nat.set_option("join_tier1", int(rng.choice([-1, 10, 11, 21, 0])))
    seen = 0
    for g, s in nat.scores_in_batches(k, gs.residues, gs.offsets, gs.genome_of, per_batch, low_memory=bool(rng.integers(0, 2))):
        H.assert_scores_equal(s.as_dict(), ora.scores(g), f"seed {seed} batches of {per_batch} genome {g}")
        seen += 1
    assert seen == ora.genomes
    nat.close()
