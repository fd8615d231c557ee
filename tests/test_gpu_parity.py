"""GPU parity: the HIP path through the C ABI (include/pandelos_amd.h) against
  * the golden vectors the reference's own library.cpp produced (tests/golden/), and
  * the CPU oracle (oracle/pangene_oracle.c) on seeded inputs.
Bar: bit-exact for every integer array and for the float32 bit patterns, emission order included."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _native(res, off, gen, k, **kw):
    from pandelos_amd.pangene_native import PangeneNative
    return PangeneNative.from_arrays(k, res, off, gen, **kw)


@pytest.mark.parametrize("name", H.SMALL_CASES)
def test_hip_matches_reference_fixture(name):
    res, off, gen, k, fx = H.load_small(name)
    nat = _native(res, off, gen, k)
    c = nat.cost
    assert (c.sequences, c.genomes) == (int(fx["sequences"]), int(fx["genomes"]))
    assert c.total_cost == int(fx["total_cost"])
    assert bool(c.hash_fallback) == bool(fx["hash_fallback"])
    assert [nat.genome_cost(g) for g in range(c.genomes)] == [int(x) for x in fx["genome_cost"]]
    H.assert_scores_equal_fixture(lambda g: nat.generate_scores_part(g).as_dict(), fx, c.genomes, name)


@pytest.mark.parametrize("name", sorted(H.DIGESTS))
def test_hip_matches_reference_digest(name):
    res, off, gen, k, d = H.load_large(name)
    nat = _native(res, off, gen, k)
    assert nat.cost.total_cost == d["total_cost"]
    assert [nat.genome_cost(g) for g in range(d["genomes"])] == d["genome_cost"]
    H.assert_scores_match_digest(lambda g: nat.generate_scores_part(g).as_dict(), d, name)


@pytest.mark.parametrize("name", ["synth_5x60x80_k3", "synth_5x60x80_k13", "synth_5x60x80_k16_hash", "low_complexity",
                                  "q1_fold_same_gene_twice"])
def test_dictionary_stage_matches_oracle(name):
    """K-hist/K-rank/K-sort/K-rle: rank table, B^(k-1) and the (rank, gene, count) records."""
    from oracle import binding as ob
    res, off, gen, k, _ = H.load_small(name)
    nat = _native(res, off, gen, k)
    ora = ob.Oracle(res, off, gen, k)
    tab, lm = nat.rank_table()
    assert np.array_equal(tab, ora.rank_values) and lm == ora.last_multiplier
    assert nat.cost.rank_base == ora.rank_base and nat.cost.kmer_occurrences == ora.kmer_occurrences
    ranks, seqs, counts = nat.dictionary()
    d = ora.dictionary()
    # the oracle re-sorts the folded last group by gene (library.cpp:312-315); compare as (rank, gene)-sorted sets
    # both sides hold the folded last group re-sorted by gene (library.cpp:312-315): the device layout must be the oracle's
    assert np.array_equal(ranks, d["rank"]) and np.array_equal(seqs, d["seq"]) and np.array_equal(counts, d["count"])
    cost, kl = nat.sequence_costs()
    assert np.array_equal(cost, ora.total_visited()) and np.array_equal(kl, ora.kseq_lengths())
    # rank-groups of >= 2 records as the reference's scan forms them (library.cpp:297-306: the last record never opens one)
    rk = np.sort(d["rank"])                          # (the fold moved the last record inside its group: back to rank order)
    heads = np.flatnonzero(np.r_[True, rk[1:-1] != rk[:-2]]) if len(d) > 1 else np.array([0])
    sizes = np.diff(np.r_[heads, max(len(d) - 1, 1)]).astype(np.int64)
    if len(d) > 1:
        sizes[-1] += 1
    assert nat.cost.groups == int((sizes >= 2).sum()) and nat.cost.shared_records == int(sizes[sizes >= 2].sum())
    assert nat.cost.total_cost == int((sizes[sizes >= 2] ** 2).sum())


@pytest.mark.parametrize("shape,k", [
    (dict(genomes=7, genes_per_genome=150, mean_len=120, sub_rate=0.12, seed=201), 3),
    (dict(genomes=12, genes_per_genome=400, mean_len=150, sub_rate=0.2, seed=202), 4),
    (dict(genomes=3, genes_per_genome=30, mean_len=2500, sub_rate=0.05, seed=203), 3),   # long genes: HBM-table rows
    (dict(genomes=20, genes_per_genome=100, mean_len=60, sub_rate=0.3, seed=204), 2),    # k=2: everything matches everything
])
def test_hip_matches_oracle_on_random_sets(shape, k):
    from oracle import binding as ob
    from pandelos_amd.synth import make_gene_set
    gs = make_gene_set(**shape)
    nat = _native(gs.residues, gs.offsets, gs.genome_of, k)
    ora = ob.Oracle(gs.residues, gs.offsets, gs.genome_of, k)
    assert nat.cost.total_cost == ora.total_cost
    for g in range(ora.genomes):
        H.assert_scores_equal(nat.generate_scores_part(g).as_dict(), ora.scores(g), f"genome {g}")


@pytest.mark.parametrize("genomes,per_genome,floor", [(100, 3, 64), (300, 3, 256), (420, 1, 300), (2100, 1, 2048)])
def test_wide_rows_keep_the_reference_emission_order(genomes, per_genome, floor):
    """K-order ranks rows of <= 256 cells inside one wave (up to 4 cells per lane), rows of <= 512 cells with a bitonic sort of
    single 64-bit words (key + index), rows of <= 2048 cells with the three-array bitonic sort, and anything wider by
    counting: `genomes` genomes sharing the same gene families give every row ~`genomes` cells."""
    from oracle import binding as ob
    from pandelos_amd.synth import make_gene_set
    gs = make_gene_set(genomes=genomes, genes_per_genome=per_genome, mean_len=90, sub_rate=0.04, seed=207)
    nat = _native(gs.residues, gs.offsets, gs.genome_of, 4)
    ora = ob.Oracle(gs.residues, gs.offsets, gs.genome_of, 4)
    widest = 0
    for g in range(ora.genomes):
        got = nat.generate_scores_part(g).as_dict()
        H.assert_scores_equal(got, ora.scores(g), f"genome {g}")
        if len(got["row"]):
            widest = max(widest, int(np.bincount(np.asarray(got["row"])).max()))
    assert widest > floor
    if genomes == 420:
        assert widest <= 512            # (the single-word network's rows)


def test_per_genome_device_copies_equal_the_host_mirror():
    """pdl_compute_scores slices a pinned host mirror of the whole result (<= 1 GiB) or, for larger results, copies each
    genome's block from the device through per-thread pinned bounce buffers; option host_mirror=0 forces the second path."""
    res, off, gen, k, fx = H.load_small(H.SMALL_CASES[0])
    nat = _native(res, off, gen, k)
    H.assert_scores_equal_fixture(lambda g: nat.generate_scores_part(g).as_dict(), fx, nat.cost.genomes, "mirror")
    nat2 = _native(res, off, gen, k)
    nat2.set_option("host_mirror", 0)
    H.assert_scores_equal_fixture(lambda g: nat2.generate_scores_part(g).as_dict(), fx, nat2.cost.genomes, "device copies")


def test_concurrent_callers_without_the_host_mirror_match_the_digest():
    """The reference's pool calls computeScores from ThreadsNum threads at once (Pangenes.java:54-66).  Here: eight threads,
    the device-copy path (blocks of several MB: more than one bounce-buffer chunk), every block against the reference's digest."""
    from concurrent.futures import ThreadPoolExecutor
    res, off, gen, k, d = H.load_large("synth_16x1000x300_k5")
    nat = _native(res, off, gen, k)
    nat.set_option("host_mirror", 0)
    with ThreadPoolExecutor(8) as ex:
        blocks = list(ex.map(lambda g: nat.generate_scores_part(g).as_dict(), range(d["genomes"])))
    H.assert_scores_match_digest(lambda g: blocks[g], d, "8 threads, device copies")


@pytest.mark.parametrize("name", ["synth_5x60x80_k3", "low_complexity"])
def test_staging_overflow_repeats_the_pass(name):
    """A staging area that is too small for the emitted cells makes the join report the size it needs and the pass is
    repeated once; rows that did not fit hold no cell in the first attempt, so K-order stays inside its buffers."""
    res, off, gen, k, fx = H.load_small(name)
    nat = _native(res, off, gen, k)
    nat.set_option("staging_cap", 16)           # far below the cells of one row
    H.assert_scores_equal_fixture(lambda g: nat.generate_scores_part(g).as_dict(), fx, nat.cost.genomes, f"{name} tiny staging")
    nat.set_option("join_tiny_tier2", 1)        # the HBM-table kernel reserves exact sizes: same rule
    H.assert_scores_equal_fixture(lambda g: nat.generate_scores_part(g).as_dict(), fx, nat.cost.genomes, f"{name} tiny staging, HBM tier")


def test_measurement_options_change_nothing_but_the_timings():
    """stage_timers = 0 drops the HIP events around the single stages (their fields read 0, the totals and the join's stay);
    join_grid_pct launches the join's first tier with fewer workgroups.  Neither touches a result."""
    res, off, gen, k, d = H.load_large("synth_16x1000x300_k5")
    nat = _native(res, off, gen, k)
    nat.set_option("stage_timers", 0)
    nat.set_option("join_grid_pct", 40)
    nat.preprocess(k, res, off, gen)
    H.assert_scores_match_digest(lambda g: nat.generate_scores_part(g).as_dict(), d, "no stage timers, 40 % grid")
    tm = nat.timings()
    assert tm["sort_rank_ms"] == 0 and tm["order_ms"] == 0 and tm["hist_ms"] == 0
    assert tm["preprocess_total_ms"] > 0 and tm["score_total_ms"] > 0 and tm["join_ms"] > 0
    nat.set_option("stage_timers", 1)
    nat.preprocess(k, res, off, gen)
    nat.generate_scores_part(0)
    tm = nat.timings()
    assert tm["sort_rank_ms"] > 0 and tm["order_ms"] > 0 and tm["join_ms"] >= tm["join_overflow_ms"]


def test_errors_mirror_reference_behaviour():
    from pandelos_amd import _lib
    res, off, gen, _, _ = H.load_small("readme4_k2")
    with pytest.raises(_lib.PdlError) as e:
        _native(res, off, gen, 0)
    assert e.value.code == _lib.PDL_ERR_KVALUE and "K value must be greater than 0" in str(e.value)
    with pytest.raises(_lib.PdlError) as e:
        _native(res, off, gen, 500)       # no gene is that long: empty dictionary
    assert e.value.code == _lib.PDL_ERR_EMPTY


def test_complexity_only_mode():
    res, off, gen, k, fx = H.load_small("synth_5x60x80_k3")
    nat = _native(res, off, gen, k, only_complexity=True)
    assert nat.cost.total_cost == int(fx["total_cost"])
    from pandelos_amd import _lib
    with pytest.raises(_lib.PdlError):
        nat.generate_scores_part(0)


def test_genome_shard_scores_only_its_genomes():
    from pandelos_amd import _lib
    from pandelos_amd.pangene_native import PangeneNative
    res, off, gen, k, fx = H.load_small("synth_5x60x80_k3")
    nat = PangeneNative.open()
    nat.set_genome_shard([1, 3])
    nat.preprocess(k, res, off, gen)
    for g in (1, 3):
        got = nat.generate_scores_part(g).as_dict()
        for f in H.FIELDS:
            assert np.array_equal(H.raw(got[f]), fx[f"g{g}_{f}"]), (g, f)
    with pytest.raises(_lib.PdlError):
        nat.generate_scores_part(0)
    counts = nat.scores_counts()
    assert counts[0] == 0 and counts[1] == len(fx["g1_scores"]) and counts[3] == len(fx["g3_scores"])
    nat.set_genome_shard([3])                     # narrowing a shard-built dictionary is fine
    assert np.array_equal(H.raw(nat.generate_scores_part(3).scores), fx["g3_scores"])
    full = _native(res, off, gen, k)              # built for all genomes: cells are shared between their two rows
    with pytest.raises(_lib.PdlError) as e:
        full.set_genome_shard([1, 3])
    assert e.value.code == _lib.PDL_ERR_STATE


def test_device_resident_inputs_via_torch():
    import torch
    from pandelos_amd.pangene_native import PangeneNative
    res, off, gen, k, fx = H.load_small("synth_5x60x80_k3")
    dev = torch.device("cuda:0")
    pad = (-len(res)) % 16
    t_res = torch.from_numpy(np.concatenate([res, np.zeros(pad + 16, np.uint8)])).to(dev)
    t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    t_gen = torch.from_numpy(gen.astype(np.int32)).to(dev)
    torch.cuda.synchronize()
    nat = PangeneNative.from_device(k, t_res.data_ptr(), t_off.data_ptr(), t_gen.data_ptr(), len(gen), len(res),
                                    stream=torch.cuda.current_stream().cuda_stream, keepalive=(t_res, t_off, t_gen))
    assert nat.cost.total_cost == int(fx["total_cost"])
    H.assert_scores_equal_fixture(lambda g: nat.generate_scores_part(g).as_dict(), fx, nat.cost.genomes, "torch-resident")


def test_canonical_order_flag_gives_same_cells_sorted_by_column():
    from pandelos_amd import _lib
    res, off, gen, k, fx = H.load_small("synth_5x60x80_k3")
    nat = _native(res, off, gen, k, flags=_lib.PDL_FLAG_CANONICAL_ORDER)
    for g in range(nat.cost.genomes):
        got = nat.generate_scores_part(g).as_dict()
        key = lambda r, c: (r.astype(np.int64) << 32) | c.astype(np.int64)
        kg = key(got["row"], got["column"])
        assert np.all(np.diff(kg) > 0)
        o = np.argsort(key(fx[f"g{g}_row"], fx[f"g{g}_column"]))
        assert np.array_equal(kg, key(fx[f"g{g}_row"], fx[f"g{g}_column"])[o])
        assert np.array_equal(H.raw(got["scores"]), fx[f"g{g}_scores"][o])


@pytest.mark.parametrize("name", ["synth_5x60x80_k3", "low_complexity", "q1_fold_same_gene_twice", "readme4_k1"])
def test_hbm_table_path_matches_fixture(name):
    """Rows whose candidate set does not fit the LDS table are redone by k_join_hbm.  A deliberately tiny
    LDS table (option join_tiny_tier2: 512 slots, 64 ranges staged per batch) exercises the multi-batch staging
    and, where a row has more than 384 candidates, that path."""
    res, off, gen, k, fx = H.load_small(name)
    nat = _native(res, off, gen, k)
    nat.set_option("join_tiny_tier2", 1)
    H.assert_scores_equal_fixture(lambda g: nat.generate_scores_part(g).as_dict(), fx, nat.cost.genomes, name)


@pytest.mark.parametrize("tier", [0, 9, 10, 11, 20, 21])
@pytest.mark.parametrize("name", ["synth_16x1000x300_k5", "protein_like_12x400x150_k4_div30"])
def test_every_first_tier_reproduces_the_reference_digests(name, tier):
    """The join's first tier is chosen by genome count (1024 slots + filter up to 320 genomes, 2048 slots + filter and eight
    chunks of lookups in flight beyond — the 512-genome set's kernel); the others exist for experiments.  Forced one by one
    on sets the reference's digests pin, each must give the same cells in the same order."""
    res, off, gen, k, d = H.load_large(name)
    nat = _native(res, off, gen, k)
    nat.set_option("join_tier1", tier)
    H.assert_scores_match_digest(lambda g: nat.generate_scores_part(g).as_dict(), d, f"{name} tier {tier}")
    tm = nat.timings()
    assert tm["aside_reloads"] == 0 and tm["aside_repeats"] == 0        # the canary of the filter tiers' put-aside lists (the other tiers keep none)


@pytest.mark.parametrize("name", H.SMALL_CASES + sorted(H.DIGESTS))
def test_partition_tier_forced_on_reproduces_the_reference(name):
    """The partition tier (k_join_part: several short rows per workgroup cycle, lookups partitioned by column instead of
    hashed one by one) is chosen by the average row length; forced on ("join_tier0" 1) it must reproduce every fixture and
    every digest of the reference — whatever it cannot take (genes of <= 2k k-mers, long rows, too many heavy lookups) it
    hands to the filter tier."""
    small = name in H.SMALL_CASES
    res, off, gen, k, fx = H.load_small(name) if small else H.load_large(name)
    nat = _native(res, off, gen, k)
    nat.set_option("join_tier0", 1)
    if small:
        H.assert_scores_equal_fixture(lambda g: nat.generate_scores_part(g).as_dict(), fx, nat.cost.genomes, name)
    else:
        H.assert_scores_match_digest(lambda g: nat.generate_scores_part(g).as_dict(), fx, name)
    tm = nat.timings()
    assert tm["tier1_rows"] <= tm["scored_rows"] and tm["aside_reloads"] == 0
    if name in ("synth_16x1000x300_k5", "synth_40x60x40_k3"):
        assert tm["tier1_rows"] < tm["scored_rows"] // 2          # (the tier did take rows)


@pytest.mark.parametrize("name", H.SMALL_CASES + sorted(H.DIGESTS))
def test_scans_in_one_launch_reproduce_the_reference(name):
    """"onepass_scan" 1: every prefix scan of the build and of the scoring pass is ONE launch (two-level decoupled look-back,
    k_scan_onepass in pdl_scan.h) instead of tile sums + scan of the sums + apply.  Off by default (measured slower on MI355X);
    it must reproduce every fixture and digest all the same, twice in a row on one context (the epoch of the look-back words
    moves on), and no look-back may have run into its poll bound (the host would have failed the call)."""
    from pandelos_amd.pangene_native import PangeneNative
    small = name in H.SMALL_CASES
    res, off, gen, k, fx = H.load_small(name) if small else H.load_large(name)
    nat = PangeneNative.open()
    nat.set_option("onepass_scan", 1)
    for _ in range(2):
        nat.preprocess(k, res, off, gen)
        if small:
            H.assert_scores_equal_fixture(lambda g: nat.generate_scores_part(g).as_dict(), fx, nat.cost.genomes, name)
        else:
            H.assert_scores_match_digest(lambda g: nat.generate_scores_part(g).as_dict(), fx, name)
    nat.close()


@pytest.mark.parametrize("name,expect", [("synth_16x1000x300_k5", True), ("protein_like_24x1500x300_k5", False)])
def test_partition_tier_is_taken_by_itself_only_where_it_pays(name, expect):
    """By itself ("join_tier0" -1) the partition tier runs in front of the filter tier on short rows of genes whose k-mers
    rarely repeat inside the gene (records with a count >= 2 at most 1 in 5000); protein-like text with low-complexity stretches
    keeps the filter tier alone (there it costs 1.05 ms against 0.64).  Same digests either way."""
    res, off, gen, k, d = H.load_large(name)
    nat = _native(res, off, gen, k)
    H.assert_scores_match_digest(lambda g: nat.generate_scores_part(g).as_dict(), d, name)
    tm = nat.timings()
    assert (tm["tier1_rows"] < tm["scored_rows"]) == expect


@pytest.mark.parametrize("tier", [10, 11])
def test_a_pass_that_saw_a_reload_is_repeated_with_fully_tagged_entries(tier):
    """The 8-byte put-aside entries carry a 10-bit tag: a stale entry passes it once in 1024.  So a pass in which any entry
    needed a second look is never returned: it is repeated with 16-byte entries that name row and launch in full
    (pdl_timings.aside_repeats).  No run has ever seen a reload; the test switch makes the next pass count as one that did."""
    res, off, gen, k, d = H.load_large("synth_16x1000x300_k5")
    nat = _native(res, off, gen, k)
    nat.set_option("join_tier1", tier)
    nat.set_option("aside_test_reload", 1)
    H.assert_scores_match_digest(lambda g: nat.generate_scores_part(g).as_dict(), d, f"tier {tier}, repeated with wide entries")
    tm = nat.timings()
    assert tm["aside_repeats"] == 1 and tm["aside_reloads"] == 0
    nat.set_option("join_tier1", tier)          # (any option marks the scores stale: the next pass is an ordinary one)
    H.assert_scores_match_digest(lambda g: nat.generate_scores_part(g).as_dict(), d, f"tier {tier}")
    assert nat.timings()["aside_repeats"] == 0


def test_more_than_320_genomes_take_the_2048_slot_tier_and_match_the_oracle():
    """The tier a 512-genome set gets by itself (no option): 2048 slots + filter, three workgroups per CU, eight chunks of
    lookups in flight.  330 small genomes, scored three times over (the wrong cells this tier once produced came and went
    with the timing), every genome against the CPU oracle."""
    from oracle import binding as ob
    from pandelos_amd.calculate_k import calculate_k
    from pandelos_amd.synth import make_gene_set
    gs = make_gene_set(genomes=330, genes_per_genome=36, mean_len=90, sub_rate=0.1, seed=3301)
    k = calculate_k(gs.residues)
    ora = ob.Oracle(gs.residues, gs.offsets, gs.genome_of, k)
    want = [ora.scores(g) for g in range(ora.genomes)]
    nat = _native(gs.residues, gs.offsets, gs.genome_of, k)
    for it in range(3):
        if it:
            nat.preprocess(k, gs.residues, gs.offsets, gs.genome_of)
        assert nat.cost.total_cost == ora.total_cost
        for g in range(ora.genomes):
            H.assert_scores_equal(nat.generate_scores_part(g).as_dict(), want[g], f"pass {it} genome {g}")


def test_hbm_table_path_matches_oracle_on_dense_set():
    """k=2 on 1500 genes: every gene shares k-mers with every other one (candidate sets ~ N)."""
    from oracle import binding as ob
    from pandelos_amd.synth import make_gene_set
    gs = make_gene_set(genomes=10, genes_per_genome=150, mean_len=60, sub_rate=0.3, seed=205)
    nat = _native(gs.residues, gs.offsets, gs.genome_of, 2)
    nat.set_option("join_tiny_tier2", 1)
    ora = ob.Oracle(gs.residues, gs.offsets, gs.genome_of, 2)
    for g in range(ora.genomes):
        H.assert_scores_equal(nat.generate_scores_part(g).as_dict(), ora.scores(g), f"genome {g}")
    assert nat.timings()["overflow_rows"] > 300      # rows only see the genes above them: the upper rows overflow


def test_shard_set_before_preprocess_builds_only_its_rows():
    """Multi-GPU path: the shard is known before the dictionary build, so range lists / costs exist for its genes only;
    two disjoint shards together reproduce the unsharded result and total cost."""
    from pandelos_amd.pangene_native import PangeneNative
    from pandelos_amd import _lib
    res, off, gen, k, fx = H.load_small("synth_5x60x80_k3")
    total = 0
    for shard in ([0, 2, 4], [1, 3]):
        nat = PangeneNative.open()
        nat.set_genome_shard(shard)
        nat.preprocess(k, res, off, gen)
        total += nat.cost.total_cost
        for g in shard:
            assert nat.genome_cost(g) == int(fx["genome_cost"][g])
            got = nat.generate_scores_part(g).as_dict()
            for f in H.FIELDS:
                assert np.array_equal(H.raw(got[f]), fx[f"g{g}_{f}"]), (g, f)
        # another shard on the same dictionary — wider, or disjoint: its range lists are built before the next scoring pass
        for other in ([0, 1, 2, 3, 4], [g for g in range(5) if g not in shard]):
            nat.set_genome_shard(other)
            for g in other:
                got = nat.generate_scores_part(g).as_dict()
                for f in H.FIELDS:
                    assert np.array_equal(H.raw(got[f]), fx[f"g{g}_{f}"]), (g, f)
                assert nat.genome_cost(g) == int(fx["genome_cost"][g])
            assert nat.timings()["reshard_ms"] > 0
        with pytest.raises(_lib.PdlError):
            nat.set_genome_shard([])                    # back to "all genomes, symmetric pass": that is another dictionary build
    assert total == int(fx["total_cost"])


def test_last_record_fold_moves_through_a_large_group():
    """The globally last record (a singleton of the largest rank) is folded into the preceding rank-group and re-sorted
    by gene (library.cpp:300-315).  Here that group has 3000 members and the folded record belongs to gene 0, so the
    device's insertion shift walks several 1024-element chunks."""
    from oracle import binding as ob
    genes = [b"YYAA"] + [b"AAYA"] * 3000 + [b"AAAA", b"AYAA"]
    # ranks (k=2, A<Y): AA < AY < YA < YY; YY only in gene 0 -> last record, folded into the YA group (genes 1..3000, 3002)
    residues = np.frombuffer(b"".join(genes), np.uint8)
    offsets = np.arange(len(genes) + 1, dtype=np.uint64) * 4
    genome_of = (np.arange(len(genes)) % 7).astype(np.uint32)
    nat = _native(residues, offsets, genome_of, 2)
    ora = ob.Oracle(residues, offsets, genome_of, 2)
    assert nat.cost.total_cost == ora.total_cost
    ranks, seqs, counts = nat.dictionary()
    d = ora.dictionary()
    assert np.array_equal(ranks, d["rank"]) and np.array_equal(seqs, d["seq"]) and np.array_equal(counts, d["count"])
    for g in range(7):
        H.assert_scores_equal(nat.generate_scores_part(g).as_dict(), ora.scores(g), f"genome {g}")


def _flat(genes, genome_ids):
    residues = np.frombuffer(b"".join(genes), np.uint8)
    offsets = np.zeros(len(genes) + 1, np.uint64)
    np.cumsum([len(g) for g in genes], out=offsets[1:])
    return residues, offsets, np.asarray(genome_ids, np.uint32)


@pytest.mark.parametrize("case", ["single_gene", "single_genome", "identical_genes", "bytes_above_127", "k1",
                                  "one_very_long_gene", "empty_genes_between"])
def test_edge_cases_match_the_oracle(case):
    from oracle import binding as ob
    rng = np.random.default_rng(9)
    aa = b"ACDEFGHIKLMNPQRSTVWY"
    rnd = lambda n: bytes(aa[i] for i in rng.integers(0, 20, n))
    if case == "single_gene":
        genes, gid, k = [rnd(50)], [0], 3
    elif case == "single_genome":                      # only intra-genome cells
        base = rnd(120)
        genes, gid, k = [base, base[:60] + rnd(60), rnd(100), base[30:] + rnd(10)], [0, 0, 0, 0], 3
    elif case == "identical_genes":                    # scores of exactly 1.0 in several genomes
        base = rnd(90)
        genes, gid, k = [base, base, base, rnd(80), base, rnd(70)], [0, 1, 2, 0, 0, 1], 4
    elif case == "bytes_above_127":                    # Latin-1 residues: rank table covers all 256 byte values
        genes = [bytes(rng.integers(128, 256, 80).astype(np.uint8)) for _ in range(6)]
        genes[3] = genes[0][:50] + genes[3][50:]
        gid, k = [0, 0, 1, 1, 2, 2], 2
    elif case == "k1":
        genes, gid, k = [rnd(30) for _ in range(8)], [0, 1, 2, 3, 0, 1, 2, 3], 1
    elif case == "one_very_long_gene":                 # 60 000 residues: tens of thousands of candidates -> big LDS / HBM tiers
        long_gene = rnd(60000)
        genes = [long_gene] + [long_gene[i * 100:i * 100 + 150] for i in range(300)] + [rnd(200) for _ in range(50)]
        gid, k = [i % 5 for i in range(len(genes))], 3
    else:                                               # genes shorter than k (even empty) in the middle of the stream
        genes, gid, k = [rnd(40), b"", b"AC", rnd(40), b"A", rnd(45), b""], [0, 0, 1, 1, 2, 2, 2], 3
        genes[3] = genes[0][:30] + genes[3][30:]
    res, off, gen = _flat(genes, gid)
    nat = _native(res, off, gen, k)
    ora = ob.Oracle(res, off, gen, k)
    assert nat.cost.total_cost == ora.total_cost
    assert nat.cost.genomes == ora.genomes
    for g in range(ora.genomes):
        H.assert_scores_equal(nat.generate_scores_part(g).as_dict(), ora.scores(g), f"{case} genome {g}")


def test_gene_with_over_a_million_kmers_uses_wide_counters():
    """>= 2^20 k-mers in one gene: the packed 21-bit accumulators could overflow, every row goes to the HBM kernel with
    32-bit counters (the reference's int arrays, library.cpp:421-423)."""
    from oracle import binding as ob
    rng = np.random.default_rng(12)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    giant = aa[rng.integers(0, 20, (1 << 20) + 5000)].tobytes()
    genes = [giant] + [giant[i * 9973:i * 9973 + 400] for i in range(40)] + [aa[rng.integers(0, 20, 300)].tobytes() for _ in range(20)]
    res, off, gen = _flat(genes, [i % 4 for i in range(len(genes))])
    nat = _native(res, off, gen, 4)
    ora = ob.Oracle(res, off, gen, 4)
    assert nat.cost.total_cost == ora.total_cost
    for g in range(4):
        H.assert_scores_equal(nat.generate_scores_part(g).as_dict(), ora.scores(g), f"genome {g}")
    assert nat.timings()["overflow_rows"] == len(genes)


@pytest.mark.parametrize("name,per_batch", [("synth_16x1000x300_k5", 5), ("protein_like_24x1500x300_k5", 7), ("synth_40x60x40_k3", 40), ("synth_40x60x40_k3", 1)])
def test_genome_batches_on_one_dictionary_reproduce_the_reference_digests(name, per_batch):
    """A set scored a batch of genomes at a time (pdl_set_genome_shard on an existing dictionary: the postings are built once,
    each batch gets its own range lists; option low_memory) gives every genome the Scores block of the whole-set pass — the
    reference's digests — although no batch sees the row/column symmetry the whole-set pass uses."""
    from pandelos_amd.pangene_native import PangeneNative
    res, off, gen, k, d = H.load_large(name)
    nat = PangeneNative.open()
    got = {}
    reshards = 0
    for g, s in nat.scores_in_batches(k, res, off, gen, per_batch):
        got[g] = s.as_dict()
        reshards += nat.timings()["reshard_ms"] > 0 and g % per_batch == 0
    assert sorted(got) == list(range(d["genomes"]))
    H.assert_scores_match_digest(lambda g: got[g], d, f"{name} in batches of {per_batch}")
    assert [nat.genome_cost(g) for g in range(d["genomes"] - 1, d["genomes"])] == d["genome_cost"][-1:]      # (the last batch's costs are on the context)
    assert reshards == (d["genomes"] - 1) // per_batch
    with pytest.raises(_lib_error()):
        nat.dictionary()                                # low_memory: the sorted k-mer stream went back after the build
    nat.close()


def _lib_error():
    from pandelos_amd import _lib
    return _lib.PdlError


@pytest.mark.parametrize("letters,k", [(22, 15), (24, 14), (11, 19), (20, 15)])
def test_ranks_that_wrap_past_64_bits_unnoticed_match_the_oracle(letters, k):
    """22 letters at k = 15 (24 at 14, 11 at 19): B^k passes 2^64, the reference's overflow test does not notice, and the ranks — the
    polynomial mod 2^64 — fill all 64 bits while rank_init counts 63 (62).  The rank sort must still match whole digits (a pass
    that matched only the bits rank_init counts put records where the histogram did not: a memory fault on seed 3078 of the wide
    fuzz run); one GPU and three ranks against the oracle."""
    from oracle import binding as ob
    from pandelos_amd.distributed import LocalRanks
    from tests.test_gpu_dist import _device_inputs
    gs = H.wrapped_rank_set(letters, seed=letters * 100 + k)
    ora = ob.Oracle(gs.residues, gs.offsets, gs.genome_of, k)
    assert not ora.hash_fallback and ora.total_cost > 0
    nat = _native(gs.residues, gs.offsets, gs.genome_of, k)
    assert nat.cost.total_cost == ora.total_cost
    want = [ora.scores(g) for g in range(ora.genomes)]
    for g in range(ora.genomes):
        H.assert_scores_equal(nat.generate_scores_part(g).as_dict(), want[g], f"{letters} letters k={k} genome {g}")
    nat.close()
    lr = LocalRanks(3)
    lr.preprocess(k, *_device_inputs(gs.residues, gs.offsets, gs.genome_of), gs.genes, len(gs.residues))
    assert lr.total_cost == ora.total_cost
    lr.score_all()
    for g in range(ora.genomes):
        H.assert_scores_equal(lr.generate_scores_part(g).as_dict(), want[g], f"{letters} letters k={k} three ranks genome {g}")
    lr.close()


@pytest.mark.parametrize("shape", [
    dict(genomes=40, genes_per_genome=200, mean_len=400, sub_rate=0.05, seed=7701),      # rows of 4-9 k lookups: both forms of tier 0, some beyond
    dict(genomes=24, genes_per_genome=150, mean_len=1200, sub_rate=0.08, seed=7702),     # genes of > 960 shared k-mers: more ranges than a cycle stages
    dict(genomes=70, genes_per_genome=120, mean_len=250, sub_rate=0.03, seed=7703),      # close homologs in 70 genomes: rows of ~8 k lookups
])
def test_long_rows_go_through_both_forms_of_the_partition_tier_and_beyond(shape):
    """Tier 0 forced on for sets whose rows do NOT fit its first form: rows that alone exceed 4096 lookups take the 512-thread
    form, rows beyond 8192 lookups or 960 ranges are handed on to the filter tier (descriptors written by the tier that hands
    them on), and every genome must still match the oracle."""
    from oracle import binding as ob
    from pandelos_amd.calculate_k import calculate_k
    from pandelos_amd.pangene_native import PangeneNative
    from pandelos_amd.synth import make_gene_set
    gs = make_gene_set(**shape)
    k = calculate_k(gs.residues)
    ora = ob.Oracle(gs.residues, gs.offsets, gs.genome_of, k)
    nat = PangeneNative.open()
    nat.set_option("join_tier0", 1)
    nat.preprocess(k, gs.residues, gs.offsets, gs.genome_of)
    assert nat.cost.total_cost == ora.total_cost
    for g in range(ora.genomes):
        H.assert_scores_equal(nat.generate_scores_part(g).as_dict(), ora.scores(g), f"{shape} genome {g}")
    tm = nat.timings()
    assert tm["aside_reloads"] == 0
    print(f"{shape['genomes']}x{shape['genes_per_genome']}x{shape['mean_len']}: rows {tm['scored_rows']}, to the filter tier {tm['tier1_rows']}, tier 2 {tm['tier2_rows']}, tier 3 {tm['overflow_rows']}")
    assert 0 < tm["tier1_rows"] < tm["scored_rows"]          # some rows stayed in tier 0, some went beyond it
    nat.close()
