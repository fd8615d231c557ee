"""Multi-GPU path on the device (pdl_dist_*): W ranks build the dictionary co-operatively (rank intervals), deal the
genomes, score only above the diagonal and exchange the mirrored cells.  Every genome's Scores block must equal the
reference's fixture bit for bit, whatever W is and whichever rank scored it.

  * LocalRanks: W contexts in one process on one device, device copies in place of the collectives (same library calls
    in the same order as the torch.distributed driver);
  * two real processes over torch.distributed (gloo, both on cuda:0): the driver itself, collectives included."""
import os
import socket

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _device_inputs(res, off, gen):
    import torch
    dev = torch.device("cuda", 0)
    pad = (-len(res)) % 16 + 16
    t_res = torch.from_numpy(np.concatenate([res, np.zeros(pad, np.uint8)])).to(dev)
    t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    t_gen = torch.from_numpy(gen.astype(np.int32)).to(dev)
    return t_res, t_off, t_gen


def _local(world, res, off, gen, k, flags=0, exchange_weights=True, sender_ranges=True):
    from pandelos_amd.distributed import LocalRanks
    t = _device_inputs(res, off, gen)
    lr = LocalRanks(world, flags=flags, exchange_weights=exchange_weights, sender_ranges=sender_ranges)
    cost = lr.preprocess(k, *t, len(gen), len(res))
    return lr, cost


@pytest.mark.parametrize("ranges", ["sender", "owner"])      # who builds the range lists: the rank that holds the run (tuples travel), or the rank that owns the gene
@pytest.mark.parametrize("world", [1, 2, 3, 5])
@pytest.mark.parametrize("name", ["synth_5x60x80_k3", "synth_5x60x80_k13", "synth_5x60x80_k16_hash", "low_complexity", "q1_fold",
                                  "q1_fold_onto_singleton", "q1_fold_same_gene_twice", "readme4_k1", "readme4_k2",
                                  "short_and_duplicate_genes", "interleaved_genomes"])
def test_ranks_together_reproduce_the_fixture(name, world, ranges):
    res, off, gen, k, fx = H.load_small(name)
    lr, cost = _local(world, res, off, gen, k, sender_ranges=ranges == "sender")
    # (the senders' flow is the library's call: a last run of exactly one record — tiny sets over many ranks — leaves the lists to the owners)
    assert not lr.used_sender_ranges or ranges == "sender"
    if ranges == "sender" and name in ("synth_5x60x80_k3", "synth_5x60x80_k13", "low_complexity"):
        assert lr.used_sender_ranges
    assert lr.total_cost == int(fx["total_cost"]) and (cost.sequences, cost.genomes) == (int(fx["sequences"]), int(fx["genomes"]))
    assert sum(lr.run_records) == cost.dictionary_records
    assert [lr.genome_cost(g) for g in range(cost.genomes)] == [int(x) for x in fx["genome_cost"]]     # each from its owner (library.cpp:535-538)
    for n in lr.ranks:                                   # counters over the whole dictionary: the same everywhere
        assert (n.cost.groups, n.cost.shared_records, n.cost.dictionary_records) == (cost.groups, cost.shared_records, cost.dictionary_records)
    lr.score_all()
    assert all(int(x) < world for x in lr.owner)
    H.assert_scores_equal_fixture(lambda g: lr.generate_scores_part(g).as_dict(), fx, cost.genomes, f"{name} W={world}")
    # what travelled: cells whose row and column live on different ranks, each once (the upper one)
    sent = int(lr.outbox_counts.sum())
    cross = 0
    for g in range(cost.genomes):
        col_owner = lr.owner[fx[f"g{g}_second_seq_genome"]]
        cross += int((col_owner != lr.owner[g]).sum())
    assert 2 * sent == cross
    lr.close()


@pytest.mark.parametrize("name", ["synth_16x1000x300_k5", "protein_like_24x1500x300_k5"])      # uniform text / protein-like composition with
@pytest.mark.parametrize("weights", [True, False])      # low-complexity stretches; the deal's weights: from the runs / computed by every rank
@pytest.mark.parametrize("world", [2, 4])
def test_ranks_together_reproduce_the_reference_digest(world, weights, name):
    res, off, gen, k, d = H.load_large(name)
    lr, cost = _local(world, res, off, gen, k, exchange_weights=weights)
    assert lr.used_sender_ranges == weights              # (without exchanged weights the owners build their lists, as in round 2)
    assert lr.total_cost == d["total_cost"]
    lr.score_all()
    H.assert_scores_match_digest(lambda g: lr.generate_scores_part(g).as_dict(), d, f"W={world}")
    # the deal balances what the join walks (lookups above the diagonal), within one genome's worth
    walked = [n.timings()["walked_lookups"] for n in lr.ranks]
    # (the deal's weights leave out the fold of the globally last record, library.cpp:300-306: at most one group's size off)
    assert abs(sum(walked) * 2 - (lr.total_cost - cost.shared_records)) <= 2 * cost.sequences
    assert max(walked) - min(walked) <= max(walked) / 2
    lr.close()


def test_canonical_order_and_tiny_staging_under_sharding():
    from pandelos_amd import _lib
    res, off, gen, k, fx = H.load_small("synth_5x60x80_k3")
    lr, cost = _local(3, res, off, gen, k, flags=_lib.PDL_FLAG_CANONICAL_ORDER)
    for n in lr.ranks:
        n.set_option("staging_cap", 8)                   # the first attempt of every rank overflows and is repeated
    lr.score_all()
    for g in range(cost.genomes):
        got = lr.generate_scores_part(g).as_dict()
        order = np.lexsort((fx[f"g{g}_column"], fx[f"g{g}_row"]))
        for f in ("scores", "percs", "tr_percs", "row", "column"):
            assert np.array_equal(H.raw(got[f]), fx[f"g{g}_{f}"][order]), f"genome {g} {f}"
        for f in ("max_genome_score", "max_genome_score_col", "scoresMaxMappings"):
            assert np.array_equal(H.raw(got[f]), fx[f"g{g}_{f}"]), f"genome {g} {f}"
    lr.close()


@pytest.mark.parametrize("seed", list(range(5000 + int(os.environ.get("PDL_FUZZ_OFFSET", "0")), 5000 + int(os.environ.get("PDL_FUZZ_OFFSET", "0")) + int(os.environ.get("PDL_FUZZ_SEEDS_DIST", "60")))))   # widen with PDL_FUZZ_SEEDS_DIST=N
def test_random_sets_sharded_match_the_oracle(seed):
    from oracle import binding as ob
    from tests.test_gpu_fuzz import _random_set
    res, off, gen, k = _random_set(seed)
    if int((np.diff(off.astype(np.int64)) >= k).sum()) == 0:
        pytest.skip("no gene holds a k-mer (undefined in the reference)")
    ora = ob.Oracle(res, off, gen, k)
    world = 2 + seed % 3
    lr, cost = _local(world, res, off, gen, k, exchange_weights=bool(seed & 4))
    assert lr.total_cost == ora.total_cost
    lr.score_all()
    for g in range(ora.genomes):
        H.assert_scores_equal(lr.generate_scores_part(g).as_dict(), ora.scores(g), f"seed {seed} W={world} genome {g}")
    lr.close()


# ---- the torch.distributed driver, two processes on one GPU ---------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, name, out, backend="gloo"):
    import torch
    import torch.distributed as dist
    from pandelos_amd.distributed import DistributedPangenes
    from pandelos_amd.pangene_native import PangeneNative
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res, off, gen, k, d = H.load_large(name)
        t = _device_inputs(res, off, gen)
        dp = DistributedPangenes(PangeneNative.open(), torch.device("cuda", 0), device_collectives=(backend == "nccl"))
        dp.preprocess(k, *t, len(gen), len(res))
        total_cost = dp.total_cost()
        dp.score_all()
        mine = dp.my_genomes()
        import hashlib
        digests = {}
        for g in mine:
            got = dp.nat.generate_scores_part(g).as_dict()
            digests[g] = (int(got["scoresCount"]), {f: hashlib.sha256(H.raw(got[f]).tobytes()).hexdigest() for f in H.FIELDS})
        out.put((rank, total_cost, digests))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("backend,world", [("gloo", 2), ("nccl", 1)])
def test_processes_over_torch_distributed_reproduce_the_digest(backend, world):
    """gloo: two ranks sharing cuda:0, exchanges staged through the host.  nccl: RCCL takes one rank per device, so on a
    one-GPU box this is a group of ONE — still the device-tensor branch of every exchange (all-gather of the run sizes,
    the cell counts and cells through all_to_all_single, the cost all-reduce) running through RCCL for real."""
    import torch.multiprocessing as mp
    name = "synth_8x300x200_k4_div25"
    d = H.DIGESTS[name]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, name, out, backend)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    got = []
    for _ in range(600):                                 # a rank that dies must fail the test at once, not after a long wait
        try:
            got.append(out.get(timeout=0.5))
        except queue.Empty:
            assert all(p.is_alive() or p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        if len(got) == world:
            break
    assert len(got) == world
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    seen = {}
    for rank, total_cost, digests in got:
        assert total_cost == d["total_cost"]
        assert digests, f"rank {rank} scored nothing"
        seen.update(digests)
    assert sorted(seen) == list(range(d["genomes"]))
    for g, (z, sha) in seen.items():
        assert z == d["scoresCount"][g] and sha == d["sha256"][g], f"genome {g}"


def test_library_memory_is_sent_in_place():
    """The driver hands the library's outboxes to the collectives as tensors that ALIAS the memory (no copy): the tensor
    library must accept foreign device memory through the CUDA array interface on this platform, else the driver copies."""
    import torch
    from pandelos_amd.distributed import device_view
    dev = torch.device("cuda", 0)
    base = torch.arange(24, dtype=torch.int32, device=dev)
    v = device_view(base.data_ptr(), (4, 6), torch.int32, dev)
    assert v is not None, "torch.as_tensor does not alias device memory here: every outbox is copied before it is sent"
    base[7] = 1234
    torch.cuda.synchronize()
    assert v.data_ptr() == base.data_ptr() and int(v[1, 1]) == 1234 and v.shape == (4, 6)
    w = device_view(base.data_ptr() + 8, (2,), torch.int64, dev)
    assert w is not None and int(w[0]) == (3 << 32 | 2)
    assert device_view(0, (4,), torch.int32, dev) is None and device_view(base.data_ptr(), (0,), torch.int32, dev) is None
