"""N > 1 host logic on CPU: two gloo ranks partition the genomes, each scores its shard (the CPU oracle stands in for
the device here — it is the checker, the thing under test is the partition + collectives), totals are all-reduced
and must equal the single-process result."""
import os
import socket

import numpy as np
import pytest

from pandelos_amd import distributed as D
from pandelos_amd.synth import make_gene_set


def test_lpt_is_a_partition_and_balanced():
    w = np.array([5, 9, 1, 7, 3, 3, 8, 2], dtype=float)
    for n in (1, 2, 3, 4, 8):
        shards = D.lpt_shards(w, n)
        assert sorted(g for s in shards for g in s) == list(range(len(w)))
        loads = [sum(w[g] for g in s) for s in shards]
        assert max(loads) - min(loads) <= max(w)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, shape, k, out):
    import torch
    import torch.distributed as dist
    from oracle import binding as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gs = make_gene_set(**shape)
        shard = D.shard_for_rank(gs.offsets, gs.genome_of, world, rank)
        owner = D.gather_genome_owner(shard, gs.genomes)
        assert all(owner[g] == rank for g in shard)
        ora = ob.Oracle(gs.residues, gs.offsets, gs.genome_of, k)
        cells = sum(int(ora.scores(g)["scoresCount"]) for g in shard)
        cost = sum(ora.genome_cost(g) for g in shard)
        tot_cells, tot_cost = D.all_reduce_sum([cells, cost])
        slowest = D.all_reduce_max(float(rank))
        if rank == 0:
            full_cells = sum(int(ora.scores(g)["scoresCount"]) for g in range(ora.genomes))
            out.put((tot_cells, tot_cost, full_cells, ora.total_cost, slowest, [int(x) for x in owner]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_gloo_ranks_cover_all_genomes_once():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    shape = dict(genomes=7, genes_per_genome=60, mean_len=70, sub_rate=0.1, seed=77)
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, shape, 3, out)) for r in range(2)]
    for p in procs:
        p.start()
    tot_cells, tot_cost, full_cells, full_cost, slowest, owner = out.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert tot_cells == full_cells and tot_cost == full_cost
    assert slowest == 1.0 and sorted(set(owner)) == [0, 1]
