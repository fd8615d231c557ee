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


# ---- the exchange logic of the multi-GPU driver, on CPU tensors ----------------------------------------------------
class _FakeRank:
    """Stands in for PangeneNative in DistributedPangenes: runs and outboxes with recognisable contents (rank and index
    encoded in every word), so that the driver's offsets, split sizes and copy directions can be checked without a GPU."""

    def __init__(self, rank, world, genomes=5):
        import torch
        self.rank, self.world, self.genomes = rank, world, genomes
        self.records = 3 + 4 * rank                                   # uneven runs; rank 0 has the shortest
        self.run = torch.arange(self.records, dtype=torch.int64) + (rank << 32)
        self.run_weights = np.arange(genomes, dtype=np.int64) * (rank + 1)
        self.run_costs = np.arange(genomes, dtype=np.int64) * (rank + 3) + 1
        self.cost = "cost"
        # range tuples for rank d: (rank + 2) * (d + 1) of them (also for myself: a rank owns genes of its own run too), or none
        self.tcounts = np.array([0 if (rank + d) % 3 == 0 else (rank + 2) * (d + 1) for d in range(world)], dtype=np.int64)     # (some pairs exchange nothing, rank 0 keeps nothing for itself)
        self.tkeys = torch.tensor([(d << 24) | (rank << 8) | i for d in range(world) for i in range(int(self.tcounts[d]))], dtype=torch.int32)
        self.tranges = torch.tensor([(rank << 40) | (d << 20) | i for d in range(world) for i in range(int(self.tcounts[d]))], dtype=torch.int64)
        self.tctr = np.array([10 + rank, 20 + rank, 30 + rank], dtype=np.int64)
        # cells for rank d: (rank + 1) * (d + 2) of them, none for myself
        self.counts = np.array([0 if d == rank else (rank + 1) * (d + 2) for d in range(world)], dtype=np.int64)
        cells = []
        for d in range(world):
            for i in range(int(self.counts[d])):
                cells.append([rank, d, i, 7, 8, 9])
        self.outbox = torch.tensor(cells, dtype=torch.int32).reshape(-1, 6)
        self.seen = {}

    def copy_device(self, dst, src, nbytes):
        import ctypes
        ctypes.memmove(dst, src, nbytes)

    def dist_preprocess_begin(self, k, *a, keepalive=None):
        return self.run.data_ptr(), self.records, self.records

    def dist_preprocess_finish(self, ptr, total, genome_weights=None, keepalive=None):
        self.seen["dictionary"] = keepalive.clone()
        self.seen["total"] = total
        self.seen["weights"] = np.asarray(genome_weights).copy()
        self.seen["flow"] = "owner"

    def dist_preprocess_ranges(self, run_records, genome_weights, genome_costs):
        self.seen["run_records"] = [int(x) for x in run_records]
        self.seen["weights"] = np.asarray(genome_weights).copy()
        self.seen["costs"] = np.asarray(genome_costs).copy()
        self.run += (1 << 48)                    # (in place: the call takes the head bits out of the run: the runs must travel AFTER it)
        return self.tkeys.data_ptr(), self.tranges.data_ptr(), self.tcounts, self.tctr

    def dist_preprocess_finish_ranges(self, ptr, total, d_keys, d_ranges, n_tuples, counter_sums, keepalive=None):
        full, rk, rr = keepalive
        self._full = full                        # (the runs may still be on their way: this call must not read the dictionary)
        self.seen["total"] = total
        self.seen["tuples"] = (rk[:n_tuples].tolist(), rr[:n_tuples].tolist())
        self.seen["sums"] = [int(x) for x in counter_sums]
        self.seen["flow"] = "sender"

    def dist_score_begin(self, world):
        if getattr(self, "_full", None) is not None:
            self.seen["dictionary"] = self._full.clone()
        return self.outbox.data_ptr(), self.counts

    def dist_score_finish(self, ptr, n, keepalive=None):
        self.seen["inbox"] = keepalive[:n].clone()


def _exchange_worker(rank, world, port, out, flow="host"):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if flow == "broadcast":
        os.environ["PDL_DIST_GATHER"] = "broadcast"
    else:
        os.environ.pop("PDL_DIST_GATHER", None)
    if flow.endswith("-owner"):               # the owners build their range lists from the gathered dictionary (no tuple exchange)
        os.environ["PDL_DIST_RANGES"] = "owner"
        flow = flow[:-6]
    else:
        os.environ.pop("PDL_DIST_RANGES", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fake = _FakeRank(rank, world)
        # flow "p2p" / "broadcast": the branch RCCL runs on device tensors (point-to-point gather of the runs in place,
        # all-to-all straight out of / into the exchange buffers), here on CPU tensors over gloo
        dp = D.DistributedPangenes(fake, torch.device("cpu"), device_collectives=flow != "host")
        t = torch.zeros(1)
        dp.preprocess(3, t, t, t, 1, 1)
        dp.score_all()
        out.put((rank, fake.seen["dictionary"].tolist(), fake.seen["total"], fake.seen["weights"].tolist(), fake.seen["inbox"].tolist(),
                 {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in fake.seen.items() if k in ("flow", "tuples", "sums", "costs", "run_records")}))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,flow", [(2, "host"), (3, "host"), (2, "p2p"), (3, "p2p"), (4, "p2p"), (3, "broadcast"), (3, "host-owner"), (3, "p2p-owner")])
def test_driver_exchanges_runs_and_cells_in_rank_order(world, flow):
    import queue
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, out, flow)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(400):
        try:
            r = out.get(timeout=0.5)
            got[r[0]] = r[1:]
        except queue.Empty:
            assert all(p.is_alive() or p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        if len(got) == world:
            break
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    fakes = [_FakeRank(r, world) for r in range(world)]
    sender = not flow.endswith("-owner")
    dictionary = [int(x) + ((1 << 48) if sender else 0) for f in fakes for x in f.run.tolist()]      # the runs in rank order = the dictionary (as "ranges" left them)
    weights = np.sum([f.run_weights for f in fakes], axis=0).tolist()
    for r in range(world):
        d, total, w, inbox, more = got[r]
        assert d[:total] == dictionary and total == len(dictionary) and w == weights
        assert more["flow"] == ("sender" if sender else "owner")
        if sender:        # the tuples every rank filed for r, source-rank major, keys and ranges alike; the counters and costs summed
            want_k = [(r << 24) | (s << 8) | i for s in range(world) for i in range(int(fakes[s].tcounts[r]))]
            want_r = [(s << 40) | (r << 20) | i for s in range(world) for i in range(int(fakes[s].tcounts[r]))]
            assert [list(x) for x in more["tuples"]] == [want_k, want_r]
            assert more["sums"] == np.sum([f.tctr for f in fakes], axis=0).tolist()
            assert more["costs"] == np.sum([f.run_costs for f in fakes], axis=0).tolist()
            assert more["run_records"] == [f.records for f in fakes]
        want = [[s, r, i, 7, 8, 9] for s in range(world) for i in range(int(fakes[s].counts[r]))]     # source-major, as all-to-all delivers
        assert inbox == want


# ---- an exchange that does not complete ends the rank, loudly ---------------------------------------------------------
def _deadline_worker(rank, world, port, sleeper, err_path):
    import sys
    import time
    import torch
    import torch.distributed as dist
    sys.stderr = open(err_path, "w")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["PDL_DIST_TIMEOUT_S"] = "4"
    os.environ.pop("PDL_DIST_GATHER", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fake = _FakeRank(rank, world)
    dp = D.DistributedPangenes(fake, torch.device("cpu"), device_collectives=True)
    t = torch.zeros(1)
    dp.preprocess(3, t, t, t, 1, 1)
    if rank == sleeper:
        time.sleep(60)                       # never reaches the cell exchange in time: the others must not wait for it for ever
    dp.score_all()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_a_rank_whose_exchange_does_not_complete_says_which_and_exits_nonzero(tmp_path):
    """One of two ranks sleeps in front of the cell exchange.  The other one's all-to-all cannot complete: after
    PDL_DIST_TIMEOUT_S it names the exchange, its peers and the bytes, and exits with ExchangeDeadline.EXIT_CODE — it does not
    hang until the launcher's own limit."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    port = _free_port()
    errs = [tmp_path / f"rank{r}.err" for r in range(2)]
    procs = [ctx.Process(target=_deadline_worker, args=(r, 2, port, 1, str(errs[r]))) for r in range(2)]
    for p in procs:
        p.start()
    procs[0].join(45)
    assert procs[0].exitcode == D.ExchangeDeadline.EXIT_CODE, procs[0].exitcode
    text = errs[0].read_text()
    assert "exchange 'cells: counts (all-to-all)'" in text and "rank 0 of 2" in text and "has not completed in 4 s" in text
    procs[1].kill()
    procs[1].join(10)
