"""Multi-GPU driver of the hot path — one process per GPU, bytes moved by RCCL over xGMI.

The reference runs its per-genome tasks on a thread pool over ONE dictionary in shared memory
(``Pangenes.java:54-66``, ``library.cpp:73``).  Here every GPU ends up with the whole dictionary in
its own HBM and scores a disjoint set of genomes; what the shared memory did is done by two exchanges:

  dictionary   every rank ranks all k-mers (the input is on every host), keeps one interval of the
               rank space, sorts + dedups it (``pdl_dist_preprocess_begin``): 1/W of the reference's
               ``library.cpp:270-287``.  The runs are all-gathered *in place* into one device array —
               W−1 sends and W−1 receives per rank, posted together, so each of the 7 xGMI links of a
               GPU carries one peer's run (no ring, no merge: the concatenation in rank order is the
               dictionary).  The genomes are dealt by longest-processing-time on exact lookup counts
               (identical on every rank).  The posting-range lists are made by the SENDERS: every rank
               builds the range tuples of all genes of its own run (groups never straddle runs) and
               files them by the rank that owns the gene (``pdl_dist_preprocess_ranges``); one
               all-to-all of 12-byte tuples later the owners only sort what they received
               (``pdl_dist_preprocess_finish_ranges``) — nobody walks the whole dictionary.  Where that
               is not available (see the header) ``pdl_dist_preprocess_finish`` builds the lists of a
               rank's genes from the gathered dictionary, as before.
  cells        rows only meet the genes above them (half the lookups, as on one GPU); a cell whose
               column is another rank's row travels there: one variable-size all-to-all of 24-byte
               cells (``pdl_dist_score_begin`` / ``pdl_dist_score_finish``).

``DistributedPangenes`` is one rank of a ``torch.distributed`` group ("nccl" = RCCL on ROCm; "gloo"
stages through host memory and is what the CPU-side tests and the one-GPU rehearsal use).
``LocalRanks`` runs W ranks inside one process on one device with device-to-device copies in place of
the collectives: the same library calls in the same order, used by the parity tests and by
``tools/shard_step_time.py`` to time each rank's share on a single MI355X.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from . import _lib


# ---- partition helpers (host only) ---------------------------------------------------------------------
def residues_per_genome(offsets: np.ndarray, genome_of: np.ndarray, genomes: int) -> np.ndarray:
    lens = np.diff(np.asarray(offsets, dtype=np.uint64)).astype(np.float64)
    return np.bincount(np.asarray(genome_of, dtype=np.int64), weights=lens, minlength=genomes)


def lpt_shards(weights: Sequence[float], n: int) -> List[List[int]]:
    """Longest-processing-time assignment of genomes to n ranks; every list ascending.  Deterministic, so
    every rank computes the same partition without talking to the others (the library does the same on
    exact lookup counts, ``pdl_dist_genome_owner``)."""
    order = np.argsort(-np.asarray(weights, dtype=np.float64), kind="stable")
    loads = [0.0] * n
    shards: List[List[int]] = [[] for _ in range(n)]
    for g in order:
        r = int(np.argmin(loads))
        shards[r].append(int(g))
        loads[r] += float(weights[g])
    return [sorted(s) for s in shards]


def shard_for_rank(offsets, genome_of, world: int, rank: int) -> List[int]:
    genomes = int(np.max(genome_of)) + 1 if len(genome_of) else 0
    return lpt_shards(residues_per_genome(offsets, genome_of, genomes), world)[rank]


def exclusive_offsets(counts: Sequence[int]) -> np.ndarray:
    out = np.zeros(len(counts) + 1, np.int64)
    np.cumsum(np.asarray(counts, dtype=np.int64), out=out[1:])
    return out


# ---- small collectives on scalars ---------------------------------------------------------------------
def all_reduce_sum(values: Sequence[float], device=None) -> List[float]:
    """Sum a few scalars over all ranks (no-op without an initialised process group)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def all_reduce_max(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_genome_owner(shard: Sequence[int], genomes: int, device=None) -> np.ndarray:
    """owner[g] = rank that scores genome g (checks that the partition is exact)."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    mine = torch.zeros(genomes, dtype=torch.int64, device=device)
    hits = torch.zeros(genomes, dtype=torch.int64, device=device)
    if len(shard):
        idx = torch.as_tensor(list(shard), dtype=torch.int64, device=device)
        mine[idx] = rank
        hits[idx] = 1
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        dist.all_reduce(hits, op=dist.ReduceOp.SUM)
    if not bool((hits == 1).all()):
        raise RuntimeError("genome shards are not a partition of the genomes")
    return mine.cpu().numpy()


# ---- library-owned device memory as a tensor (no copy) -----------------------------------------------------------------
class _DeviceArray:
    """What ``torch.as_tensor`` needs to alias device memory it did not allocate (the CUDA array interface, version 2)."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_view(ptr: int, shape, dtype, device):
    """A tensor over ``ptr`` (an outbox of the library: valid until the next call on that context), or None where the
    interface is not available — the caller then copies."""
    import torch
    typestr = {torch.int32: "<i4", torch.int64: "<i8"}[dtype]
    if not ptr or int(np.prod(shape)) == 0 or getattr(device, "type", "cpu") != "cuda":
        return None
    try:
        t = torch.as_tensor(_DeviceArray(ptr, shape, typestr), device=device)
        return t if t.data_ptr() == ptr and t.dtype == dtype else None
    except Exception:
        return None


# ---- a deadline for every exchange --------------------------------------------------------------------------------
class _Watchdog:
    """ONE daemon thread per process looks after every exchange that is under way (a timer thread per exchange cost ~0.1 ms of
    thread start-up each, five times per step)."""

    def __init__(self):
        import threading
        self.lock = threading.Lock()
        self.open = {}                   # id -> (expiry on the monotonic clock, ExchangeDeadline)
        self.thread = None

    def _run(self):
        import time
        while True:
            time.sleep(0.2)
            now = time.monotonic()
            with self.lock:
                late = [d for (t, d) in self.open.values() if t <= now]
            if late:
                late[0]._expired()

    def add(self, deadline):
        import threading
        import time
        with self.lock:
            self.open[id(deadline)] = (time.monotonic() + deadline.seconds, deadline)
            if self.thread is None:
                self.thread = threading.Thread(target=self._run, name="pandelos-exchange-watchdog", daemon=True)
                self.thread.start()

    def remove(self, deadline):
        with self.lock:
            self.open.pop(id(deadline), None)


_watchdog = _Watchdog()


class ExchangeDeadline:
    """``with ExchangeDeadline(...)`` around a blocking exchange: a rank whose exchange has not completed in ``seconds`` says
    which exchange, with which peers and how many bytes, and EXITS NON-ZERO (``os._exit``: the main thread sits inside the
    collective and cannot be unwound; the launcher then tears the job down instead of waiting for its own limit).  A fresh
    launch may set ``PDL_DIST_GATHER=broadcast`` to swap the point-to-point gather for broadcasts.  ``PDL_DIST_TIMEOUT_S``
    sets the deadline (default 120 s, 0 = none).  ``detail`` may be a callable: the text is only made when it is needed."""

    EXIT_CODE = 87

    def __init__(self, what: str, rank: int, world: int, detail="", seconds: float = None):
        import os
        self.what, self.rank, self.world, self.detail = what, rank, world, detail
        self.seconds = float(os.environ.get("PDL_DIST_TIMEOUT_S", "120")) if seconds is None else float(seconds)

    def _expired(self):
        import os
        import sys
        detail = self.detail() if callable(self.detail) else self.detail
        sys.stderr.write(f"pandelos_amd: rank {self.rank} of {self.world}: exchange '{self.what}' has not completed in "
                         f"{self.seconds:g} s ({detail}); giving up (exit {self.EXIT_CODE}).  "
                         f"A fresh launch may set PDL_DIST_GATHER=broadcast for the dictionary gather.\n")
        sys.stderr.flush()
        os._exit(self.EXIT_CODE)

    def __enter__(self):
        if self.seconds > 0:
            _watchdog.add(self)
        return self

    def __exit__(self, *exc):
        if self.seconds > 0:
            _watchdog.remove(self)
        return False


# ---- one rank of a torch.distributed group -----------------------------------------------------------------
class DistributedPangenes:
    """The hot path on rank ``dist.get_rank()`` of an initialised process group.

    ``device_collectives`` (backend "nccl"): the exchange buffers are device tensors handed to RCCL.
    Otherwise ("gloo") they are staged through host tensors — same calls, same order."""

    def __init__(self, nat, device, device_collectives: bool):
        import torch.distributed as dist
        self.nat = nat
        self.dev = device
        self.on_device = device_collectives
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.exchange_s = {"dictionary": 0.0, "ranges": 0.0, "cells": 0.0}
        self.sender_ranges = None          # after preprocess: the range lists came from the senders
        import os
        self.p2p_gather = os.environ.get("PDL_DIST_GATHER", "p2p") != "broadcast"     # how the runs are gathered on device tensors
        self.use_sender = os.environ.get("PDL_DIST_RANGES", "sender") != "owner"      # "owner": every rank builds its range lists from the gathered dictionary

    def _sync(self):
        """The collectives run on torch's current stream of the device; the library works on its own: wait on the host."""
        if self.on_device and getattr(self.dev, "type", "cpu") == "cuda":
            import torch
            torch.cuda.current_stream(self.dev).synchronize()

    # exchange of the dictionary runs: full[offs[p] : offs[p+1]] <- rank p's run, for every p
    def _all_gather_runs(self, full, offs, wait=True):
        """-> the requests still in flight (``wait=False`` on device tensors: the caller works beside the transfer and waits later)."""
        import torch
        import torch.distributed as dist
        r, W = self.rank, self.world
        pending = []
        if self.on_device and self.p2p_gather:
            ops = []
            mine = full[offs[r]:offs[r + 1]]
            for step in range(1, W):        # peer order staggered per rank: every link is busy from the start
                dst, src = (r + step) % W, (r - step) % W
                if offs[r + 1] > offs[r]:
                    ops.append(dist.P2POp(dist.isend, mine, dst))
                if offs[src + 1] > offs[src]:
                    ops.append(dist.P2POp(dist.irecv, full[offs[src]:offs[src + 1]], src))
            if ops:
                pending = list(dist.batch_isend_irecv(ops))
        elif self.on_device:                # PDL_DIST_GATHER=broadcast: one broadcast per run, in place (no point-to-point calls)
            for p in range(W):
                if offs[p + 1] > offs[p]:
                    pending.append(dist.broadcast(full[offs[p]:offs[p + 1]], src=p, async_op=True))
        if self.on_device:
            if wait:
                for req in pending:
                    req.wait()
                pending = []
            return pending
        else:
            host = torch.empty(int(offs[-1]), dtype=torch.int64)
            host[offs[r]:offs[r + 1]] = full[offs[r]:offs[r + 1]].cpu()
            for p in range(W):
                if offs[p + 1] > offs[p]:
                    dist.broadcast(host[offs[p]:offs[p + 1]], src=p)
            full.copy_(host)
        return []

    def _all_to_all_rows(self, recv, send, recv_counts, send_counts):
        """Variable-size all-to-all of the rows of two device tensors (host-staged without device collectives).  What a rank
        files for itself does not go through the collective: RCCL moves a self-addressed chunk at a fraction of a device copy's
        rate (a group of one: 31 GB/s), so it is copied; two calls carry the rest: first every rank's rows for the ranks below it, then those
        for the ranks above it (each call's input and output are then contiguous stretches)."""
        import torch
        import torch.distributed as dist
        r, W = self.rank, self.world
        rs, ss = [int(x) for x in recv_counts], [int(x) for x in send_counts]
        n_in, n_out = sum(rs), sum(ss)
        if not self.on_device:
            h_recv = torch.empty((n_in,) + tuple(recv.shape[1:]), dtype=recv.dtype)
            dist.all_to_all_single(h_recv, send[:n_out].cpu(), output_split_sizes=rs, input_split_sizes=ss)
            recv[:n_in].copy_(h_recv)
            if getattr(self.dev, "type", "cpu") == "cuda":
                torch.cuda.synchronize(self.dev)
            return
        if ss[r] != rs[r]:
            raise RuntimeError(f"rank {r} files {ss[r]} rows for itself and expects {rs[r]}")
        so, ro = sum(ss[:r]), sum(rs[:r])
        if ss[r]:
            recv[ro:ro + rs[r]].copy_(send[so:so + ss[r]])
        zero = [0] * W
        if W > 1:      # every rank makes both calls.  "down": what goes to lower ranks, i.e. arrives from higher ones; then "up"
            dist.all_to_all_single(recv[ro + rs[r]:n_in], send[:so], output_split_sizes=zero[:r + 1] + rs[r + 1:], input_split_sizes=ss[:r] + zero[r:])
            dist.all_to_all_single(recv[:ro], send[so + ss[r]:n_out], output_split_sizes=rs[:r] + zero[r:], input_split_sizes=zero[:r + 1] + ss[r + 1:])

    def preprocess(self, k, t_res, t_off, t_gen, n_genes, n_residues):
        import time
        import torch
        import torch.distributed as dist
        nat, W, r = self.nat, self.world, self.rank
        ptr, records, _ = nat.dist_preprocess_begin(k, t_res.data_ptr(), t_off.data_ptr(), t_gen.data_ptr(), n_genes, n_residues,
                                                    W, r, keepalive=(t_res, t_off, t_gen))
        t0 = time.perf_counter()
        # one small all-gather: [records of my run | every genome's lookups inside it: above the diagonal, and as the reference counts
        # them]; summed over the ranks the former deal the genomes, the latter are "Genome g cost"
        G = len(nat.run_weights)
        mine = torch.from_numpy(np.concatenate([[records], nat.run_weights, nat.run_costs]).astype(np.int64))
        cdev = self.dev if self.on_device else None
        if self.on_device:
            mine = mine.to(self.dev)
        allv = torch.empty(W * mine.numel(), dtype=torch.int64, device=mine.device)       # (flat: gloo takes no 2-D output)
        with ExchangeDeadline("dictionary: record counts, genome weights and costs (all-gather)", r, W, f"{mine.numel() * 8} bytes per rank"):
            dist.all_gather_into_tensor(allv, mine)
            allv = allv.cpu().numpy().reshape(W, -1)
        run_records = allv[:, 0]
        offs = exclusive_offsets(run_records)
        weights, costs = allv[:, 1:1 + G].sum(axis=0), allv[:, 1 + G:1 + 2 * G].sum(axis=0)
        total = int(offs[-1])
        self.exchange_s["dictionary"] = time.perf_counter() - t0
        # the range tuples of my run, filed by owner (takes the group-head bits out of the run: the runs travel afterwards)
        made = nat.dist_preprocess_ranges(run_records, weights, costs) if self.use_sender else None
        self.sender_ranges = made is not None
        t0 = time.perf_counter()
        recv_k = recv_r = None
        if made is not None:
            kptr, rptr, send_counts, ctr = made
            mine2 = torch.from_numpy(np.concatenate([send_counts, ctr]).astype(np.int64))
            if self.on_device:
                mine2 = mine2.to(self.dev)
            allc = torch.empty(W * mine2.numel(), dtype=torch.int64, device=mine2.device)
            with ExchangeDeadline("ranges: tuple counts and run counters (all-gather)", r, W, f"{mine2.numel() * 8} bytes per rank"):
                dist.all_gather_into_tensor(allc, mine2)
                allc = allc.cpu().numpy().reshape(W, -1)
            recv_counts, sums = allc[:, r].copy(), allc[:, W:].sum(axis=0)
            n_out, n_in = int(send_counts.sum()), int(recv_counts.sum())
            recv_k = torch.empty(max(n_in, 1), dtype=torch.int32, device=self.dev)
            recv_r = torch.empty(max(n_in, 1), dtype=torch.int64, device=self.dev)
            # the outbox is sent from where the library left it (a copy only where the tensor library cannot alias foreign memory)
            send_k, send_r = device_view(kptr, (n_out,), torch.int32, self.dev), device_view(rptr, (n_out,), torch.int64, self.dev)
            if send_k is None or send_r is None:
                send_k = torch.empty(max(n_out, 1), dtype=torch.int32, device=self.dev)
                send_r = torch.empty(max(n_out, 1), dtype=torch.int64, device=self.dev)
                if n_out:
                    nat.copy_device(send_k.data_ptr(), kptr, n_out * 4)
                    nat.copy_device(send_r.data_ptr(), rptr, n_out * 8)
            detail = lambda: ("sending " + ", ".join(f"{int(c) * 12} to rank {d}" for d, c in enumerate(send_counts) if d != r) +
                              " bytes; receiving " + ", ".join(f"{int(c) * 12} from rank {p}" for p, c in enumerate(recv_counts) if p != r) + " bytes")
            with ExchangeDeadline("ranges: 12-byte tuples (two all-to-alls: keys, packed ranges)", r, W, detail):
                self._all_to_all_rows(recv_k, send_k, recv_counts, send_counts)
                self._all_to_all_rows(recv_r, send_r, recv_counts, send_counts)
                self._sync()
            del send_k, send_r
            self.exchange_s["ranges"] = time.perf_counter() - t0
            t0 = time.perf_counter()
        full = torch.empty(max(total, 1), dtype=torch.int64, device=self.dev)      # 8-byte records {gene, count|flag}
        if records:
            nat.copy_device(full.data_ptr() + int(offs[r]) * 8, ptr, records * 8)
        with ExchangeDeadline("dictionary: runs gathered in place (" + ("point-to-point" if self.on_device and self.p2p_gather else "broadcasts") + ")", r, W,
                              lambda: "sending %d bytes to each peer; receiving %s bytes" % (records * 8, ", ".join(f"{int(offs[p + 1] - offs[p]) * 8} from rank {p}" for p in range(W) if p != r))):
            if made is not None:
                # the owners' finish sorts the tuples and only NOTES where the dictionary is: the runs travel beside it
                pending = self._all_gather_runs(full, offs, wait=False)
                t1 = time.perf_counter()
                nat.dist_preprocess_finish_ranges(full.data_ptr(), total, recv_k.data_ptr(), recv_r.data_ptr(), n_in, sums,
                                                  keepalive=(full, recv_k, recv_r))
                t0 += time.perf_counter() - t1          # (the library call is not exchange time)
                for req in pending:
                    req.wait()
                self._sync()
            else:
                self._all_gather_runs(full, offs)
                self._sync()
        self.exchange_s["dictionary"] += time.perf_counter() - t0
        if made is None:
            nat.dist_preprocess_finish(full.data_ptr(), total, genome_weights=weights, keepalive=full)
        return nat.cost

    def score_all(self):
        import time
        import torch
        import torch.distributed as dist
        nat, W = self.nat, self.world
        ptr, send_counts = nat.dist_score_begin(W)
        t0 = time.perf_counter()
        cdev = self.dev if self.on_device else None
        sc = torch.as_tensor(send_counts, dtype=torch.int64, device=cdev)
        rc = torch.empty(W, dtype=torch.int64, device=cdev)
        with ExchangeDeadline("cells: counts (all-to-all)", self.rank, W, f"{W * 8} bytes"):
            dist.all_to_all_single(rc, sc)
            recv_counts = rc.cpu().numpy()
        n_out, n_in = int(send_counts.sum()), int(recv_counts.sum())
        recv = torch.empty((max(n_in, 1), 6), dtype=torch.int32, device=self.dev)
        send = device_view(ptr, (n_out, 6), torch.int32, self.dev)                      # pdl_dist_cell = 6 x 4 bytes, sent from the library's outbox
        if send is None:
            send = torch.empty((max(n_out, 1), 6), dtype=torch.int32, device=self.dev)
            if n_out:
                nat.copy_device(send.data_ptr(), ptr, n_out * _lib.DIST_CELL_BYTES)
        detail = lambda: ("sending " + ", ".join(f"{int(c) * _lib.DIST_CELL_BYTES} to rank {d}" for d, c in enumerate(send_counts) if d != self.rank) +
                          " bytes; receiving " + ", ".join(f"{int(c) * _lib.DIST_CELL_BYTES} from rank {p}" for p, c in enumerate(recv_counts) if p != self.rank) + " bytes")
        with ExchangeDeadline("cells: 24-byte cells (all-to-all)", self.rank, W, detail):
            self._all_to_all_rows(recv, send, recv_counts, send_counts)
            self._sync()
        self.exchange_s["cells"] = time.perf_counter() - t0
        nat.dist_score_finish(recv.data_ptr(), n_in, keepalive=recv)

    def total_cost(self) -> int:
        """"Total cost: P lookups" (library.cpp:349): every rank holds the costs of its own genomes; their sum over the ranks."""
        return int(round(all_reduce_sum([float(self.nat.cost.total_cost)], device=self.dev if self.on_device else None)[0]))

    def my_genomes(self) -> List[int]:
        owner = self.nat.dist_genome_owner()
        return [int(g) for g in np.nonzero(owner == self.rank)[0]]


# ---- W ranks inside one process on one device -----------------------------------------------------------------
class LocalRanks:
    """W contexts on ONE device driven in lockstep; device-to-device copies stand in for the collectives.

    Same library calls in the same order as ``DistributedPangenes``; per-rank device times come from the
    contexts' own HIP-event timings (``tools/shard_step_time.py``)."""

    def __init__(self, world: int, device: int = -1, stream=None, flags: int = 0, exchange_weights: bool = True, sender_ranges: bool = True):
        from .pangene_native import PangeneNative
        self.world = world
        self.exchange_weights = exchange_weights      # False: every rank computes the deal's weights itself (one more pass)
        self.sender_ranges = sender_ranges and exchange_weights     # the range lists come from the senders where the library can (else: owners)
        self.ranks = [PangeneNative.open(device=device, stream=stream, flags=flags) for _ in range(world)]

    def close(self):
        for n in self.ranks:
            n.close()

    def preprocess(self, k, t_res, t_off, t_gen, n_genes, n_residues):
        import torch
        W = self.world
        runs = [n.dist_preprocess_begin(k, t_res.data_ptr(), t_off.data_ptr(), t_gen.data_ptr(), n_genes, n_residues, W, r,
                                        keepalive=(t_res, t_off, t_gen)) for r, n in enumerate(self.ranks)]
        offs = exclusive_offsets([rec for _, rec, _ in runs])
        total = int(offs[-1])
        self.run_records = [rec for _, rec, _ in runs]
        weights = np.sum([n.run_weights for n in self.ranks], axis=0) if self.exchange_weights else None
        made = None
        if self.sender_ranges:               # every rank: the tuples of its run, by owner (before the runs are "gathered": it takes their head bits out)
            costs = np.sum([n.run_costs for n in self.ranks], axis=0)
            made = [n.dist_preprocess_ranges(self.run_records, weights, costs) for n in self.ranks]
            assert all(m is None for m in made) or all(m is not None for m in made), "ranks disagree on who builds the range lists"
            if made[0] is None:
                made = None
        self.used_sender_ranges = made is not None
        self.dictionaries = []
        for r, n in enumerate(self.ranks):
            full = torch.empty(max(total, 1), dtype=torch.int64, device=t_res.device)
            for p, (ptr, rec, _) in enumerate(runs):      # "all-gather": every rank's copy of every run
                if rec:
                    n.copy_device(full.data_ptr() + int(offs[p]) * 8, ptr, rec * 8)
            self.dictionaries.append(full)
        if made is None:
            for n, full in zip(self.ranks, self.dictionaries):
                n.dist_preprocess_finish(full.data_ptr(), total, genome_weights=weights, keepalive=full)
        else:
            cmat = np.stack([m[2] for m in made])            # [src][dst] tuples
            sums = np.sum([m[3] for m in made], axis=0)
            self.tuple_counts = cmat
            inboxes = []
            for d, n in enumerate(self.ranks):       # "all-to-all": column d of the count matrix, source-rank major (every inbox is
                n_in = int(cmat[:, d].sum())         # filled before any rank finishes: a finish reuses the memory of its rank's outbox)
                recv_k = torch.empty(max(n_in, 1), dtype=torch.int32, device=t_res.device)
                recv_r = torch.empty(max(n_in, 1), dtype=torch.int64, device=t_res.device)
                at = 0
                for s_ in range(W):
                    cnt = int(cmat[s_, d])
                    if cnt:
                        o = int(cmat[s_, :d].sum())
                        n.copy_device(recv_k.data_ptr() + at * 4, made[s_][0] + o * 4, cnt * 4)
                        n.copy_device(recv_r.data_ptr() + at * 8, made[s_][1] + o * 8, cnt * 8)
                        at += cnt
                inboxes.append((recv_k, recv_r, n_in))
            for n, full, (recv_k, recv_r, n_in) in zip(self.ranks, self.dictionaries, inboxes):
                n.dist_preprocess_finish_ranges(full.data_ptr(), total, recv_k.data_ptr(), recv_r.data_ptr(), n_in, sums, keepalive=(full, recv_k, recv_r))
        self.owner = self.ranks[0].dist_genome_owner()
        for n in self.ranks[1:]:
            assert np.array_equal(n.dist_genome_owner(), self.owner), "ranks disagree on the genome deal"
        # a rank reports the lookups of ITS genomes ("Genome g cost", library.cpp:535-538); "Total cost" is their sum over the ranks
        self.total_cost = sum(int(n.cost.total_cost) for n in self.ranks)
        return self.ranks[0].cost

    def genome_cost(self, genome: int) -> int:
        return self.ranks[int(self.owner[genome])].genome_cost(genome)

    def score_all(self):
        import torch
        W = self.world
        boxes = [n.dist_score_begin(W) for n in self.ranks]
        self.outbox_counts = np.stack([c for _, c in boxes])               # [src][dst]
        dev = self.dictionaries[0].device
        for d, n in enumerate(self.ranks):                                  # "all-to-all": column d of the count matrix
            n_in = int(self.outbox_counts[:, d].sum())
            recv = torch.empty((max(n_in, 1), 6), dtype=torch.int32, device=dev)
            at = 0
            for s in range(W):
                cnt = int(self.outbox_counts[s, d])
                if cnt:
                    src_off = int(self.outbox_counts[s, :d].sum())
                    n.copy_device(recv.data_ptr() + at * _lib.DIST_CELL_BYTES, boxes[s][0] + src_off * _lib.DIST_CELL_BYTES,
                                  cnt * _lib.DIST_CELL_BYTES)
                    at += cnt
            n.dist_score_finish(recv.data_ptr(), n_in, keepalive=recv)

    def generate_scores_part(self, genome: int):
        return self.ranks[int(self.owner[genome])].generate_scores_part(genome)


class RanksInTurn:
    """The W ranks of a job played ONE AFTER THE OTHER on a single context — for sets whose W contexts would not fit one
    device side by side (BASELINE configs[4]: every rank ranks all 0.9 G k-mers).  Same library calls as
    ``DistributedPangenes``; what the exchanges would carry is kept in device tensors between the passes:

      pass 1   every rank: begin                                                   -> record counts, genome weights and costs
      pass 2   every rank: begin, ranges                                           -> its run and its tuples are kept
      pass 3   every rank: begin, ranges, finish, score_begin                      -> its outbox is kept (the "all-to-all")
      pass 4   every rank: begin, ranges, finish, score_begin, score_finish        -> ``visit(rank, nat, genomes)`` sees its results

    (``sender_ranges=False``, or a build the library cannot make that way: no "ranges", the owners build their lists in
    "finish".  ``tools/shard_step_time.py`` is the timing version of the same walk.)"""

    def __init__(self, world: int, nat=None, sender_ranges: bool = True):
        from .pangene_native import PangeneNative
        self.world = world
        self.nat = nat or PangeneNative.open()
        self.sender_ranges = sender_ranges

    def run(self, k, t_res, t_off, t_gen, n_genes, n_residues, visit):
        import torch
        nat, W, dev = self.nat, self.world, t_res.device

        def begin(r):
            return nat.dist_preprocess_begin(k, t_res.data_ptr(), t_off.data_ptr(), t_gen.data_ptr(), n_genes, n_residues, W, r)

        weights = costs = None
        records = []
        for r in range(W):
            _, rec, _ = begin(r)
            records.append(rec)
            weights = nat.run_weights.copy() if weights is None else weights + nat.run_weights
            costs = nat.run_costs.copy() if costs is None else costs + nat.run_costs
        self.run_records = records
        total = int(sum(records))
        sender = self.sender_ranges
        saved, keys, rngs, tcounts, ctrs = [], [], [], [], []
        for r in range(W):
            ptr, rec, _ = begin(r)
            if sender:
                made = nat.dist_preprocess_ranges(records, weights, costs)
                if made is None:
                    assert r == 0, "ranks disagree on who builds the range lists"
                    sender = False
                else:
                    n_t = int(made[2].sum())
                    kt = torch.empty(max(n_t, 1), dtype=torch.int32, device=dev)
                    rt = torch.empty(max(n_t, 1), dtype=torch.int64, device=dev)
                    if n_t:
                        nat.copy_device(kt.data_ptr(), made[0], n_t * 4)
                        nat.copy_device(rt.data_ptr(), made[1], n_t * 8)
                    keys.append(kt[:n_t]); rngs.append(rt[:n_t]); tcounts.append(made[2]); ctrs.append(made[3])
            run_t = torch.empty(max(rec, 1), dtype=torch.int64, device=dev)
            if rec:
                nat.copy_device(run_t.data_ptr(), ptr, rec * 8)
            saved.append(run_t[:rec])
        self.used_sender_ranges = sender
        full0 = torch.cat(saved) if total else torch.zeros(1, dtype=torch.int64, device=dev)
        del saved
        full = full0 if sender else torch.empty_like(full0)     # the owners' finish works in place (head bits, the fold of the last record): a fresh copy per rank
        if sender:
            tmat, sums = np.stack(tcounts), np.sum(ctrs, axis=0)
            self.tuple_counts = tmat

        def upto_score_begin(r):
            begin(r)
            if sender:
                nat.dist_preprocess_ranges(records, weights, costs)
                n_in = int(tmat[:, r].sum())
                rk = torch.empty(max(n_in, 1), dtype=torch.int32, device=dev)
                rr = torch.empty(max(n_in, 1), dtype=torch.int64, device=dev)
                at = 0
                for s_ in range(W):               # source-rank major, as the all-to-all delivers
                    c = int(tmat[s_, r])
                    if c:
                        o = int(tmat[s_, :r].sum())
                        rk[at:at + c] = keys[s_][o:o + c]
                        rr[at:at + c] = rngs[s_][o:o + c]
                        at += c
                torch.cuda.synchronize(dev)
                nat.dist_preprocess_finish_ranges(full.data_ptr(), total, rk.data_ptr(), rr.data_ptr(), n_in, sums, keepalive=(full, rk, rr))
            else:
                full.copy_(full0)
                torch.cuda.synchronize(dev)
                nat.dist_preprocess_finish(full.data_ptr(), total, genome_weights=weights)
            return nat.dist_score_begin(W)

        outbox, counts = [], []
        self.total_cost, self.owner = 0, None
        for r in range(W):
            ptr, cnt = upto_score_begin(r)
            if self.owner is None:
                self.owner = nat.dist_genome_owner()
            self.total_cost += int(nat.cost.total_cost)
            n_out = int(cnt.sum())
            box = torch.empty((max(n_out, 1), 6), dtype=torch.int32, device=dev)
            if n_out:
                nat.copy_device(box.data_ptr(), ptr, n_out * _lib.DIST_CELL_BYTES)
            outbox.append(box)
            counts.append(cnt)
        cmat = np.stack(counts)                  # [src][dst]
        self.outbox_counts = cmat
        for r in range(W):
            n_in = int(cmat[:, r].sum())
            inbox = torch.empty((max(n_in, 1), 6), dtype=torch.int32, device=dev)
            at = 0
            for s_ in range(W):
                c = int(cmat[s_, r])
                if c:
                    o = int(cmat[s_, :r].sum())
                    inbox[at:at + c] = outbox[s_][o:o + c]
                    at += c
            upto_score_begin(r)
            nat.dist_score_finish(inbox.data_ptr(), n_in, keepalive=inbox)
            visit(r, nat, [int(g) for g in np.nonzero(self.owner == r)[0]])
            del inbox
        del outbox, full, full0
