"""Multi-GPU sharding of the scoring path — one process per GPU.

The unit of work is the reference's per-genome task (``generateScoresPart(g)``, Pangenes.java:60-66):
tasks are independent given the dictionary and their outputs (cells, per-row and per-column maxima)
need no cross-rank reduction, so genomes are dealt to ranks once, before the dictionary build, by
longest-processing-time on a cost proxy that is known up front (residues per genome; the reference's
own per-genome cost needs the dictionary).  Each rank then

  * builds the dictionary postings (round-1 status: every rank builds all of them itself),
  * builds posting-range lists / costs only for ITS genes (``pdl_set_genome_shard`` before
    ``pdl_preprocess``), and scores only its genomes.

Collectives (``torch.distributed``: RCCL on GPUs, gloo in the CPU tests) carry only the scalar
totals that a whole-job report needs.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np


def residues_per_genome(offsets: np.ndarray, genome_of: np.ndarray, genomes: int) -> np.ndarray:
    lens = np.diff(np.asarray(offsets, dtype=np.uint64)).astype(np.float64)
    return np.bincount(np.asarray(genome_of, dtype=np.int64), weights=lens, minlength=genomes)


def lpt_shards(weights: Sequence[float], n: int) -> List[List[int]]:
    """Longest-processing-time assignment of genomes to n ranks; every list ascending.  Deterministic, so
    every rank computes the same partition without talking to the others."""
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
