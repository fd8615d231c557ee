#!/usr/bin/env python3
"""bench.py — scored gene-pairs/s of the PanDelos hot path on MI355X (BASELINE.json metric).

One *step* = one complete pass of the hot path over the workload, inputs already resident in HBM:
    pdl_preprocess_device   K-hist, K-rank, K-sort, K-rle, K-groups, K-ranges   (library.cpp:189-371)
    pdl_score_all           K-join (+HBM-table pass for overflow rows), K-order (library.cpp:409-527)
Outputs (all per-genome Scores blocks) stay in HBM; the PCIe-inclusive rate is reported in DESIGN.md.

Workload at N=1: BASELINE.json configs[2], the canonical 64-genome set.  The real 64-Mycoplasma
.faa cannot be fetched offline, so the stand-in of BASELINE.md §4 is generated (64 genomes x 750
genes x 370 aa, 25 % substitutions, seed 6401); `--faa FILE` runs a real file instead.

value = N*(N-1) ordered gene pairs / seconds per step  (the reference scores every
row gene against all N columns).

N > 1 (launched by torch.distributed.run, one rank per GPU): the genome tasks are sharded over the
ranks (LPT on residues per genome, fixed before the dictionary build); the workload is the same set,
so scaling is "strong".  Round-1 status: every rank builds the dictionary postings itself, range lists
and scoring are per shard; collectives carry only scalar totals.

Extra objects on the JSON line:
  roofline     dominant kernel = K-join; achieved = algorithmic bytes of the join launch
               (8 B per lookup + 20 B per emitted cell + 8 B per (gene, genome) maximum) divided by
               its HIP-event duration measured inside the library on the launch stream.
  cpu_baseline the reference's own library.cpp (oracle/_ref, kind "reference") — or the C
               restatement (kind "port") when that build is absent — timed on this host's cores
               on the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(gs, k, pairs, max_threads):
    """Time the reference (or the port) on the host cores on the same workload."""
    from oracle import binding as ob
    threads = max(1, min(max_threads, os.cpu_count() or 1))
    sample = f"whole workload ({gs.genes} genes, {gs.genomes} genomes), preprocess 1 thread + scoring {threads} threads"
    if ob.have_reference():
        with tempfile.TemporaryDirectory() as td:
            faa = Path(td) / "bench.faa"
            gs.write_faa(faa)
            info = ob.run_harness(ob.REF_SO, faa, k, threads=threads, timeout=1500)
        secs = info["preprocess_s"] + info["scores_s"]
        return {"value": pairs / secs, "unit": "gene-pairs/s", "cores": threads, "kind": "reference",
                "sample": sample, "preprocess_s": info["preprocess_s"], "scores_s": info["scores_s"],
                "lookups_per_s": (info["total_cost"] or 0) / max(info["scores_s"], 1e-9)}
    # port: the C restatement, scoring threaded over genomes from Python (ctypes releases the GIL)
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()
    o = ob.Oracle(gs.residues, gs.offsets, gs.genome_of, k)
    t1 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda g: o.scores(g)["scoresCount"], range(o.genomes)))
    t2 = time.perf_counter()
    return {"value": pairs / (t2 - t0), "unit": "gene-pairs/s", "cores": threads, "kind": "port",
            "sample": sample, "preprocess_s": t1 - t0, "scores_s": t2 - t1,
            "lookups_per_s": o.total_cost / max(t2 - t1, 1e-9)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="mycoplasma64_standin")
    ap.add_argument("--faa", default=None, help="run a real .faa instead of the synthetic stand-in")
    ap.add_argument("--k", type=int, default=0, help="override k (default: calculate_k)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=64)
    args = ap.parse_args()

    import torch
    from pandelos_amd.calculate_k import calculate_k
    from pandelos_amd.pangene_idata import PangeneIData
    from pandelos_amd.pangene_native import PangeneNative
    from pandelos_amd.synth import CONFIGS, GeneSet, make_gene_set

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    # PDL_BENCH_BACKEND=gloo + PDL_BENCH_ONE_DEVICE=1 rehearse the N > 1 flow on a one-GPU box (every rank on cuda:0,
    # collectives on CPU tensors); the driver's runs use RCCL ("nccl"), one rank per GPU.
    backend = os.environ.get("PDL_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("PDL_BENCH_ONE_DEVICE") == "1" else local_rank
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    if distributed:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    n_gpus = world if distributed else 1
    coll_dev = dev if backend == "nccl" else None      # where the scalar collectives live

    # ---- workload -----------------------------------------------------------------------------------
    if args.faa:
        data = PangeneIData.read_from_file(args.faa)
        res, off, gen = data.flatten()
        gs = GeneSet(res, off, gen, np.zeros(len(gen), np.int64))
        workload = f"real .faa {os.path.basename(args.faa)}"
        data_kind = "real"
    else:
        shape = CONFIGS[args.config]
        gs = make_gene_set(**shape)
        workload = (f"{args.config}: synthetic {shape['genomes']} genomes x {shape['genes_per_genome']} genes x "
                    f"{shape['mean_len']} aa, {int(shape['sub_rate'] * 100)}% substitutions, seed {shape['seed']}")
        data_kind = "synthetic"
    k = args.k or calculate_k(gs.residues)
    n_genes, n_genomes = gs.genes, gs.genomes
    pairs = float(n_genes) * float(n_genes - 1)

    pad = (-len(gs.residues)) % 16 + 16
    t_res = torch.from_numpy(np.concatenate([gs.residues, np.zeros(pad, np.uint8)])).to(dev)
    t_off = torch.from_numpy(gs.offsets.astype(np.int64)).to(dev)
    t_gen = torch.from_numpy(gs.genome_of.astype(np.int32)).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()

    from pandelos_amd import distributed as D

    # one context per rank for the whole run: every step redoes all the work, only allocations are reused.
    # N > 1: genomes are dealt to ranks once (LPT on residues per genome); the shard is in force before the
    # dictionary build, so a rank builds range lists for, and scores, only its own genes.
    nat = PangeneNative.open(stream=stream)
    if n_gpus > 1:
        shard = D.shard_for_rank(gs.offsets, gs.genome_of, n_gpus, rank)
        D.gather_genome_owner(shard, n_genomes, device=coll_dev)      # the shards must partition the genomes
        nat.set_genome_shard(shard)

    def one_step():
        nat.preprocess_device(k, t_res.data_ptr(), t_off.data_ptr(), t_gen.data_ptr(), n_genes, len(gs.residues))
        nat.score_all()
        return nat

    def sync():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    sync()
    t0 = time.perf_counter()
    join_ms, pre_ms, score_ms = [], [], []
    for _ in range(args.steps):
        one_step()
        tm = nat.timings()
        join_ms.append(tm["join_ms"] + tm["join_overflow_ms"])
        pre_ms.append(tm["preprocess_total_ms"])
        score_ms.append(tm["score_total_ms"])
    sync()
    elapsed = time.perf_counter() - t0
    elapsed = D.all_reduce_max(elapsed, device=coll_dev)
    sec_per_step = elapsed / max(args.steps, 1)

    cost = nat.cost
    tm = nat.timings()
    # algorithmic bytes of this rank's join launch (SURVEY.md §8d join terms)
    p_l, z_l, rows_l = tm["scored_lookups"], tm["emitted_cells"], tm["scored_rows"]
    join_bytes = 8.0 * p_l + 20.0 * z_l + 8.0 * rows_l * n_genomes
    join_s = (sum(join_ms) / len(join_ms)) / 1e3 if join_ms else 0.0
    achieved = join_bytes / join_s / 1e9 if join_s > 0 else 0.0
    z_total, p_total = D.all_reduce_sum([float(z_l), float(cost.total_cost)], device=coll_dev)
    bytes_alg_total = (cost.residues + 16.0 * cost.kmer_occurrences + 16.0 * cost.dictionary_records +
                       8.0 * p_total + 20.0 * z_total + 8.0 * n_genes * n_genomes)

    # HBM-side traffic of the join launch from the committed PMC profile of this workload (FETCH_SIZE / WRITE_SIZE
    # are collected in separate rocprofv3 --pmc passes, corrected as MI355X_MICROARCH.md prescribes; see
    # profiles/*_pmc_traffic_join.json).  null when no profile of this workload/sharding is on file.
    traffic = None
    if n_gpus == 1 and not args.faa:
        for pf in sorted((ROOT / "profiles").glob("r*b_pmc_traffic_join.json"), reverse=True):
            try:
                prof = json.loads(pf.read_text())
                if prof.get("workload") == args.config:
                    traffic = sum(v.get("hbm_bytes_per_launch_corrected", 0.0) for kname, v in prof["kernels"].items()
                                  if "k_join" in kname)
                    break
            except Exception:
                pass

    out = {
        "metric": "scored gene-pairs/sec (whole node)",
        "value": pairs / sec_per_step,
        "unit": "gene-pairs/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": sec_per_step * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u32 accumulate + f32 finalize",
        "data": data_kind,
        "config": {"workload": workload, "genes": n_genes, "genomes": n_genomes, "residues": int(cost.residues),
                   "k": int(k), "kmer_occurrences": int(cost.kmer_occurrences),
                   "dictionary_records": int(cost.dictionary_records), "lookups": int(p_total),
                   "emitted_cells": int(z_total), "sharding": f"genome tasks over {n_gpus} GPU(s) (LPT on residues); postings built on every rank, "
                               "range lists and scoring per shard"},
        "achieved_hbm_GBps_whole_path": bytes_alg_total / sec_per_step / 1e9,
        "lookups_per_s": p_total / sec_per_step,
        "stage_ms": {"preprocess": sum(pre_ms) / len(pre_ms), "score": sum(score_ms) / len(score_ms),
                     "hist": tm["hist_ms"], "rank": tm["rank_ms"], "sort_rank": tm["sort_rank_ms"], "dict": tm["dict_ms"],
                     "sort_seq": tm["sort_seq_ms"], "ranges": tm["ranges_ms"], "join": tm["join_ms"],
                     "join_overflow": tm["join_overflow_ms"], "order": tm["order_ms"],
                     "tier2_rows": tm["tier2_rows"], "overflow_rows": tm["overflow_rows"]},
        "roofline": {"bound": "hbm", "kernel": "k_join_lds (+k_join_hbm)", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "bytes_per_launch": join_bytes, "launch_ms": join_s * 1e3},
    }
    nat.close()
    if rank == 0:
        if n_gpus == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(gs, k, pairs, args.cpu_threads)
            except Exception as e:  # the baseline must never take the measurement down
                out["cpu_baseline"] = {"value": None, "unit": "gene-pairs/s", "cores": 0, "kind": "unavailable",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
