#!/usr/bin/env python3
"""bench.py — scored gene-pairs/s of the PanDelos hot path on MI355X (BASELINE.json metric).

One *step* = one complete pass of the hot path over the workload, inputs already resident in HBM:
    N = 1   pdl_preprocess_device   K-hist, K-rank, K-sort, K-rle, K-groups, K-ranges   (library.cpp:189-371)
            pdl_score_all           K-join (+HBM-table pass for overflow rows), K-order (library.cpp:409-527)
    N > 1   pandelos_amd.distributed.DistributedPangenes, one rank per GPU: rank-interval dictionary build,
            all-gather of the runs (RCCL send/recv, one peer per xGMI link), genome deal, upper-triangle join,
            all-to-all of the mirrored cells, K-order.  Same dataset at every N: "scaling": "strong".
Outputs (all per-genome Scores blocks) stay in HBM; `host_path` in the same line is SURVEY.md §8d's wall time
(host arrays in -> every Scores block on the host, PCIe both ways) measured by this run at N = 1.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
(torch.distributed.run on 127.0.0.1) BEFORE anything touches a GPU; under torchrun it is one of the ranks.
With fewer than N GPUs visible (a one-GPU box) the ranks share device 0 and use gloo: a rehearsal, flagged
`"rehearsal": true`, whose timings mean nothing.

Workload: BASELINE.json configs[2], the canonical 64-genome set — the stand-in of BASELINE.md §4 (64 genomes x 750
genes x 370 aa, 25 % substitutions, seed 6401; the real Mycoplasma .faa cannot be fetched offline); `--faa FILE` runs a
real file.  value = N*(N-1) ordered gene pairs / seconds per step (the reference scores every row
gene against all N columns).  `scale_set` repeats the measurement on configs[3] (128 x 4000 x 300, the set BASELINE.json
shards over 8 GPUs) at the same N, so that a 1/2/4/8 series has a workload large enough to show scaling
(`--no-scale-set` skips it).

Extra objects on the JSON line:
  roofline             dominant kernel = K-join; achieved = algorithmic bytes of the join launch (8 B per lookup as the
                       reference counts them + 20 B per emitted cell + 8 B per (gene, genome) maximum) / its HIP-event
                       duration measured inside the library on the launch stream.  `traffic` = HBM bytes of the same
                       launch from the PMC counters, measured by THIS run at N = 1: two child runs of this script under
                       `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, FETCH doubled on gfx950, as
                       MI355X_MICROARCH.md prescribes) after the timed region; `--no-traffic` skips them, and without
                       rocprofv3 the committed profile of the workload is quoted instead (`traffic_source` says which).
  roofline_whole_path  the same for the whole step: SURVEY.md §8d's bytes_alg / step time.
  cpu_baseline         the reference's own library.cpp (oracle/_ref, kind "reference") — or the C restatement
                       (kind "port") when that build is absent — timed on this host's cores on the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SCALE_SET = "synthetic_128x4000x300"


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
        return {"value": pairs / secs, "unit": "gene-pairs/s", "cores": threads, "host_cores": os.cpu_count(), "kind": "reference",
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
    return {"value": pairs / (t2 - t0), "unit": "gene-pairs/s", "cores": threads, "host_cores": os.cpu_count(), "kind": "port",
            "sample": sample, "preprocess_s": t1 - t0, "scores_s": t2 - t1,
            "lookups_per_s": o.total_cost / max(t2 - t1, 1e-9)}


def measure_traffic(config: str, options):
    """HBM-side bytes per K-join launch, from the counters: FETCH_SIZE and WRITE_SIZE do not fit one pass (TCC slots), so two
    child runs of this script under rocprofv3, each with --kernel-trace only beside --pmc.  Units are KiB; on gfx950
    FETCH_SIZE tallies 128-byte requests as 64 -> doubled (MI355X_MICROARCH.md, HBM section).  -> (bytes, how) or (None, why)."""
    import csv
    import glob
    import shutil
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ):
        return None, "this run is itself being profiled"
    per_launch = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            cmd = [prof, "--pmc", counter, "--kernel-trace", "-d", td, "--output-format", "csv", "--", sys.executable,
                   str(Path(__file__).resolve()), "--steps", "2", "--warmup", "1", "--config", config, "--no-cpu-baseline", "--no-scale-set",
                   "--no-host-path", "--no-traffic"] + [x for o in options for x in ("--option", o)]
            try:
                r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=300)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 --pmc {counter}: timed out"
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter}: exit {r.returncode}"
            tot, launches = {}, {}
            for f in glob.glob(td + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    name = row["Kernel_Name"]
                    if row["Counter_Name"] == counter and "k_join" in name:
                        tot[name] = tot.get(name, 0.0) + float(row["Counter_Value"])
                        launches.setdefault(name, set()).add(row["Dispatch_Id"])
            if not tot:
                return None, f"rocprofv3 --pmc {counter}: no k_join dispatch in the output"
            per_launch[counter] = sum(tot[n] / len(launches[n]) for n in tot)     # a scoring pass launches each tier's kernel once
    return ((2.0 * per_launch["FETCH_SIZE"] + per_launch["WRITE_SIZE"]) * 1024.0,
            "measured by this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two child passes of this script, FETCH doubled (gfx950)")


def launch_ranks(args) -> int:
    """--gpus N without a launcher: start N rank processes (fresh children; this process never touches a GPU)."""
    import torch
    visible = torch.cuda.device_count()          # (counts devices without initialising the runtime)
    env = dict(os.environ)
    if visible < args.gpus:
        if args.gpus > 6:
            print(f"bench.py: {args.gpus} ranks asked for, {visible} GPU(s) visible, and a one-GPU rehearsal takes at most 6 ranks", file=sys.stderr)
            return 2
        env["PDL_BENCH_BACKEND"] = "gloo"
        env["PDL_BENCH_ONE_DEVICE"] = "1"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


class Runner:
    """One rank's view of the job: device-resident inputs of a workload and the step function."""

    def __init__(self, torch, dev, stream, n_gpus, rank, on_device_collectives, distributed=None):
        self.torch, self.dev, self.stream = torch, dev, stream
        self.n_gpus, self.rank, self.on_dev = n_gpus, rank, on_device_collectives
        self.distributed = (n_gpus > 1) if distributed is None else distributed
        self.options = []

    def load(self, gs, k):
        torch = self.torch
        from pandelos_amd.pangene_native import PangeneNative
        self.gs, self.k = gs, k
        pad = (-len(gs.residues)) % 16 + 16
        self.t_res = torch.from_numpy(np.concatenate([gs.residues, np.zeros(pad, np.uint8)])).to(self.dev)
        self.t_off = torch.from_numpy(gs.offsets.astype(np.int64)).to(self.dev)
        self.t_gen = torch.from_numpy(gs.genome_of.astype(np.int32)).to(self.dev)
        torch.cuda.synchronize()
        # one context per rank for the whole run: every step redoes all the work, only allocations are reused
        # N > 1: the library works on a stream of its own — on torch's default (null) stream every launch would synchronise with
        # RCCL's stream and back (measured with a group of one: 9.2 ms per step instead of 3.3)
        self.nat = PangeneNative.open(stream=None if self.distributed else self.stream)
        for name, value in self.options:
            self.nat.set_option(name, value)
        self.step_args = (self.k, self.t_res.data_ptr(), self.t_off.data_ptr(), self.t_gen.data_ptr(), gs.genes, len(gs.residues))
        self.dp = None
        if self.distributed:
            from pandelos_amd.distributed import DistributedPangenes
            self.dp = DistributedPangenes(self.nat, self.dev, self.on_dev)

    def step(self):
        if self.dp is None:
            self.nat.preprocess_device(*self.step_args)
            self.nat.score_all()
        else:
            gs = self.gs
            self.dp.preprocess(self.k, self.t_res, self.t_off, self.t_gen, gs.genes, len(gs.residues))
            self.dp.score_all()

    def sync(self):
        if self.distributed:
            import torch.distributed as dist
            dist.barrier()
        self.torch.cuda.synchronize()

    def measure(self, steps, warmup, coll_dev):
        from pandelos_amd import distributed as D
        for _ in range(warmup):
            self.step()
        self.nat.set_option("stage_timers", 0)      # the timed steps carry the events of the totals and of the join only
        self.step()
        self.sync()
        t0 = time.perf_counter()
        join_ms, pre_ms, score_ms, xd, xc, xr = [], [], [], [], [], []
        for _ in range(steps):
            self.step()
            ts = self.nat.timings_struct()           # (HIP-event times of this step's launches; three fields, no dictionary: the loop is timed)
            join_ms.append(ts.join_ms)               # (all three tiers)
            pre_ms.append(ts.preprocess_total_ms)
            score_ms.append(ts.score_total_ms)
            if self.dp is not None:
                xd.append(self.dp.exchange_s["dictionary"] * 1e3); xc.append(self.dp.exchange_s["cells"] * 1e3); xr.append(self.dp.exchange_s["ranges"] * 1e3)
        self.sync()
        elapsed = D.all_reduce_max(time.perf_counter() - t0, device=coll_dev)
        sec_per_step = elapsed / max(steps, 1)
        self.nat.set_option("stage_timers", 1)      # one more step, untimed, for the per-stage breakdown (stage_ms)
        self.step()
        self.sync()
        gs, cost, tm = self.gs, self.nat.cost, self.nat.timings()
        mean = lambda v: sum(v) / len(v) if v else 0.0
        # algorithmic bytes of this rank's join launch (SURVEY.md §8d join terms; lookups as the reference counts them)
        p_l, z_l, rows_l = tm["scored_lookups"], tm["emitted_cells"], tm["scored_rows"]
        join_bytes = 8.0 * p_l + 20.0 * z_l + 8.0 * rows_l * gs.genomes
        join_s = mean(join_ms) / 1e3
        # (N > 1: a rank reports the cells, the walked postings and the reference-counted lookups of ITS genomes)
        z_total, walked, p_total = D.all_reduce_sum([float(z_l), float(tm["walked_lookups"]), float(cost.total_cost)], device=coll_dev)
        bytes_alg_total = (cost.residues + 16.0 * cost.kmer_occurrences + 16.0 * cost.dictionary_records +
                           8.0 * p_total + 20.0 * z_total + 8.0 * gs.genes * gs.genomes)
        pairs = float(gs.genes) * float(gs.genes - 1)
        stage = {"preprocess": mean(pre_ms), "score": mean(score_ms), "hist": tm["hist_ms"], "rank": tm["rank_ms"],
                 "sort_rank": tm["sort_rank_ms"], "dict": tm["dict_ms"], "sort_seq": tm["sort_seq_ms"], "ranges": tm["ranges_ms"],
                 "join": tm["join_ms"], "join_overflow": tm["join_overflow_ms"], "order": tm["order_ms"],
                 "tier1_rows": tm["tier1_rows"], "tier2_rows": tm["tier2_rows"], "overflow_rows": tm["overflow_rows"], "aside_reloads": tm["aside_reloads"]}
        if self.dp is not None:
            stage.update({"dist_begin": tm["dist_begin_ms"], "dist_ranges": tm["dist_ranges_ms"], "dist_finish": tm["dist_finish_ms"], "dist_score_begin": tm["dist_score_begin_ms"],
                          "dist_score_finish": tm["dist_score_finish_ms"], "range_lists_by": "senders" if self.dp.sender_ranges else "owners",
                          "exchange_dictionary_wall": mean(xd), "exchange_ranges_wall": mean(xr), "exchange_cells_wall": mean(xc),
                          "outbox_cells_rank0": tm["outbox_cells"]})
        return {"pairs": pairs, "sec_per_step": sec_per_step, "cost": cost, "stage_ms": stage, "join_bytes": join_bytes, "join_s": join_s,
                "bytes_alg_total": bytes_alg_total, "z_total": z_total, "p_total": p_total, "walked": walked}

    def host_path(self, iters):
        """SURVEY.md §8d wall time: host arrays -> dictionary -> scores -> every genome's Scores block on the host.  One host
        thread makes the G calls (a pool of Python threads, tried in place of the reference's Java pool, only adds
        interpreter contention: 10.0 ms against 6.0)."""
        gs = self.gs
        times, cells = [], 0
        for _ in range(iters + 1):
            t0 = time.perf_counter()
            self.nat.preprocess(self.k, gs.residues, gs.offsets, gs.genome_of)
            cells = 0
            for g in range(gs.genomes):
                cells += int(self.nat.generate_scores_part(g).scoresCount)
            times.append(time.perf_counter() - t0)
        times = sorted(times[1:])                           # first pass allocates the pinned mirror
        # (the mean is not quoted: one pass in ten or so is interrupted by the interpreter's collector or the host scheduler
        #  and takes 2-7x as long — see p90 / max; the C-ABI figure below has no such pass)
        return {"ms": 1e3 * times[len(times) // 2], "ms_min": 1e3 * times[0], "ms_p90": 1e3 * times[min(len(times) - 1, int(0.9 * len(times)))],
                "ms_max": 1e3 * times[-1], "iterations": len(times), "cells": cells,
                "through": "pdl_preprocess + pdl_compute_scores for every genome, via the Python binding (one extra copy per array)"}

    def host_path_native(self, iters, threads):
        """The same wall time through the C ABI alone (pandelos_amd/lib/host_path: pdl_preprocess + a pool of host threads
        making the G pdl_compute_scores calls, Pangenes.java:54-66) — what a compiled host pays, without the binding's copies."""
        exe = ROOT / "pandelos_amd" / "lib" / "host_path"
        if not exe.exists():
            return None
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            faa = Path(td) / "set.faa"
            self.gs.write_faa(faa)
            try:
                r = subprocess.run([str(exe), str(faa), str(self.k), str(threads), str(iters)], capture_output=True, text=True, timeout=300)
            except subprocess.TimeoutExpired:
                return None
        if r.returncode != 0:
            return {"error": (r.stderr or "").strip()[-200:]}
        out = json.loads(r.stdout.strip().splitlines()[-1])
        out["through"] = "pdl_preprocess + pdl_compute_scores for every genome from a pool of host threads, C ABI only (pandelos_amd/lib/host_path)"
        return out

    def pipeline(self, iters):
        """north_star's whole path on this set, ".faa in, edge list out", through the native host (pandelos_amd/lib/pangenes, what
        pandelos_mi355x.sh runs in place of the reference's `java ... Pangenes -i -k -o`): .faa -> HBM (pdl_ingest_faa, k on the
        way) -> dictionary -> scores -> best-hit filter on the device, edges to the host (pdl_compute_edges) -> network
        container and .net text.  Wall times of the stages, medians over `iters` passes in one process."""
        exe = ROOT / "pandelos_amd" / "lib" / "pangenes"
        if not exe.exists():
            return None
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            faa, net, tim = Path(td) / "set.faa", Path(td) / "set.net", Path(td) / "timings.json"
            self.gs.write_faa(faa)
            try:
                r = subprocess.run([str(exe), "-i", str(faa), "-k", str(self.k), "-o", str(net), "--timings", str(tim), "--repeat", str(iters + 1)],
                                   capture_output=True, text=True, timeout=600)
            except subprocess.TimeoutExpired:
                return None
            if r.returncode != 0 or not tim.exists():
                return {"error": (r.stderr or r.stdout or "").strip()[-200:]}
            out = json.loads(tim.read_text())
            out["net_lines"] = sum(1 for _ in open(net))
        out["through"] = ("pandelos_amd/lib/pangenes (native host over the C ABI): pdl_ingest_faa, pdl_preprocess_ingested, pdl_score_all, "
                          "pdl_compute_edges for every genome, PangeneNet container + .net text; medians, first pass left out")
        return out

    def close(self):
        self.nat.close()
        del self.t_res, self.t_off, self.t_gen


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="mycoplasma64_standin")
    ap.add_argument("--faa", default=None, help="run a real .faa instead of the synthetic stand-in")
    ap.add_argument("--k", type=int, default=0, help="override k (default: calculate_k)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scale-set", action="store_true")
    ap.add_argument("--no-host-path", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child passes that measure roofline.traffic")
    ap.add_argument("--cpu-threads", type=int, default=64)
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE", help="pdl_set_option on every context (experiments)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    import torch
    from pandelos_amd.calculate_k import calculate_k
    from pandelos_amd.pangene_idata import PangeneIData
    from pandelos_amd.synth import CONFIGS, GeneSet, make_gene_set

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # (PDL_BENCH_FORCE_DIST=1 under a one-rank torchrun: the N > 1 code path, RCCL calls included, with a group of one)
    distributed = world > 1 or os.environ.get("PDL_BENCH_FORCE_DIST") == "1"
    # PDL_BENCH_BACKEND=gloo + PDL_BENCH_ONE_DEVICE=1 rehearse the N > 1 flow on a one-GPU box (every rank on cuda:0,
    # exchanges staged through host tensors); the driver's runs use RCCL ("nccl"), one rank per GPU.
    backend = os.environ.get("PDL_BENCH_BACKEND", "nccl")
    rehearsal = os.environ.get("PDL_BENCH_ONE_DEVICE") == "1"
    dev = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(dev)
    if distributed:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    n_gpus = world if distributed else 1
    coll_dev = dev if backend == "nccl" else None      # where the scalar collectives live
    stream = torch.cuda.current_stream().cuda_stream
    run = Runner(torch, dev, stream, n_gpus, rank, backend == "nccl", distributed)
    run.options = [(o.split("=")[0], int(o.split("=")[1])) for o in args.option]

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
    run.load(gs, k)
    m = run.measure(args.steps, args.warmup, coll_dev)
    cost = m["cost"]
    achieved = m["join_bytes"] / m["join_s"] / 1e9 if m["join_s"] > 0 else 0.0
    whole = m["bytes_alg_total"] / m["sec_per_step"] / 1e9

    # HBM-side traffic of the join launch from the committed PMC profile of this workload (FETCH_SIZE / WRITE_SIZE
    # are collected in separate rocprofv3 --pmc passes, corrected as MI355X_MICROARCH.md prescribes; see
    # profiles/*_pmc_traffic_join.json).  null when no profile of this workload/sharding is on file.
    traffic, traffic_src = None, None
    if n_gpus == 1 and not args.faa:
        if not args.no_traffic:
            traffic, traffic_src = measure_traffic(args.config, args.option)
        if traffic is None:
            why = traffic_src
            for pf in sorted((ROOT / "profiles").glob("r*b_pmc_traffic_join.json"), reverse=True):
                try:
                    prof = json.loads(pf.read_text())
                    if prof.get("workload") == args.config:
                        traffic = sum(v.get("hbm_bytes_per_launch_corrected", 0.0) for kname, v in prof["kernels"].items()
                                      if "k_join" in kname)
                        traffic_src = f"profiles/{pf.name} (committed profile" + (f"; live measurement: {why})" if why else ")")
                        break
                except Exception:
                    pass

    sharding = ("whole dataset on one GPU" if n_gpus == 1 else
                f"{n_gpus} ranks: rank-interval dictionary build + all-gather of the runs, genomes dealt by lookups above the diagonal, "
                "upper-triangle join + all-to-all of mirrored cells")
    out = {
        "metric": "scored gene-pairs/sec (whole node)",
        "value": m["pairs"] / m["sec_per_step"],
        "unit": "gene-pairs/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": m["sec_per_step"] * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u32 accumulate + f32 finalize",
        "data": data_kind,
        "config": {"workload": workload, "genes": gs.genes, "genomes": gs.genomes, "residues": int(cost.residues),
                   "k": int(k), "kmer_occurrences": int(cost.kmer_occurrences),
                   "dictionary_records": int(cost.dictionary_records), "lookups": int(m["p_total"]),
                   "lookups_walked": int(m["walked"]), "emitted_cells": int(m["z_total"]),
                   "cells_per_row": m["z_total"] / max(gs.genes, 1), "sharding": sharding},
        "timed_region": "inputs and outputs resident in HBM (host_path = SURVEY §8d host-to-host wall time); timed steps carry the HIP events of the two totals and of the join only, stage_ms comes from one more step outside the timed region",
        "lookups_per_s": m["p_total"] / m["sec_per_step"],
        "stage_ms": m["stage_ms"],
        "roofline": {"bound": "hbm", "kernel": "k_join_part / k_join_lds (+k_join_hbm): every tier of the join, one event pair", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                     "bytes_per_launch": m["join_bytes"], "launch_ms": m["join_s"] * 1e3},
        "roofline_whole_path": {"bound": "hbm", "achieved": whole, "peak": HBM_PEAK_GBPS * n_gpus, "unit": "GB/s",
                                "frac": whole / (HBM_PEAK_GBPS * n_gpus), "bytes_alg": m["bytes_alg_total"],
                                "formula": "1*R + 16*M + 16*U + 8*P + 20*Z + 8*N*G (SURVEY.md §8d)"},
    }
    if rehearsal:
        out["rehearsal"] = True
    if n_gpus == 1 and not args.no_host_path:
        hp = run.host_path(9)
        hp["value"] = m["pairs"] / (hp["ms"] / 1e3)
        hp["unit"] = "gene-pairs/s"
        out["host_path"] = hp
        hn = run.host_path_native(9, max(1, min(8, os.cpu_count() or 1)))
        if hn and "ms" in hn:
            hn["value"] = m["pairs"] / (hn["ms"] / 1e3)
            hn["unit"] = "gene-pairs/s"
        if hn:
            out["host_path_native"] = hn
        if hn and "value" in hn:
            # SURVEY §8d / BASELINE.md §3 wall time (residues in host memory -> every Scores block on the host) beside `value`
            # (inputs and outputs resident in HBM, the task contract's definition): the C-ABI figure, as a compiled host pays it
            out["value_host_to_host"] = hn["value"]
        pl = run.pipeline(5)
        if pl:
            if "faa_to_net_ms" in pl:
                pl["value"] = m["pairs"] / (pl["faa_to_net_ms"] / 1e3)
                pl["unit"] = "gene-pairs/s"
            out["pipeline"] = pl
    run.close()

    # ---- the set BASELINE.json shards over 8 GPUs, at this N ------------------------------------------------------
    if not args.no_scale_set and not args.faa and args.config != SCALE_SET:
        shape = CONFIGS[SCALE_SET]
        gs2 = make_gene_set(**shape)
        k2 = calculate_k(gs2.residues)
        run.load(gs2, k2)
        m2 = run.measure(max(2, min(args.steps, 5)), 1, coll_dev)
        a2 = m2["join_bytes"] / m2["join_s"] / 1e9 if m2["join_s"] > 0 else 0.0
        out["scale_set"] = {
            "workload": f"{SCALE_SET}: synthetic 128 genomes x 4000 genes x 300 aa, 8% substitutions, seed {shape['seed']}",
            "value": m2["pairs"] / m2["sec_per_step"], "unit": "gene-pairs/s", "n_gpus": n_gpus, "ms_per_step": m2["sec_per_step"] * 1e3,
            "genes": gs2.genes, "k": int(k2), "lookups": int(m2["p_total"]), "lookups_walked": int(m2["walked"]),
            "emitted_cells": int(m2["z_total"]), "stage_ms": m2["stage_ms"],
            "roofline_join_frac": a2 / HBM_PEAK_GBPS,
            "roofline_whole_path_frac": m2["bytes_alg_total"] / m2["sec_per_step"] / 1e9 / (HBM_PEAK_GBPS * n_gpus)}
        run.close()
        del gs2

    if rank == 0:
        if n_gpus == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(gs, k, m["pairs"], args.cpu_threads)
            except Exception as e:  # the baseline must never take the measurement down
                out["cpu_baseline"] = {"value": None, "unit": "gene-pairs/s", "cores": 0, "kind": "unavailable",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
