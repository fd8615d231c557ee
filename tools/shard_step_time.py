#!/usr/bin/env python3
"""Step time of ONE rank of a W-rank run, measured on a single GPU: rank r's genome shard (the LPT deal bench.py uses) is
set before the dictionary build, exactly as in `bench.py --gpus W`.  The slowest rank bounds the W-GPU step.
usage: shard_step_time.py [--world 8] [--config mycoplasma64_standin] [--steps 10]"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--config", default="mycoplasma64_standin")
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import torch
    from pandelos_amd import distributed as D
    from pandelos_amd.calculate_k import calculate_k
    from pandelos_amd.pangene_native import PangeneNative
    from pandelos_amd.synth import CONFIGS, make_gene_set
    gs = make_gene_set(**CONFIGS[args.config])
    k = calculate_k(gs.residues)
    dev = torch.device("cuda", 0)
    pad = (-len(gs.residues)) % 16 + 16
    t_res = torch.from_numpy(np.concatenate([gs.residues, np.zeros(pad, np.uint8)])).to(dev)
    t_off = torch.from_numpy(gs.offsets.astype(np.int64)).to(dev)
    t_gen = torch.from_numpy(gs.genome_of.astype(np.int32)).to(dev)
    out = {"config": args.config, "world": args.world, "ranks": []}
    for rank in range(args.world):
        nat = PangeneNative.open(stream=torch.cuda.current_stream().cuda_stream)
        if args.world > 1:
            nat.set_genome_shard(D.shard_for_rank(gs.offsets, gs.genome_of, args.world, rank))

        def step():
            nat.preprocess_device(k, t_res.data_ptr(), t_off.data_ptr(), t_gen.data_ptr(), gs.genes, len(gs.residues))
            nat.score_all()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        tm = nat.timings()
        out["ranks"].append({"rank": rank, "ms_per_step": ms, "preprocess_ms": tm["preprocess_total_ms"], "score_ms": tm["score_total_ms"],
                             "join_ms": tm["join_ms"], "sort_seq_ms": tm["sort_seq_ms"], "ranges_ms": tm["ranges_ms"],
                             "lookups": tm["scored_lookups"], "rows": tm["scored_rows"]})
        nat.close()
    out["slowest_ms"] = max(r["ms_per_step"] for r in out["ranks"])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
