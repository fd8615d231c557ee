#!/usr/bin/env python3
"""Time of the ranking kernel under the reference's hash fallback (k so large that B^(k-1) overflows 64 bits,
library.cpp:81-86,110-121): one gene per lane, k_rank_hash.  Prints the stage times of a preprocess at that k beside the
ordinary k of the set.   usage: python tools/hash_rank_time.py [--config mycoplasma64_standin] [--k 16]"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="mycoplasma64_standin")
    ap.add_argument("--k", type=int, default=16)
    args = ap.parse_args()
    from pandelos_amd.calculate_k import calculate_k
    from pandelos_amd.pangene_native import PangeneNative
    from pandelos_amd.synth import CONFIGS, make_gene_set
    gs = make_gene_set(**CONFIGS[args.config])
    out = {"workload": args.config, "genes": gs.genes}
    for k in (int(calculate_k(gs.residues)), args.k):
        nat = PangeneNative.open()
        best = None
        for _ in range(4):
            nat.preprocess(k, gs.residues, gs.offsets, gs.genome_of)
            tm = nat.timings()
            if best is None or tm["rank_ms"] < best["rank_ms"]:
                best = {f: round(float(tm[f]), 4) for f in ("rank_ms", "sort_rank_ms", "dict_ms", "preprocess_total_ms")}
        best["kmer_occurrences"] = int(nat.cost.kmer_occurrences)
        best["dictionary_records"] = int(nat.cost.dictionary_records)
        out[f"k={k}"] = best
        nat.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
