#!/usr/bin/env python3
"""Join time with the partition tier on / off / by itself (auto) on a few sets.  usage: tier0_compare.py [config ...]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from pandelos_amd.calculate_k import calculate_k
from pandelos_amd.pangene_native import PangeneNative
from pandelos_amd.synth import CONFIGS, make_gene_set
from tests import helpers as H
names = sys.argv[1:] or ["salmonella7_standin", "xanthomonas14_standin", "mycoplasma64_standin", "synth_16x1000x300_k5", "protein_like_24x1500x300_k5"]
for name in names:
    if name in CONFIGS:
        gs = make_gene_set(**CONFIGS[name]); res, off, gen = gs.residues, gs.offsets, gs.genome_of; k = calculate_k(res)
    else:
        res, off, gen, k, _ = H.load_large(name)
    for mode in (-1, 0, 1):
        nat = PangeneNative.open()
        nat.set_option("join_tier0", mode)
        best = None
        for it in range(4):
            nat.preprocess(k, res, off, gen); nat.score_all()
            t = nat.timings()
            if best is None or t["join_ms"] < best["join_ms"]: best = t
        print(f"{name:32s} tier0={mode:2d} join {best['join_ms']:.3f} ms  score {best['score_total_ms']:.3f}  rows {best['scored_rows']} tier1_rows {best['tier1_rows']} tier2 {best['tier2_rows']} walked/row {best['walked_lookups'] / max(1, best['scored_rows']):.0f}", flush=True)
        nat.close()
