#!/usr/bin/env python3
"""How many rows of a set each join tier took (one GPU).  usage: tier_counts.py <config> [option=value ...]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from pandelos_amd.calculate_k import calculate_k
from pandelos_amd.pangene_native import PangeneNative
from pandelos_amd.synth import CONFIGS, make_gene_set
name = sys.argv[1]
gs = make_gene_set(**CONFIGS[name]); k = calculate_k(gs.residues)
nat = PangeneNative.open()
for o in sys.argv[2:]:
    n, v = o.split("="); nat.set_option(n, int(v))
for it in range(3):
    nat.preprocess(k, gs.residues, gs.offsets, gs.genome_of); nat.score_all()
t = nat.timings()
print({x: t[x] for x in ("scored_rows", "tier1_rows", "tier2_rows", "overflow_rows", "join_ms", "join_overflow_ms", "order_ms", "score_total_ms", "preprocess_total_ms", "emitted_cells", "aside_reloads")})
