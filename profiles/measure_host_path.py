#!/usr/bin/env python3
"""PCIe-inclusive rate of the boundary as the JNI/C-ABI host sees it: host arrays in, every genome's Scores block
back in host memory (pdl_preprocess + G x pdl_compute_scores).  Not the bench metric (bench.py times the
device-resident path); quoted in DESIGN.md §4."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from pandelos_amd.calculate_k import calculate_k
from pandelos_amd.pangene_native import PangeneNative
from pandelos_amd.synth import CONFIGS, make_gene_set

name = sys.argv[1] if len(sys.argv) > 1 else "mycoplasma64_standin"
gs = make_gene_set(**CONFIGS[name])
k = calculate_k(gs.residues)
nat = PangeneNative.open()
best = None
for rep in range(4):
    t0 = time.perf_counter()
    nat.preprocess(k, gs.residues, gs.offsets, gs.genome_of)
    t1 = time.perf_counter()
    cells = 0
    for g in range(gs.genomes):
        cells += nat.generate_scores_part(g).scoresCount
    t2 = time.perf_counter()
    r = {"preprocess_s": t1 - t0, "scores_to_host_s": t2 - t1, "total_s": t2 - t0}
    if best is None or r["total_s"] < best["total_s"]:
        best = r
pairs = gs.genes * (gs.genes - 1)
best.update(workload=name, genes=gs.genes, genomes=gs.genomes, k=k, cells=cells, gene_pairs_per_s=pairs / best["total_s"])
print(json.dumps(best))
