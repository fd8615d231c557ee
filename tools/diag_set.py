#!/usr/bin/env python3
"""Diagnostic: one stand-in set scored several times on the GPU, every genome's Scores compared field by field with the
CPU oracle (test infrastructure); prints what differs (values or order).  usage: diag_set.py [config] [passes]"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import binding as ob
from pandelos_amd.calculate_k import calculate_k
from pandelos_amd.pangene_native import PangeneNative
from pandelos_amd.synth import CONFIGS, make_gene_set

name = sys.argv[1] if len(sys.argv) > 1 else "salmonella7_standin"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
gs = make_gene_set(**CONFIGS[name]); k = calculate_k(gs.residues)
ora = ob.Oracle(gs.residues, gs.offsets, gs.genome_of, k)
want = [ora.scores(g) for g in range(gs.genomes)]
F = ("scores", "percs", "tr_percs", "row", "column", "max_genome_score", "max_genome_score_col")
nat = PangeneNative.from_arrays(k, gs.residues, gs.offsets, gs.genome_of)
for it in range(passes):
    if it:
        nat.preprocess(k, gs.residues, gs.offsets, gs.genome_of)
    bad = 0
    for g in range(gs.genomes):
        s = nat.generate_scores_part(g)
        w = want[g]
        if s.scoresCount != int(w["scoresCount"]):
            print(f"pass {it} genome {g}: count {s.scoresCount} != {int(w['scoresCount'])}"); bad += 1; continue
        for f in F:
            a, b = np.asarray(getattr(s, f)).reshape(-1), np.asarray(w[f]).reshape(-1)
            if a.dtype == np.float32: a, b = a.view(np.uint32), b.view(np.uint32)
            d = np.nonzero(a != b)[0]
            if len(d):
                bad += 1
                print(f"pass {it} genome {g} field {f}: {len(d)} of {len(a)} differ, first at {d[:6].tolist()}")
                if f == "scores":
                    i = int(d[0]); lo, hi = max(0, i - 2), i + 4
                    print("   got  rows", s.row[lo:hi].tolist(), "cols", s.column[lo:hi].tolist(), "scores", s.scores[lo:hi].tolist())
                    print("   want rows", w["row"][lo:hi].tolist(), "cols", w["column"][lo:hi].tolist(), "scores", w["scores"][lo:hi].tolist())
                    # same multiset of cells?
                    ka = np.sort((s.row.astype(np.int64) << 32) | s.column); kb = np.sort((w["row"].astype(np.int64) << 32) | w["column"])
                    print("   same (row, column) set:", bool(np.array_equal(ka, kb)))
    print(f"pass {it}: {bad} differing fields; put-aside entries loaded again: {nat.timings()['aside_reloads']}", flush=True)
