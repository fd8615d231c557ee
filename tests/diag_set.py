#!/usr/bin/env python3
"""Diagnostic (test infrastructure: it runs the CPU oracle, so it lives under tests/): one stand-in set scored several times on the GPU, every genome's Scores compared field by field with the
CPU oracle (test infrastructure); prints what differs (values or order).  usage: python tests/diag_set.py [config] [passes] [option=value ...]"""
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
options = [o.split("=") for o in sys.argv[3:]]                  # name=value ... (pdl_set_option)
if name in CONFIGS:
    shape = CONFIGS[name]
else:
    import json
    shape = json.loads((ROOT / "tests" / "golden" / "digests.json").read_text())[name]["shape"]
gs = make_gene_set(**shape); k = calculate_k(gs.residues)
ora = ob.Oracle(gs.residues, gs.offsets, gs.genome_of, k)
want = [ora.scores(g) for g in range(gs.genomes)]
F = ("scores", "percs", "tr_percs", "row", "column", "max_genome_score", "max_genome_score_col")
nat = PangeneNative.open()
for o, v in options:
    nat.set_option(o, int(v))
for it in range(passes):
    nat.preprocess(k, gs.residues, gs.offsets, gs.genome_of)
    bad = 0
    for g in range(gs.genomes):
        s = nat.generate_scores_part(g)
        w = want[g]
        if s.scoresCount != int(w["scoresCount"]):
            print(f"pass {it} genome {g}: count {s.scoresCount} != {int(w['scoresCount'])}"); bad += 1
            ka = (s.row.astype(np.int64) << 32) | s.column; kb = (w["row"].astype(np.int64) << 32) | w["column"]
            extra, missing = np.setdiff1d(ka, kb), np.setdiff1d(kb, ka)
            for tag, ks, src in (("extra", extra, s), ("missing", missing, None)):
                for kk in ks[:6]:
                    r, c = int(kk >> 32), int(kk & 0xffffffff)
                    line = f"   {tag} cell row {r} (genome {int(gs.genome_of[r])}) col {c} (genome {int(gs.genome_of[c])})"
                    if src is not None:
                        i = int(np.nonzero(ka == kk)[0][0]); line += f" score {src.scores[i]:.6f} perc {src.percs[i]:.6f} tr {src.tr_percs[i]:.6f}"
                    else:
                        i = int(np.nonzero(kb == kk)[0][0]); line += f" score {w['scores'][i]:.6f} perc {w['percs'][i]:.6f} tr {w['tr_percs'][i]:.6f}"
                    print(line)
            continue
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
