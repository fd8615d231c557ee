#!/usr/bin/env python3
"""Checks too long for every suite run, for the end of a round (GPU box): configs[3] at full size against the reference digests
with every scan in one launch and in genome batches of 32; the 64-genome set over eight ranks ten times over.
usage: python tests/extra_checks.py"""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np
from tests import helpers as H
from pandelos_amd.synth import make_gene_set
from pandelos_amd.pangene_native import PangeneNative
BASE = json.loads((H.GOLDEN / "digests_baseline.json").read_text())
# 1. configs[3] with every scan in one launch, and (again) in genome batches of 32 under low_memory
d = BASE["synthetic_128x4000x300"]; gs = make_gene_set(**d["shape"])
nat = PangeneNative.open(); nat.set_option("onepass_scan", 1)
nat.preprocess(d["k"], gs.residues, gs.offsets, gs.genome_of)
assert nat.cost.total_cost == d["total_cost"]
H.assert_scores_match_digest(lambda g: nat.generate_scores_part(g).as_dict(), d, "configs[3] onepass_scan")
nat.close(); print("configs[3] with one-launch scans: digests OK", flush=True)
import hashlib
nat = PangeneNative.open(); n = 0
for g, s in nat.scores_in_batches(d["k"], gs.residues, gs.offsets, gs.genome_of, 32):
    got = s.as_dict()
    assert int(got["scoresCount"]) == d["scoresCount"][g]
    for f in H.FIELDS:
        assert hashlib.sha256(H.raw(got[f]).tobytes()).hexdigest() == d["sha256"][g][f], (g, f)
    n += 1
nat.close(); print("configs[3] in batches of 32:", n, "genomes, digests OK", flush=True)
# 2. the 64-genome set over eight ranks, ten times
from tests.test_gpu_dist import _local
d = BASE["mycoplasma64_standin"]; gs = make_gene_set(**d["shape"])
for it in range(10):
    lr, cost = _local(8, gs.residues, gs.offsets, gs.genome_of, d["k"])
    assert lr.used_sender_ranges and lr.total_cost == d["total_cost"]
    lr.score_all()
    H.assert_scores_match_digest(lambda g: lr.generate_scores_part(g).as_dict(), d, f"W=8 pass {it}")
    lr.close()
print("64-genome set over eight ranks, ten times: digests OK", flush=True)
