"""Shared test helpers: golden fixture access and bit-exact Scores comparison."""
from __future__ import annotations

import hashlib
import json
import tempfile
from pathlib import Path

import numpy as np

from pandelos_amd.pangene_idata import PangeneIData
from pandelos_amd.synth import make_gene_set

GOLDEN = Path(__file__).resolve().parent / "golden"
FIELDS = ("scores", "percs", "tr_percs", "row", "column", "first_seq_genome", "second_seq_genome",
          "max_genome_score", "max_genome_score_col", "scoresMaxMappings")
SMALL_CASES = sorted(p.stem for p in GOLDEN.glob("*.npz"))
DIGESTS = json.loads((GOLDEN / "digests.json").read_text())


def raw(a):
    """float32 arrays are compared as bit patterns; everything else as is."""
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def load_small(name):
    """-> (residues, offsets, genome_of, k, fixture dict) for one tests/golden/<name>.npz"""
    fx = dict(np.load(GOLDEN / f"{name}.npz"))
    with tempfile.NamedTemporaryFile(suffix=".faa") as f:
        f.write(fx["faa"].tobytes())
        f.flush()
        data = PangeneIData.read_from_file(f.name)
    res, off, gen = data.flatten()
    return res, off, gen, int(fx["k"]), fx


def load_large(name):
    d = DIGESTS[name]
    gs = make_gene_set(**d["shape"])
    return gs.residues, gs.offsets, gs.genome_of, d["k"], d


def assert_scores_equal_fixture(get_scores, fx, genomes, label=""):
    """get_scores(g) -> dict of numpy arrays with the Scores field names."""
    for g in range(genomes):
        got = get_scores(g)
        for f in FIELDS:
            want = fx[f"g{g}_{f}"]
            have = raw(got[f])
            assert have.shape == want.shape, f"{label} genome {g} field {f}: shape {have.shape} != {want.shape}"
            assert np.array_equal(have, want), f"{label} genome {g} field {f} differs"


def assert_scores_match_digest(get_scores, d, label=""):
    for g in range(d["genomes"]):
        got = get_scores(g)
        assert int(got["scoresCount"]) == d["scoresCount"][g], f"{label} genome {g} scoresCount"
        for f in FIELDS:
            h = hashlib.sha256(raw(got[f]).tobytes()).hexdigest()
            assert h == d["sha256"][g][f], f"{label} genome {g} field {f} digest differs"


def assert_scores_equal(a: dict, b: dict, label=""):
    for f in FIELDS:
        x, y = raw(a[f]), raw(b[f])
        assert x.shape == y.shape, f"{label} field {f}: shape {x.shape} != {y.shape}"
        if not np.array_equal(x, y):
            bad = np.nonzero(x.reshape(-1) != y.reshape(-1))[0]
            raise AssertionError(f"{label} field {f}: {len(bad)} mismatches, first at {bad[:5]}")
