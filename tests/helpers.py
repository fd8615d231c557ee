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


# ---- genome-pair sampling of a set too large for the reference (configs[4]) ------------------------------------------
# A cell's three values depend on the two genes alone (their k-mer multisets under the same k and the same alphabet ranks):
# the cells of rows of genome A against columns of genome B are the same whether the dictionary was built from all 512
# genomes or from a handful of them — except around the reference's fold of the globally LAST record into the preceding
# rank-group (library.cpp:300-306), which touches only cells of the gene that holds that record.  So the reference run on a
# few genomes pins those genome pairs of the full run, once the genes holding the largest-rank k-mer (of the subset and of
# the full set) are left out on both sides.
LETTERS = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)


def genes_holding_the_largest_kmer(residues, offsets, k, gene_ids=None):
    """Global ids of the genes that contain an occurrence of the largest-rank k-mer (ranks = base-20 numbers over LETTERS in
    ascending letter order, library.cpp:96-100,134-150) among `gene_ids` (default: all genes).  Chunked: the full
    512-genome set is 0.9 G residues."""
    code = np.full(256, 255, np.uint8)
    code[LETTERS] = np.arange(20, dtype=np.uint8)
    off = np.asarray(offsets, dtype=np.int64)
    ids = np.arange(len(off) - 1, dtype=np.int64) if gene_ids is None else np.asarray(gene_ids, dtype=np.int64)
    best, holders = -1, []
    step = 4096
    for i0 in range(0, len(ids), step):
        part = ids[i0:i0 + step]
        contiguous = len(part) and part[-1] - part[0] + 1 == len(part)
        for lo, hi, genes in ([(int(off[part[0]]), int(off[part[-1] + 1]), part)] if contiguous
                              else [(int(off[g]), int(off[g + 1]), np.array([g])) for g in part]):
            c = code[residues[lo:hi]].astype(np.int64)
            assert c.max(initial=0) < 20
            n = len(c) - k + 1
            if n <= 0:
                continue
            v = np.zeros(n, np.int64)
            for j in range(k):
                v = v * 20 + c[j:j + n]
            # k-mers must not straddle genes: position p (chunk-local) belongs to gene `gi`, valid iff p + k <= end of that gene
            ends = off[genes + 1] - lo
            gi = np.searchsorted(ends, np.arange(n), side="right")
            v[np.arange(n) + k > ends[np.minimum(gi, len(ends) - 1)]] = -1
            m = int(v.max())
            if m > best:
                best, holders = m, []
            if m == best and m >= 0:
                holders.extend(int(genes[x]) for x in np.unique(gi[v == m]))
    return sorted(set(holders))


def pair_cells_digest(block, first_gene_a, first_gene_b, genome_b, excluded_a=(), excluded_b=()):
    """SHA-256 (and count) of the cells of one Scores block (rows of genome A) whose column lies in genome B, as sorted
    (row - first gene of A, column - first gene of B, score bits, perc bits, tr bits); cells of excluded local genes left out."""
    sel = np.asarray(block["second_seq_genome"]) == genome_b
    r = np.asarray(block["row"])[sel].astype(np.int64) - first_gene_a
    c = np.asarray(block["column"])[sel].astype(np.int64) - first_gene_b
    keep = ~np.isin(r, np.asarray(list(excluded_a), dtype=np.int64)) & ~np.isin(c, np.asarray(list(excluded_b), dtype=np.int64))
    r, c = r[keep], c[keep]
    o = np.lexsort((c, r))
    h = hashlib.sha256()
    h.update(r[o].astype("<i4").tobytes()); h.update(c[o].astype("<i4").tobytes())
    for f in ("scores", "percs", "tr_percs"):
        h.update(raw(np.asarray(block[f])[sel][keep][o]).astype("<u4").tobytes())
    return h.hexdigest(), int(len(r))


def wrapped_rank_set(letters: int, seed: int, genomes: int = 4, per_genome: int = 12, length: int = 70):
    """A small set over `letters` distinct ASCII letters — for k so large that B^k passes 2^64 while rank_init's overflow test
    (library.cpp:104-111: wrapped products, one multiplication ahead) does not notice, e.g. 22 letters at k = 15, 24 at k = 14:
    the ranks are then the polynomial mod 2^64 and fill all 64 bits although rank_init sees fewer.  -> pandelos_amd.synth.GeneSet"""
    from pandelos_amd.synth import GeneSet
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz", dtype=np.uint8)[:letters]
    fams = [alpha[rng.integers(0, letters, int(rng.integers(length // 2, length * 2)))] for _ in range(per_genome)]
    genes, gen, fam_of = [], [], []
    for g in range(genomes):
        for f, base in enumerate(fams):
            if rng.random() < 0.15:
                continue
            s = base.copy()
            m = rng.random(len(s)) < 0.05
            s[m] = alpha[rng.integers(0, letters, int(m.sum()))]
            genes.append(s); gen.append(g); fam_of.append(f)
    genes[0][:letters] = alpha                       # every letter occurs
    offsets = np.zeros(len(genes) + 1, np.uint64)
    np.cumsum([len(x) for x in genes], out=offsets[1:])
    return GeneSet(np.concatenate(genes).astype(np.uint8), offsets, np.asarray(gen, np.uint32), np.asarray(fam_of, np.int64))
