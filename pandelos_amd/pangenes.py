"""``Pangenes`` — host-side mirror of ``ig/infoasys/cli/pangenes/Pangenes.java`` (the reference's ``main``) over the
MI355X native path: read the ``.faa``, build the dictionary, score every genome, keep bidirectional best hits,
write the ``.net`` edge list that ``netclu_ng.py`` consumes.

    python -m pandelos_amd.pangenes -i in.faa -k K -o out.net [-j THREADS] [-c]

Flags are the reference's (``Cli.java:13-57``): ``-i/--input``, ``-k/--kvalue``, ``-o/--output`` required,
``-c/--complexity`` (cost model only), ``-j/--threads`` (accepted; the device pass is not threaded), ``-h``.

The best-hit filter follows ``Pangenes.java:98-176`` and the network ``PangeneNet.java:49-62,159-179`` in
single-thread order (``-j 1``; with more threads the reference's own ``.net`` line order is not deterministic).
It is array code (numpy), not a port of the Java loops; ``oracle/pangenes_host.py`` holds the loop-by-loop
restatement the tests compare it with.  No JVM exists in the build image and the reference has no tests for
its host code, so this part's parity is pinned only against that restatement.
"""
from __future__ import annotations

import argparse
import sys
from typing import List, Sequence

import numpy as np

from .scores import Scores

INT32_MAX = np.iinfo(np.int32).max


def bbh_edges(s: Scores):
    """Edges one genome task adds to the network, in insertion order -> (src, dst, score) arrays.

    Phase 1 (Pangenes.java:98-128): an inter-genome cell is a bidirectional best hit when its score equals both
    the row's best against that genome and the column's best against this genome; both directions are inserted.
    Phase 2 (:146-176): an intra-genome cell (row < column) is kept when it is the best of both genes inside the
    genome and not below the row's threshold = the smallest, over the row's best-hit genomes, of that genome's
    largest best-hit score below 1."""
    scores, row, col = s.scores, s.row.astype(np.int64), s.column.astype(np.int64)
    g1, g2 = s.first_seq_genome.astype(np.int64), s.second_seq_genome.astype(np.int64)
    mp = s.scoresMaxMappings.astype(np.int64)
    genomes = s.max_genome_score.shape[1] if s.max_genome_score.ndim == 2 else 0
    n_seq = len(mp)
    if s.scoresCount == 0:
        e = np.zeros(0, np.int64)
        return e, e, np.zeros(0, np.float32)
    row_best = s.max_genome_score[mp[row], g2]                      # max_genome_score[scoresMaxMappings[row]][second genome]
    inter = g1 != g2
    bbh = inter & (scores == row_best) & (scores == s.max_genome_score_col[col])
    # inter_max_score[sg]: largest best-hit score below 1 (float compare, :116-118)
    inter_max = np.zeros(genomes, np.float32)
    below1 = bbh & (scores.astype(np.float64) < 1.0)
    np.maximum.at(inter_max, g2[below1], scores[below1])
    thr = np.full(n_seq, np.inf, np.float32)                        # scoresRowThreshold (:146-155)
    np.minimum.at(thr, row[bbh], inter_max[g2[bbh]])
    col_in_genome = np.where(mp[col] == INT32_MAX, 0, mp[col])
    intra = ((row < col) & ~inter & (scores == row_best) & (mp[col] != INT32_MAX)
             & (scores == s.max_genome_score[col_in_genome, g2]) & (scores >= thr[row]))
    b = np.nonzero(bbh)[0]
    src = np.empty(2 * len(b), np.int64); dst = np.empty(2 * len(b), np.int64); sc = np.empty(2 * len(b), np.float32)
    src[0::2], dst[0::2], sc[0::2] = row[b], col[b], scores[b]      # addConnection(row, column) then (column, row), :103-104
    src[1::2], dst[1::2], sc[1::2] = col[b], row[b], scores[b]
    t = np.nonzero(intra)[0]
    return np.concatenate([src, row[t]]), np.concatenate([dst, col[t]]), np.concatenate([sc, scores[t]])


def _java_double_str(values: np.ndarray) -> List[str]:
    """Double.toString of float32 scores widened to double: shortest round-trip digits, decimal layout for
    1e-3 <= x < 1e7 and ``d.dddE-n`` otherwise."""
    out = []
    for v in values.astype(np.float64):
        r = repr(float(v))
        if 1e-3 <= v < 1e7:
            if "e" in r:                      # repr switches to exponent form below 1e-4 only; not reached in this range
                r = np.format_float_positional(v, unique=True, trim="0")
            out.append(r if "." in r else r + ".0")
        elif v == 0.0:
            out.append("0.0")
        else:
            m, e = np.format_float_scientific(v, unique=True, trim="0", exp_digits=1).split("e")
            if "." not in m:
                m += ".0"
            out.append(f"{m}E{int(e)}")
    return out


def net_lines(src: np.ndarray, dst: np.ndarray, score: np.ndarray) -> List[str]:
    """PangeneNet: first insert per (src, dst) wins (TreeSet keyed by dest, :49-62); undirected save (:167-175) walks the
    sources in java.util.HashMap order (buckets of the final power-of-two table, insertion order inside a bucket) and a
    source's edges by ascending dest, writing those with src <= dst."""
    if len(src) == 0:
        return []
    key = (src << 32) | dst
    _, first = np.unique(key, return_index=True)
    first.sort()
    src, dst, score = src[first], dst[first], score[first]           # one entry per (src, dst), insertion order kept
    usrc, src_first = np.unique(src, return_index=True)              # insertion order of the map keys
    cap = 16
    while len(usrc) > 0.75 * cap:
        cap *= 2
    h = usrc & 0xFFFFFFFF
    bucket = (h ^ (h >> 16)) & (cap - 1)
    pos = np.empty(int(usrc.max()) + 1, np.int64)
    pos[usrc[np.lexsort((src_first, bucket))]] = np.arange(len(usrc))
    keep = src <= dst
    src, dst, score = src[keep], dst[keep], score[keep]
    order = np.lexsort((dst, pos[src]))
    txt = _java_double_str(score[order])
    return [f"{a}\t{b}\t{t}\n" for a, b, t in zip(src[order].tolist(), dst[order].tolist(), txt)]


def run(native, nof_genomes: int) -> List[str]:
    """Pangenes.main after the dictionary exists (:54-66,185-194,222-227), one worker."""
    # the filter runs where the cells are: on the device (pdl_compute_edges); a `native` without that entry point (the
    # tests' stand-in around the CPU oracle) gets the array form above applied to its Scores blocks
    if hasattr(native, "generate_edges_part"):
        parts = [native.generate_edges_part(g) for g in range(nof_genomes)]
    else:
        parts = [bbh_edges(native.generate_scores_part(g, False)) for g in range(nof_genomes)]
    if not parts:
        return []
    return net_lines(np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]),
                     np.concatenate([p[2] for p in parts]))


def main(argv: Sequence[str] | None = None) -> int:
    ap = argparse.ArgumentParser(prog="PanDelos [OPTIONS]", add_help=False)
    ap.add_argument("-i", "--input", required=True, help="Input file (.faa) to process")
    ap.add_argument("-k", "--kvalue", required=True, type=int, help="Length of the kmers used by the algorithm")
    ap.add_argument("-c", "--complexity", action="store_true",
                    help="Compute the required number of operations without computing the network (fast)")
    ap.add_argument("-o", "--output", required=True, help="Output file for the network")
    ap.add_argument("-h", "--help", action="help", help="Print this help message")
    ap.add_argument("-j", "--threads", type=int, default=None,
                    help="Number of threads to use for the computation, defaults to # of processors")
    try:
        args = ap.parse_args(argv)
    except SystemExit as e:              # Cli.java:83-87: message, help, exit(1)
        if e.code not in (0, None):
            print("Error while parsing cli arguments!")
            ap.print_help()
            return 1
        return 0
    from . import _lib
    from .pangene_native import PangeneNative
    nativ = PangeneNative.open()
    try:
        ing = nativ.ingest_faa(args.input)                      # Pangenes.java:26-31 (PangeneIData.readFromFile), streamed to HBM
    except _lib.PdlError as e:                                  # the reference prints the stack trace and returns
        print(f"{type(e).__name__}: {e}", file=sys.stderr)
        return 0
    nativ.preprocess_ingested(args.kvalue, only_complexity=args.complexity)      # :39 (-c: PangeneNative.printComplexity, :33-36)
    print("------------\nCOMPUTATIONAL COSTS: ")
    print(f"Total cost: {nativ.cost.total_cost} lookups")
    print(f"Linear ratio: {nativ.cost.linear_ratio:g}\n------------\n")
    if args.complexity:
        return 0
    lines = run(nativ, ing["genomes"])
    print("----------")
    print(f"writing into {args.output}")
    with open(args.output, "w") as f:
        f.writelines(lines)
    return 0


if __name__ == "__main__":
    sys.exit(main())
