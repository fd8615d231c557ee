"""CPU ORACLE of the Java host around the native boundary (test infrastructure — never imported by the product).

Literal, loop-by-loop restatement of
  ig/infoasys/cli/pangenes/Pangenes.java:60-183   (per-genome task: BBH filter, both phases)
  ig/infoasys/cli/pangenes/PangeneNet.java:17-62,159-179  (Edge / addConnection / saveToFile)
in single-thread order (``-j 1``: genomes ascending; the reference itself is order-nondeterministic with more
threads).  Parity status: **unpinned by the reference** — no JVM exists in the build image and the reference
holds no test or fixture for its host code; what is pinned is the Scores input (golden
vectors of the reference's native library).  java.util.HashMap's iteration order and Double.toString are
emulated from their documented behaviour (see the functions below).
"""
from __future__ import annotations

import math
import struct
from typing import Dict, List

import numpy as np


# ---- java.lang.Double.toString of a float widened to double --------------------------------------------
def java_double_to_string(x: float) -> str:
    """Shortest-repr digits (what JDK >= 19 prints; older JDKs may print one more digit in rare cases) laid out
    like Double.toString: plain decimal for 1e-3 <= |x| < 1e7, computerized scientific notation otherwise."""
    if x != x:
        return "NaN"
    if x in (math.inf, -math.inf):
        return "Infinity" if x > 0 else "-Infinity"
    if x == 0:
        return "-0.0" if math.copysign(1.0, x) < 0 else "0.0"
    sign = "-" if x < 0 else ""
    ax = abs(x)
    digits = repr(ax)                             # shortest round-trip digits
    # normalise repr to (digit string, decimal exponent)
    if "e" in digits:
        m, e = digits.split("e")
        e10 = int(e)
    else:
        m, e10 = digits, 0
    if "." in m:
        ip, fp = m.split(".")
    else:
        ip, fp = m, ""
    ds = (ip + fp).lstrip("0")
    # decimal exponent of the first significant digit
    first = len(ip.lstrip("0")) if ip.strip("0") else -(len(fp) - len(fp.lstrip("0")))
    e10 = e10 + (first - 1 if ip.strip("0") else first - 1)
    ds = ds.rstrip("0") or "0"
    if 1e-3 <= ax < 1e7:
        if e10 >= 0:
            ip_d = ds[: e10 + 1].ljust(e10 + 1, "0")
            fp_d = ds[e10 + 1:] or "0"
        else:
            ip_d = "0"
            fp_d = "0" * (-e10 - 1) + ds
        return f"{sign}{ip_d}.{fp_d}"
    return f"{sign}{ds[0]}.{ds[1:] or '0'}E{e10}"


# ---- java.util.HashMap<Integer, ...> iteration order -------------------------------------------------
def java_hashmap_key_order(keys_in_insertion_order: List[int]) -> List[int]:
    """Iteration order of a HashMap<Integer,V> after inserting the given distinct keys in this order (no removals).
    Table: power of two, 16 initially, doubled while size > 0.75 * capacity; bucket = (h ^ (h >>> 16)) & (cap - 1)
    with h = Integer.hashCode() = the value; buckets are walked in index order, a bucket's nodes in insertion order
    (resize splits preserve relative order; tree bins keep the next-pointer order)."""
    n = len(keys_in_insertion_order)
    cap = 16
    while n > 0.75 * cap:
        cap *= 2
    def bucket(k: int) -> int:
        h = k & 0xFFFFFFFF
        return (h ^ (h >> 16)) & (cap - 1)
    return [k for _, k in sorted(((bucket(k), i), k) for i, k in enumerate(keys_in_insertion_order))]


class PangeneNet:
    """PangeneNet.java:38-62: adjacency HashMap<Integer, TreeSet<Edge>>; an Edge compares by node only, so the first
    insert per (src, dest) wins."""

    def __init__(self):
        self.adj: Dict[int, Dict[int, float]] = {}      # insertion-ordered dict = key insertion order

    def add_connection(self, src: int, dest: int, score: float) -> None:
        edges = self.adj.get(src)
        if edges is None:
            self.adj[src] = {dest: score}
        elif dest not in edges:
            edges[dest] = score

    def lines(self) -> List[str]:
        """saveToFile(file, directed=false), PangeneNet.java:167-175"""
        out = []
        for src in java_hashmap_key_order(list(self.adj.keys())):
            edges = self.adj[src]
            for dest in sorted(edges):                   # TreeSet order
                if src <= dest:
                    out.append(f"{src}\t{dest}\t{java_double_to_string(float(edges[dest]))}\n")
        return out


def process_genome(pnet: PangeneNet, s: dict, nof_genomes: int, sequences_count: int) -> None:
    """The body of the pool task, Pangenes.java:64-176, for one Scores block `s` (dict of the Scores fields)."""
    f32 = np.float32
    n = int(s["scoresCount"])
    scores, percs, tr_percs = s["scores"], s["percs"], s["tr_percs"]
    row, column = s["row"], s["column"]
    g1, g2 = s["first_seq_genome"], s["second_seq_genome"]
    mgs, mgsc, mp = s["max_genome_score"], s["max_genome_score_col"], s["scoresMaxMappings"]
    inter_max_score = [f32(0.0)] * nof_genomes
    should_add = [False] * n
    for i in range(n):                                                   # :98-128
        if g1[i] != g2[i]:
            if scores[i] == mgs[mp[row[i]]][g2[i]] and scores[i] == mgsc[column[i]]:
                pnet.add_connection(int(row[i]), int(column[i]), float(scores[i]))
                pnet.add_connection(int(column[i]), int(row[i]), float(scores[i]))
                should_add[i] = True
                sg = int(g2[i])
                score = scores[i]
                if float(score) < 1.0 and score > inter_max_score[sg]:   # :116-118
                    inter_max_score[sg] = score
    thr = {}                                                             # scoresRowThreshold, default +inf  :146-155
    for i in range(n):
        if should_add[i]:
            r = int(row[i])
            sg = int(g2[i])
            thr[r] = min(thr.get(r, f32(np.inf)), inter_max_score[sg])
    for i in range(n):                                                   # :164-176
        if (row[i] < column[i] and g1[i] == g2[i]
                and scores[i] == mgs[mp[row[i]]][g2[i]]
                and scores[i] == mgs[mp[column[i]]][g2[i]]
                and scores[i] >= thr.get(int(row[i]), f32(np.inf))):
            pnet.add_connection(int(row[i]), int(column[i]), float(scores[i]))


def build_net(get_scores, nof_genomes: int, sequences_count: int) -> List[str]:
    """Pangenes.java:60-66,185-194,222-227 with one worker: genomes in ascending order, then saveToFile."""
    pnet = PangeneNet()
    for g in range(nof_genomes):
        process_genome(pnet, get_scores(g), nof_genomes, sequences_count)
    return pnet.lines()
