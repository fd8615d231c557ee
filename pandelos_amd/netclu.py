"""De-clustering of the gene network into gene families — the stage after the hot path (``netclu_ng.py`` of the pipeline,
SURVEY.md §8f #3), so that a machine without the reference checkout (and without networkx) can go from ``.net`` to ``.clus``.

What the pipeline computes (``netclu_ng.py:41-175`` + the text filter of ``pandelos.sh:79``):

  * nodes / undirected edges from the ``.net`` lines (self edges dropped, ``:41-56``);
  * every connected component is a family unless it holds a *collision*: two genes of one genome that are not
    adjacent (``get_max_collision``, ``:75-92``);
  * a component with a collision is split by the first level of UNWEIGHTED Girvan–Newman (edges of highest betweenness
    removed one by one until the component falls apart, ``:97-111``), recursively, until no part holds a collision;
  * genes in no family are singletons; the families are printed ``F{ a ; b ; c}``, filtered, sorted and de-duplicated.

The result is a set of families, so only the partition matters — but which edge Girvan–Newman removes first on a tie of
betweenness values does change the partition, and the reference leaves that to networkx: ``max()`` over a dict in
``G.edges()`` order, on floats summed in Brandes' accumulation order over graphs that ``girvan_newman`` has rebuilt twice
(``G.copy().to_undirected()``).  This module follows those orders: adjacency as insertion-ordered dicts, the same rebuilds,
the same traversal and summation order, the same normalisation (``1 / (n (n-1))``) — and the order a subgraph VIEW is
walked in, which for a component much smaller than the network is the iteration order of the Python set that filters it
(so the sets are built here with the same insertion sequences).  It is pinned by the ``.clus`` fixtures under
``tests/golden/net`` (produced with the reference's script and networkx 3.4.2 on CPython 3.10 in the build container).

    python -m pandelos_amd.netclu in.faa in.net > out.clus
"""
from __future__ import annotations

import sys
from collections import deque
from typing import Dict, Iterable, List, Sequence, Tuple

Adj = Dict[int, Dict[int, None]]          # node -> neighbours, both in insertion order


def read_names(faa_path) -> Tuple[List[str], List[str]]:
    """Gene names and genome labels from the header lines, by RAW line parity like the script (netclu_ng.py:17-29)."""
    names, genomes = [], []
    with open(faa_path, "r") as f:
        for i, line in enumerate(f):
            if i % 2 == 0:
                cols = line.strip().split("\t")
                genomes.append(cols[0])
                names.append(cols[1])
    return names, genomes


def read_net(net_path) -> Adj:
    """netclu_ng.py:41-56: nodes in first-mention order, every edge entered in both directions (the second is a no-op
    for the order)."""
    adj: Adj = {}
    with open(net_path, "r") as f:
        for line in f:
            cols = line.strip().split("\t")
            if len(cols) < 2 or not cols[0]:
                continue
            a, b = int(cols[0]), int(cols[1])
            if a not in adj:
                adj[a] = {}
            if b not in adj and a != b:
                adj[b] = {}
            if a != b:
                adj[a][b] = None
                adj[b][a] = None
    return adj


def view_filter(nodes: Iterable[int], member) -> set:
    """The node filter of ``G.subgraph(nodes)``: a NEW set filled in the iteration order of ``nodes`` (networkx:
    ``show_nodes(self.nbunch_iter(nodes))``).  Its own iteration order (CPython's, a function of the insertion sequence)
    is what a small subgraph view is walked in, so the same object has to be built the same way here."""
    return set(n for n in nodes if member(n))


def view_adjacency(root: Adj, filt: set) -> Adj:
    """Nodes and neighbours of the subgraph view in the order networkx iterates them (``FilterAtlas.__iter__``): through the
    FILTER SET when it is less than half as long as the underlying dict, through the dict otherwise."""
    order = [n for n in filt if n in root] if 2 * len(filt) < len(root) else [n for n in root if n in filt]
    out: Adj = {}
    for u in order:
        nb = root[u]
        out[u] = {v: None for v in filt if v in nb} if 2 * len(filt) < len(nb) else {v: None for v in nb if v in filt}
    return out


def rebuilt(adj: Adj) -> Adj:
    """``Graph.copy()`` / ``to_undirected()``: nodes in order, then every (u, v) of the adjacency in order re-inserted —
    a node's neighbours end up in the order the edges touching it were first added."""
    out: Adj = {u: {} for u in adj}
    for u, nbrs in adj.items():
        for v in nbrs:
            out[u][v] = None
            out[v][u] = None
    return out


def connected_components(adj: Adj) -> List[set]:
    seen, comps = set(), []
    for s in adj:
        if s in seen:
            continue
        comp, queue = {s}, deque([s])
        while queue:
            v = queue.popleft()
            for w in adj[v]:
                if w not in comp:
                    comp.add(w)
                    queue.append(w)
        seen |= comp
        comps.append(comp)
    return comps


def edge_list(adj: Adj) -> List[Tuple[int, int]]:
    """``G.edges()`` of an undirected graph: every edge once, from the endpoint that comes first in node order."""
    seen, out = set(), []
    for u, nbrs in adj.items():
        for v in nbrs:
            if v not in seen:
                out.append((u, v))
        seen.add(u)
    return out


def most_central_edge(adj: Adj) -> Tuple[int, int]:
    """Edge of highest shortest-path betweenness (Brandes, unweighted), first one in ``G.edges()`` order on a tie."""
    bet: Dict[Tuple[int, int], float] = {e: 0.0 for e in edge_list(adj)}
    for s in adj:
        # single-source shortest paths by BFS
        order: List[int] = []
        pred: Dict[int, List[int]] = {v: [] for v in adj}
        sigma = dict.fromkeys(adj, 0.0)
        dist = {s: 0}
        sigma[s] = 1.0
        queue = deque([s])
        while queue:
            v = queue.popleft()
            order.append(v)
            dv, sv = dist[v], sigma[v]
            for w in adj[v]:
                if w not in dist:
                    queue.append(w)
                    dist[w] = dv + 1
                if dist[w] == dv + 1:
                    sigma[w] += sv
                    pred[w].append(v)
        # accumulation, farthest first
        delta = dict.fromkeys(order, 0)
        while order:
            w = order.pop()
            coeff = (1 + delta[w]) / sigma[w]
            for v in pred[w]:
                c = sigma[v] * coeff
                if (v, w) in bet:
                    bet[(v, w)] += c
                else:
                    bet[(w, v)] += c
                delta[v] += c
    n = len(adj)
    if n > 1:
        scale = 1 / (n * (n - 1))
        for e in bet:
            bet[e] *= scale
    best, best_val = None, None
    for e, val in bet.items():
        if best is None or val > best_val:
            best, best_val = e, val
    return best


def girvan_newman_first_level(adj: Adj) -> List[set]:
    """First tuple of ``girvan_newman(G)``: remove most central edges until the number of components grows."""
    g = rebuilt(rebuilt(adj))                                 # G.copy().to_undirected()
    if not any(g[u] for u in g):
        return connected_components(g)
    start = len(connected_components(g))
    while True:
        u, v = most_central_edge(g)
        del g[u][v]
        del g[v][u]
        comps = connected_components(g)
        if len(comps) > start:
            return comps


def max_collision(nodes: Sequence[int], adj: Adj, genome_of: Sequence[str]) -> int:
    """netclu_ng.py:75-92: over the genes of one genome inside the set, the most same-genome genes one of them is NOT
    adjacent to."""
    by_genome: Dict[str, List[int]] = {}
    for s in nodes:
        by_genome.setdefault(genome_of[s], []).append(s)
    worst = 0
    for members in by_genome.values():
        if len(members) < 2:
            continue
        for s1 in members:
            k = sum(1 for s2 in members if s2 != s1 and s2 not in adj[s1])
            worst = max(worst, k)
    return worst


def split_until_clean(filt: set, root: Adj, genome_of: Sequence[str]) -> List[List[int]]:
    """netclu_ng.py:97-111 (``split_until_max_k``) on the subgraph view with node filter ``filt``."""
    out: List[List[int]] = []
    for com in (sorted(c) for c in girvan_newman_first_level(view_adjacency(root, filt))):
        if max_collision(com, root, genome_of) > 0:
            out += split_until_clean(view_filter(com, filt.__contains__), root, genome_of)      # a view of a view: filter on the root graph
        else:
            out.append(com)
    return out


def families(names: Sequence[str], genome_of: Sequence[str], adj: Adj) -> Tuple[List[List[int]], List[int]]:
    """-> (families that come from network components, genes that are in no component)."""
    fams: List[List[int]] = []
    placed = set()
    for comp in connected_components(adj):                    # (a set filled in breadth-first order, like networkx's)
        if max_collision(comp, adj, genome_of) > 0:
            parts = split_until_clean(view_filter(comp, adj.__contains__), adj, genome_of)
        else:
            parts = [sorted(comp)]
        for part in parts:
            fams.append(sorted(part))
            placed.update(part)
    return fams, [g for g in range(len(names)) if g not in placed]


def clus_text(names: Sequence[str], fams: Sequence[Sequence[int]], singletons: Sequence[int]) -> str:
    """What ``grep "F{ " | sed s/F{\\ //g | sed s/}//g | sed s/\\ \\;//g | sort | uniq`` leaves (pandelos.sh:79): one family per
    line, names joined by a blank; the script prints leftover genes as ``F{ name }``, which keeps a trailing blank.  Lines
    are sorted by code point (``sort`` under the C locale) and de-duplicated."""
    lines = {" ".join(names[g] for g in fam) for fam in fams}
    lines |= {names[g] + " " for g in singletons}
    return "".join(l + "\n" for l in sorted(lines))


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 2:
        print("usage: python -m pandelos_amd.netclu dataset.faa network.net > families.clus", file=sys.stderr)
        return 1
    names, genome_of = read_names(argv[0])
    fams, singles = families(names, genome_of, read_net(argv[1]))
    sys.stdout.write(clus_text(names, fams, singles))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
