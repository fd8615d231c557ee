"""De-clustering (.net -> .clus) against the reference's netclu_ng.py: the .clus fixtures under tests/golden/net were
produced by the reference's script (networkx 3.4.2) in the build container from the .net fixtures beside them
(make_golden_net.py, make_golden_baseline.py).  pandelos_amd/netclu.py must give the same gene families — including
which edge Girvan-Newman removes first when betweenness values tie."""
import gzip
import json

import pytest

from pandelos_amd import netclu
from pandelos_amd.synth import make_gene_set
from tests import helpers as H
from tests.test_host_net import CASES

NET = H.GOLDEN / "net"


@pytest.mark.parametrize("name", sorted(CASES))
def test_families_equal_the_reference_clus(name, tmp_path):
    shape, _ = CASES[name]
    faa = tmp_path / "in.faa"
    make_gene_set(**shape).write_faa(faa)
    names, genome_of = netclu.read_names(faa)
    fams, singles = netclu.families(names, genome_of, netclu.read_net(NET / f"{name}.net"))
    assert netclu.clus_text(names, fams, singles) == (NET / f"{name}.clus").read_text()


@pytest.mark.timeout(600)
def test_canonical_64_genome_set_equals_the_reference_clus(tmp_path):
    name = "mycoplasma64_standin"
    shape = json.loads((H.GOLDEN / "digests_baseline.json").read_text())[name]["shape"]
    faa, net = tmp_path / "in.faa", tmp_path / "in.net"
    make_gene_set(**shape).write_faa(faa)
    net.write_bytes(gzip.open(NET / f"{name}.net.gz", "rb").read())
    names, genome_of = netclu.read_names(faa)
    fams, singles = netclu.families(names, genome_of, netclu.read_net(net))
    assert netclu.clus_text(names, fams, singles) == gzip.open(NET / f"{name}.clus.gz", "rt").read()


def test_collision_is_split_by_edge_betweenness():
    """Two triangles of three genomes joined by one edge: genes 0 and 3 are both of genome A and not adjacent -> collision;
    the bridge has the highest betweenness and goes first."""
    genome_of = ["A", "B", "C", "A", "B", "C"]
    names = [f"g{i}" for i in range(6)]
    adj = {}
    for a, b in [(0, 1), (1, 2), (0, 2), (3, 4), (4, 5), (3, 5), (2, 3)]:
        adj.setdefault(a, {})[b] = None
        adj.setdefault(b, {})[a] = None
    assert netclu.max_collision(sorted(adj), adj, genome_of) == 1
    fams, singles = netclu.families(names, genome_of, adj)
    assert sorted(fams) == [[0, 1, 2], [3, 4, 5]] and singles == []
    assert netclu.clus_text(names, fams, singles + [0]) == "g0 \ng0 g1 g2\ng3 g4 g5\n"      # leftover genes keep the script's trailing blank
