"""The host half after the native boundary (BBH filter + .net edge list).

CPU: the product's array implementation (pandelos_amd/pangenes.py) against the loop-by-loop restatement of the Java
(oracle/pangenes_host.py) and against the committed .net/.clus fixtures (tests/golden/net, produced with the reference's
netclu_ng.py in the build container).  GPU: the whole path .faa -> .net on the device equals the fixture byte for byte,
which makes the gene families (.clus) identical too.  NOTE: nothing of the reference pins the Java host (no JVM here, no
reference tests): these fixtures pin our reading of it, not the Java itself."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from pandelos_amd import pangenes as PH
from pandelos_amd.scores import Scores
from pandelos_amd.synth import make_gene_set
from tests import helpers as H

NET = H.GOLDEN / "net"
CASES = {
    "synth_5x60x80_k3": (dict(genomes=5, genes_per_genome=60, mean_len=80, sub_rate=0.08, seed=1), 3),
    "synth_8x300x200_k4_div25": (dict(genomes=8, genes_per_genome=300, mean_len=200, sub_rate=0.25, seed=11), 4),
    "synth_12x100x100_k3_near_identical": (dict(genomes=12, genes_per_genome=100, mean_len=100, sub_rate=0.02, seed=5), 3),
    "paralogs_6x80x120_k3": (dict(genomes=6, genes_per_genome=80, mean_len=120, sub_rate=0.12, seed=21, paralogs=0.35), 3),
    "paralogs_10x60x90_k3_div20": (dict(genomes=10, genes_per_genome=60, mean_len=90, sub_rate=0.20, seed=22, paralogs=0.5), 3),
}


class _OracleNative:
    """Scores blocks from the CPU oracle, shaped like PangeneNative (test scaffolding)."""
    def __init__(self, gs, k):
        from oracle import binding as ob
        self.o = ob.Oracle(gs.residues, gs.offsets, gs.genome_of, k)

    def generate_scores_part(self, g, multithread=False):
        d = self.o.scores(g)
        return Scores(**{f: d[f] for f in ("scoresCount",) + H.FIELDS})


@pytest.mark.parametrize("name", sorted(CASES))
def test_product_host_matches_oracle_host_and_fixture(name):
    from oracle import pangenes_host as oh
    shape, k = CASES[name]
    gs = make_gene_set(**shape)
    nat = _OracleNative(gs, k)
    got = PH.run(nat, gs.genomes)
    want = oh.build_net(lambda g: nat.o.scores(g), gs.genomes, gs.genes)
    assert got == want
    assert "".join(got) == (NET / f"{name}.net").read_text()


def test_java_hashmap_order_and_double_format():
    from oracle import pangenes_host as oh
    # 13 keys -> capacity 32 (13 > 0.75*16); bucket = key & 31 for small keys; collisions keep insertion order
    keys = [40, 8, 33, 1, 65, 2, 97, 3, 4, 5, 6, 7, 9]
    assert oh.java_hashmap_key_order(keys) == [33, 1, 65, 97, 2, 3, 4, 5, 6, 7, 40, 8, 9]
    lines = PH.net_lines(np.array(keys + [8], np.int64), np.array([100] * 13 + [100], np.int64),
                         np.array([0.5] * 13 + [0.25], np.float32))
    assert [int(l.split("\t")[0]) for l in lines] == [33, 1, 65, 97, 2, 3, 4, 5, 6, 7, 40, 8, 9]
    assert lines[11] == "8\t100\t0.5\n"                         # first insert per (src, dst) wins
    for v, s in [(0.849056601524353, "0.849056601524353"), (1.0, "1.0"), (0.00095, "9.5E-4"), (1e-4, "1.0E-4"),
                 (0.001, "0.001"), (float(np.float32(1 / 3)), "0.3333333432674408")]:
        assert oh.java_double_to_string(v) == s
        assert PH._java_double_str(np.array([v], np.float64))[0] == s


@pytest.mark.skipif(not Path("/root/reference/netclu_ng.py").exists(), reason="reference scripts not present")
def test_clus_from_product_net_equals_fixture(tmp_path):
    """pandelos.sh:76-79 on our .net: the reference's netclu_ng.py + grep/sed/sort/uniq."""
    name = "synth_5x60x80_k3"
    shape, k = CASES[name]
    gs = make_gene_set(**shape)
    faa, net = tmp_path / "in.faa", tmp_path / "out.net"
    gs.write_faa(faa)
    net.write_text("".join(PH.run(_OracleNative(gs, k), gs.genomes)))
    p = subprocess.run([sys.executable, "/root/reference/netclu_ng.py", str(faa), str(net)], capture_output=True, text=True, check=True)
    fams = sorted({l.replace("F{ ", "").replace("}", "").replace(" ;", "") for l in p.stdout.splitlines() if "F{ " in l})
    assert "".join(f + "\n" for f in fams) == (NET / f"{name}.clus").read_text()
    planted = {}
    for i, f in enumerate(gs.family_of):
        planted.setdefault(int(f), []).append(i)
    assert len(fams) == len(planted)                              # every planted family comes back as one gene family


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_device_pipeline_writes_the_fixture_net(name, tmp_path):
    shape, k = CASES[name]
    gs = make_gene_set(**shape)
    faa, net = tmp_path / "in.faa", tmp_path / "out.net"
    gs.write_faa(faa)
    assert PH.main(["-i", str(faa), "-k", str(k), "-o", str(net)]) == 0
    assert net.read_text() == (NET / f"{name}.net").read_text()


@pytest.mark.gpu
def test_native_host_binary_writes_the_fixture_net(tmp_path):
    """pandelos_amd/lib/pangenes: the C++ host over the C ABI (what replaces `java ... Pangenes`, pandelos.sh:73)."""
    from pandelos_amd import _lib
    name = "synth_8x300x200_k4_div25"
    shape, k = CASES[name]
    gs = make_gene_set(**shape)
    faa, net = tmp_path / "in.faa", tmp_path / "out.net"
    gs.write_faa(faa)
    p = subprocess.run([str(_lib.LIB_DIR / "pangenes"), "-i", str(faa), "-k", str(k), "-o", str(net)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert net.read_text() == (NET / f"{name}.net").read_text()
    assert "Total cost:" in p.stdout
    bad = subprocess.run([str(_lib.LIB_DIR / "pangenes"), "-i", str(faa)], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 1 and "Error while parsing cli arguments!" in bad.stdout      # Cli.java:83-87


@pytest.mark.gpu
@pytest.mark.parametrize("per_batch", [3, 1])
def test_native_host_in_genome_batches_writes_the_same_net(per_batch, tmp_path):
    """--genome-batch N: the genomes scored N at a time on one dictionary (for sets whose results do not fit the device
    together); a genome's task — its Scores block, the best-hit filter over it — does not depend on the batch it is in, so
    the .net is the fixture's, byte for byte."""
    from pandelos_amd import _lib
    name = "synth_8x300x200_k4_div25"
    shape, k = CASES[name]
    gs = make_gene_set(**shape)
    faa, net = tmp_path / "in.faa", tmp_path / "out.net"
    gs.write_faa(faa)
    p = subprocess.run([str(_lib.LIB_DIR / "pangenes"), "-i", str(faa), "-k", str(k), "-o", str(net), "--genome-batch", str(per_batch)],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert net.read_text() == (NET / f"{name}.net").read_text()
    assert "genome batches of" in p.stdout and p.stdout.count("Filtered count:") == shape["genomes"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["synth_5x60x80_k3", "paralogs_6x80x120_k3"])
def test_wrapper_script_goes_from_faa_to_the_fixture_clus(name, tmp_path):
    """pandelos_mi355x.sh <in.faa> <prefix>: k selection, the native stage on the GPU, de-clustering — without a PanDelos
    checkout (this repository's own calculate_k / netclu).  The .clus equals the one the reference's scripts made."""
    import os
    shape, k = CASES[name]
    gs = make_gene_set(**shape)
    faa = tmp_path / "in.faa"
    gs.write_faa(faa)
    env = {kk: v for kk, v in os.environ.items() if kk != "PANDELOS_PATH"}
    p = subprocess.run(["bash", str(H.GOLDEN.parents[1] / "pandelos_mi355x.sh"), str(faa), str(tmp_path / "out")], capture_output=True, text=True,
                       cwd=tmp_path, env=env, timeout=600)
    assert p.returncode == 0 and "Finish!" in p.stdout, p.stdout + p.stderr
    import re
    assert re.search(rf"k =\s+{k}\b", p.stdout), p.stdout
    assert (tmp_path / "out.clus").read_text() == (NET / f"{name}.clus").read_text()
    assert sorted(x.name for x in tmp_path.iterdir()) == ["in.faa", "out.clus"]          # temporaries are gone


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_device_filter_emits_the_hosts_edges_in_insertion_order(name):
    """pdl_compute_edges (K-bbh on the device) against the array form of the filter applied to the same Scores blocks, edge
    for edge and in order — and both against the loop-by-loop restatement of the Java through the .net fixture above."""
    from pandelos_amd.pangene_native import PangeneNative
    shape, k = CASES[name]
    gs = make_gene_set(**shape)
    nat = PangeneNative.from_arrays(k, gs.residues, gs.offsets, gs.genome_of)
    for g in range(gs.genomes):
        src, dst, sc = nat.generate_edges_part(g)
        want = PH.bbh_edges(nat.generate_scores_part(g))
        assert np.array_equal(src, want[0]) and np.array_equal(dst, want[1]) and np.array_equal(H.raw(sc), H.raw(want[2])), f"genome {g}"


@pytest.mark.gpu
def test_native_host_path_timer_counts_every_cell(tmp_path):
    """pandelos_amd/lib/host_path (what bench.py's host_path_native runs): pdl_scan_faa -> pdl_preprocess -> the G
    pdl_compute_scores calls from four host threads; the cells it received are the fixture's."""
    import json
    from pandelos_amd import _lib
    fx = dict(np.load(H.GOLDEN / "synth_5x60x80_k3.npz"))
    faa = tmp_path / "in.faa"
    faa.write_bytes(fx["faa"].tobytes())
    p = subprocess.run([str(_lib.LIB_DIR / "host_path"), str(faa), str(int(fx["k"])), "4", "2"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["cells"] == sum(len(fx[f"g{g}_scores"]) for g in range(int(fx["genomes"]))) and out["genomes"] == int(fx["genomes"])
