#!/usr/bin/env python3
"""Regenerate tests/golden/net/*.net and *.clus (build container only).

.net  = Scores of the reference's own library.cpp (oracle/_ref via the JVM-less harness) pushed through the
        loop-by-loop restatement of the Java host (oracle/pangenes_host.py; the Java itself cannot run here:
        no JVM — this half of the pipeline is NOT pinned by the reference)
.clus = the reference's netclu_ng.py run as a subprocess on (.faa, .net), followed by the text filter of
        pandelos.sh:79 (grep "F{ " | sed ... | sort | uniq)
Only inputs/outputs are stored.
"""
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import binding as ob, pangenes_host as oh            # noqa: E402
from pandelos_amd.synth import make_gene_set                      # noqa: E402

OUT = Path(__file__).resolve().parent / "net"
NETCLU = Path("/root/reference/netclu_ng.py")

CASES = {
    "synth_5x60x80_k3": (dict(genomes=5, genes_per_genome=60, mean_len=80, sub_rate=0.08, seed=1), 3),
    "synth_8x300x200_k4_div25": (dict(genomes=8, genes_per_genome=300, mean_len=200, sub_rate=0.25, seed=11), 4),
    "synth_12x100x100_k3_near_identical": (dict(genomes=12, genes_per_genome=100, mean_len=100, sub_rate=0.02, seed=5), 3),
    # in-genome duplicates: components that hold two genes of one genome -> netclu_ng.py's Girvan-Newman splitting
    "paralogs_6x80x120_k3": (dict(genomes=6, genes_per_genome=80, mean_len=120, sub_rate=0.12, seed=21, paralogs=0.35), 3),
    "paralogs_10x60x90_k3_div20": (dict(genomes=10, genes_per_genome=60, mean_len=90, sub_rate=0.20, seed=22, paralogs=0.5), 3),
}


def clus_of(faa: Path, net: Path) -> str:
    p = subprocess.run([sys.executable, str(NETCLU), str(faa), str(net)], capture_output=True, text=True, check=True)
    fams = []
    for line in p.stdout.splitlines():                            # pandelos.sh:79
        if "F{ " in line:
            fams.append(line.replace("F{ ", "").replace("}", "").replace(" ;", ""))
    return "".join(f + "\n" for f in sorted(set(fams)))


def main():
    OUT.mkdir(exist_ok=True)
    for name, (shape, k) in CASES.items():
        gs = make_gene_set(**shape)
        with tempfile.TemporaryDirectory() as td:
            faa = Path(td) / "in.faa"
            gs.write_faa(faa)
            ob.run_harness(ob.REF_SO, faa, k, dump=Path(td) / "ref.bin")
            ref = ob.read_dump(Path(td) / "ref.bin")
            lines = oh.build_net(lambda g: ref["per_genome"][g], ref["genomes"], ref["sequences"])
            net = OUT / f"{name}.net"
            net.write_text("".join(lines))
            clus = clus_of(faa, net)
            (OUT / f"{name}.clus").write_text(clus)
        fams = [l.split() for l in clus.splitlines()]
        print(name, "edges", len(lines), "families", len(fams), "genes", gs.genes,
              "largest", max(len(f) for f in fams), "singletons", sum(len(f) == 1 for f in fams))


if __name__ == "__main__":
    main()
