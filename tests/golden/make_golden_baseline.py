#!/usr/bin/env python3
"""Pin the BASELINE.json configs to the reference (build container only: needs /root/reference).

For every stand-in set of BASELINE.md §4 that the reference can process in this container's memory
(configs[0..3]; configs[4] needs ~60 GB for the reference's 16-byte records + per-pass copy + 24-byte
range triples and is left out) this script

  * writes the set as a .faa,
  * takes k from the reference's own calculate_k.py (run as a subprocess on that .faa),
  * runs the reference's library.cpp (oracle/_ref via the JVM-less harness, scoring threaded) and stores
    per-genome SHA-256 digests of every Scores array + scoresCount + "Total cost" / "Genome g cost"
    in digests_baseline.json,
  * for the canonical 64-genome set also pushes the Scores through the restatement of the Java host
    (oracle/pangenes_host.py) and the reference's netclu_ng.py: net/<name>.net.gz, net/<name>.clus.gz.

Only inputs' (shape, seed) and outputs are stored; nothing of the reference's source.
usage: make_golden_baseline.py [config ...]      (default: all four)
"""
import gzip
import hashlib
import json
import re
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import binding as ob, pangenes_host as oh          # noqa: E402
from pandelos_amd.synth import CONFIGS, make_gene_set           # noqa: E402

HERE = Path(__file__).resolve().parent
OUT = HERE / "digests_baseline.json"
REF_K = Path("/root/reference/calculate_k.py")
NETCLU = Path("/root/reference/netclu_ng.py")
FIELDS = ("scores", "percs", "tr_percs", "row", "column", "first_seq_genome", "second_seq_genome",
          "max_genome_score", "max_genome_score_col", "scoresMaxMappings")
DEFAULT = ["salmonella7_standin", "xanthomonas14_standin", "mycoplasma64_standin", "synthetic_128x4000x300"]
WITH_NET = {"mycoplasma64_standin"}


def raw(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def reference_k(faa: Path) -> int:
    p = subprocess.run([sys.executable, str(REF_K), str(faa)], capture_output=True, text=True, check=True)
    return int(re.search(r"^k =\s+(\d+)", p.stdout, re.M).group(1))


def genome_costs(info, genomes):
    gc = info["genome_cost"]
    if all(g in gc for g in range(genomes)) and sum(gc[g] for g in range(genomes)) == info["total_cost"]:
        return [gc[g] for g in range(genomes)]
    return None


def clus_of(faa: Path, net: Path) -> str:
    p = subprocess.run([sys.executable, str(NETCLU), str(faa), str(net)], capture_output=True, text=True, check=True)
    fams = [l.replace("F{ ", "").replace("}", "").replace(" ;", "") for l in p.stdout.splitlines() if "F{ " in l]   # pandelos.sh:79
    return "".join(f + "\n" for f in sorted(set(fams)))


def main():
    assert ob.have_reference(), "run `make -C oracle` in the build container first"
    names = sys.argv[1:] or DEFAULT
    digests = json.loads(OUT.read_text()) if OUT.exists() else {}
    for name in names:
        shape = CONFIGS[name]
        t0 = time.time()
        gs = make_gene_set(**shape)
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            faa = Path(td) / "in.faa"
            gs.write_faa(faa)
            k = reference_k(faa)
            info = ob.run_harness(ob.REF_SO, faa, k, threads=8, dump=Path(td) / "ref.bin", timeout=6 * 3600)
            ref = ob.read_dump(Path(td) / "ref.bin")
            digests[name] = {
                "shape": shape, "k": k, "sequences": ref["sequences"], "genomes": ref["genomes"],
                "total_cost": info["total_cost"],
                # (the reference prints "Genome g cost" from its scoring threads without a lock: on large sets lines can
                #  interleave; the list is stored only when every line came through whole and the sum equals "Total cost")
                "genome_cost": genome_costs(info, ref["genomes"]),
                "scoresCount": [int(d["scoresCount"]) for d in ref["per_genome"]],
                "sha256": [{f: hashlib.sha256(raw(d[f]).tobytes()).hexdigest() for f in FIELDS} for d in ref["per_genome"]],
                "reference_s": {"preprocess": info["preprocess_s"], "scores_8_threads": info["scores_s"]},
            }
            OUT.write_text(json.dumps(digests, indent=1))
            print(name, "genes", ref["sequences"], "k", k, "cost", info["total_cost"], "cells", sum(digests[name]["scoresCount"]),
                  f"ref {info['preprocess_s']:.1f}+{info['scores_s']:.1f} s, total {time.time() - t0:.0f} s", flush=True)
            if name in WITH_NET:
                lines = oh.build_net(lambda g: ref["per_genome"][g], ref["genomes"], ref["sequences"])
                net = Path(td) / "out.net"
                net.write_text("".join(lines))
                clus = clus_of(faa, net)
                (HERE / "net").mkdir(exist_ok=True)
                with gzip.GzipFile(HERE / "net" / f"{name}.net.gz", "wb", mtime=0) as f:
                    f.write(net.read_bytes())
                with gzip.GzipFile(HERE / "net" / f"{name}.clus.gz", "wb", mtime=0) as f:
                    f.write(clus.encode())
                fams = [l.split() for l in clus.splitlines()]
                print(name, "edges", len(lines), "families", len(fams), "largest", max(len(f) for f in fams),
                      "singletons", sum(len(f) == 1 for f in fams), f"total {time.time() - t0:.0f} s", flush=True)
            del ref


if __name__ == "__main__":
    main()
