#!/usr/bin/env python3
"""Pin sampled genome pairs of BASELINE configs[4] (512 x 5000 x 350) to the reference (build container only).

The reference cannot process the whole set in this container (~60 GB of 16-byte records, per-pass copies and 24-byte range
triples in 64 GB without swap).  A cell's values depend on its two genes alone, so the reference's library.cpp (oracle/_ref
through the JVM-less harness) is run on a few 3-4-genome subsets of the set with k forced to the full set's k, and for every
ordered genome pair (A, B) of a subset the SHA-256 of the sorted cells (local gene indices, float bits) is stored.  The GPU
test scores the full set and must reproduce each digest from its own cells of those genome pairs
(tests/test_gpu_config5.py).  Left out on both sides: cells of the genes that hold the largest-rank k-mer of the subset or of
the full set — the only cells the reference's fold of the globally last record (library.cpp:300-306) can change.

Only (shape, seed), genome ids and digests are stored.   usage: make_golden_config5_pairs.py
"""
import json
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import binding as ob                                  # noqa: E402
from pandelos_amd.calculate_k import calculate_k                  # noqa: E402
from pandelos_amd.synth import CONFIGS, GeneSet, make_gene_set    # noqa: E402
from tests import helpers as H                                    # noqa: E402

NAME = "synthetic_512x5000x350"
SUBSETS = [[0, 1, 255, 511], [5, 130, 383, 384], [2, 256, 510]]
OUT = Path(__file__).resolve().parent / "config5_pairs.json"


def subset_of(gs, genomes):
    first = np.searchsorted(gs.genome_of, np.arange(gs.genomes + 1))          # genes of a genome are consecutive ids
    ids = np.concatenate([np.arange(first[g], first[g + 1]) for g in genomes])
    lens = (gs.offsets[ids + 1] - gs.offsets[ids]).astype(np.int64)
    res = np.concatenate([gs.residues[int(gs.offsets[first[g]]):int(gs.offsets[first[g + 1]])] for g in genomes])
    off = np.zeros(len(ids) + 1, np.uint64)
    np.cumsum(lens, out=off[1:])
    gen = np.concatenate([np.full(first[g + 1] - first[g], i, np.uint32) for i, g in enumerate(genomes)])
    return GeneSet(res, off, gen, gs.family_of[ids]), ids, first


def main():
    assert ob.have_reference(), "run `make -C oracle` in the build container first"
    t0 = time.time()
    gs = make_gene_set(**CONFIGS[NAME])
    k = calculate_k(gs.residues)
    print(f"{NAME}: {gs.genes} genes, {len(gs.residues)} residues, k = {k}  ({time.time() - t0:.0f} s)", flush=True)
    full_holders = H.genes_holding_the_largest_kmer(gs.residues, gs.offsets, k)
    print("genes holding the largest k-mer of the full set:", full_holders, f"({time.time() - t0:.0f} s)", flush=True)
    out = {"config": NAME, "shape": CONFIGS[NAME], "k": int(k), "sequences": gs.genes, "genomes": gs.genomes,
           "largest_kmer_genes_full_set": full_holders, "subsets": []}
    for genomes in SUBSETS:
        sub, ids, first = subset_of(gs, genomes)
        holders = set(full_holders) & set(int(x) for x in ids)
        holders |= set(int(ids[x]) for x in H.genes_holding_the_largest_kmer(sub.residues, sub.offsets, k))
        # excluded genes as (genome, index inside the genome)
        excl = {g: sorted(int(x - first[g]) for x in holders if first[g] <= x < first[g + 1]) for g in genomes}
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            faa = Path(td) / "sub.faa"
            sub.write_faa(faa)
            info = ob.run_harness(ob.REF_SO, faa, k, threads=4, dump=Path(td) / "ref.bin", timeout=3600)
            ref = ob.read_dump(Path(td) / "ref.bin")
        sfirst = np.searchsorted(sub.genome_of, np.arange(len(genomes) + 1))
        pairs = []
        for ia, a in enumerate(genomes):
            blk = ref["per_genome"][ia]
            for ib, b in enumerate(genomes):
                dig, cnt = H.pair_cells_digest(blk, int(sfirst[ia]), int(sfirst[ib]), ib, excl[a], excl[b])
                pairs.append({"row_genome": a, "col_genome": b, "cells": cnt, "sha256": dig})
        out["subsets"].append({"genomes": genomes, "excluded_local_genes": {str(g): v for g, v in excl.items()},
                               "reference_total_cost": info["total_cost"], "reference_cells": int(sum(d["scoresCount"] for d in ref["per_genome"])),
                               "pairs": pairs})
        print(genomes, "ref cost", info["total_cost"], "cells", out["subsets"][-1]["reference_cells"], "excluded", excl,
              f"({time.time() - t0:.0f} s)", flush=True)
        OUT.write_text(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
