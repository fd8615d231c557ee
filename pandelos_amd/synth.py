"""Seeded synthetic amino-acid gene sets (benchmark / test inputs).

Shape follows SURVEY.md §8(d): a 20-letter uniform alphabet, ``F`` ancestral gene families
whose lengths are N(mean, mean/4) clipped at 20, each genome carrying each family with
probability 0.85 and a per-residue substitution rate on every copy.  The generator is this
project's own (numpy PCG64), so a (seed, shape) pair names one exact byte sequence.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

ALPHABET = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)


@dataclass
class GeneSet:
    residues: np.ndarray    # uint8 [R]   concatenated gene sequences (ASCII letters)
    offsets: np.ndarray     # uint64 [N+1]
    genome_of: np.ndarray   # uint32 [N]  dense genome ids, first-seen order
    family_of: np.ndarray   # int64 [N]   planted family (ground truth for sanity checks)

    @property
    def genes(self) -> int:
        return len(self.genome_of)

    @property
    def genomes(self) -> int:
        return int(self.genome_of.max()) + 1 if len(self.genome_of) else 0

    def write_faa(self, path) -> None:
        """Two-line records ``genome<TAB>gene<TAB>product`` / sequence (README.md:24-36 format)."""
        res = self.residues.tobytes()
        with open(path, "wb") as f:
            for i in range(self.genes):
                g = int(self.genome_of[i])
                fam = int(self.family_of[i])
                f.write(b"G%d\tg%d_%d@G%d:1\tprod %d\n" % (g, g, i, g, fam))
                f.write(res[int(self.offsets[i]):int(self.offsets[i + 1])])
                f.write(b"\n")


def make_gene_set(genomes: int, genes_per_genome: int, mean_len: int, sub_rate: float, seed: int,
                  presence: float = 0.85, paralogs: float = 0.0) -> GeneSet:
    """``paralogs``: fraction of a genome's families that get a second, independently mutated copy in that genome
    (in-genome duplicates make the gene network hold same-genome genes in one component: the case netclu splits)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    fams = max(1, int(round(genes_per_genome / presence)))
    lens = np.clip(rng.normal(mean_len, mean_len / 4.0, fams).astype(np.int64), 20, None)
    anc_off = np.zeros(fams + 1, np.int64)
    np.cumsum(lens, out=anc_off[1:])
    ancestors = rng.integers(0, 20, int(anc_off[-1]), dtype=np.uint8)

    chunks, gene_lens, genome_ids, fam_ids = [], [], [], []
    for g in range(genomes):
        present = np.nonzero(rng.random(fams) < presence)[0]
        if len(present) == 0:
            present = np.array([g % fams])
        if paralogs > 0.0:                      # (no random draw without paralogs: earlier (shape, seed) pairs keep their bytes)
            present = np.concatenate([present, present[rng.random(len(present)) < paralogs]])
        ln = lens[present]
        total = int(ln.sum())
        excl = np.zeros(len(ln), np.int64)
        np.cumsum(ln[:-1], out=excl[1:])
        src = np.arange(total, dtype=np.int64) - np.repeat(excl, ln) + np.repeat(anc_off[present], ln)
        seq = ancestors[src]
        mut = rng.random(total) < sub_rate
        seq = np.where(mut, rng.integers(0, 20, total, dtype=np.uint8), seq)
        chunks.append(ALPHABET[seq])
        gene_lens.append(ln)
        genome_ids.append(np.full(len(ln), g, np.uint32))
        fam_ids.append(present)
    residues = np.concatenate(chunks)
    gl = np.concatenate(gene_lens)
    offsets = np.zeros(len(gl) + 1, np.uint64)
    np.cumsum(gl, out=offsets[1:])
    return GeneSet(residues, offsets, np.concatenate(genome_ids), np.concatenate(fam_ids))


# BASELINE.md §4 stand-ins (real .faa files are not available offline)
CONFIGS = {
    "salmonella7_standin": dict(genomes=7, genes_per_genome=4500, mean_len=310, sub_rate=0.02, seed=701),
    "xanthomonas14_standin": dict(genomes=14, genes_per_genome=4300, mean_len=340, sub_rate=0.10, seed=1401),
    "mycoplasma64_standin": dict(genomes=64, genes_per_genome=750, mean_len=370, sub_rate=0.25, seed=6401),
    "synthetic_128x4000x300": dict(genomes=128, genes_per_genome=4000, mean_len=300, sub_rate=0.08, seed=4001),
    "synthetic_512x5000x350": dict(genomes=512, genes_per_genome=5000, mean_len=350, sub_rate=0.08, seed=5001),
}
