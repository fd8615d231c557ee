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


# Average amino-acid composition of proteins (per cent, UniProtKB/Swiss-Prot statistics, letters of ALPHABET order
# A C D E F G H I K L M N P Q R S T V W Y) — for the "protein-like" stand-ins
PROTEIN_COMPOSITION = np.array([8.25, 1.38, 5.46, 6.72, 3.86, 7.07, 2.27, 5.91, 5.80, 9.65,
                                2.41, 4.06, 4.74, 3.93, 5.53, 6.64, 5.36, 6.86, 1.10, 2.92])
LOW_COMPLEXITY_MOTIFS = (b"Q", b"GS", b"A", b"PE", b"KE", b"SR", b"N")


def make_gene_set(genomes: int, genes_per_genome: int, mean_len: int, sub_rate: float, seed: int,
                  presence: float = 0.85, paralogs: float = 0.0, protein_like: bool = False) -> GeneSet:
    """``paralogs``: fraction of a genome's families that get a second, independently mutated copy in that genome
    (in-genome duplicates make the gene network hold same-genome genes in one component: the case netclu splits).
    ``protein_like``: residues drawn with the average composition of real proteins instead of uniformly (substitutions
    too), a fifth of the families carry a low-complexity stretch (poly-Q, GS-linkers, ...: k-mers shared by unrelated
    genes, many times over) and genomes differ in size — the cases uniform random text is kindest to."""
    rng = np.random.Generator(np.random.PCG64(seed))
    fams = max(1, int(round(genes_per_genome / presence)))
    lens = np.clip(rng.normal(mean_len, mean_len / 4.0, fams).astype(np.int64), 20, None)
    anc_off = np.zeros(fams + 1, np.int64)
    np.cumsum(lens, out=anc_off[1:])
    ancestors = rng.integers(0, 20, int(anc_off[-1]), dtype=np.uint8)
    comp = None
    genome_presence = np.full(genomes, presence)
    if protein_like:
        comp = PROTEIN_COMPOSITION / PROTEIN_COMPOSITION.sum()
        ancestors = rng.choice(20, int(anc_off[-1]), p=comp).astype(np.uint8)
        letter = {int(c): i for i, c in enumerate(ALPHABET)}
        for f in np.nonzero(rng.random(fams) < 0.2)[0]:                    # low-complexity stretches
            motif = LOW_COMPLEXITY_MOTIFS[int(rng.integers(0, len(LOW_COMPLEXITY_MOTIFS)))]
            n = int(min(lens[f] - 2, rng.integers(8, 40)))
            at = int(anc_off[f] + rng.integers(1, lens[f] - n))
            ancestors[at:at + n] = np.array([letter[motif[i % len(motif)]] for i in range(n)], np.uint8)
        genome_presence = np.clip(presence * rng.uniform(0.55, 1.1, genomes), 0.05, 1.0)   # genomes of different sizes

    chunks, gene_lens, genome_ids, fam_ids = [], [], [], []
    for g in range(genomes):
        present = np.nonzero(rng.random(fams) < genome_presence[g])[0]
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
        if comp is None:
            seq = np.where(mut, rng.integers(0, 20, total, dtype=np.uint8), seq)
        else:
            seq = np.where(mut, rng.choice(20, total, p=comp).astype(np.uint8), seq)
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
    # not a BASELINE config: a set of more than 320 genomes the reference CAN process here, so that the join tier configs[4]
    # runs on (2048 slots + filter) is pinned by reference digests too (tests/golden/digests_baseline.json)
    "manygenomes_384x400x160": dict(genomes=384, genes_per_genome=400, mean_len=160, sub_rate=0.08, seed=3841),
}
