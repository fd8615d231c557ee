"""``Scores`` — host-side mirror of ``ig/infoasys/cli/pangenes/Scores.java:4-34`` (same field names)."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

FIELDS = ("scores", "percs", "tr_percs", "row", "column", "first_seq_genome", "second_seq_genome",
          "max_genome_score", "max_genome_score_col", "scoresMaxMappings")


@dataclass
class Scores:
    scoresCount: int
    scores: np.ndarray                # float32 [scoresCount]
    percs: np.ndarray                 # float32 [scoresCount]
    tr_percs: np.ndarray              # float32 [scoresCount]
    row: np.ndarray                   # int32   [scoresCount]
    column: np.ndarray                # int32   [scoresCount]
    first_seq_genome: np.ndarray      # int32   [scoresCount]
    second_seq_genome: np.ndarray     # int32   [scoresCount]
    max_genome_score: np.ndarray      # float32 [rows][genomes]
    max_genome_score_col: np.ndarray  # float32 [sequences]
    scoresMaxMappings: np.ndarray     # int32   [sequences]

    def as_dict(self) -> dict:
        d = {f: getattr(self, f) for f in FIELDS}
        d["scoresCount"] = self.scoresCount
        d["rows"] = self.max_genome_score.shape[0]
        return d
