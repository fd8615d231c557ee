"""k selection, as the pipeline's ``calculate_k.py`` computes it (calculate_k.py:9-30).

k = floor( log_A(total residues) / H_A ), A = alphabet size, H_A = Shannon entropy of the
residue distribution in base A.  The reference script takes every odd line of the file
(raw line parity, no blank-line skipping); ``calculate_k_faa`` keeps that, ``calculate_k``
works on already-flattened residues.
"""
from __future__ import annotations

import math

import numpy as np


def _k_from_counts(counts) -> int:
    counts = [int(c) for c in counts if c > 0]
    total = sum(counts)
    a = len(counts)
    if a < 2 or total == 0:
        raise ValueError("calculate_k needs at least two distinct residues")
    h = 0.0
    for c in counts:
        h += -math.log(c / total, a) * (c / total)
    return math.floor(math.log(total, a) / h)


def calculate_k(residues: np.ndarray) -> int:
    return _k_from_counts(np.bincount(np.asarray(residues, dtype=np.uint8), minlength=256))


def calculate_k_faa(path) -> int:
    # dict preserves first-seen order like the reference's `alphabet` dict, so the float
    # summation order of the entropy is the same (calculate_k.py:24-26)
    alphabet: dict = {}
    with open(path, "r") as f:
        for i, line in enumerate(f):
            if i % 2 != 0:
                for s in line.strip():
                    alphabet[s] = alphabet.get(s, 0) + 1
    return _k_from_counts(alphabet.values())


if __name__ == "__main__":      # the line pandelos.sh greps for (pandelos.sh:67-68)
    import sys
    print("k = ", calculate_k_faa(sys.argv[1]))
