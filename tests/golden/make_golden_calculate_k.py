#!/usr/bin/env python3
"""Expected k values FROM THE REFERENCE's calculate_k.py (build container only: runs /root/reference/calculate_k.py as a
subprocess on every case and stores its "k = N" line).  Small inputs are stored as .faa text beside the result; the
synthetic ones are regenerated from (shape, seed).  tests/test_calculate_k.py compares pandelos_amd.calculate_k with these.

Cases include what the script's raw line parity does (calculate_k.py:24-30 takes every odd LINE, blank ones included):
a leading blank line makes it read the header lines instead of the sequences."""
import json
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from pandelos_amd.synth import CONFIGS, make_gene_set           # noqa: E402
from tests.golden.make_golden import README4, LOWC               # noqa: E402

HERE = Path(__file__).resolve().parent / "calculate_k"
REF = Path("/root/reference/calculate_k.py")

TEXT = {
    "readme4": README4,
    "low_complexity": LOWC,
    "leading_blank_line": b"\nG0\ta\tp\nACDEFGHIKLACDEFGHIKL\nG1\tb\tp\nACDEFGHIKMACDEFGHIKM\n",
    "blank_between_records": b"G0\ta\tp\nACDEFGHIKL\n\nG1\tb\tp\nACDEFGHIKM\nG0\tc\tp\nCDEFGHIKLA\n",
    "crlf_and_spaces": b"G0\ta\tp\r\n  ACDEFGHIKLMNPQ  \r\nG1\tb\tp\r\nACDEFGHIKLMNPQRSTVWY\r\n",
    "two_letters": b"x\tg\tp\nAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAC\ny\tg\tp\nACACACACACACACACACACACACACACACAC\n",
}
SYNTH = {
    "synth_5x60x80": dict(genomes=5, genes_per_genome=60, mean_len=80, sub_rate=0.08, seed=1),
    "synth_16x1000x300": dict(genomes=16, genes_per_genome=1000, mean_len=300, sub_rate=0.08, seed=7),
    "mycoplasma64_standin": CONFIGS["mycoplasma64_standin"],
    "xanthomonas14_standin": CONFIGS["xanthomonas14_standin"],
}


def reference_k(path):
    p = subprocess.run([sys.executable, str(REF), str(path)], capture_output=True, text=True)
    m = re.search(r"^k =\s+(-?\d+)", p.stdout, re.M)
    return int(m.group(1)) if (p.returncode == 0 and m) else None      # None: the script raised (e.g. one-letter alphabet)


def main():
    HERE.mkdir(exist_ok=True)
    out = {"text": {}, "synthetic": {}}
    for name, faa in TEXT.items():
        (HERE / f"{name}.faa").write_bytes(faa)
        out["text"][name] = reference_k(HERE / f"{name}.faa")
    for name, shape in SYNTH.items():
        with tempfile.TemporaryDirectory() as td:
            f = Path(td) / "in.faa"
            make_gene_set(**shape).write_faa(f)
            out["synthetic"][name] = {"shape": shape, "k": reference_k(f)}
    (HERE / "expected.json").write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
