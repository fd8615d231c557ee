"""k selection against the reference's own calculate_k.py: tests/golden/calculate_k/expected.json holds the "k = N" line the
reference script printed for every case (make_golden_calculate_k.py, build container).  The file reader keeps the
script's raw line parity (calculate_k.py:24-30: every odd LINE, blank lines counted), the array form works on
flattened residues."""
import json
import tempfile
from pathlib import Path

import pytest

from pandelos_amd.calculate_k import calculate_k, calculate_k_faa
from pandelos_amd.pangene_idata import PangeneIData
from pandelos_amd.synth import make_gene_set
from tests import helpers as H

DIR = H.GOLDEN / "calculate_k"
EXPECTED = json.loads((DIR / "expected.json").read_text())


@pytest.mark.parametrize("name", sorted(EXPECTED["text"]))
def test_file_reader_matches_the_reference_script(name):
    want = EXPECTED["text"][name]
    assert want is not None
    assert calculate_k_faa(DIR / f"{name}.faa") == want


@pytest.mark.parametrize("name", sorted(EXPECTED["synthetic"]))
def test_synthetic_sets_match_the_reference_script(name, tmp_path):
    case = EXPECTED["synthetic"][name]
    gs = make_gene_set(**case["shape"])
    faa = tmp_path / "in.faa"
    gs.write_faa(faa)
    assert calculate_k_faa(faa) == case["k"]
    assert calculate_k(gs.residues) == case["k"]              # what bench.py and the tests use on flattened residues


def test_line_parity_is_the_scripts_not_the_parsers():
    """With a leading blank line the reference script reads the HEADER lines (odd line numbers), the .faa parser of the
    Java host (PangeneIData.java:30-75) skips blank lines: the two disagree on what the residues are, and the k of the
    pipeline is the script's."""
    faa = DIR / "leading_blank_line.faa"
    res, _, _ = PangeneIData.read_from_file(faa).flatten()
    assert calculate_k_faa(faa) == EXPECTED["text"]["leading_blank_line"]
    assert bytes(res[:10]) == b"ACDEFGHIKL"                   # the parser still sees the sequences
