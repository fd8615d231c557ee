"""K-ingest: the library's `.faa` parser against the reader that mirrors PangeneIData.readFromFile
(PangeneIData.java:30-75) and against the k values the reference's calculate_k.py printed (tests/golden/calculate_k).

CPU: pdl_scan_faa (the parser without a device).  GPU: pdl_ingest_faa streams the same bytes into HBM through its pinned
staging buffers; what arrives is read back and compared, and the dictionary + Scores built from it reproduce the reference
fixtures / digests."""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import pytest

from pandelos_amd import _lib
from pandelos_amd.calculate_k import calculate_k_faa
from pandelos_amd.pangene_idata import PangeneIData
from pandelos_amd.synth import CONFIGS, make_gene_set

from .helpers import DIGESTS, FIELDS, GOLDEN, SMALL_CASES, assert_scores_equal_fixture, assert_scores_match_digest, load_large

KDIR = GOLDEN / "calculate_k"
EXPECTED = json.loads((KDIR / "expected.json").read_text())

EDGE_TEXTS = {
    "no_final_newline": b"A\ta1\tp\nMKV\nB\tb1\tp\nMKL",
    "lone_cr_terminators": b"A\ta1\tp\rMKVA\rB\tb1\tp\rMKLA\r",
    "mixed_terminators": b"A\ta1\tp\r\nMKVA\nA\ta2\tp\rMKIA\r\n\r\n  \t \nB\tb1\tp q\n  MKLA \t\n",
    "controls_trimmed": b"\x01A\ta1\tp\x1f\n\x00MKVA\x02\nB\tb1\tp\nMK LA\n",
    "interleaved_genomes": b"B\tb1\tp\nMKV\nA\ta1\tp\nMKL\nB\tb2\tp\nMKI\nC\tc1\tp\nMRV\nA\ta2\tp\nMRL\n",
    "header_without_sequence_at_eof": b"A\ta1\tp\nMKV\nB\tb1\tp\n",
    "empty_file": b"",
    "only_blank_lines": b"\n\n \n\t\n",
    "tabs_inside_product": b"A\ta1\tp\tq\tr\nMKV\nA\ta2\t\tq\nMKL\n",
}


def _same(path, want_k=None):
    d = PangeneIData.read_from_file(path)
    res, off, gen = d.flatten()
    s = PangeneIData.scan(path)
    assert s["sequences"] == len(gen) and s["genomes"] == len(d.genomeNames)
    assert np.array_equal(s["residues"], res) and np.array_equal(s["offsets"], off) and np.array_equal(s["genome_of"], gen)
    if want_k is not None:
        assert s["k_suggested"] == want_k
    return s


@pytest.mark.parametrize("name", sorted(EXPECTED["text"]))
def test_scan_matches_the_reader_and_the_reference_k_on_the_text_cases(name):
    _same(KDIR / f"{name}.faa", EXPECTED["text"][name])


@pytest.mark.parametrize("name", sorted(EDGE_TEXTS))
def test_scan_line_and_trim_rules(name, tmp_path):
    p = tmp_path / f"{name}.faa"
    p.write_bytes(EDGE_TEXTS[name])
    s = _same(p)
    try:
        want = calculate_k_faa(p)
    except (ValueError, ZeroDivisionError):
        want = 0                                       # the script dies there (log of 0 or 1 letters)
    assert s["k_suggested"] == want


def test_scan_small_fixtures(tmp_path):
    for name in SMALL_CASES:
        fx = np.load(GOLDEN / f"{name}.npz")
        p = tmp_path / f"{name}.faa"
        p.write_bytes(fx["faa"].tobytes())
        _same(p)


@pytest.mark.parametrize("name", sorted(EXPECTED["synthetic"]))
def test_scan_k_of_the_stand_in_sets(name, tmp_path):
    case = EXPECTED["synthetic"][name]
    gs = make_gene_set(**case["shape"])
    p = tmp_path / "in.faa"
    gs.write_faa(p)
    s = PangeneIData.scan(p)
    assert s["k_suggested"] == case["k"]
    assert np.array_equal(s["residues"], gs.residues) and np.array_equal(s["offsets"], gs.offsets) and np.array_equal(s["genome_of"], gs.genome_of)


def _messy_faa(seed, records, genomes=9):
    """A few MB of .faa text with everything the reader's rules are about, spread over the whole file (so that every chunk of
    the parallel parse starts in a different state): all three terminators, blank and whitespace-only lines between and INSIDE
    records (the raw line parity calculate_k.py goes by drifts away from the header / sequence alternation), padding to trim,
    genomes that come back after others, letters that first appear late."""
    rng = np.random.default_rng(seed)
    out = []
    term = [b"\n", b"\r\n", b"\r"]
    letters = b"ACDEFGHIKLMNPQRSTVWY"
    for i in range(records):
        g = int(rng.integers(0, min(genomes, 2 + i * genomes // max(records // 2, 1))))
        pad = [b"", b" ", b"\t ", b"\x01"][int(rng.integers(0, 4))]
        out += [pad, b"genome_%d\tgene_%d\tproduct %d" % (g, i, i), pad, term[int(rng.integers(0, 3))]]
        while rng.random() < 0.1:
            out += [[b"", b"  ", b"\t"][int(rng.integers(0, 3))], term[int(rng.integers(0, 3))]]
        n = int(rng.integers(1, 400))
        seq = bytes(letters[int(x)] for x in rng.integers(0, 20 if i > records // 3 else 12, n))
        if i == records - 7:
            seq += b"XBZ"                              # letters seen for the first time near the end
        out += [pad, seq, pad, term[int(rng.integers(0, 3))]]
        while rng.random() < 0.1:
            out += [[b"", b" "][int(rng.integers(0, 2))], term[int(rng.integers(0, 3))]]
    text = b"".join(out)
    return text[:-1] if seed % 2 and text.endswith(b"\n") else text     # (sometimes no terminator at the very end)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_parallel_scan_of_a_messy_file_equals_the_reader(seed, tmp_path):
    """Files of several MB are cut into chunks at line starts and parsed by a team of threads (pdl_ingest.hip): same sequences,
    offsets, genome ids in first-seen order and k as the one-pass reader that mirrors the Java, whatever state a chunk starts in."""
    p = tmp_path / "messy.faa"
    text = _messy_faa(seed, 14000 + 1000 * seed)
    assert len(text) > 2 * (1 << 20)
    p.write_bytes(text)
    s = _same(p)
    # calculate_k.py opens the file in text mode: universal newlines, the same raw lines
    assert s["k_suggested"] == calculate_k_faa(p)
    assert s["genomes"] == 9


def test_parallel_scan_reports_the_first_malformed_header(tmp_path):
    import ctypes as C
    lib = _lib.load()
    text = _messy_faa(7, 15000)
    lines = text.split(b"\n")
    # break two headers far apart (both behind the first chunk): the one the one-pass reader meets first is reported
    hdr = [i for i, l in enumerate(lines) if l.strip(b" \t\x01\r").startswith(b"genome_") and b"\r" not in l.strip(b"\r")]
    a, b = hdr[len(hdr) // 2], hdr[-5]
    for i in (a, b):
        lines[i] = lines[i].replace(b"\tproduct", b" product")
    p = tmp_path / "bad.faa"
    p.write_bytes(b"\n".join(lines))
    ing = _lib.PdlIngest()
    assert lib.pdl_scan_faa(str(p).encode(), C.byref(ing), None, 0, None, None, 0) == _lib.PDL_ERR_ARGUMENT
    msg = lib.pdl_last_error(None).decode()
    with open(p, "r") as f:                            # raw line number as the reader counts (universal newlines)
        want = next(n for n, l in enumerate(f, 1) if l.strip(" \t\x01\n").startswith("genome_") and l.count("\t") - l.strip(" \t\x01\n").count("\t") >= 0 and l.strip(" \t\x01\n").count("\t") < 2)
    assert f"line {want}:" in msg, msg


def test_scan_errors(tmp_path):
    import ctypes as C
    lib = _lib.load()
    ing = _lib.PdlIngest()
    assert lib.pdl_scan_faa(str(tmp_path / "missing.faa").encode(), C.byref(ing), None, 0, None, None, 0) == _lib.PDL_ERR_ARGUMENT
    assert b"missing.faa" in lib.pdl_last_error(None)
    p = tmp_path / "two_fields.faa"
    p.write_bytes(b"A\ta1\tp\nMKV\nB\tb1\nMKL\n")          # PangeneIData.java:49-51 indexes cc[2]: the Java reader throws here
    assert lib.pdl_scan_faa(str(p).encode(), C.byref(ing), None, 0, None, None, 0) == _lib.PDL_ERR_ARGUMENT
    assert b"line 3" in lib.pdl_last_error(None)
    with pytest.raises(IndexError):
        PangeneIData.read_from_file(p)
    p = tmp_path / "ok.faa"
    p.write_bytes(b"A\ta1\tp\nMKV\n")
    buf = np.zeros(2, np.uint8)
    assert lib.pdl_scan_faa(str(p).encode(), C.byref(ing), buf.ctypes.data, 2, None, None, 0) == _lib.PDL_ERR_ARGUMENT   # buffer too small


# ---- GPU --------------------------------------------------------------------------------------------------------------
def _device_bytes(nat, ptr, nbytes):
    import torch
    t = torch.empty(max(nbytes, 1), dtype=torch.uint8, device="cuda")
    nat.copy_device(t.data_ptr(), ptr, nbytes)
    return t[:nbytes].cpu().numpy()


def _scores_dict(s):
    d = {f: getattr(s, f) for f in FIELDS}
    d["scoresCount"] = s.scoresCount
    return d


@pytest.mark.gpu
@pytest.mark.parametrize("name", SMALL_CASES)
def test_ingested_file_scores_like_the_fixture(name, tmp_path):
    from pandelos_amd.pangene_native import PangeneNative
    fx = dict(np.load(GOLDEN / f"{name}.npz"))
    p = tmp_path / "in.faa"
    p.write_bytes(fx["faa"].tobytes())
    ref = PangeneIData.scan(p)
    nat = PangeneNative.open()
    ing = nat.ingest_faa(p)
    assert (ing["sequences"], ing["genomes"], ing["residues"]) == (ref["sequences"], ref["genomes"], len(ref["residues"]))
    assert np.array_equal(ing["offsets"], ref["offsets"]) and np.array_equal(ing["genome_of"], ref["genome_of"])
    assert np.array_equal(_device_bytes(nat, ing["d_residues"], ing["residues"]), ref["residues"])
    assert np.array_equal(_device_bytes(nat, ing["d_offsets"], 8 * (ing["sequences"] + 1)).view(np.uint64), ref["offsets"])
    assert np.array_equal(_device_bytes(nat, ing["d_genome_of"], 4 * ing["sequences"]).view(np.uint32), ref["genome_of"])
    nat.preprocess_ingested(int(fx["k"]))
    assert nat.cost.total_cost == int(fx["total_cost"])
    assert_scores_equal_fixture(lambda g: _scores_dict(nat.generate_scores_part(g)), fx, ing["genomes"], name)
    nat.close()


@pytest.mark.gpu
def test_ingest_streams_a_file_larger_than_its_staging_buffers(tmp_path):
    """17.6 MB of residues = three fills of the two 8-MB pinned buffers; what arrives equals what was written, k is the
    reference script's, the Scores are the reference's digests; a second, smaller file on the same context replaces it."""
    from pandelos_amd.pangene_native import PangeneNative
    base = json.loads((GOLDEN / "digests_baseline.json").read_text())["mycoplasma64_standin"]
    gs = make_gene_set(**CONFIGS["mycoplasma64_standin"])
    p = tmp_path / "in.faa"
    gs.write_faa(p)
    nat = PangeneNative.open()
    ing = nat.ingest_faa(p)
    assert ing["k_suggested"] == base["k"] and ing["sequences"] == gs.genes and ing["genomes"] == gs.genomes
    assert np.array_equal(_device_bytes(nat, ing["d_residues"], ing["residues"]), gs.residues)
    assert np.array_equal(ing["offsets"], gs.offsets) and np.array_equal(ing["genome_of"], gs.genome_of)
    assert ing["genome_names"][:3] == ["G0", "G1", "G2"]
    nat.preprocess_ingested(ing["k_suggested"])
    assert nat.cost.total_cost == base["total_cost"]
    assert_scores_match_digest(lambda g: _scores_dict(nat.generate_scores_part(g)), base, "ingested 64-genome set")
    # the same context, another file
    res, off, gen, k, d = load_large("synth_8x300x200_k4_div25")
    small = make_gene_set(**d["shape"])
    small.write_faa(p)
    ing = nat.ingest_faa(p)
    assert np.array_equal(_device_bytes(nat, ing["d_residues"], ing["residues"]), res)
    nat.preprocess_ingested(k)
    assert_scores_match_digest(lambda g: _scores_dict(nat.generate_scores_part(g)), d, "second ingest")
    nat.close()


@pytest.mark.gpu
def test_preprocess_ingested_needs_an_ingest():
    from pandelos_amd.pangene_native import PangeneNative
    nat = PangeneNative.open()
    with pytest.raises(_lib.PdlError) as e:
        nat.preprocess_ingested(3)
    assert e.value.code == _lib.PDL_ERR_STATE
    with pytest.raises(_lib.PdlError):
        nat.ingest_faa("/nonexistent/file.faa")
    nat.close()
