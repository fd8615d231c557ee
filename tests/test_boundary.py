"""The drop-in boundary without a GPU: the C-ABI library loads, exports every symbol include/pandelos_amd.h
declares, the JNI shim exports the reference's two symbols, and creating a context without a device fails
loudly (there is no CPU fallback)."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HEADER = (ROOT / "include" / "pandelos_amd.h").read_text()
DECLARED = sorted(set(re.findall(r"PDL_API[^;]*?\b(pdl_\w+)\s*\(", HEADER)))


def test_header_and_binding_agree():
    from pandelos_amd import _lib
    assert sorted(_lib.EXPORTS) == DECLARED


def test_core_library_exports_every_declared_symbol():
    from pandelos_amd import _lib
    lib = C.CDLL(str(_lib.LIB_PATH))
    for name in DECLARED:
        assert hasattr(lib, name), f"{name} missing from libpandelos_amd.so"


def test_jni_shim_exports_the_reference_symbols():
    from pandelos_amd import _lib
    shim = C.CDLL(str(_lib.LIB_DIR / "libnative.so"))
    # ig/native/pangene_native.h:16-25
    assert hasattr(shim, "Java_infoasys_cli_pangenes_PangeneNative_preprocessSequences")
    assert hasattr(shim, "Java_infoasys_cli_pangenes_PangeneNative_computeScores")
    for name in DECLARED:   # the shim is self-contained: it also carries the C ABI
        assert hasattr(shim, name)


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from pandelos_amd import _lib
    lib = _lib.load()
    assert not lib.pdl_create(None)
    assert b"no CPU path" in lib.pdl_last_error(None) or b"HIP" in lib.pdl_last_error(None)
    from pandelos_amd.pangene_native import PangeneNative
    from pandelos_amd.pangene_idata import PangeneIData
    import numpy as np
    d = PangeneIData.from_arrays(np.frombuffer(b"ACDEFG", np.uint8), [0, 6], [0])
    with pytest.raises(_lib.PdlError):
        PangeneNative(2, d)


def test_jni_slot_numbers_match_a_jdk_header_when_one_is_available():
    """include/pdl_jni_abi.h names JNI function-table slots from the specification; check them against a real
    jni.h (the build container has one inside the reference tree; nothing is copied from it)."""
    jni_h = Path("/root/reference/ig/native/jni/jni.h")
    if not jni_h.exists():
        pytest.skip("no jni.h to compare with")
    src = jni_h.read_text()
    body = src[src.index("struct JNINativeInterface_ {"):]
    body = body[:body.index("};")]
    names = [m.group(1) or m.group(2) for m in re.finditer(r"void \*(reserved\d)|\(JNICALL \*(\w+)\)", body)]
    ours = dict(re.findall(r"PJ_(\w+)\s*=\s*(\d+)", (ROOT / "include" / "pdl_jni_abi.h").read_text()))
    slots = int(ours.pop("TABLE_SLOTS"))
    assert slots == len(names)
    for name, idx in ours.items():
        assert names[int(idx)] == name, (name, idx, names[int(idx)])


def test_faa_reader_follows_the_java_reader(tmp_path):
    from pandelos_amd.pangene_idata import PangeneIData
    p = tmp_path / "x.faa"
    p.write_bytes(b"\n  \nG0\ta\tp q\n  ACDEFGHIKL  \n\nG1\tb\tp\r\nACDEFGHIKM\r\n\n\nG0\tc\tp\nCDEFGHIKLA\n")
    d = PangeneIData.read_from_file(p)
    assert d.sequences == [b"ACDEFGHIKL", b"ACDEFGHIKM", b"CDEFGHIKLA"]
    assert d.sequenceGenome == [0, 1, 0] and d.genomeNames == ["G0", "G1"]
    assert d.sequenceName == ["a", "b", "c"] and d.sequenceDescription[0] == "p q"
    res, off, gen = d.flatten()
    assert list(off) == [0, 10, 20, 30] and list(gen) == [0, 1, 0] and res.tobytes() == b"".join(d.sequences)


def test_calculate_k_matches_the_reference_formula(tmp_path):
    import math
    from pandelos_amd.calculate_k import calculate_k, calculate_k_faa
    from pandelos_amd.synth import make_gene_set
    gs = make_gene_set(genomes=5, genes_per_genome=60, mean_len=80, sub_rate=0.08, seed=1)
    p = tmp_path / "s.faa"
    gs.write_faa(p)
    k1, k2 = calculate_k(gs.residues), calculate_k_faa(p)
    assert k1 == k2 == 3          # this shape gives k = 3
    # closed form for a uniform 20-letter alphabet: floor(log20(R) / ~1)
    assert k1 == math.floor(math.log(len(gs.residues), 20) / 0.9999) or k1 == math.floor(math.log(len(gs.residues), 20))
