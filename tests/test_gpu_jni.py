"""The JNI boundary on the GPU: oracle/jni_harness plays the JVM (same calls the Java host makes,
Pangenes.java:39,64-66) against pandelos_amd/lib/libnative.so and the Scores objects it fills are compared,
field by field and bit by bit, with the golden vectors of the reference's libnative."""
import tempfile
from pathlib import Path

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["readme4_k2", "q1_fold_same_gene_twice", "synth_5x60x80_k3", "synth_5x60x80_k16_hash",
                                  "blank_lines_and_spaces", "interleaved_genomes"])
@pytest.mark.parametrize("threads", [1, 4])
def test_jni_shim_fills_scores_like_the_reference(name, threads, tmp_path):
    from oracle import binding as ob
    from pandelos_amd import _lib
    fx = dict(np.load(H.GOLDEN / f"{name}.npz"))
    faa = tmp_path / "in.faa"
    faa.write_bytes(fx["faa"].tobytes())
    info = ob.run_harness(_lib.LIB_DIR / "libnative.so", faa, int(fx["k"]), threads=threads, dump=tmp_path / "out.bin", timeout=300)
    got = ob.read_dump(tmp_path / "out.bin")
    assert info["total_cost"] == int(fx["total_cost"])                      # "Total cost: N lookups" line
    assert [info["genome_cost"][g] for g in range(got["genomes"])] == [int(x) for x in fx["genome_cost"]]
    assert info["hash_fallback"] == bool(fx["hash_fallback"])
    H.assert_scores_equal_fixture(lambda g: got["per_genome"][g], fx, got["genomes"], f"{name} via JNI")


def test_jni_shim_k_zero_exits_like_the_reference(tmp_path):
    import subprocess
    from oracle import binding as ob
    from pandelos_amd import _lib
    fx = dict(np.load(H.GOLDEN / "readme4_k2.npz"))
    faa = tmp_path / "in.faa"
    faa.write_bytes(fx["faa"].tobytes())
    p = subprocess.run([str(ob.HARNESS), "--lib", str(_lib.LIB_DIR / "libnative.so"), "-i", str(faa), "-k", "0"],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 1 and "K value must be greater than 0." in p.stdout     # library.cpp:90-93
