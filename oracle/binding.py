"""ctypes binding of the CPU ORACLE (test infrastructure — never imported by the product).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  It loads ``oracle/_build/libpangene_oracle.so`` (plain-C restatement of
``ig/native/library.cpp``, see ``pangene_oracle.c``) and drives ``oracle/_build/jni_harness``
against either ``oracle/_ref/libnative_ref.so`` (the reference's own library.cpp, compiled in
place by ``oracle/Makefile``) or the product's JNI shim.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import re
import struct
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
BUILD = HERE / "_build"
REF = HERE / "_ref"
ORACLE_SO = BUILD / "libpangene_oracle.so"
HARNESS = BUILD / "jni_harness"
REF_SO = REF / "libnative_ref.so"


def build(quiet: bool = True) -> None:
    """(Re)build the checker binaries with oracle/Makefile (gcc/g++ only)."""
    subprocess.run(["make", "-s", "-C", str(HERE), "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def have_reference() -> bool:
    return REF_SO.exists() and HARNESS.exists()


class _PoScores(C.Structure):
    _fields_ = [("count", C.c_uint32), ("rows", C.c_uint32), ("genomes", C.c_uint32),
                ("sequences", C.c_uint32),
                ("scores", C.POINTER(C.c_float)), ("percs", C.POINTER(C.c_float)),
                ("tr_percs", C.POINTER(C.c_float)),
                ("row", C.POINTER(C.c_int32)), ("column", C.POINTER(C.c_int32)),
                ("first_seq_genome", C.POINTER(C.c_int32)), ("second_seq_genome", C.POINTER(C.c_int32)),
                ("max_genome_score", C.POINTER(C.c_float)),
                ("max_genome_score_col", C.POINTER(C.c_float)),
                ("scoresMaxMappings", C.POINTER(C.c_int32))]


_KMER_DT = np.dtype([("rank", "<u8"), ("seq", "<u4"), ("count", "<u4")])
_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    if not ORACLE_SO.exists():
        build()
    lib = C.CDLL(str(ORACLE_SO))
    lib.po_create.restype = C.c_void_p
    lib.po_destroy.argtypes = [C.c_void_p]
    lib.po_preprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int]
    lib.po_preprocess.restype = C.c_int
    lib.po_compute_scores.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(_PoScores)]
    lib.po_compute_scores.restype = C.c_int
    lib.po_free_scores.argtypes = [C.POINTER(_PoScores)]
    for name, res in [("po_sequences", C.c_uint32), ("po_genomes", C.c_uint32), ("po_rank_base", C.c_uint32),
                      ("po_rank_byte_order", C.c_uint32), ("po_hash_fallback", C.c_int),
                      ("po_last_multiplier", C.c_uint64), ("po_rank_values", C.c_void_p),
                      ("po_dict_size", C.c_uint64), ("po_dict", C.c_void_p),
                      ("po_kmer_occurrences", C.c_uint64), ("po_kseq_lengths", C.c_void_p),
                      ("po_total_visited", C.c_void_p), ("po_total_cost", C.c_uint64)]:
        getattr(lib, name).argtypes = [C.c_void_p]
        getattr(lib, name).restype = res
    lib.po_genome_cost.argtypes = [C.c_void_p, C.c_uint32]
    lib.po_genome_cost.restype = C.c_uint64
    lib.po_rank_gene.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.po_rank_gene.restype = C.c_int
    _lib = lib
    return lib


def _np_from(ptr, dtype, n):
    if n == 0:
        return np.zeros(0, dtype)
    addr = C.cast(ptr, C.c_void_p).value
    buf = (C.c_char * (np.dtype(dtype).itemsize * n)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


SCORE_FIELDS = ("scores", "percs", "tr_percs", "row", "column", "first_seq_genome", "second_seq_genome")


class Oracle:
    """CPU restatement of preprocessSequences + computeScores on flattened sequences."""

    def __init__(self, residues: np.ndarray, offsets: np.ndarray, genome_of: np.ndarray, k: int,
                 only_complexity: bool = False):
        lib = _load()
        self._lib = lib
        self._res = np.ascontiguousarray(residues, dtype=np.uint8)
        self._off = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._gen = np.ascontiguousarray(genome_of, dtype=np.uint32)
        self._ctx = C.c_void_p(lib.po_create())
        self.status = lib.po_preprocess(self._ctx, self._res.ctypes.data, self._off.ctypes.data,
                                        self._gen.ctypes.data, len(self._gen), int(k), int(only_complexity))
        self.k = int(k)

    def close(self):
        if self._ctx:
            self._lib.po_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- dictionary-stage introspection ------------------------------------------------------
    @property
    def sequences(self): return self._lib.po_sequences(self._ctx)
    @property
    def genomes(self): return self._lib.po_genomes(self._ctx)
    @property
    def rank_base(self): return self._lib.po_rank_base(self._ctx)
    @property
    def rank_byte_order(self): return self._lib.po_rank_byte_order(self._ctx)
    @property
    def hash_fallback(self): return bool(self._lib.po_hash_fallback(self._ctx))
    @property
    def last_multiplier(self): return self._lib.po_last_multiplier(self._ctx)
    @property
    def rank_values(self): return _np_from(self._lib.po_rank_values(self._ctx), np.uint8, 256)
    @property
    def kmer_occurrences(self): return self._lib.po_kmer_occurrences(self._ctx)
    @property
    def total_cost(self): return self._lib.po_total_cost(self._ctx)
    def genome_cost(self, g): return self._lib.po_genome_cost(self._ctx, g)
    def dictionary(self): return _np_from(self._lib.po_dict(self._ctx), _KMER_DT, self._lib.po_dict_size(self._ctx))
    def kseq_lengths(self): return _np_from(self._lib.po_kseq_lengths(self._ctx), np.uint32, self.sequences)
    def total_visited(self): return _np_from(self._lib.po_total_visited(self._ctx), np.uint64, self.sequences)

    def rank_gene(self, chars: bytes) -> np.ndarray:
        out = np.zeros(max(len(chars) - self.k + 1, 0), np.uint64)
        buf = np.frombuffer(chars, np.uint8)
        n = self._lib.po_rank_gene(self._ctx, buf.ctypes.data, len(chars), out.ctypes.data)
        return out[:n]

    # -- scoring ---------------------------------------------------------------------------
    def scores(self, genome: int) -> dict:
        s = _PoScores()
        rc = self._lib.po_compute_scores(self._ctx, genome, C.byref(s))
        if rc != 0:
            raise ValueError(f"po_compute_scores({genome}) -> {rc}")
        z = s.count
        out = {"scoresCount": z, "rows": s.rows}
        for f in ("scores", "percs", "tr_percs"):
            out[f] = _np_from(getattr(s, f), np.float32, z)
        for f in ("row", "column", "first_seq_genome", "second_seq_genome"):
            out[f] = _np_from(getattr(s, f), np.int32, z)
        out["max_genome_score"] = _np_from(s.max_genome_score, np.float32, s.rows * s.genomes).reshape(s.rows, s.genomes)
        out["max_genome_score_col"] = _np_from(s.max_genome_score_col, np.float32, s.sequences)
        out["scoresMaxMappings"] = _np_from(s.scoresMaxMappings, np.int32, s.sequences)
        self._lib.po_free_scores(C.byref(s))
        return out


# ---- JVM-less harness --------------------------------------------------------------------------

def read_dump(path) -> dict:
    """Parse the binary Scores dump written by jni_harness --dump."""
    b = Path(path).read_bytes()
    assert b[:8] == b"PDLSCOR1", "bad dump magic"
    n, g_count, k = struct.unpack_from("<III", b, 8)
    off = 20
    genomes = []
    for g in range(g_count):
        gg, z, rows = struct.unpack_from("<III", b, off)
        off += 12
        assert gg == g
        d = {"scoresCount": z, "rows": rows}
        for f in ("scores", "percs", "tr_percs"):
            d[f] = np.frombuffer(b, "<f4", z, off).copy(); off += 4 * z
        for f in ("row", "column", "first_seq_genome", "second_seq_genome"):
            d[f] = np.frombuffer(b, "<i4", z, off).copy(); off += 4 * z
        d["max_genome_score"] = np.frombuffer(b, "<f4", rows * g_count, off).reshape(rows, g_count).copy()
        off += 4 * rows * g_count
        d["max_genome_score_col"] = np.frombuffer(b, "<f4", n, off).copy(); off += 4 * n
        d["scoresMaxMappings"] = np.frombuffer(b, "<i4", n, off).copy(); off += 4 * n
        genomes.append(d)
    assert off == len(b)
    return {"sequences": n, "genomes": g_count, "k": k, "per_genome": genomes}


def run_harness(lib_so, faa, k, threads=1, dump=None, complexity=False, env=None, timeout=None) -> dict:
    """Run jni_harness against a library exporting the two JNI symbols.

    Returns the harness's JSON timing line plus whatever the library printed in the
    reference's cost-model format (``Total cost``, ``Genome g cost``) parsed from stdout.
    """
    if not HARNESS.exists():
        build()
    cmd = [str(HARNESS), "--lib", str(lib_so), "-i", str(faa), "-k", str(k), "-j", str(threads)]
    if dump:
        cmd += ["--dump", str(dump)]
    if complexity:
        cmd += ["-c"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout)
    if p.returncode != 0:
        raise RuntimeError(f"jni_harness failed ({p.returncode}): {p.stderr[-2000:]}")
    info = json.loads(p.stderr.strip().splitlines()[-1])
    m = re.search(r"Total cost: (\d+) lookups", p.stdout)
    info["total_cost"] = int(m.group(1)) if m else None
    info["genome_cost"] = {int(a): int(b) for a, b in re.findall(r"Genome (\d+) cost = (\d+)", p.stdout)}
    info["hash_fallback"] = "Hashing fallback!" in p.stdout
    info["stdout"] = p.stdout
    return info
