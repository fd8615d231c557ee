"""ctypes binding of include/pandelos_amd.h (libpandelos_amd.so, built in-tree by __graft_entry__.build()).

There is no CPU fallback: importing this module without the built library, or creating a context
without a HIP device, raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

LIB_DIR = Path(__file__).resolve().parent / "lib"
LIB_PATH = LIB_DIR / "libpandelos_amd.so"

PDL_OK = 0
PDL_ERR_KVALUE, PDL_ERR_EMPTY, PDL_ERR_ARGUMENT, PDL_ERR_DEVICE, PDL_ERR_STATE, PDL_ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6
PDL_FLAG_CANONICAL_ORDER = 1


class PdlError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"pandelos_amd error {code}: {message}")
        self.code = code


class PdlConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("stream", C.c_void_p), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class PdlCost(C.Structure):
    _fields_ = [("residues", C.c_uint64), ("kmer_occurrences", C.c_uint64), ("dictionary_records", C.c_uint64),
                ("shared_records", C.c_uint64), ("groups", C.c_uint64), ("total_cost", C.c_uint64),
                ("linear_ratio", C.c_float), ("sequences", C.c_uint32), ("genomes", C.c_uint32),
                ("rank_base", C.c_uint32), ("rank_bits", C.c_uint32), ("hash_fallback", C.c_int32),
                ("kvalue", C.c_int32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class PdlScores(C.Structure):
    _fields_ = [("scoresCount", C.c_uint32), ("rows", C.c_uint32), ("genomes", C.c_uint32), ("sequences", C.c_uint32),
                ("scores", C.POINTER(C.c_float)), ("percs", C.POINTER(C.c_float)), ("tr_percs", C.POINTER(C.c_float)),
                ("row", C.POINTER(C.c_int32)), ("column", C.POINTER(C.c_int32)),
                ("first_seq_genome", C.POINTER(C.c_int32)), ("second_seq_genome", C.POINTER(C.c_int32)),
                ("max_genome_score", C.POINTER(C.c_float)), ("max_genome_score_col", C.POINTER(C.c_float)),
                ("scoresMaxMappings", C.POINTER(C.c_int32))]


class PdlTimings(C.Structure):
    _fields_ = [("hist_ms", C.c_float), ("rank_ms", C.c_float), ("sort_rank_ms", C.c_float), ("dict_ms", C.c_float),
                ("sort_seq_ms", C.c_float), ("ranges_ms", C.c_float), ("join_ms", C.c_float),
                ("join_overflow_ms", C.c_float), ("order_ms", C.c_float), ("preprocess_total_ms", C.c_float),
                ("score_total_ms", C.c_float), ("emitted_cells", C.c_uint64), ("scored_rows", C.c_uint64),
                ("scored_lookups", C.c_uint64), ("overflow_rows", C.c_uint64), ("join_launches", C.c_uint32),
                ("tier2_rows", C.c_uint32),
                ("dist_begin_ms", C.c_float), ("dist_finish_ms", C.c_float), ("dist_score_begin_ms", C.c_float),
                ("dist_score_finish_ms", C.c_float), ("walked_lookups", C.c_uint64), ("outbox_cells", C.c_uint64),
                ("inbox_cells", C.c_uint64), ("aside_reloads", C.c_uint64), ("aside_repeats", C.c_uint32), ("tier1_rows", C.c_uint32), ("reshard_ms", C.c_float), ("dist_ranges_ms", C.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if not n.startswith("reserved")}


class PdlEdges(C.Structure):
    _fields_ = [("count", C.c_uint32), ("src", C.POINTER(C.c_int32)), ("dst", C.POINTER(C.c_int32)), ("score", C.POINTER(C.c_float))]


class PdlDistSlice(C.Structure):
    _fields_ = [("d_postings", C.c_void_p), ("records", C.c_uint64), ("kmers", C.c_uint64),
                ("genome_weights", C.POINTER(C.c_uint64)), ("genomes", C.c_uint32), ("genome_costs", C.POINTER(C.c_uint64))]


class PdlDistRanges(C.Structure):
    _fields_ = [("available", C.c_int32), ("d_keys", C.c_void_p), ("d_ranges", C.c_void_p), ("counts", C.POINTER(C.c_uint64)),
                ("total", C.c_uint64), ("shared_records", C.c_uint64), ("groups", C.c_uint64), ("repeat_sample", C.c_uint64)]


class PdlDistOutbox(C.Structure):
    _fields_ = [("d_cells", C.c_void_p), ("counts", C.POINTER(C.c_uint64)), ("total", C.c_uint64)]


class PdlIngest(C.Structure):
    _fields_ = [("file_bytes", C.c_uint64), ("residues", C.c_uint64), ("sequences", C.c_uint32), ("genomes", C.c_uint32),
                ("k_suggested", C.c_int32), ("reserved", C.c_uint32), ("parse_ms", C.c_double),
                ("offsets", C.POINTER(C.c_uint64)), ("genome_of", C.POINTER(C.c_uint32)),
                ("d_residues", C.c_void_p), ("d_offsets", C.c_void_p), ("d_genome_of", C.c_void_p)]


DIST_CELL_BYTES = 24      # pdl_dist_cell: 3 x f32 + 3 x u32


# every symbol include/pandelos_amd.h declares (tests/test_boundary.py checks the export table against this)
EXPORTS = ("pdl_create", "pdl_destroy", "pdl_last_error", "pdl_preprocess", "pdl_preprocess_device",
           "pdl_genome_cost", "pdl_sequence_costs", "pdl_set_genome_shard", "pdl_score_all", "pdl_compute_scores",
           "pdl_free_scores", "pdl_scores_counts", "pdl_get_dictionary", "pdl_get_rank_table", "pdl_get_timings",
           "pdl_version", "pdl_set_option", "pdl_dist_preprocess_begin", "pdl_dist_preprocess_finish",
           "pdl_dist_preprocess_ranges", "pdl_dist_preprocess_finish_ranges",
           "pdl_dist_genome_owner", "pdl_dist_score_begin", "pdl_dist_score_finish", "pdl_copy_device",
           "pdl_compute_edges", "pdl_free_edges", "pdl_ingest_faa", "pdl_ingest_genome_name", "pdl_preprocess_ingested",
           "pdl_scan_faa", "pdl_pin_arrived", "pdl_pin_checksum")

_lib = None


def load():
    """Load libpandelos_amd.so; raises when it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: the PyTorch-ROCm wheel carries its own libamdhip64.so.7 /
    # libhsa-runtime64; if it is going to be used at all (device buffers, streams, RCCL) it must be
    # the copy that gets loaded, so that this library binds to the same one by soname.  Without
    # torch the system ROCm runtime is used (the JNI / C++ host case).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not LIB_PATH.exists():
        raise ImportError(f"{LIB_PATH} is missing: run `python __graft_entry__.py` (hipcc --offload-arch=gfx950) first")
    lib = C.CDLL(str(LIB_PATH))
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    lib.pdl_create.argtypes = [C.POINTER(PdlConfig)]; lib.pdl_create.restype = vp
    lib.pdl_destroy.argtypes = [vp]; lib.pdl_destroy.restype = None
    lib.pdl_last_error.argtypes = [vp]; lib.pdl_last_error.restype = C.c_char_p
    lib.pdl_preprocess.argtypes = [vp, vp, vp, vp, u32, i32, i32, C.POINTER(PdlCost)]; lib.pdl_preprocess.restype = i32
    lib.pdl_preprocess_device.argtypes = [vp, vp, vp, vp, u32, u64, i32, i32, C.POINTER(PdlCost)]
    lib.pdl_preprocess_device.restype = i32
    lib.pdl_genome_cost.argtypes = [vp, u32, C.POINTER(u64)]; lib.pdl_genome_cost.restype = i32
    lib.pdl_sequence_costs.argtypes = [vp, vp, vp]; lib.pdl_sequence_costs.restype = i32
    lib.pdl_set_genome_shard.argtypes = [vp, vp, u32]; lib.pdl_set_genome_shard.restype = i32
    lib.pdl_score_all.argtypes = [vp]; lib.pdl_score_all.restype = i32
    lib.pdl_compute_scores.argtypes = [vp, u32, C.POINTER(PdlScores)]; lib.pdl_compute_scores.restype = i32
    lib.pdl_free_scores.argtypes = [C.POINTER(PdlScores)]; lib.pdl_free_scores.restype = None
    lib.pdl_scores_counts.argtypes = [vp, vp]; lib.pdl_scores_counts.restype = i32
    lib.pdl_get_dictionary.argtypes = [vp, vp, vp, vp]; lib.pdl_get_dictionary.restype = i32
    lib.pdl_get_rank_table.argtypes = [vp, vp, C.POINTER(u64)]; lib.pdl_get_rank_table.restype = i32
    lib.pdl_get_timings.argtypes = [vp, C.POINTER(PdlTimings)]; lib.pdl_get_timings.restype = i32
    lib.pdl_version.argtypes = []; lib.pdl_version.restype = C.c_char_p
    lib.pdl_set_option.argtypes = [vp, C.c_char_p, C.c_int64]; lib.pdl_set_option.restype = i32
    lib.pdl_dist_preprocess_begin.argtypes = [vp, vp, vp, vp, u32, u64, i32, u32, u32, C.POINTER(PdlDistSlice)]
    lib.pdl_dist_preprocess_begin.restype = i32
    lib.pdl_dist_preprocess_finish.argtypes = [vp, vp, u64, vp, C.POINTER(PdlCost)]; lib.pdl_dist_preprocess_finish.restype = i32
    lib.pdl_dist_preprocess_ranges.argtypes = [vp, vp, vp, vp, C.POINTER(PdlDistRanges)]; lib.pdl_dist_preprocess_ranges.restype = i32
    lib.pdl_dist_preprocess_finish_ranges.argtypes = [vp, vp, u64, vp, vp, u64, vp, C.POINTER(PdlCost)]; lib.pdl_dist_preprocess_finish_ranges.restype = i32
    lib.pdl_dist_genome_owner.argtypes = [vp, vp]; lib.pdl_dist_genome_owner.restype = i32
    lib.pdl_dist_score_begin.argtypes = [vp, C.POINTER(PdlDistOutbox)]; lib.pdl_dist_score_begin.restype = i32
    lib.pdl_dist_score_finish.argtypes = [vp, vp, u64]; lib.pdl_dist_score_finish.restype = i32
    lib.pdl_copy_device.argtypes = [vp, vp, vp, u64]; lib.pdl_copy_device.restype = i32
    lib.pdl_compute_edges.argtypes = [vp, u32, C.POINTER(PdlEdges)]; lib.pdl_compute_edges.restype = i32
    lib.pdl_free_edges.argtypes = [C.POINTER(PdlEdges)]; lib.pdl_free_edges.restype = None
    lib.pdl_ingest_faa.argtypes = [vp, C.c_char_p, C.POINTER(PdlIngest)]; lib.pdl_ingest_faa.restype = i32
    lib.pdl_ingest_genome_name.argtypes = [vp, u32]; lib.pdl_ingest_genome_name.restype = C.c_char_p
    lib.pdl_preprocess_ingested.argtypes = [vp, i32, i32, C.POINTER(PdlCost)]; lib.pdl_preprocess_ingested.restype = i32
    lib.pdl_scan_faa.argtypes = [C.c_char_p, C.POINTER(PdlIngest), vp, u64, vp, vp, u32]; lib.pdl_scan_faa.restype = i32
    lib.pdl_pin_arrived.argtypes = [vp, vp, vp, u32, u32, u32]; lib.pdl_pin_arrived.restype = i32
    lib.pdl_pin_checksum.argtypes = [vp, vp, vp, u32]; lib.pdl_pin_checksum.restype = u32
    _lib = lib
    return lib
