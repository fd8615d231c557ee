"""``PangeneNative`` — host-side mirror of ``ig/infoasys/cli/pangenes/PangeneNative.java`` over the C ABI.

Same surface as the Java class: the constructor preprocesses (PangeneNative.java:5-7),
``print_complexity`` is the ``-c`` mode (:10-12), ``generate_scores_part(genome, multithread)``
returns one ``Scores`` block (:17-21).  Underneath are ``pdl_preprocess`` / ``pdl_compute_scores``
of ``include/pandelos_amd.h``; errors come back as ``PdlError`` instead of the reference's
``exit(1)`` (library.cpp:90-93).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from .pangene_idata import PangeneIData
from .scores import Scores


def _np_copy(ptr, dtype, n):
    if n == 0:
        return np.zeros(0, dtype)
    buf = (C.c_char * (np.dtype(dtype).itemsize * n)).from_address(C.cast(ptr, C.c_void_p).value)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


class PangeneNative:
    def __init__(self, k: int, data: PangeneIData, only_complexity: bool = False, device: int = -1,
                 stream: Optional[int] = None, flags: int = 0):
        residues, offsets, genome_of = data.flatten()
        self._init(k, residues, offsets, genome_of, only_complexity, device, stream, flags)

    @classmethod
    def from_arrays(cls, k, residues, offsets, genome_of, only_complexity=False, device=-1, stream=None, flags=0):
        self = cls.__new__(cls)
        self._init(k, residues, offsets, genome_of, only_complexity, device, stream, flags)
        return self

    @classmethod
    def from_device(cls, k, d_residues: int, d_offsets: int, d_genome_of: int, n_sequences: int, n_residues: int,
                    only_complexity=False, device=-1, stream=None, flags=0, keepalive=None):
        """Device-resident inputs (raw device pointers, e.g. ``tensor.data_ptr()``)."""
        self = cls.__new__(cls)
        self._open(device, stream, flags)
        self.preprocess_device(k, d_residues, d_offsets, d_genome_of, n_sequences, n_residues, only_complexity, keepalive)
        return self

    def preprocess_device(self, k, d_residues: int, d_offsets: int, d_genome_of: int, n_sequences: int,
                          n_residues: int, only_complexity=False, keepalive=None):
        """(Re)run preprocessSequences on this context from device-resident inputs; like the reference's
        entry point it resets all state of the context first (library.cpp:192).  Work buffers are reused."""
        self._keep = keepalive
        self.cost = _lib.PdlCost()
        rc = self._lib.pdl_preprocess_device(self._ctx, d_residues, d_offsets, d_genome_of, n_sequences, n_residues,
                                             int(k), int(only_complexity), C.byref(self.cost))
        self._check(rc)

    @classmethod
    def open(cls, device=-1, stream=None, flags=0) -> "PangeneNative":
        """A context without a dictionary yet (set a genome shard, then preprocess)."""
        self = cls.__new__(cls)
        self._open(device, stream, flags)
        self.cost = _lib.PdlCost()
        return self

    @staticmethod
    def print_complexity(k: int, data: PangeneIData) -> "PangeneNative":
        """PangeneNative.printComplexity (PangeneNative.java:10-12): cost model only."""
        nat = PangeneNative(k, data, only_complexity=True)
        c = nat.cost
        print("------------\nCOMPUTATIONAL COSTS: ")
        print(f"Total cost: {c.total_cost} lookups")
        print(f"Linear ratio: {c.linear_ratio:g}\n------------\n")
        return nat

    # ------------------------------------------------------------------------------------------
    def _open(self, device, stream, flags):
        self._lib = _lib.load()
        cfg = _lib.PdlConfig(device=device, stream=stream or None, flags=flags, reserved=0)
        ctx = self._lib.pdl_create(C.byref(cfg))
        if not ctx:
            raise _lib.PdlError(_lib.PDL_ERR_DEVICE, self._lib.pdl_last_error(None).decode())
        self._ctx = C.c_void_p(ctx)

    def _init(self, k, residues, offsets, genome_of, only_complexity, device, stream, flags):
        self._open(device, stream, flags)
        self.preprocess(k, residues, offsets, genome_of, only_complexity)

    def preprocess(self, k, residues, offsets, genome_of, only_complexity=False):
        """(Re)run preprocessSequences on this context from host arrays."""
        res = np.ascontiguousarray(residues, dtype=np.uint8)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        gen = np.ascontiguousarray(genome_of, dtype=np.uint32)
        self.cost = _lib.PdlCost()
        rc = self._lib.pdl_preprocess(self._ctx, res.ctypes.data, off.ctypes.data, gen.ctypes.data, len(gen), int(k),
                                      int(only_complexity), C.byref(self.cost))
        self._check(rc)

    def ingest_faa(self, path) -> dict:
        """`.faa` -> HBM in one pass (``pdl_ingest_faa``: PangeneIData.readFromFile + calculate_k.py on the way, pinned
        staging buffers, copies overlapped with the parse).  -> sizes, ``k_suggested``, ``genome_names``, host ``offsets`` /
        ``genome_of`` and the device pointers ``preprocess_ingested`` works on."""
        ing = _lib.PdlIngest()
        self._check(self._lib.pdl_ingest_faa(self._ctx, str(path).encode(), C.byref(ing)))
        n = ing.sequences
        return {"file_bytes": ing.file_bytes, "residues": ing.residues, "sequences": n, "genomes": ing.genomes,
                "k_suggested": ing.k_suggested, "parse_ms": ing.parse_ms,
                "offsets": _np_copy(ing.offsets, np.uint64, n + 1), "genome_of": _np_copy(ing.genome_of, np.uint32, n),
                "genome_names": [self._lib.pdl_ingest_genome_name(self._ctx, g).decode("latin-1") for g in range(ing.genomes)],
                "d_residues": ing.d_residues or 0, "d_offsets": ing.d_offsets or 0, "d_genome_of": ing.d_genome_of or 0}

    def preprocess_ingested(self, k, only_complexity=False):
        """preprocessSequences on what ``ingest_faa`` left in HBM."""
        self.cost = _lib.PdlCost()
        self._check(self._lib.pdl_preprocess_ingested(self._ctx, int(k), int(only_complexity), C.byref(self.cost)))

    def _check(self, rc):
        if rc != _lib.PDL_OK:
            raise _lib.PdlError(rc, self._lib.pdl_last_error(self._ctx).decode())

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.pdl_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- reference surface ---------------------------------------------------------------------------
    def generate_scores_part(self, genome: int, multithread: bool = False) -> Scores:
        """PangeneNative.generateScoresPart (PangeneNative.java:17-21); ``multithread`` only selected the
        ignored step_size in the reference (library.cpp:454)."""
        s = _lib.PdlScores()
        self._check(self._lib.pdl_compute_scores(self._ctx, int(genome), C.byref(s)))
        try:
            z, rows, g, n = s.scoresCount, s.rows, s.genomes, s.sequences
            out = Scores(
                scoresCount=z,
                scores=_np_copy(s.scores, np.float32, z), percs=_np_copy(s.percs, np.float32, z),
                tr_percs=_np_copy(s.tr_percs, np.float32, z),
                row=_np_copy(s.row, np.int32, z), column=_np_copy(s.column, np.int32, z),
                first_seq_genome=_np_copy(s.first_seq_genome, np.int32, z),
                second_seq_genome=_np_copy(s.second_seq_genome, np.int32, z),
                max_genome_score=_np_copy(s.max_genome_score, np.float32, rows * g).reshape(rows, g),
                max_genome_score_col=_np_copy(s.max_genome_score_col, np.float32, n),
                scoresMaxMappings=_np_copy(s.scoresMaxMappings, np.int32, n))
        finally:
            self._lib.pdl_free_scores(C.byref(s))
        return out

    def generate_edges_part(self, genome: int):
        """The best-hit filter of the Java host (Pangenes.java:98-176) for one genome task, run on the device: -> (src, dst,
        score) of the edges the task adds to the network, in the host's insertion order."""
        e = _lib.PdlEdges()
        self._check(self._lib.pdl_compute_edges(self._ctx, int(genome), C.byref(e)))
        try:
            n = e.count
            return (_np_copy(e.src, np.int32, n).astype(np.int64), _np_copy(e.dst, np.int32, n).astype(np.int64), _np_copy(e.score, np.float32, n))
        finally:
            self._lib.pdl_free_edges(C.byref(e))

    # -- beyond the reference surface (device-resident batch, sharding, introspection) ------------------
    def score_all(self) -> None:
        self._check(self._lib.pdl_score_all(self._ctx))

    def set_option(self, name: str, value: int) -> None:
        """Tuning / test switch of this context (``pdl_set_option``; the library reads no environment variable)."""
        self._check(self._lib.pdl_set_option(self._ctx, name.encode(), int(value)))

    # -- multi-GPU passes (one context per GPU; pandelos_amd.distributed moves the bytes in between) ------------
    def dist_preprocess_begin(self, k, d_residues: int, d_offsets: int, d_genome_of: int, n_sequences: int,
                              n_residues: int, world: int, rank: int, keepalive=None):
        """-> (device pointer, records, k-mers) of this rank's run of the dictionary; ``self.run_weights`` = every genome's
        lookups above the diagonal inside the run (int64 [G])."""
        self._keep = keepalive
        sl = _lib.PdlDistSlice()
        self._check(self._lib.pdl_dist_preprocess_begin(self._ctx, d_residues, d_offsets, d_genome_of, n_sequences, n_residues,
                                                        int(k), int(world), int(rank), C.byref(sl)))
        self.run_weights = np.ctypeslib.as_array(sl.genome_weights, shape=(sl.genomes,)).astype(np.int64) if sl.genomes else np.zeros(0, np.int64)
        self.run_costs = np.ctypeslib.as_array(sl.genome_costs, shape=(sl.genomes,)).astype(np.int64) if sl.genomes else np.zeros(0, np.int64)
        return sl.d_postings or 0, int(sl.records), int(sl.kmers)

    def dist_preprocess_ranges(self, run_records, genome_weights, genome_costs):
        """The range tuples of this rank's run, filed by the rank that owns the gene (between begin and finish; the runs are
        gathered AFTER it).  ``run_records`` [world]; ``genome_weights`` / ``genome_costs`` [G]: the ranks' ``run_weights`` /
        ``run_costs`` summed.  -> None when this build cannot go that way (every rank gets the same answer: use
        ``dist_preprocess_finish``), else (device pointer of the keys, of the packed ranges, tuples per destination rank [world],
        counters [3] of this run: shared records, groups, repeat statistic)."""
        rr = np.ascontiguousarray(run_records, dtype=np.uint64)
        w = np.ascontiguousarray(genome_weights, dtype=np.uint64)
        cs = np.ascontiguousarray(genome_costs, dtype=np.uint64)
        out = _lib.PdlDistRanges()
        self._check(self._lib.pdl_dist_preprocess_ranges(self._ctx, rr.ctypes.data, w.ctypes.data, cs.ctypes.data, C.byref(out)))
        if not out.available:
            return None
        counts = np.array([out.counts[d] for d in range(len(rr))], dtype=np.int64)
        assert int(counts.sum()) == out.total
        return out.d_keys or 0, out.d_ranges or 0, counts, np.array([out.shared_records, out.groups, out.repeat_sample], dtype=np.int64)

    def dist_preprocess_finish_ranges(self, d_postings_all: int, total_records: int, d_keys: int, d_ranges: int, n_tuples: int,
                                      counter_sums, keepalive=None) -> None:
        """Adopts the gathered dictionary and sorts the received tuples (source-rank major) by gene.  ``counter_sums`` [3]: the
        ranks' counters summed.  The three device arrays must stay alive until the next preprocess (``keepalive``)."""
        self._keep_dict = keepalive
        self.cost = _lib.PdlCost()
        sums = np.ascontiguousarray(counter_sums, dtype=np.uint64)
        self._check(self._lib.pdl_dist_preprocess_finish_ranges(self._ctx, d_postings_all, int(total_records), d_keys, d_ranges, int(n_tuples),
                                                                sums.ctypes.data, C.byref(self.cost)))

    def dist_preprocess_finish(self, d_postings_all: int, total_records: int, genome_weights=None, keepalive=None) -> None:
        """``genome_weights``: the ranks' ``run_weights`` summed (identical on every rank); None lets the library compute them."""
        self._keep_dict = keepalive
        self.cost = _lib.PdlCost()
        w = None
        if genome_weights is not None:
            w = np.ascontiguousarray(genome_weights, dtype=np.uint64)
        self._check(self._lib.pdl_dist_preprocess_finish(self._ctx, d_postings_all, int(total_records),
                                                         w.ctypes.data if w is not None else None, C.byref(self.cost)))

    def dist_genome_owner(self) -> np.ndarray:
        out = np.zeros(self.cost.genomes, np.uint32)
        self._check(self._lib.pdl_dist_genome_owner(self._ctx, out.ctypes.data))
        return out

    def dist_score_begin(self, world: int):
        """-> (device pointer of the outbox, cells per destination rank [world])."""
        ob = _lib.PdlDistOutbox()
        self._check(self._lib.pdl_dist_score_begin(self._ctx, C.byref(ob)))
        counts = np.array([ob.counts[d] for d in range(world)], dtype=np.int64)
        assert int(counts.sum()) == ob.total
        return ob.d_cells or 0, counts

    def dist_score_finish(self, d_inbox: int, n_inbox: int, keepalive=None) -> None:
        self._keep_inbox = keepalive
        self._check(self._lib.pdl_dist_score_finish(self._ctx, d_inbox, int(n_inbox)))

    def copy_device(self, d_dst: int, d_src: int, nbytes: int) -> None:
        self._check(self._lib.pdl_copy_device(self._ctx, d_dst, d_src, int(nbytes)))

    def set_genome_shard(self, genomes: Sequence[int]) -> None:
        g = np.ascontiguousarray(genomes, dtype=np.uint32)
        self._check(self._lib.pdl_set_genome_shard(self._ctx, g.ctypes.data, len(g)))

    def scores_in_batches(self, k, residues, offsets, genome_of, genomes_per_batch: int, low_memory: bool = True):
        """Score a set a batch of genomes at a time and yield ``(genome, Scores)`` for every genome in ascending order — for sets
        whose maxima, staging and cells do not fit the device together (the reference's own granularity: one task per genome
        with private scratch, Pangenes.java:60-66, library.cpp:417-428).  The dictionary is built once, with the first batch as
        the genome shard; every further batch costs the two passes over the postings that form its genes' range lists
        (``pdl_set_genome_shard`` on an existing dictionary).  Batches are scored without the row/column symmetry of the
        whole-set pass (a cell's mirror belongs to another batch), i.e. with the reference's full lookup count."""
        gen = np.ascontiguousarray(genome_of, dtype=np.uint32)
        n_genomes = int(gen.max()) + 1 if len(gen) else 0
        if low_memory:
            self.set_option("low_memory", 1)
        for g0 in range(0, n_genomes, max(1, int(genomes_per_batch))):
            batch = list(range(g0, min(n_genomes, g0 + max(1, int(genomes_per_batch)))))
            self.set_genome_shard(batch)
            if g0 == 0:
                self.preprocess(k, residues, offsets, gen)
            for g in batch:
                yield g, self.generate_scores_part(g)

    def genome_cost(self, genome: int) -> int:
        v = C.c_uint64()
        self._check(self._lib.pdl_genome_cost(self._ctx, genome, C.byref(v)))
        return v.value

    def sequence_costs(self):
        n = self.cost.sequences
        cost = np.zeros(n, np.uint64)
        kl = np.zeros(n, np.uint32)
        self._check(self._lib.pdl_sequence_costs(self._ctx, cost.ctypes.data, kl.ctypes.data))
        return cost, kl

    def scores_counts(self) -> np.ndarray:
        out = np.zeros(self.cost.genomes, np.uint32)
        self._check(self._lib.pdl_scores_counts(self._ctx, out.ctypes.data))
        return out

    def dictionary(self):
        u = self.cost.dictionary_records
        ranks, seqs, counts = np.zeros(u, np.uint64), np.zeros(u, np.uint32), np.zeros(u, np.uint32)
        self._check(self._lib.pdl_get_dictionary(self._ctx, ranks.ctypes.data, seqs.ctypes.data, counts.ctypes.data))
        return ranks, seqs, counts

    def rank_table(self):
        tab = np.zeros(256, np.uint8)
        lm = C.c_uint64()
        self._check(self._lib.pdl_get_rank_table(self._ctx, tab.ctypes.data, C.byref(lm)))
        return tab, lm.value

    def timings(self) -> dict:
        return self.timings_struct().as_dict()

    def timings_struct(self):
        """pdl_timings as the ctypes structure (no dictionary is built: for callers inside a timed loop)."""
        t = _lib.PdlTimings()
        self._check(self._lib.pdl_get_timings(self._ctx, C.byref(t)))
        return t
