"""`.faa` ingest — host-side mirror of ``ig/infoasys/cli/pangenes/PangeneIData.java``.

``PangeneIData.read_from_file`` follows ``readFromFile`` (PangeneIData.java:30-75): lines are
trimmed (Java ``String.trim``: every char <= U+0020 at either end), blank lines are skipped,
the remaining lines alternate header / sequence, a header is ``genome<TAB>gene<TAB>product``
and genome ids are dense integers in first-seen order.  On top of the Java fields it keeps the
flattened form the C ABI takes (``residues`` / ``offsets`` / ``genome_of``).
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from typing import List

import numpy as np

_JAVA_TRIM = bytes(range(0x21))  # String.trim() strips chars <= ' '
_READLINE = re.compile(rb"\r\n|\r|\n")


@dataclass
class PangeneIData:
    sequences: List[bytes] = field(default_factory=list)            # PangeneIData.java:13
    sequenceName: List[str] = field(default_factory=list)           # :14
    sequenceDescription: List[str] = field(default_factory=list)    # :15
    sequenceGenome: List[int] = field(default_factory=list)         # :17
    genomeNames: List[str] = field(default_factory=list)            # :18

    @staticmethod
    def read_from_file(path) -> "PangeneIData":
        d = PangeneIData()
        genome_id: dict = {}
        name_line = True
        genome_name = seq_name = product = None
        with open(path, "rb") as f:
            raw_lines = _READLINE.split(f.read())                # BufferedReader.readLine: \n, \r and \r\n end a line
        if raw_lines and raw_lines[-1] == b"":
            raw_lines.pop()
        for raw in raw_lines:
            line = raw.strip(_JAVA_TRIM)
            if not line:
                continue
            if name_line:
                cc = line.decode("latin-1").split("\t")
                # Java indexes cc[1], cc[2] unconditionally (PangeneIData.java:49-51)
                genome_name, seq_name, product = cc[0], cc[1], cc[2]
            else:
                d.sequences.append(line)
                d.sequenceName.append(seq_name)
                gid = genome_id.get(genome_name)
                if gid is None:
                    gid = len(genome_id)
                    genome_id[genome_name] = gid
                d.sequenceGenome.append(gid)
                d.sequenceDescription.append(product)
            name_line = not name_line
        d.genomeNames = [None] * len(genome_id)
        for name, gid in genome_id.items():
            d.genomeNames[gid] = name
        return d

    @staticmethod
    def from_arrays(residues, offsets, genome_of) -> "PangeneIData":
        d = PangeneIData()
        res = np.asarray(residues, dtype=np.uint8).tobytes()
        off = np.asarray(offsets, dtype=np.uint64)
        for i in range(len(genome_of)):
            d.sequences.append(res[int(off[i]):int(off[i + 1])])
            d.sequenceName.append(f"s{i}")
            d.sequenceDescription.append("")
            d.sequenceGenome.append(int(genome_of[i]))
        ng = (max(d.sequenceGenome) + 1) if d.sequenceGenome else 0
        d.genomeNames = [f"G{g}" for g in range(ng)]
        return d

    @staticmethod
    def scan(path) -> dict:
        """The library's own parser (``pdl_scan_faa``: the one ``pdl_ingest_faa`` streams to the device with) run on the
        host only -> flattened arrays + sizes + the k ``calculate_k.py`` prints for the file.  Needs no GPU."""
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        ing = _lib.PdlIngest()
        rc = lib.pdl_scan_faa(str(path).encode(), C.byref(ing), None, 0, None, None, 0)
        if rc != _lib.PDL_OK:
            raise _lib.PdlError(rc, lib.pdl_last_error(None).decode())
        res = np.zeros(ing.residues, np.uint8); off = np.zeros(ing.sequences + 1, np.uint64); gen = np.zeros(ing.sequences, np.uint32)
        rc = lib.pdl_scan_faa(str(path).encode(), C.byref(ing), res.ctypes.data, len(res), off.ctypes.data, gen.ctypes.data, len(gen))
        if rc != _lib.PDL_OK:
            raise _lib.PdlError(rc, lib.pdl_last_error(None).decode())
        return {"residues": res, "offsets": off, "genome_of": gen, "sequences": ing.sequences, "genomes": ing.genomes,
                "k_suggested": ing.k_suggested, "file_bytes": ing.file_bytes}

    # -- flattened form for the C ABI ----------------------------------------------------------
    def flatten(self):
        n = len(self.sequences)
        offsets = np.zeros(n + 1, np.uint64)
        if n:
            np.cumsum(np.fromiter((len(s) for s in self.sequences), np.uint64, n), out=offsets[1:])
        residues = np.frombuffer(b"".join(self.sequences), dtype=np.uint8).copy()
        genome_of = np.asarray(self.sequenceGenome, dtype=np.uint32)
        return residues, offsets, genome_of
