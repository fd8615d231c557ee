#!/bin/bash
# pandelos_mi355x.sh — same plugin surface as the reference's pandelos.sh (pandelos.sh:44-81):
#     bash pandelos_mi355x.sh <dataset.faa> <out_prefix>      ->   <out_prefix>.clus
# Three stages, as there: k selection, the gene network (.net), de-clustering into gene families (.clus).
#   * The Java stage in the middle (pandelos.sh:73) is the native MI355X host built by `python __graft_entry__.py`
#     (pandelos_amd/lib/pangenes; the JVM route is to point -Djava.library.path at pandelos_amd/lib instead, see
#     INTEGRATION.md).
#   * With PANDELOS_PATH set to a PanDelos checkout, k selection and de-clustering are the reference's own scripts
#     (calculate_k.py, netclu_ng.py — needs networkx); without it, k comes from the native host's own ingest pass
#     (-k auto: pdl_ingest_faa computes calculate_k.py's value on the way) and de-clustering from pandelos_amd/netclu.py;
#     the tests pin both to the reference's outputs.
sdir="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ref="${PANDELOS_PATH:-}"
idb="$1"; oprefix="$2"
if [ ! -f "$idb" ]; then echo "ERROR: input dataset file not found: $idb !"; echo "usage is: pandelos_mi355x.sh dataset.faa out_prefix"; exit; fi
if [ -z "$oprefix" ]; then echo "ERROR: output prefix not given !"; echo "usage is: pandelos_mi355x.sh dataset.faa out_prefix"; exit; fi
tmp=$(mktemp -p ./ -t "$(basename "$idb" .faa).XXXXXX")
dnet="${tmp}.net"; clus="${oprefix}.clus"
echo "calculating k ..."
if [ -n "$ref" ]; then
    python3 "$ref/calculate_k.py" "$idb" > "$tmp"
    k=$(grep -E "^k =" "$tmp" | sed s/k\ =\ //g)
else
    k=auto      # the native host's ingest pass computes calculate_k.py's value while the file streams to the GPU and prints "k = N"
fi
[ "$k" != auto ] && echo "k = $k"
echo "clustering ..."
"$sdir/pandelos_amd/lib/pangenes" -i "$idb" -k $k -o "$dnet" > "$tmp" || { echo "ERROR: the native stage failed"; cat "$tmp"; rm -f "$tmp" "$dnet"; exit 1; }
[ "$k" = auto ] && grep -E "^k = " "$tmp"
echo "de-clustering ..."
if [ -n "$ref" ]; then
    python3 "$ref/netclu_ng.py" "$idb" "$dnet" >> "$tmp"
    echo "writing gene gene families in $clus ..."
    grep "F{ " "$tmp" | sed s/F{\ //g | sed s/}//g | sed s/\ \;//g | LC_ALL=C sort | uniq > "$clus"
else
    echo "writing gene gene families in $clus ..."
    PYTHONPATH="$sdir" python3 -m pandelos_amd.netclu "$idb" "$dnet" > "$clus"
fi
rm -f "$tmp" "$dnet"
echo "Finish!"
