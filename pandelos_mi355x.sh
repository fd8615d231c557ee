#!/bin/bash
# pandelos_mi355x.sh — same plugin surface as the reference's pandelos.sh (pandelos.sh:44-81):
#     bash pandelos_mi355x.sh <dataset.faa> <out_prefix>      ->   <out_prefix>.clus
# The k selection and the de-clustering stay the reference's scripts (calculate_k.py, netclu_ng.py: set
# PANDELOS_PATH to a PanDelos checkout); the Java stage in the middle (pandelos.sh:73) is replaced by the native
# MI355X host built by `python __graft_entry__.py` (pandelos_amd/lib/pangenes; the JVM route is to point
# -Djava.library.path at pandelos_amd/lib instead).
sdir="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ref="${PANDELOS_PATH:?set PANDELOS_PATH to the PanDelos checkout that holds calculate_k.py and netclu_ng.py}"
idb="$1"; oprefix="$2"
if [ ! -f "$idb" ]; then echo "ERROR: input dataset file not found: $idb !"; echo "usage is: pandelos_mi355x.sh dataset.faa out_prefix"; exit; fi
if [ -z "$oprefix" ]; then echo "ERROR: output prefix not given !"; echo "usage is: pandelos_mi355x.sh dataset.faa out_prefix"; exit; fi
tmp=$(mktemp -p ./ -t "$(basename "$idb" .faa).XXXXXX")
dnet="${tmp}.net"; clus="${oprefix}.clus"
python3 "$ref/calculate_k.py" "$idb" > "$tmp"
k=$(grep -E "^k =" "$tmp" | sed s/k\ =\ //g)
echo "k = $k"
"$sdir/pandelos_amd/lib/pangenes" -i "$idb" -k $k -o "$dnet" > "$tmp"
python3 "$ref/netclu_ng.py" "$idb" "$dnet" >> "$tmp"
grep "F{ " "$tmp" | sed s/F{\ //g | sed s/}//g | sed s/\ \;//g | sort | uniq > "$clus"
rm "$tmp"
echo "Finish!"
