#!/usr/bin/env python3
"""Idle time of the stream between consecutive kernels / copies of one bench step, from a rocprofv3 trace.
usage: rocprofv3 --kernel-trace --memory-copy-trace -d DIR --output-format csv -- python3 bench.py --steps 5 ...
       python tools/trace_gaps.py DIR [first kernel of a step = k_scan_apply<KseqFlag]"""
import csv, glob, sys
root = sys.argv[1]
ev = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
for f in glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
ev.sort()
# steps: split at the first kernel of the preprocess (K-len scan)
marker = sys.argv[2] if len(sys.argv) > 2 else "KseqFlag"
starts = [i for i, e in enumerate(ev) if marker in e[2] and "tile_sums" in e[2]]
if len(starts) < 3:
    print("no steps found"); sys.exit(1)
a, b = starts[-2], starts[-1]          # the last complete step
step = ev[a:b]
busy = sum(e[1] - e[0] for e in step)
span = ev[b][0] - step[0][0]
print(f"step: {len(step)} activities, span {span/1e3:.1f} us, busy {busy/1e3:.1f} us, idle {(span-busy)/1e3:.1f} us")
gaps = []
for i in range(len(step)):
    nxt = step[i + 1][0] if i + 1 < len(step) else ev[b][0]
    gaps.append((nxt - step[i][1], step[i][2], step[i + 1][2] if i + 1 < len(step) else ev[b][2]))
gaps.sort(reverse=True)
for g, x, y in gaps[:14]:
    print(f"  {g/1e3:7.1f} us  after {x}  before {y}")
small = [g for g, _, _ in gaps if g < 6000]
print(f"  {len(small)} gaps under 6 us: {sum(small)/1e3:.1f} us, mean {sum(small)/max(1,len(small))/1e3:.2f} us")
