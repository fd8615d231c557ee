#!/usr/bin/env python3
"""Per-rank device time of a W-rank run, measured on ONE MI355X.

The multi-GPU path (include/pandelos_amd.h, pdl_dist_*) is four library calls per rank with two exchanges in between.
This tool plays every rank of a W-rank job in turn on a single context (so even the 512-genome set fits: one rank's
buffers at a time), with device copies in place of the collectives, and reports for every rank the device time of
each call (HIP events inside the library) and the bytes it would send and receive:

  pass 1   every rank: pdl_dist_preprocess_begin                 -> record counts, genome weights and costs
  pass 2   every rank: begin, ranges                              -> its run and its range tuples are kept (the "all-gather", the "all-to-all")
  pass 3   every rank: begin, ranges, finish, score_begin        -> its outbox is kept
  pass 4   every rank: begin, ranges, finish, score_begin, score_finish (inbox = what the others listed for it)
           [--check: every genome's Scores block against tests/golden/digests_baseline.json]

Projection printed with the measurements: step(W) = slowest rank's device time + the two exchanges at --link-gbps per
xGMI link (every peer's run / cells arrive over that peer's own link, so the time is the largest single message).
W = 1 is the ordinary single-GPU path (pdl_preprocess_device + pdl_score_all).

usage: python tools/shard_step_time.py --config synthetic_128x4000x300 --world 1 2 4 8 [--check] [--out FILE]
"""
from __future__ import annotations

import argparse
import hashlib
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="mycoplasma64_standin")
    ap.add_argument("--world", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--repeat", type=int, default=2, help="timed repetitions of every call (the fastest counts)")
    ap.add_argument("--link-gbps", type=float, default=100.0, help="effective one-way rate of one xGMI link for the projection (peak ~153)")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--ranges", choices=["sender", "owner"], default="sender", help="who builds the range lists (owner: every rank from the gathered dictionary, the flow of round 2)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    import torch
    from pandelos_amd import _lib
    from pandelos_amd.calculate_k import calculate_k
    from pandelos_amd.distributed import exclusive_offsets
    from pandelos_amd.pangene_native import PangeneNative
    from pandelos_amd.synth import CONFIGS, make_gene_set

    dev = torch.device("cuda", 0)
    gs = make_gene_set(**CONFIGS[args.config])
    k = calculate_k(gs.residues)
    pad = (-len(gs.residues)) % 16 + 16
    t_res = torch.from_numpy(np.concatenate([gs.residues, np.zeros(pad, np.uint8)])).to(dev)
    t_off = torch.from_numpy(gs.offsets.astype(np.int64)).to(dev)
    t_gen = torch.from_numpy(gs.genome_of.astype(np.int32)).to(dev)
    n, n_res = gs.genes, len(gs.residues)
    digests = None
    if args.check:
        from tests import helpers as H
        digests = json.loads((H.GOLDEN / "digests_baseline.json").read_text()).get(args.config)
        if digests is None:
            print(f"--check: no reference digests for {args.config}", file=sys.stderr)

    nat = PangeneNative.open()
    report = {"workload": args.config, "genes": n, "genomes": gs.genomes, "k": int(k), "link_gbps_assumed": args.link_gbps, "worlds": {}}

    def begin(W, r):
        return nat.dist_preprocess_begin(k, t_res.data_ptr(), t_off.data_ptr(), t_gen.data_ptr(), n, n_res, W, r)

    for W in args.world:
        if W == 1:
            best = None
            for _ in range(args.repeat + 1):
                nat.preprocess_device(k, t_res.data_ptr(), t_off.data_ptr(), t_gen.data_ptr(), n, n_res)
                nat.score_all()
                tm = nat.timings()
                tot = tm["preprocess_total_ms"] + tm["score_total_ms"]
                if best is None or tot < best["device_ms"]:
                    best = {"device_ms": tot, "preprocess_ms": tm["preprocess_total_ms"], "score_ms": tm["score_total_ms"], "join_ms": tm["join_ms"],
                            "order_ms": tm["order_ms"], "walked_lookups": tm["walked_lookups"]}
            report["worlds"]["1"] = {"ranks": [best], "slowest_device_ms": best["device_ms"], "projected_step_ms": best["device_ms"], "speedup": 1.0}
            print(f"W=1: {best['device_ms']:.3f} ms (preprocess {best['preprocess_ms']:.3f}, score {best['score_ms']:.3f})", flush=True)
            continue
        # ---- pass 1: record counts, genome weights and costs --------------------------------------------------------------
        runs, begin_ms = [], []
        for r in range(W):
            t = []
            for _ in range(args.repeat):
                ptr, rec, kmers = begin(W, r)
                t.append(nat.timings()["dist_begin_ms"])
            runs.append((rec, kmers))
            begin_ms.append(min(t))
            weights = nat.run_weights.copy() if r == 0 else weights + nat.run_weights
            costs = nat.run_costs.copy() if r == 0 else costs + nat.run_costs
        records = [rec for rec, _ in runs]
        offs = exclusive_offsets(records)
        total = int(offs[-1])
        # ---- pass 2: the range tuples of every run (sender flow), the runs kept -------------------------------------------
        sender = args.ranges == "sender"
        saved, keys, rngs, tcounts, ctrs, ranges_ms = [], [], [], [], [], []
        for r in range(W):
            t = [0.0]
            ptr, rec, _ = begin(W, r)
            if sender:
                made = nat.dist_preprocess_ranges(records, weights, costs)
                t = [nat.timings()["dist_ranges_ms"]]
                for _ in range(args.repeat - 1):
                    ptr, rec, _ = begin(W, r)
                    made = nat.dist_preprocess_ranges(records, weights, costs)
                    t.append(nat.timings()["dist_ranges_ms"])
                if made is None:
                    sender = False
                else:
                    n_t = int(made[2].sum())
                    kt = torch.empty(max(n_t, 1), dtype=torch.int32, device=dev)
                    rt = torch.empty(max(n_t, 1), dtype=torch.int64, device=dev)
                    if n_t:
                        nat.copy_device(kt.data_ptr(), made[0], n_t * 4)
                        nat.copy_device(rt.data_ptr(), made[1], n_t * 8)
                    keys.append(kt[:n_t]); rngs.append(rt[:n_t]); tcounts.append(made[2]); ctrs.append(made[3])
            ranges_ms.append(min(t))
            run_t = torch.empty(max(rec, 1), dtype=torch.int64, device=dev)      # (the next begin overwrites the context's run)
            if rec:
                nat.copy_device(run_t.data_ptr(), ptr, rec * 8)
            saved.append(run_t[:rec])
        full0 = torch.cat(saved) if total else torch.zeros(1, dtype=torch.int64, device=dev)
        del saved
        full = full0 if sender else torch.empty_like(full0)
        if sender:
            tmat, sums = np.stack(tcounts), np.sum(ctrs, axis=0)

        def tuples_for(r):
            n_in = int(tmat[:, r].sum())
            rk = torch.empty(max(n_in, 1), dtype=torch.int32, device=dev)
            rr = torch.empty(max(n_in, 1), dtype=torch.int64, device=dev)
            at = 0
            for s_ in range(W):
                c = int(tmat[s_, r])
                if c:
                    o = int(tmat[s_, :r].sum())
                    rk[at:at + c] = keys[s_][o:o + c]
                    rr[at:at + c] = rngs[s_][o:o + c]
                    at += c
            return rk, rr, n_in

        def upto_finish(r):
            begin(W, r)
            if sender:
                nat.dist_preprocess_ranges(records, weights, costs)
                rk, rr, n_in = tuples_for(r)               # (the sort works in these buffers: a fresh copy per repetition)
                torch.cuda.synchronize()
                nat.dist_preprocess_finish_ranges(full.data_ptr(), total, rk.data_ptr(), rr.data_ptr(), n_in, sums, keepalive=(full, rk, rr))
            else:
                full.copy_(full0)
                torch.cuda.synchronize()
                nat.dist_preprocess_finish(full.data_ptr(), total, genome_weights=weights)

        # ---- pass 3: finish + score_begin, outboxes kept --------------------------------------------------------------
        finish_ms, sbegin_ms, outbox, out_counts, ranks_info = [], [], [], [], []
        owner = None
        for r in range(W):
            tf, ts = [], []
            for _ in range(args.repeat):
                upto_finish(r)
                tf.append(nat.timings()["dist_finish_ms"])
                ptr, counts = nat.dist_score_begin(W)
                ts.append(nat.timings()["dist_score_begin_ms"])
            if owner is None:
                owner = nat.dist_genome_owner()
            tm = nat.timings()
            n_out = int(counts.sum())
            box = torch.empty((max(n_out, 1), 6), dtype=torch.int32, device=dev)
            if n_out:
                nat.copy_device(box.data_ptr(), ptr, n_out * _lib.DIST_CELL_BYTES)
            outbox.append(box); out_counts.append(counts)
            finish_ms.append(min(tf)); sbegin_ms.append(min(ts))
            ranks_info.append({"rank": r, "genomes": int((owner == r).sum()), "run_records": runs[r][0], "run_kmers": runs[r][1],
                               "rows": int(tm["scored_rows"]), "walked_lookups": int(tm["walked_lookups"]), "join_ms": tm["join_ms"],
                               "sort_rank_ms": tm["sort_rank_ms"], "rank_ms": tm["rank_ms"], "sort_seq_ms": tm["sort_seq_ms"], "ranges_ms": tm["ranges_ms"],
                               "outbox_cells": n_out, "aside_repeats": int(tm["aside_repeats"]), "aside_reloads": int(tm["aside_reloads"]), "tier1_rows": int(tm["tier1_rows"])})
            if sender:
                ranks_info[-1].update({"tuples_made": int(tmat[r].sum()), "tuples_received": int(tmat[:, r].sum())})
        cmat = np.stack(out_counts)                  # [src][dst]
        # ---- pass 4: score_finish (+ check) -------------------------------------------------------------------------------
        sfinish_ms = []
        ok = True
        for r in range(W):
            n_in = int(cmat[:, r].sum())
            inbox = torch.empty((max(n_in, 1), 6), dtype=torch.int32, device=dev)
            at = 0
            for s in range(W):
                c = int(cmat[s, r])
                if c:
                    o = int(cmat[s, :r].sum())
                    inbox[at:at + c] = outbox[s][o:o + c]
                    at += c
            t = []
            for _ in range(args.repeat):
                upto_finish(r)
                nat.dist_score_begin(W)
                nat.dist_score_finish(inbox.data_ptr(), n_in)
                t.append(nat.timings()["dist_score_finish_ms"])
            tm = nat.timings()
            sfinish_ms.append(min(t))
            ranks_info[r].update({"inbox_cells": n_in, "order_ms": tm["order_ms"], "emitted_cells": int(tm["emitted_cells"])})
            if digests is not None:
                from tests import helpers as H
                for g in np.nonzero(owner == r)[0]:
                    got = nat.generate_scores_part(int(g)).as_dict()
                    if int(got["scoresCount"]) != digests["scoresCount"][g]:
                        ok = False
                    for f in H.FIELDS:
                        if hashlib.sha256(H.raw(got[f]).tobytes()).hexdigest() != digests["sha256"][g][f]:
                            ok = False
        link = args.link_gbps * 1e9
        for r in range(W):
            info = ranks_info[r]
            info.update({"begin_ms": begin_ms[r], "ranges_ms_call": ranges_ms[r], "finish_ms": finish_ms[r], "score_begin_ms": sbegin_ms[r], "score_finish_ms": sfinish_ms[r]})
            info["device_ms"] = begin_ms[r] + ranges_ms[r] + finish_ms[r] + sbegin_ms[r] + sfinish_ms[r]
            # what arrives at rank r: every other run over that peer's link; every peer's cells over that peer's link
            info["dictionary_recv_bytes"] = int((total - runs[r][0]) * 8)
            info["cells_recv_bytes"] = int(ranks_info[r]["inbox_cells"] * _lib.DIST_CELL_BYTES)
        biggest_run = max(rec for rec, _ in runs) * 8
        biggest_cells = int(cmat.max()) * _lib.DIST_CELL_BYTES
        biggest_tuples = int(max(tmat[s_, d] for s_ in range(W) for d in range(W) if s_ != d)) * 12 if sender and W > 1 else 0
        xd, xc, xt = biggest_run / link * 1e3, biggest_cells / link * 1e3, biggest_tuples / link * 1e3
        # sender flow: the owners' finish does not read the dictionary, so the gather of the runs proceeds beside it — what is left
        # of the gather once the slowest finish is over counts
        xd_exposed = max(0.0, xd - max(finish_ms)) if sender else xd
        slow = max(i["device_ms"] for i in ranks_info)
        w1 = report["worlds"].get("1", {}).get("slowest_device_ms")
        proj = slow + xt + xd_exposed + xc
        entry = {"ranks": ranks_info, "range_lists_by": "senders" if sender else "owners", "slowest_device_ms": slow,
                 "exchange_dictionary_ms_model": xd, "exchange_dictionary_ms_model_beside_finish": xd_exposed,
                 "exchange_ranges_ms_model": xt, "exchange_cells_ms_model": xc,
                 "dictionary_bytes_total": total * 8, "range_tuple_bytes_total": int(tmat.sum()) * 12 if sender else 0, "cells_exchanged_total": int(cmat.sum()),
                 "projected_step_ms": proj, "speedup": (w1 / proj) if w1 else None,
                 "speedup_device_only": (w1 / slow) if w1 else None,
                 "note": "per-rank device times measured on ONE GPU; exchanges are a model (largest single message per link at the assumed rate): a projection, unmeasured on N GPUs"}
        if digests is not None:
            entry["matches_reference_digests"] = ok
        report["worlds"][str(W)] = entry
        print(f"W={W} ({entry['range_lists_by']}): slowest rank {slow:.3f} ms device (begin {max(begin_ms):.3f}, ranges {max(ranges_ms):.3f}, finish {max(finish_ms):.3f}, "
              f"score_begin {max(sbegin_ms):.3f}, score_finish {max(sfinish_ms):.3f}); exchanges ~{xt:.3f} (tuples) + {xd_exposed:.3f} of {xd:.3f} (runs) + {xc:.3f} (cells) ms at {args.link_gbps:.0f} GB/s/link; "
              f"projected speed-up {entry['speedup'] if entry['speedup'] else float('nan'):.2f}"
              + (f"; digests {'OK' if ok else 'MISMATCH'}" if digests is not None else ""), flush=True)
        del full, full0, outbox, keys, rngs
        torch.cuda.empty_cache()
    nat.close()
    text = json.dumps(report, indent=1)
    if args.out:
        Path(args.out).write_text(text + "\n")
    else:
        print(text)


if __name__ == "__main__":
    main()
