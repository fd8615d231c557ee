#!/usr/bin/env python3
"""Where the wall time of one multi-GPU step goes on the host side: the torch.distributed driver (pandelos_amd/distributed.py)
under RCCL with a group of ONE on cuda:0 — every collective is issued, nothing travels — with the library calls, the small
all-gathers, the all-to-alls, the allocations and the views of library memory timed one by one (each bracketed by a device
synchronisation, so the sum is more than the step).   usage: python tools/dist_driver_profile.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MASTER_ADDR"]="127.0.0.1"; os.environ["MASTER_PORT"]="29533"
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
from pandelos_amd.distributed import DistributedPangenes, device_view
from pandelos_amd.pangene_native import PangeneNative
from pandelos_amd.synth import CONFIGS, make_gene_set
from pandelos_amd.calculate_k import calculate_k
gs = make_gene_set(**CONFIGS["mycoplasma64_standin"]); k = calculate_k(gs.residues)
dev = torch.device("cuda",0)
pad = (-len(gs.residues)) % 16 + 16
t_res = torch.from_numpy(np.concatenate([gs.residues, np.zeros(pad, np.uint8)])).to(dev)
t_off = torch.from_numpy(gs.offsets.astype(np.int64)).to(dev); t_gen = torch.from_numpy(gs.genome_of.astype(np.int32)).to(dev)
dp = DistributedPangenes(PangeneNative.open(), dev, True)
import pandelos_amd.distributed as D
orig_view = D.device_view
def timed(name, fn):
    def w(*a, **kw):
        torch.cuda.synchronize(); t=time.perf_counter(); r=fn(*a, **kw); torch.cuda.synchronize(); acc[name]=acc.get(name,0)+time.perf_counter()-t; return r
    return w
acc={}
D.device_view = timed("device_view", orig_view)
dp._all_to_all_rows = timed("a2a_rows", dp._all_to_all_rows)
dp._all_gather_runs = timed("gather_runs", dp._all_gather_runs)
dp.nat.dist_preprocess_ranges = timed("lib_ranges", dp.nat.dist_preprocess_ranges)
dp.nat.dist_preprocess_finish_ranges = timed("lib_finish", dp.nat.dist_preprocess_finish_ranges)
dp.nat.dist_preprocess_begin = timed("lib_begin", dp.nat.dist_preprocess_begin)
dp.nat.copy_device = timed("copy_device", dp.nat.copy_device)
oe = torch.empty
def te(*a, **kw):
    t=time.perf_counter(); r=oe(*a, **kw); acc["torch.empty"]=acc.get("torch.empty",0)+time.perf_counter()-t; return r
torch.empty = te
oag = dist.all_gather_into_tensor
dist.all_gather_into_tensor = timed("all_gather_small", oag)
dp.nat.dist_score_begin = timed('lib_score_begin', dp.nat.dist_score_begin)
dp.nat.dist_score_finish = timed('lib_score_finish', dp.nat.dist_score_finish)
for i in range(8):
    acc.clear(); torch.cuda.synchronize(); t=time.perf_counter()
    dp.preprocess(k, t_res, t_off, t_gen, gs.genes, len(gs.residues)); torch.cuda.synchronize()
    tp=time.perf_counter()-t
    dp.score_all(); torch.cuda.synchronize()
    tot=time.perf_counter()-t
    print("   preprocess %.2f ms score %.2f ms"%(tp*1e3,(tot-tp)*1e3))
    print(i, "total %.2f ms"%(tot*1e3), {n: round(v*1e3,3) for n,v in acc.items()}, {n: round(v*1e3,3) for n,v in dp.exchange_s.items()}, flush=True)
dist.destroy_process_group()
