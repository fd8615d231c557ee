import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pandelos_amd.pangene_native import PangeneNative
from pandelos_amd.pangene_idata import PangeneIData
from pandelos_amd.synth import CONFIGS, make_gene_set
for name in ("mycoplasma64_standin", "synthetic_128x4000x300"):
    gs = make_gene_set(**CONFIGS[name]); p = f"/tmp/{name}.faa"; gs.write_faa(p)
    nat = PangeneNative.open()
    for it in range(3):
        t0 = time.perf_counter(); ing = nat.ingest_faa(p); t1 = time.perf_counter()
        nat.preprocess_ingested(ing["k_suggested"]); t2 = time.perf_counter()
    t3 = time.perf_counter(); d = PangeneIData.read_from_file(p); r = d.flatten(); t4 = time.perf_counter()
    print(name, "file MB", ing["file_bytes"] / 1e6, "ingest ms", ing["parse_ms"], "(wall", (t1 - t0) * 1e3, ") preprocess_ingested ms", (t2 - t1) * 1e3, "| python reader + flatten ms", (t4 - t3) * 1e3, "k", ing["k_suggested"], flush=True)
    nat.close()
