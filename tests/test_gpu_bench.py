"""bench.py is what the driver runs: ONE JSON line with the contract's fields, on one GPU and — started by the script itself when
no launcher did — on N ranks (here: two ranks rehearsed on the one device over gloo, which walks the same N > 1 flow: senders'
range tuples, the three exchanges, per-rank results reduced)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
ROOT = Path(__file__).resolve().parents[1]

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline")


def _bench(*args):
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip() and not l.startswith("[Gloo]")]      # (the rehearsal's transport announces itself on stdout)
    assert len(lines) == 1, f"bench.py must print ONE line, got {len(lines)}"
    return json.loads(lines[0])


def test_default_run_prints_the_contract_line():
    d = _bench()
    for k in CONTRACT + ("cpu_baseline", "value_host_to_host", "pipeline", "stage_ms", "roofline_whole_path"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"].startswith("synthetic")
    assert d["metric"] and d["unit"] == "gene-pairs/s" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 / (d["config"]["genes"] * (d["config"]["genes"] - 1)) - 1) < 1e-6      # value = pairs / step time
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.05 < r["frac"] < 1.0 and (r["traffic"] is None or r["traffic"] > r["bytes_per_launch"] * 0.5)
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["stage_ms"]["aside_reloads"] == 0 and d["value_host_to_host"] < d["value"]
    assert d["pipeline"]["faa_to_net_ms"] > d["pipeline"]["dictionary_ms"] > 0


def test_two_ranks_print_the_contract_line():
    d = _bench("--gpus", "2", "--steps", "3", "--warmup", "1")
    for k in CONTRACT:
        assert k in d, k
    import torch
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert bool(d.get("rehearsal", False)) == (torch.cuda.device_count() < 2)        # (a box with two GPUs runs the two ranks over RCCL for real)
    st = d["stage_ms"]
    assert st["range_lists_by"] == "senders" and st["dist_ranges"] > 0 and st["dist_finish"] > 0 and st["aside_reloads"] == 0
    assert d["scale_set"]["n_gpus"] == 2 and d["scale_set"]["emitted_cells"] == 55236320       # (configs[3]: the reference's count, summed over the ranks)
