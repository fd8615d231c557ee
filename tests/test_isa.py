"""The emitted gfx950 ISA of the shipped library, checked where two timing-dependent wrong-result bugs were actually seen
(DESIGN.md section 4):

  * "a barrier that did not wait": hipcc leaves `s_waitcnt lgkmcnt(0)` out in front of an `s_barrier` when the wave needs no
    LDS result any more; on MI355X a wave's queued LDS writes can then land behind another wave's reads.  Every workgroup
    barrier of the library is pdl_sync() (pdl_common.h): the wait, then the barrier.  Here: every `s_barrier` of every kernel
    has an `s_waitcnt ... lgkmcnt(0)` in front of it with no LDS instruction and no branch target in between.
  * "sc1 stores overtaken by sc1 loads": the join's put-aside list and HBM tables are written with PLAIN stores (the line stays
    in the XCD's L2) and read with sc1 loads.  Here: no `global_store ... sc1` in k_join_lds / k_join_hbm.

The negative control compiles pdl_join.hip with pdl_sync() reduced to a bare __syncthreads() (-DPDL_PLAIN_SYNCTHREADS) and
must be flagged — so the rule is known to see what it is looking for.  CPU box only: llvm-objdump of /opt/rocm, no GPU."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
LIB = ROOT / "pandelos_amd" / "lib" / "libpandelos_amd.so"
LLVM = Path("/opt/rocm/lib/llvm/bin")
OBJDUMP = LLVM / "llvm-objdump"
HIPCC = Path("/opt/rocm/bin/hipcc")

pytestmark = pytest.mark.skipif(not OBJDUMP.exists(), reason="no llvm-objdump (ROCm LLVM) on this machine")


def _disassemble_code_object(co: Path) -> str:
    return subprocess.run([str(OBJDUMP), "-d", str(co)], check=True, capture_output=True, text=True).stdout


def _disassemble_host_binary(path: Path, tmp: Path) -> str:
    """Every gfx950 code object bundled into a host .so / .o (llvm-objdump --offloading unpacks them beside its input)."""
    work = tmp / path.name
    shutil.copy(path, work)
    subprocess.run([str(OBJDUMP), "--offloading", str(work)], check=True, capture_output=True, cwd=tmp)
    cos = sorted(tmp.glob(path.name + ".*gfx950"))
    assert cos, "no gfx950 code object found in " + str(path)
    return "\n".join(_disassemble_code_object(c) for c in cos)


def _functions(dis: str):
    """-> {symbol: [instruction text, ...]} ; a basic-block label is kept as an entry that ends with ':'"""
    out, cur = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:\s*$", line)
        if m:
            name = m.group(1)
            if re.match(r"^L\d+|^\.L", name):          # (block labels, when the disassembler prints them this way)
                if cur is not None:
                    out[cur].append(name + ":")
                continue
            cur = name
            out[cur] = []
            continue
        if cur is None:
            continue
        t = line.split("//")[0].strip()
        if t:
            out[cur].append(t)
    return out


def barrier_violations(dis: str):
    """[(kernel, index of the s_barrier, why)] for every s_barrier without `s_waitcnt ... lgkmcnt(0)` in front of it (scanning
    back over anything but LDS instructions, labels and branches)."""
    bad = []
    n_barriers = 0
    for fn, ins in _functions(dis).items():
        for i, t in enumerate(ins):
            if not t.startswith("s_barrier"):
                continue
            n_barriers += 1
            why = "start of the kernel reached"
            for j in range(i - 1, -1, -1):
                p = ins[j]
                if p.startswith("s_waitcnt") and "lgkmcnt(0)" in p:
                    why = None
                    break
                if p.startswith("ds_"):
                    why = "LDS instruction `%s` between the last lgkmcnt(0) wait and the barrier" % p
                    break
                if p.endswith(":") or p.startswith("s_cbranch") or p.startswith("s_branch") or p.startswith("s_barrier"):
                    why = "`%s` between the last lgkmcnt(0) wait and the barrier" % p
                    break
            if why:
                bad.append((fn, i, why))
    return bad, n_barriers


def sc1_store_violations(dis: str):
    bad = []
    for fn, ins in _functions(dis).items():
        if "k_join_lds" not in fn and "k_join_hbm" not in fn:
            continue
        bad += [(fn, t) for t in ins if t.startswith("global_store") and re.search(r"\bsc1\b", t)]
    return bad


@pytest.fixture(scope="module")
def shipped(tmp_path_factory):
    assert LIB.exists(), "build the library first (python __graft_entry__.py)"
    return _disassemble_host_binary(LIB, tmp_path_factory.mktemp("isa"))


def test_every_barrier_waits_for_the_waves_lds_operations(shipped):
    bad, n = barrier_violations(shipped)
    assert n > 200, f"only {n} s_barrier found: the disassembly was not read properly"
    assert not bad, "s_barrier without an LDS wait in front:\n" + "\n".join(f"  {fn} @{i}: {why}" for fn, i, why in bad[:20])


def test_no_sc1_store_in_the_join_kernels(shipped):
    fns = [f for f in _functions(shipped) if "k_join_lds" in f or "k_join_hbm" in f]
    assert len(fns) >= 7, fns                      # five first tiers + tier 2 (+ tiny) + two HBM flavours
    bad = sc1_store_violations(shipped)
    assert not bad, "\n".join(f"  {fn}: {t}" for fn, t in bad[:20])
    # ... while the loads that read those stores back ARE sc1 (served by L2, never by L1)
    assert any(t.startswith("global_load") and " sc1" in t for f in fns for t in _functions(shipped)[f])


@pytest.mark.skipif(not HIPCC.exists(), reason="no hipcc")
def test_negative_control_a_bare_syncthreads_is_flagged(tmp_path):
    co = tmp_path / "join_plain.co"
    subprocess.run([str(HIPCC), "--offload-arch=gfx950", "-O3", "-std=c++17", "-DPDL_PLAIN_SYNCTHREADS", "--cuda-device-only",
                    "--no-gpu-bundle-output", "-c", str(ROOT / "pandelos_amd" / "csrc" / "pdl_join.hip"), "-o", str(co)], check=True)
    bad, n = barrier_violations(_disassemble_code_object(co))
    assert n > 50
    assert any("k_join_lds" in fn for fn, _, _ in bad), "the compiler emitted an LDS wait in front of every barrier by itself: the control no longer bites"


@pytest.mark.skipif(not HIPCC.exists(), reason="no hipcc")
def test_diagnostic_builds_compile_and_keep_the_barrier_rule(tmp_path):
    """-DPDL_JOIN_PHASES (phase timers and counts inside the join tiers) and -DPDL_JOIN_CHECK (lane -> range mapping checked
    against the prefix array, wave lag) are how the findings of DESIGN.md section 4 were made: they must keep compiling for
    gfx950, and the barrier rule holds for them too."""
    co = tmp_path / "join_diag.co"
    subprocess.run([str(HIPCC), "--offload-arch=gfx950", "-O3", "-std=c++17", "-DPDL_JOIN_PHASES", "-DPDL_JOIN_CHECK", "--cuda-device-only",
                    "--no-gpu-bundle-output", "-c", str(ROOT / "pandelos_amd" / "csrc" / "pdl_join.hip"), "-o", str(co)], check=True)
    bad, n = barrier_violations(_disassemble_code_object(co))
    assert n > 50 and not bad, bad[:5]
