#!/bin/bash
# The host side of the .faa ingest (pdl_ingest.hip: chunks at line starts, a team of threads, three passes) under the sanitizers
# of the ROCm clang, on the CPU build only (the GPU pool takes no sanitizer runs):
#   1. AddressSanitizer + UBSan: libpandelos_amd.so with an instrumented pdl_ingest.o, the CPU tests of tests/test_ingest.py
#   2. ThreadSanitizer: a small driver calling pdl_scan_faa (count pass + fill pass) on two messy 3-MB files, six times
# usage: bash tools/sanitize_ingest.sh        (needs a built pandelos_amd/lib/obj; leaves the shipped library as it was)
set -e
cd "$(dirname "$0")/.."
W=/tmp/pdl_sanitize; rm -rf $W; mkdir -p $W
OBJS=""; for f in pdl_sort pdl_dict pdl_join pdl_bbh pdl_api; do OBJS="$OBJS pandelos_amd/lib/obj/$f.o"; done
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fvisibility=hidden -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer \
    -c pandelos_amd/csrc/pdl_ingest.hip -o $W/ing_asan.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -o $W/libpandelos_amd_asan.so $OBJS $W/ing_asan.o
cp pandelos_amd/lib/libpandelos_amd.so $W/shipped.so
trap 'cp $W/shipped.so pandelos_amd/lib/libpandelos_amd.so' EXIT
cp $W/libpandelos_amd_asan.so pandelos_amd/lib/libpandelos_amd.so
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python3 -m pytest tests/test_ingest.py -q -m "not gpu" -x
cp $W/shipped.so pandelos_amd/lib/libpandelos_amd.so
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from tests.test_ingest import _messy_faa
open("/tmp/pdl_sanitize/messy1.faa", "wb").write(_messy_faa(1, 15000))
open("/tmp/pdl_sanitize/messy2.faa", "wb").write(_messy_faa(2, 16000))
PY
cat > $W/drv.cpp <<'CPP'
#include "pandelos_amd.h"
#include <cstdio>
#include <vector>
int main(int argc, char **argv) {
    for (int rep = 0; rep < 6; rep++) for (int i = 1; i < argc; i++) {
        pdl_ingest a{}, b{};
        if (pdl_scan_faa(argv[i], &a, nullptr, 0, nullptr, nullptr, 0) != 0) { printf("count pass failed: %s\n", pdl_last_error(nullptr)); return 1; }
        std::vector<uint8_t> res(a.residues + 64); std::vector<uint64_t> off(a.sequences + 1); std::vector<uint32_t> gen(a.sequences);
        if (pdl_scan_faa(argv[i], &b, res.data(), res.size(), off.data(), gen.data(), a.sequences) != 0) { printf("fill pass failed: %s\n", pdl_last_error(nullptr)); return 1; }
        if (rep == 0) printf("%s: %u sequences, %llu residues, %u genomes, k %d\n", argv[i], b.sequences, (unsigned long long) b.residues, b.genomes, b.k_suggested);
    }
    return 0;
}
CPP
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=thread -fno-gpu-sanitize -c pandelos_amd/csrc/pdl_ingest.hip -o $W/ing_tsan.o 2>/dev/null
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=thread -Iinclude -c $W/drv.cpp -o $W/drv.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=thread $W/drv.o $W/ing_tsan.o $OBJS -o $W/drv_tsan -lpthread 2>/dev/null
TSAN_OPTIONS=halt_on_error=1 $W/drv_tsan $W/messy1.faa $W/messy2.faa
echo "sanitizers: no report"
