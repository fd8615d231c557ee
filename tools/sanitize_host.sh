#!/bin/bash
# Host code under the sanitizers, on the CPU build only (the GPU pool takes no sanitizer runs).  The .faa ingest (pdl_ingest.hip:
# chunks at line starts, a team of threads, three passes) under those of the ROCm clang:
#   1. AddressSanitizer + UBSan: libpandelos_amd.so with an instrumented pdl_ingest.o, the CPU tests of tests/test_ingest.py
#   2. ThreadSanitizer: a small driver calling pdl_scan_faa (count pass + fill pass) on two messy 3-MB files, six times
#   3. the native host's network container + .net text (pangenes_main.cpp, struct Net: counting sorts, eight formatting threads) cut
#      out of its source into a harness with 479 k synthetic edges: g++ ASan + UBSan, then ThreadSanitizer
# usage: bash tools/sanitize_host.sh        (needs a built pandelos_amd/lib/obj; leaves the shipped library as it was)
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
python3 - <<'PY'
src = open("pandelos_amd/csrc/pangenes_main.cpp").read()
body = src[src.index("namespace {"):src.index("void usage()")]
open("/tmp/pdl_sanitize/net.cpp", "w").write("""#include <algorithm>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <string>
#include <thread>
#include <vector>
""" + body + """}
int main() {
    const int N = 47891; std::mt19937 rng(7);
    Net net;
    for (int g = 0; g < N; g++) for (int j = 0; j < 5; j++) {
        const int o = (g + (int) (rng() % 64 + 1) * 750) % N; const float sc = (rng() % 100000) / 100000.0f * 0.9f + 0.05f;
        net.add(g, o, sc); net.add(o, g, sc);
    }
    size_t bytes = 0;
    for (int rep = 0; rep < 3; rep++) { bytes = 0; for (const std::string &p : net.text()) bytes += p.size(); }
    printf("net text: %zu bytes\\n", bytes);
    return 0;
}
""")
PY
g++ -O1 -g -std=c++17 -pthread -fsanitize=address,undefined -fno-sanitize-recover=all $W/net.cpp -o $W/net_asan && $W/net_asan
g++ -O1 -g -std=c++17 -pthread -fsanitize=thread $W/net.cpp -o $W/net_tsan && TSAN_OPTIONS=halt_on_error=1 $W/net_tsan
echo "sanitizers: no report"
