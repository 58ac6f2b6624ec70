#!/bin/bash
# ThreadSanitizer build of the host parser (the WPP-row and the tile-row parallel entropy decode) driven with a few synthetic
# pictures on 4 threads.  CPU only.  usage (repo root): tools/tsan/run.sh
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${TMPDIR:-/tmp}/hm_tsan
mkdir -p "$OUT"
C=$ROOT/heif-decoder-lib_amd/csrc
python3 - "$OUT" <<'PY'
import sys
sys.path.insert(0, sys.argv[0] and "."); sys.path.insert(0, "tests")
import synthutil, corpus
out = sys.argv[1]
n = 0
for seed, kw in corpus.structure_sweep(12, first_seed=9700):
    for extra in (dict(wpp=0, tile_cols=2, tile_rows=3), dict(wpp=1, tile_cols=1, tile_rows=1)):
        try:
            open(f"{out}/p{n}.hevc", "wb").write(synthutil.picture(seed, **dict(kw, **extra)))
            n += 1
        except RuntimeError:
            pass
print(n, "pictures")
PY
g++ -std=c++17 -O1 -g -fsanitize=thread -I"$ROOT/include" -I"$C" -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ "$ROOT/tools/tsan/driver.cpp" \
    "$C/hevc_parse.cpp" "$C/hevc_headers.cpp" "$C/common.cpp" -o "$OUT/drv_tsan" -lpthread -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
"$OUT/drv_tsan" 4 "$OUT"/p*.hevc 2>&1 | grep -E "WARNING|SUMMARY|parsed" | sort | uniq -c
