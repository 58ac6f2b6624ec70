#!/bin/bash
# ASan + UBSan build of the host-side parsers and a mutation run over the committed test files (CPU only).
set -e
cd "$(dirname "$0")/../.."
S=heif-decoder-lib_amd/csrc
OUT=${TMPDIR:-/tmp}/hm_asan_fuzz
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined \
    -Iinclude -I$S -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
    tools/asan/fuzz_host.cpp $S/heif_file.cpp $S/hevc_headers.cpp $S/hevc_parse.cpp $S/stream_check.cpp $S/common.cpp \
    -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lpthread -o $OUT
# seeds: the committed HEIC files + synthesised streams with rare syntax and the range-extension tools (length-prefixed NALs)
SEEDS=${TMPDIR:-/tmp}/hm_asan_seeds
mkdir -p $SEEDS
python3 - "$SEEDS" <<'PY'
import sys
sys.path.insert(0, "tests")
import corpus
for name in ("pcm_bypass_sl_wpp", "slices_headers", "wpp_tiles_slices", "rext_ts_tools", "rext_ts_bypass_422_10", "rext_nosmooth_rice",
             "rext_chroma_qp_list6_422", "rext_cross_444_all", "rext_mono_rice_rdpcm"):
    open(f"{sys.argv[1]}/{name}.lp", "wb").write(corpus.stream(name))
PY
ASAN_OPTIONS=detect_leaks=0 $OUT tests/data/colors-no-alpha.heic tests/data/colors-with-alpha.heic tests/data/example.heic $SEEDS/*.lp "$@"
