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
ASAN_OPTIONS=detect_leaks=0 $OUT tests/data/colors-no-alpha.heic tests/data/colors-with-alpha.heic tests/data/example.heic "$@"
