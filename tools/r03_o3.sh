#!/bin/bash
# host objects at -O2 (default) / -O3: parse rate on this box's CPU
mkdir -p gpurun_out
cd heif-decoder-lib_amd/csrc
{
for o in ${OPTS:-"-O2" "-O3" "-O2"}; do
  rm -f build/hevc_parse.o build/hevc_headers.o
  make CXXFLAGS="-std=c++17 ${o//+/ } -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -Wno-unused-result -I../../include -I. -I/opt/rocm/include -D__HIP_PLATFORM_AMD__" >/dev/null 2>&1
  echo "== $o"; (cd ../.. && for i in 1 2 3; do python3 tools/parse_bench.py 48 9; done)
done
} > ../../gpurun_out/r03_o3.log 2>&1
rm -f build/hevc_parse.o build/hevc_headers.o; make >/dev/null 2>&1
