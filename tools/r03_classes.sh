#!/bin/bash
# which reconstruction path serves which picture class best now: the default record order (hevc_syntax.h: quad_class) against
# split chains for every class (HM_QUAD_CLASS=1), 512x512 tiles
mkdir -p gpurun_out
{
for n in 1536 18432; do
  echo "== $n tiles, default classes"; HM_CLASS_TILES=$n timeout 900 python3 tools/bench_classes.py 2>&1 | grep -v amdgpu.ids | tr -d '\n' | sed 's/}, /},\n/g'; echo
  echo "== $n tiles, split chains for every class"; HM_QUAD_CLASS=1 HM_CLASS_TILES=$n timeout 900 python3 tools/bench_classes.py 2>&1 | grep -v amdgpu.ids | tr -d '\n' | sed 's/}, /},\n/g'; echo
done
} > gpurun_out/r03_classes.log 2>&1
