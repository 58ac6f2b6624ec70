#!/bin/bash
# One gpurun call, several named actions; every action writes gpurun_out/<tag>_<action>.log.
# usage (repo root, on the GPU box): tools/gpu_round.sh <tag> <action> [<action> ...]
#   tests        python -m pytest tests -m gpu -x -q
#   smoke        __graft_entry__.smoke()
#   bench        python3 bench.py (the default line) -> <tag>_bench.json
#   quick        bench.py --quick --steps 10 (headline kernels only)
#   stats        rocprofv3 --kernel-trace --stats of bench.py --quick -> <tag>_kernel_stats.csv
#   traffic      tools/pmc_traffic.sh (TCC FETCH_SIZE / WRITE_SIZE passes at the bench's load: PMC_IMAGES, default 384) -> <tag>_pmc_traffic.json
#   sq           tools/pmc_sq.sh (SQ issue / stall counters at the bench's load: PMC_IMAGES, default 384) -> <tag>_pmc_sq.json
#   timing       chain.hip built with -DHM_CHAIN_TIMING: where a wave's cycles go, at full load and for few pictures
#   counts       the same build with -DHM_CHAIN_TIMING=2: events per wave instead of cycles (service phases, CTU flushes, window
#                take-overs, 4x4 passes, wave-wide blocks, iterations)
#   probes       VARIANTS / OBJ / MODE of tools/probe_chain.sh from the environment
#   classes      tools/bench_classes.py (picture classes x batch sizes)
#   sh:<file>    any other script, run with bash
tag=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
for a in "$@"; do
  log=gpurun_out/${tag}_${a//[:\/]/_}.log
  echo "=== $a ($(date +%T))"
  case $a in
  tests)   timeout 1500 python3 -m pytest tests -m gpu -x -q > $log 2>&1; tail -3 $log ;;
  smoke)   timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $log 2>&1; tail -2 $log ;;
  bench)   timeout 900 python3 bench.py > gpurun_out/${tag}_bench.json 2> $log; tail -c 600 gpurun_out/${tag}_bench.json ;;
  quick)   timeout 600 python3 bench.py --quick --steps 10 2> $log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k: v['ms_per_step'] for k, v in d['kernels'].items()}, d['roofline']['frac'])" ;;
  stats)   rm -rf /tmp/prof_$tag; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag --output-format csv -- python3 $OLDPWD/bench.py --quick --no-parity --steps 10 > $OLDPWD/$log 2>&1)
           f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_kernel_stats.csv && head -8 $f ;;
  traffic) tools/pmc_traffic.sh $tag ${PMC_IMAGES:-384} > $log 2>&1; tail -30 $log ;;
  sq)      tools/pmc_sq.sh $tag --steps 3 --warmup 1 --images ${PMC_IMAGES:-384} > $log 2>&1; tail -5 $log ;;
  timing)  (cd heif-decoder-lib_amd/csrc && rm -f build/hip_chain.o && make HIPFLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -I../../include -I. -I/opt/rocm/include -DHM_CHAIN_TIMING" >/dev/null 2>&1)
           { echo "== full load (384 images)"; HM_CHAIN_TIMING_PRINT=1 timeout 600 python3 bench.py --quick --no-parity --steps 3 2>&1 | grep "k_chain phases" | tail -2
             echo "== few pictures"; HM_CHAIN_TIMING_PRINT=1 timeout 600 python3 tools/few_pictures_probe.py 2>&1 | grep "k_chain phases" | sort | uniq -c | sort -rn | head -6; } > $log 2>&1
           (cd heif-decoder-lib_amd/csrc && rm -f build/hip_chain.o && make >/dev/null 2>&1); cat $log ;;
  counts)  (cd heif-decoder-lib_amd/csrc && rm -f build/hip_chain.o && make HIPFLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -I../../include -I. -I/opt/rocm/include -DHM_CHAIN_TIMING=2" >/dev/null 2>&1)
           { echo "== full load (384 images): service phases, CTU flushes, window take-overs, 4x4 passes, wave-wide blocks, iterations"; HM_CHAIN_TIMING_PRINT=1 timeout 600 python3 bench.py --quick --no-parity --steps 3 2>&1 | grep "k_chain phases" | tail -1; } > $log 2>&1
           (cd heif-decoder-lib_amd/csrc && rm -f build/hip_chain.o && make >/dev/null 2>&1); cat $log ;;
  probes)  tools/probe_chain.sh > $log 2>&1; cat $log ;;
  classes) { echo "== 1536 tiles per class"; timeout 1200 python3 tools/bench_classes.py 2>/dev/null | tr -d '\n' | sed 's/},/},\n/g'; echo
             echo "== 18432 tiles per class"; HM_CLASS_TILES=18432 timeout 1200 python3 tools/bench_classes.py 2>/dev/null | tr -d '\n' | sed 's/},/},\n/g'; echo; } > $log 2>&1; cat $log ;;
  sh:*)    bash ${a#sh:} > $log 2>&1; tail -30 $log ;;
  *)       echo "unknown action $a" ;;
  esac
done
