#!/bin/bash
# Round-4 check in one GPU call: (optional) GPU test suite + smoke, A/B bench lines (RIDERS_LAZY_BN=0/1 ...), rocprofv3 kernel statistics.
# Output -> gpurun_out/r04_<tag>/    usage: tools/r04_check.sh <tag> [tests|lazytests|notests] [prof|noprof] [extra bench flags]
tag=${1:-a}; what=${2:-tests}; prof=${3:-prof}; shift; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/r04_$tag; mkdir -p $out
cd $GRAFT_REPO_ROOT
if [ "$what" = "tests" ]; then
  timeout 1500 python -m pytest tests -q -m gpu -x > $out/gpu_tests.log 2>&1
  tail -5 $out/gpu_tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1
  tail -1 $out/smoke.log
elif [ "$what" = "lazytests" ]; then
  timeout 900 python -m pytest tests -q -m gpu -x -k "lazy or decoder or resnet or rcnet_e2e or conv" > $out/gpu_tests.log 2>&1
  tail -5 $out/gpu_tests.log
fi
for i in 1 2; do
  RIDERS_LAZY_BN=${LZA:-0} timeout 600 python bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline --no-sml "$@" 2>$out/bench_base.err | grep metric | sed 's/^/lazy0 /' | cut -c1-260
  RIDERS_LAZY_BN=${LZB:-1} timeout 600 python bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline --no-sml "$@" 2>$out/bench_new.err | grep metric | sed 's/^/lazy1 /' | cut -c1-260
done
timeout 900 python bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline --detail $out/per_shape.txt "$@" > $out/bench.json 2> $out/bench.err
tail -3 $out/bench.err; cut -c1-300 $out/bench.json
if [ "$prof" = "prof" ]; then
  export TMPDIR=/tmp; cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o rc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --settle-seconds 0 --no-cpu-baseline --no-sml --timer-repeat 1 "$@" > $out/prof.log 2>&1
  find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
  rm -rf $out/prof
fi
