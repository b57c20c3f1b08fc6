cd $GRAFT_REPO_ROOT; out=gpurun_out/r05_comm; mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu -x -k "rccl or roi_pool" > $out/tests.log 2>&1; tail -5 $out/tests.log
for c in c_abi torch; do
 for i in 1 2; do
  timeout 600 python bench.py --gpus 1 --steps 100 --force-ddp --comm $c --no-sml --no-legs --no-cpu-baseline --full-json $out/full_$c.json 2>$out/err_$c.log | cut -c1-330
 done
done
timeout 600 python bench.py --gpus 1 --steps 100 --no-sml --no-legs --no-cpu-baseline --full-json $out/full_plain.json 2>$out/err_plain.log | cut -c1-330
tail -3 $out/err_c_abi.log
