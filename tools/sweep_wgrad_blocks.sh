for b in 512 768 1024 1536 2048; do
  for cfg in "240,15,6 384 256" "240,15,6 256 256" "8,16,32 128 128"; do set -- $cfg
    RD_WGRAD_BLOCKS=$b RD_NHW=$1 python3 tools/bench_wgrad.py $2 $3 bf16 wgrad 2>/dev/null | sed "s/^/blocks=$b /"
  done
  RD_WGRAD_BLOCKS=$b python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep metric | cut -c1-130 | sed "s/^/blocks=$b /"
done
