#!/bin/bash
# control flow of the N = 2 bench path on a ONE-GPU box: two rank processes sharing cuda:0, torch.distributed over gloo (never a measurement)
out=$GRAFT_REPO_ROOT/gpurun_out/r06_ddp2; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --gpus 2 --share-gpu --pg-backend gloo --steps 10 --warmup 2 --settle-seconds 0.5 --no-cpu-baseline --full-json $out/chain_world2.json > $out/chain_world2.line 2> $out/chain_world2.err; echo "spawned ranks rc=$?"; cut -c1-500 $out/chain_world2.line; grep -v "Warn\|warn" $out/chain_world2.err | tail -4 | cut -c1-300
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --share-gpu --pg-backend gloo --steps 10 --warmup 2 --settle-seconds 0.5 --no-cpu-baseline --config3 --full-json $out/chain_world2_torchrun.json > $out/chain_world2_torchrun.line 2> $out/chain_world2_torchrun.err; echo "torchrun rc=$?"; cut -c1-400 $out/chain_world2_torchrun.line; grep -v "Warn\|warn" $out/chain_world2_torchrun.err | tail -4 | cut -c1-300
