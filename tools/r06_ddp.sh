#!/bin/bash
# the N > 1 code path of the default (chain) workload on ONE GPU: RCCL process group of one rank, two reducers on one C-ABI communicator, captured buckets
out=$GRAFT_REPO_ROOT/gpurun_out/r06_ddp; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --gpus 1 --force-ddp --steps 30 --no-legs --no-cpu-baseline --no-children --full-json $out/chain_forceddp.json > $out/chain_forceddp.line 2> $out/chain_forceddp.err; echo rc=$?; cut -c1-700 $out/chain_forceddp.line; tail -3 $out/chain_forceddp.err | cut -c1-300
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-ddp --comm torch --steps 20 --no-legs --no-cpu-baseline --no-children --full-json $out/chain_torchrun.json > $out/chain_torchrun.line 2> $out/chain_torchrun.err; echo rc=$?; cut -c1-400 $out/chain_torchrun.line; tail -3 $out/chain_torchrun.err | cut -c1-300
timeout 600 python bench.py --gpus 1 --force-ddp --allreduce rs_ag --config3 --steps 20 --no-legs --no-cpu-baseline --no-children --full-json $out/chain_config3.json > $out/chain_config3.line 2> $out/chain_config3.err; echo rc=$?; cut -c1-400 $out/chain_config3.line; tail -3 $out/chain_config3.err | cut -c1-300
