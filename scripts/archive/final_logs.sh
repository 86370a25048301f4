#!/bin/bash
# 1-rank torch.distributed launch + plain launch of bench.py (profiles/r02b_torchrun_1rank.md)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/torchrun_1rank.log 2>&1
python bench.py > $O/bench_plain.log 2>&1
tail -1 $O/torchrun_1rank.log | cut -c1-300; tail -1 $O/bench_plain.log | cut -c1-300
