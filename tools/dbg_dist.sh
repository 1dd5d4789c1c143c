#!/bin/bash
cd $GRAFT_REPO_ROOT
PYTHONFAULTHANDLER=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29511 tests/dist_scripts/sharded_lm_gpu.py /tmp/o.json 2>&1 | grep -v "^$" | grep -B2 -A25 "Fatal Python" | head -80
