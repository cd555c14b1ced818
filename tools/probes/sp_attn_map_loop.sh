#!/bin/bash
# usage: loop.sh <repo root> <n>
cd $1
for i in $(seq 1 $2); do
  PORT=$((29500 + RANDOM % 2000))
  OMP_NUM_THREADS=4 WANQ_REHEARSE_CONFIG=w8a8_all_linears_attn_map.yaml WANQ_REHEARSE_NO_CFG_PARALLEL=1 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $PORT tests/sp_rehearsal_worker.py 2>&1 | grep -o "RANK [01] sp_rel=[0-9.e+-]*\|RANK [01] fsdp_rel=[0-9.e+-]*" | tr '\n' ' '
  echo
done
