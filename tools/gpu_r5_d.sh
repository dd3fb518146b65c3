#!/bin/bash
# round 5: the whole GPU suite, then the driver's bench line and the two-rank gloo record
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5
( time timeout 2400 python -m pytest tests/ -x -q -m gpu -s ) > $OUT/d_tests.log 2>&1; echo "tests rc=$?" >> $OUT/d_tests.log
grep -E "passed|failed|error|rc=|real|config4|configs\[" $OUT/d_tests.log | tail -30
( time timeout 1500 python bench.py ) > $OUT/d_bench.json 2> $OUT/d_bench.err; echo "bench rc=$?"
tail -3 $OUT/d_bench.err; tail -1 $OUT/d_bench.json | cut -c1-1500
GSWORLD_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 16 --warmup 6 --no-extras --no-cpu-baseline > $OUT/d_bench_two_ranks_gloo.json 2> $OUT/d_bench_two_ranks_gloo.err; echo "gloo rc=$?"
tail -1 $OUT/d_bench_two_ranks_gloo.json | cut -c1-600
