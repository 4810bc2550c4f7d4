#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python scripts/time_path.py 4096 3000 0 > $OUT/time.log 2>&1; cat $OUT/time.log
python scripts/time_path.py 3072 3000 0 >> $OUT/time.log 2>&1; python scripts/time_path.py 2048 3000 0 >> $OUT/time.log 2>&1; tail -2 $OUT/time.log
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_horizon.py tests/test_gpu_shard.py -m gpu -q -k "accel or plummer or full_size or ranks_on_one or config5" ) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
python scripts/step_span.py > $OUT/step_span.log 2>&1; head -2 $OUT/step_span.log
EPH_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 50 --warmup 5 > $OUT/bench_2rank.json 2> $OUT/bench_2rank.err; cut -c1-300 $OUT/bench_2rank.json; tail -2 $OUT/bench_2rank.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_k20.json 2>/dev/null; cut -c1-200 $OUT/bench_k20.json
