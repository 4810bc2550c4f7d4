#!/bin/bash
# Round 2, third GPU call: fast-path fixes, layout 2, the three bench lines, the debug of the fast propagator.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( time timeout 1200 python -m pytest tests/test_gpu_fast.py tests/test_gpu_craft.py -m gpu -q -k "fast or erkn" ) > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
python scripts/dbg_fast_prop.py > $OUT/dbg_fast_prop.log 2>&1; cat $OUT/dbg_fast_prop.log
for S in 16 32 64; do EPH_FAST_SLICES=$S python scripts/time_path.py 4096 2000 4; done > $OUT/fast_time.log 2>&1
python scripts/time_path.py 1024 2000 4 >> $OUT/fast_time.log 2>&1
python scripts/time_path.py 16384 200 4 >> $OUT/fast_time.log 2>&1
cat $OUT/fast_time.log
for LAY in 0 1 2; do EPH_WG_LAYOUT=$LAY python scripts/time_path.py 4096 3000 0; done > $OUT/layout_time.log 2>&1
cat $OUT/layout_time.log
EPH_WG_LAYOUT=2 EPH_DEBUG_WG=4 python scripts/wg_cycles.py 4096 > $OUT/wg_cycles.log 2>&1; cat $OUT/wg_cycles.log
python bench.py --path fast > $OUT/bench_fast.json 2> $OUT/bench_fast.err; tail -c 1500 $OUT/bench_fast.json; tail -3 $OUT/bench_fast.err
python bench.py --workload craft --steps 3 > $OUT/bench_craft.json 2> $OUT/bench_craft.err; tail -c 1500 $OUT/bench_craft.json; tail -3 $OUT/bench_craft.err
python bench.py --workload nbody-sharded --steps 20 --warmup 3 > $OUT/bench_sharded.json 2> $OUT/bench_sharded.err; tail -c 1500 $OUT/bench_sharded.json; tail -3 $OUT/bench_sharded.err
EPH_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --workload craft --steps 2 --craft 20000 > $OUT/bench_craft_2rank.json 2> $OUT/bench_craft_2rank.err; tail -c 600 $OUT/bench_craft_2rank.json; tail -3 $OUT/bench_craft_2rank.err
