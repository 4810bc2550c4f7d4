#!/bin/bash
# PMC counters of k_craft_wave at a small batch (usage on the GPU box: scripts/prof_wave.sh <tag> <n_craft>)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export EPH_CRAFT_FORM=wave SIZES=$2
CMD="python $GRAFT_REPO_ROOT/scripts/bench_craft_small.py run"
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- $CMD > $OUT/stats.log 2>&1
timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH GRBM_GUI_ACTIVE -d $OUT -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1
rm -f $OUT/*_agent_info.csv
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pmc*_counter_collection.csv")):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_craft_wave" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print(f.split("/")[-1], dict(acc))
for r in csv.DictReader(open("$OUT/stats_kernel_stats.csv")):
    if "craft" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
tail -2 $OUT/stats.log
