#!/bin/bash
# second counter pass on the default step kernel: where the issue cycles go (VALU / LDS / scalar / misc), LDS stalls, instruction fetch
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02pmc2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/scripts/time_path.py 4096 2000 0"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_ANY -d $OUT -o p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE -d $OUT -o p2 -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES -d $OUT -o p3 -- $CMD > $OUT/p3.log 2>&1
python - <<'PY'
import csv, collections, json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r02pmc2"
res = {}
for p in ("p1", "p2", "p3"):
    f = f"{out}/{p}_counter_collection.csv"
    if not os.path.exists(f):
        import glob
        g = glob.glob(f"{out}/**/{p}_counter_collection.csv", recursive=True)
        if not g: print("missing", p); continue
        f = g[0]
    agg = collections.defaultdict(float); disp = set()
    for r in csv.DictReader(open(f)):
        if "k_lm_step_wg" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    for k, v in agg.items(): res[k] = v / max(1, len(disp))
    res[p + "_dispatches"] = len(disp)
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -f $OUT/*_agent_info.csv $OUT/*_kernel_trace.csv
