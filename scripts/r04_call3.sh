#!/bin/bash
# round 4, third GPU call: what occupies the stream during the slow first summary() of a dealt batch (rocprofv3 timeline)
set -u
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/r04c; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
EPH_TRACE_SUMMARY=1 rocprofv3 --kernel-trace --hip-trace --memory-copy-trace --output-format csv -d $OUT -o sweep -- python $GRAFT_REPO_ROOT/scripts/time_sweep_parts2.py 262144 fresh > $OUT/sweep_traced.txt 2>&1
ls -la $OUT
python - <<PY
import csv, glob
k = list(csv.DictReader(open(glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0])))
t0 = min(int(r["Start_Timestamp"]) for r in k)
rows = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"][:60]) for r in k]
rows.sort()
# the last 60 kernels before the end, with gaps
prev_end = 0
out = open("$OUT/kernel_timeline.txt", "w")
for s, d, n in rows:
    if "craft" in n or "copy16" in n or "knot0" in n:
        out.write(f"+{s/1e6:10.3f} ms  dur {d/1e6:9.3f} ms  gap {max(0, s - prev_end)/1e6:8.3f} ms  {n}\n")
    prev_end = max(prev_end, s + d)
out.close()
PY
tail -n 40 $OUT/kernel_timeline.txt; tail -n 12 $OUT/sweep_traced.txt
