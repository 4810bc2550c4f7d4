#!/bin/bash
# Samples shader clock, power and temperature of GPU 0 every 0.2 s while a command runs (usage: sample_clocks.sh out.csv cmd...)
OUT=$1; shift
( while true; do
    echo "$(date +%s.%N),$(rocm-smi -d 0 --showclocks --showpower --showtemp --csv 2>/dev/null | tail -n +2 | tr '\n' ';')"
    sleep 0.2
  done ) > "$OUT" &
SAMPLER=$!
"$@"
RC=$?
kill $SAMPLER 2>/dev/null
exit $RC
