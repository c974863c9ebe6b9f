#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py workload; leaves <out>/kernel_stats.csv (per-kernel totals) and bench.json
#   usage: tools/profile_bench.sh <workload> <steps> <warmup> <tag>
set -e -o pipefail
WL=${1:-cfg3}; STEPS=${2:-4}; WARM=${3:-1}; TAG=${4:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${WL}_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $ROOT/bench.py --workload $WL --steps $STEPS --warmup $WARM --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
f = glob.glob(out + "/**/p_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(out + "/kernel_stats.csv", "w") as fh:
    fh.write("name,calls,total_ms,avg_us,percent\n")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        name = r["Name"].replace(",", ";")[:140]
        fh.write(f"{name},{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e3:.2f},{100*float(r['TotalDurationNs'])/tot:.2f}\n")
print(open(out + "/kernel_stats.csv").read()[:3500])
PY
rm -f $OUT/*/*kernel_trace.csv $OUT/*kernel_trace.csv $OUT/*/*agent_info.csv 2>/dev/null || true
