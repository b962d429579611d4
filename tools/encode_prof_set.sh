#!/bin/bash
# usage: encode_prof_set.sh TAG genome|difference|intersection : per-kernel averages of that set's encode
# (tools/encode_prof_sets.py under rocprofv3 --kernel-trace --stats)
set -e -o pipefail
TAG=$1; WHICH=$2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${KSH_ROUND:-r03}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
D=/tmp/encs_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o enc -- python3 $R/tools/encode_prof_sets.py $WHICH > $O/encs_$TAG.log 2>&1
S=$(find $D -name "*kernel_stats.csv" | head -1)
cp $S $O/encs_${TAG}_kernel_stats.csv
python3 - "$S" <<'PY'
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "ksh::" in n:
        rows.append((int(r["TotalDurationNs"]) / 3e3, n.split("ksh::")[1].split("(")[0][:40], int(r["Calls"])))
rows.sort(reverse=True)
print("per encode (us), of 3:")
for t, n, c in rows[:30]:
    print("%-40s calls %4d  %9.1f us" % (n, c, t))
print("sum %.1f us" % sum(r[0] for r in rows))
PY
tail -1 $O/encs_$TAG.log
