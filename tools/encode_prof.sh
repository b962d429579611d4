#!/bin/bash
# Per-kernel times of one SPSS encode of a 1e8-k-mer genome set and of a small difference set
# (tools/encode_profile.py) under rocprofv3 --kernel-trace --stats.  usage: encode_prof.sh TAG [env...]
set -e -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${KSH_ROUND:-r03}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_$TAG -o enc -- python3 $R/tools/encode_profile.py > $O/enc_$TAG.log 2>&1
S=$(find $O/enc_$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$S" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "ksh::" in n:
        print("%-40s calls %4d total %9.3f ms  max %9.1f us" % (n.split("ksh::")[1].split("(")[0][:40], int(r["Calls"]), int(r["TotalDurationNs"]) / 1e6, int(r["MaxNs"]) / 1e3))
PY
cp $S $O/enc_${TAG}_kernel_stats.csv
rm -rf $O/enc_$TAG
