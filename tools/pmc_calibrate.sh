#!/bin/bash
# Calibration of FETCH_SIZE / WRITE_SIZE for 4-byte accesses at random addresses (the access shape of
# the SPSS encode kernels): tools/random_access_rate.hip issues 1e8 reads (then writes, ...) per launch
# at hashed indices of arrays of known size; the counters per launch / 1e8 = bytes counted per access.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${KSH_ROUND:-r03}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/random_access_rate.hip -o /tmp/random_access_rate
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/cal_fetch -o cal -- /tmp/random_access_rate > $O/cal_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/cal_write -o cal -- /tmp/random_access_rate > $O/cal_write.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, json, sys
o = sys.argv[1]
res = {}
for tag, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = glob.glob(o + "/cal_%s/**/*counter_collection.csv" % tag, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    res[tag] = [{"kernel": r["Kernel_Name"][:40], "dispatch": int(r["Dispatch_Id"]), "kb": float(r["Counter_Value"]),
                 "bytes_per_access": float(r["Counter_Value"]) * 1024 / 1e8} for r in rows]
json.dump(res, open(o + "/pmc_calibration.json", "w"), indent=1)
for tag in res:
    for r in res[tag][:12]:
        print(tag, r)
PY
rm -rf $O/cal_fetch $O/cal_write
