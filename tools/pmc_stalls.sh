#!/bin/bash
# Where the waves of the encode's kernels spend their cycles: SQ counters of one 1e8-k-mer genome-set encode
# (tools/encode_prof_sets.py) in a PMC pass of its own (rocprofv3 --kernel-trace --pmc, no other trace domain).
# usage: tools/pmc_stalls.sh TAG [genome|difference|intersection]
set -e -o pipefail
TAG=$1; WHICH=${2:-genome}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${KSH_ROUND:-r03}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
D=/tmp/pmc_stalls_$TAG
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS \
  --output-format csv -d $D -o st -- python3 $R/tools/encode_prof_sets.py $WHICH > $O/stalls_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU \
  --output-format csv -d ${D}b -o st -- python3 $R/tools/encode_prof_sets.py $WHICH >> $O/stalls_$TAG.log 2>&1
python3 - "$(find $D -name '*counter_collection.csv' | head -1)" "$(find ${D}b -name '*counter_collection.csv' | head -1)" <<'PY' | tee $O/stalls_$TAG.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
calls = collections.defaultdict(int)
for path in sys.argv[1:]:
    seen = set()
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "ksh::" not in n:
            continue
        n = n.split("ksh::")[1].split("(")[0][:36]
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], path)
        if key not in seen and path == sys.argv[1]:
            seen.add(key)
            dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            calls[n] += 1
rows = sorted(acc.items(), key=lambda kv: -dur[kv[0]])
print("%-36s %6s %9s | of wave cycles: %6s %6s %6s | LDS: %7s %7s | per wave: %7s %7s %7s" %
      ("kernel", "calls", "us", "wait", "stall", "active", "confl%", "busy%", "valu", "lds", "vmem"))
for n, c in rows[:24]:
    wc = max(c.get("SQ_WAVE_CYCLES", 0), 1)
    waves = max(c.get("SQ_WAVES", 0), 1)
    idx = max(c.get("SQ_LDS_IDX_ACTIVE", 0), 1)
    busy = max(c.get("SQ_BUSY_CYCLES", 0), 1)
    print("%-36s %6d %9.1f | %17.2f %6.2f %6.2f | %7.1f %7.1f | %7.0f %7.0f %7.0f" %
          (n, calls[n], dur[n], c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc,
           c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / idx,
           100 * c.get("SQ_LDS_IDX_ACTIVE", 0) / busy, c.get("SQ_INSTS_VALU", 0) / waves, c.get("SQ_INSTS_LDS", 0) / waves,
           (c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)) / waves))
PY
