#!/usr/bin/env python3
"""Per-kernel (or per-stage) HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE;
separate runs, as MI355X_MICROARCH.md prescribes).

  pmc_kernel.py <fetch counter_collection.csv> <write counter_collection.csv> <name[,name...]> [out.json]
                [--units-from <name>]

Several names = one stage made of several kernels (the LDS-staged neighbour probe): the counters
are summed over all of them; the k-mers come from the dispatches of --units-from (one thread per
k-mer: Grid_Size), by default the first name.

FETCH_SIZE: the guide says gfx950 reports half the bytes of a wide streaming read and to calibrate
other shapes; tools/pmc_calibrate.sh did (profiles/r02_pmc_calibration.json): a 4-byte read at a random
address counts 63.7 B (one 64-byte line, exact), a 4-byte write at a random address 32.0 B of
WRITE_SIZE.  The encode kernels are random-access, so `bytes_per_kmer` = FETCH_SIZE (as read) +
WRITE_SIZE; `bytes_per_kmer_streaming_rule` = 2 x FETCH_SIZE + WRITE_SIZE is what the guide's
streaming correction would give (an upper bound here).
"""
import csv
import json
import sys


def rows(path, names, counter):
    out = {}
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            for nm in names:
                if ("ksh::" + nm + "<") in r["Kernel_Name"] or ("ksh::" + nm + "(") in r["Kernel_Name"]:
                    out.setdefault(nm, []).append((int(r["Grid_Size"]), float(r["Counter_Value"]),
                                                   int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def main():
    args = [a for a in sys.argv[1:]]
    units_from = None
    units_scale = 1
    if "--units-from" in args:
        i = args.index("--units-from")
        units_from = args[i + 1]
        del args[i:i + 2]
    if "--units-scale" in args:      # k-mers per thread of the --units-from kernel (k_decode_l2: 8 keys per thread)
        i = args.index("--units-scale")
        units_scale = int(args[i + 1])
        del args[i:i + 2]
    fetch_csv, write_csv, names = args[0], args[1], args[2].split(",")
    out_path = args[3] if len(args) > 3 else None
    units_from = units_from or names[0]
    # (--units-from may name a kernel outside the stage: one with a thread per k-mer, launched once per
    # encode like the stage's kernels, when none of those has that shape)
    extra = [] if units_from in names else [units_from]
    f = rows(fetch_csv, names + extra, "FETCH_SIZE")
    w = rows(write_csv, names + extra, "WRITE_SIZE")
    units_f = sum(x[0] for x in f.get(units_from, [])) * units_scale
    units_w = sum(x[0] for x in w.get(units_from, [])) * units_scale
    for nm in extra:
        f.pop(nm, None)
        w.pop(nm, None)
    fetch_kb = sum(x[1] for v in f.values() for x in v)
    write_kb = sum(x[1] for v in w.values() for x in v)
    fb = fetch_kb * 1024 / max(units_f, 1)
    wb = write_kb * 1024 / max(units_w, 1)
    res = {
        "kernels": names,
        "dispatches": {nm: len(v) for nm, v in f.items()},
        "kmers": units_f,
        "fetch_size_kb": fetch_kb,
        "write_size_kb": write_kb,
        "fetch_bytes_per_kmer": fb,
        "write_bytes_per_kmer": wb,
        "bytes_per_kmer": fb + wb,
        "bytes_per_kmer_streaming_rule": 2 * fb + wb,
        "per_kernel_fetch_bytes_per_kmer": {nm: sum(x[1] for x in v) * 1024 / max(units_f, 1) for nm, v in f.items()},
        "per_kernel_write_bytes_per_kmer": {nm: sum(x[1] for x in v) * 1024 / max(units_w, 1) for nm, v in w.items()},
        "ns_per_kmer_under_pmc": sum(x[2] for v in f.values() for x in v) / max(units_f, 1),
        "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; sums over "
                  "every dispatch of the kernel(s); bytes = (FETCH_SIZE + WRITE_SIZE) * 1024, FETCH_SIZE as read: "
                  "calibrated for 4-byte accesses at random addresses (one 64-byte line each, "
                  "profiles/r02_pmc_calibration.json); the guide's streaming rule (2 x FETCH_SIZE) is kept beside it",
    }
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
