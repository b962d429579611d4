#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes).

  pmc_kernel.py <fetch counter_collection.csv> <write counter_collection.csv> <kernel substring> [out.json]

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE is doubled per the guide's gfx950 note (it
reports half the bytes of wide streaming reads; for narrow random reads the factor is calibrated with
tools/random_access_rate.hip under the same counter, see the "calibration" block this script copies
in when a third CSV is given).  Units (k-mers) of a dispatch = its Grid_Size (one thread per k-mer).
"""
import csv
import json
import sys


def rows(path, needle, counter):
    out = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if needle in r["Kernel_Name"] and r["Counter_Name"] == counter:
                out.append((int(r["Grid_Size"]), float(r["Counter_Value"]),
                            int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def main():
    fetch_csv, write_csv, needle = sys.argv[1:4]
    out_path = sys.argv[4] if len(sys.argv) > 4 else None
    f = rows(fetch_csv, needle, "FETCH_SIZE")
    w = rows(write_csv, needle, "WRITE_SIZE")
    units_f, units_w = sum(x[0] for x in f), sum(x[0] for x in w)
    fetch_kb, write_kb = sum(x[1] for x in f), sum(x[1] for x in w)
    res = {
        "kernel": needle,
        "dispatches": len(f),
        "threads": units_f,
        "fetch_size_kb": fetch_kb,
        "write_size_kb": write_kb,
        "fetch_bytes_per_kmer_raw": fetch_kb * 1024 / max(units_f, 1),
        "write_bytes_per_kmer": write_kb * 1024 / max(units_w, 1),
        "bytes_per_kmer": (2 * fetch_kb * 1024 / max(units_f, 1)) + write_kb * 1024 / max(units_w, 1),
        "ns_per_kmer_under_pmc": sum(x[2] for x in f) / max(units_f, 1),
        "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; sums over "
                  "every dispatch of the kernel; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 "
                  "(MI355X_MICROARCH.md: gfx950 FETCH_SIZE reports half the bytes)",
    }
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
