"""Reduce two rocprofv3 counter-collection CSVs (one pass with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: separate passes, kernel-trace only alongside) to per-kernel HBM
traffic per launch.  Counter unit: KiB per dispatch.  gfx950 correction from the same guide: FETCH_SIZE under-reports
streaming reads by 2x (calibrated for 16 B/lane loads; the ODE kernels load 4 B/lane, so the corrected read figure is
an upper estimate); WRITE_SIZE is taken as is.

usage: python profiles/make_pmc_traffic.py FETCH.csv WRITE.csv OUT.json "<command that was profiled>"
"""
import csv
import json
import statistics
import sys
from collections import defaultdict

from srcsha import csrc_sha16


def medians(path, counter):
    per = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "vihds" in r["Kernel_Name"]:
            per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (statistics.median(v), len(v)) for k, v in per.items()}


def main():
    fetch_csv, write_csv, out, cmd = sys.argv[1:5]
    f, w = medians(fetch_csv, "FETCH_SIZE"), medians(write_csv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        fk, n = f.get(k, (0.0, 0))
        wk, _ = w.get(k, (0.0, 0))
        kernels[k] = {"FETCH_SIZE_KiB_median": fk, "WRITE_SIZE_KiB_median": wk, "dispatches": n,
                      "hbm_bytes_raw": int((fk + wk) * 1024), "hbm_bytes_corrected": int((2.0 * fk + wk) * 1024)}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes of `%s` (kernel-trace only "
                       "alongside). Counter unit: KiB per dispatch; medians over the dispatches. gfx950 correction per "
                       "/opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE x2 for streaming reads "
                       "(calibrated for 16 B/lane loads; ours are 4 B/lane: upper estimate), WRITE_SIZE as is." % cmd,
               "csrc_sha16": csrc_sha16(), "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in kernels.items():
        print("%-90s %10d B corrected" % (k[:90], v["hbm_bytes_corrected"]))


if __name__ == "__main__":
    main()
