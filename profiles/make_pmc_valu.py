"""Reduce a rocprofv3 counter-collection CSV of a `--pmc SQ_INSTS_VALU` pass (kernel-trace only alongside) to VALU
wavefront-instructions per launch for every vihds kernel (median over the dispatches): the numerator of bench.py's
`issue_bound` object.

usage: python profiles/make_pmc_valu.py COUNTERS.csv OUT.json "<command that was profiled>"
"""
import csv
import json
import statistics
import sys
from collections import defaultdict

from srcsha import csrc_sha16


def main():
    path, out, cmd = sys.argv[1:4]
    per = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == "SQ_INSTS_VALU" and "vihds" in r["Kernel_Name"]:
            per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    kernels = {k: {"SQ_INSTS_VALU": statistics.median(v), "dispatches": len(v)} for k, v in per.items()}
    json.dump({"note": "rocprofv3 --pmc SQ_INSTS_VALU in a pass of its own of `%s`; wavefront-level VALU instructions per "
                       "dispatch, summed over the chip, medians over the dispatches" % cmd, "csrc_sha16": csrc_sha16(), "kernels": kernels},
              open(out, "w"), indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"]):
        print("%-100s %12.0f VALU insts" % (k[:100], v["SQ_INSTS_VALU"]))


if __name__ == "__main__":
    main()
