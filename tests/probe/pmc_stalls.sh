#!/bin/bash
# usage: bash tests/probe/pmc_stalls.sh [--workload configN]  -- wavefront-cycle breakdown of the step's kernels (two --pmc passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="--steps 20 --warmup 5 --eager --no-cpu-baseline --no-other-configs --roofline-steps 2 $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/st1 -o p -- python $R/bench.py $ARGS > /tmp/st1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_IFETCH --kernel-trace --output-format csv -d /tmp/st2 -o p -- python $R/bench.py $ARGS > /tmp/st2.log 2>&1
python - $(find /tmp/st1 -name "*counter_collection.csv" | head -1) $(find /tmp/st2 -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-48:]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "vihds" in k and sum(d.get("SQ_WAVE_CYCLES", [0])) / max(len(d.get("SQ_WAVE_CYCLES", [1])), 1) > 2e6:
        m = {c: sum(v) / len(v) for c, v in d.items()}
        wc = m.get("SQ_WAVE_CYCLES", 1)
        print(k)
        for c in sorted(m):
            print("   %-22s %14.0f   %5.1f %% of wave cycles" % (c, m[c], 100 * m[c] / wc))
PY
