#!/bin/bash
# usage: [WORKLOADS="config5 ..."] [MFMA=0] bash tests/probe/profile_workloads.sh <tag>
# The other BASELINE configurations (bench.py --workload ...): bench line + rocprofv3 kernel stats each, and the MFMA-busy
# counters of the dr_blackbox step.  Output in gpurun_out/<tag>/ (copy what is to be kept to profiles/).
TAG=$1
WL=${WORKLOADS:-config3_train config3_eval config3_eval_stored config4 config5}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in $WL; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$W -o $W -- python $R/bench.py --workload $W --steps 100 --warmup 10 --no-cpu-baseline > $O/stats_$W.log 2>&1
  cp $(find $O/stats_$W -name "*kernel_stats.csv" | head -1) $O/${TAG}_${W}_kernel_stats.csv
  rm -rf $O/stats_$W
  # HBM traffic of the workload's kernels: FETCH_SIZE and WRITE_SIZE in two separate passes (round 4)
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_${W}_$C -o pmc -- python $R/bench.py --workload $W --steps 10 --warmup 3 --eager --no-cpu-baseline --roofline-steps 4 > $O/pmc_${W}_$C.log 2>&1
  done
  F=$(find $O/pmc_${W}_FETCH_SIZE -name "*counter_collection.csv" | head -1); WR=$(find $O/pmc_${W}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  head -1 $F > $O/${TAG}_${W}_pmc_fetch_counter_collection.csv; grep vihds $F >> $O/${TAG}_${W}_pmc_fetch_counter_collection.csv
  head -1 $WR > $O/${TAG}_${W}_pmc_write_counter_collection.csv; grep vihds $WR >> $O/${TAG}_${W}_pmc_write_counter_collection.csv
  (cd $R && python profiles/make_pmc_traffic.py $F $WR profiles/${TAG}_${W}_pmc_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --workload $W --steps 10 --warmup 3 --eager --no-cpu-baseline --roofline-steps 4" > $O/traffic_$W.log 2>&1; cp profiles/${TAG}_${W}_pmc_hbm_traffic.json $O/)
  rm -rf $O/pmc_${W}_FETCH_SIZE $O/pmc_${W}_WRITE_SIZE
done
# matrix-core busy fraction of the dr_blackbox kernels: BASELINE config 4 itself (450 groups of 16 trajectories: fewer than the
# chip's SIMDs) and the same kernels with the chip full (config4_s1000: 2 250 groups)
if [ "${MFMA:-1}" = "1" ]; then
for W in config4 config4_s1000; do
  cd /tmp
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma_$W -o mfma -- python $R/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 4 > $O/mfma_$W.log 2>&1
  F=$(find $O/mfma_$W -name "*counter_collection.csv" | head -1)
  head -1 $F > $O/${TAG}_${W}_pmc_mfma_counter_collection.csv; grep "vihds" $F >> $O/${TAG}_${W}_pmc_mfma_counter_collection.csv
  rm -rf $O/mfma_$W
  cd $R
  python - $O/${TAG}_${W}_pmc_mfma_counter_collection.csv profiles/${TAG}_${W}_mfma_busy.json $W <<'PY'
import csv, json, sys, collections
busy, act = collections.defaultdict(list), collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    (busy if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES" else act)[k].append(float(r["Counter_Value"]))
out = {}
for k in busy:
    if k in act and sum(busy[k]) > 0:
        b, a = sum(busy[k]) / len(busy[k]), sum(act[k]) / len(act[k])
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs share the MFMA-busy cycles
        out[k] = {"SQ_VALU_MFMA_BUSY_CYCLES": b, "GRBM_GUI_ACTIVE": a, "mfma_busy_frac": b / (a / 8 * 1024), "dispatches": len(busy[k])}
json.dump({"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python bench.py --workload %s "
                   "--steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 4; mfma_busy_frac = MFMA-busy cycles / (GUI-active / 8 XCDs "
                   "x 1024 SIMDs)" % sys.argv[3], "kernels": out}, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
PY
  cp profiles/${TAG}_${W}_mfma_busy.json $O/
done
cp profiles/${TAG}_config4_mfma_busy.json profiles/config4_mfma_busy.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4s -o c4s -- python $R/bench.py --workload config4_s1000 --steps 50 --warmup 10 --no-cpu-baseline > $O/stats_c4s.log 2>&1
cp $(find $O/stats_c4s -name "*kernel_stats.csv" | head -1) $O/${TAG}_config4_s1000_kernel_stats.csv
rm -rf $O/stats_c4s
WL="$WL config4_s1000"
fi
cd $R
for W in $WL; do
  python bench.py --workload $W > $O/${TAG}_${W}_bench.json 2> $O/${W}.err
  tail -1 $O/${TAG}_${W}_bench.json | cut -c1-160
done
