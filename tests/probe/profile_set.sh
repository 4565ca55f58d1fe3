#!/bin/bash
# usage: bash tests/probe/profile_set.sh <tag> [extra bench.py args]
# The measurement set of one bench configuration, written to gpurun_out/<tag>/ (copy what is to be kept to profiles/):
#   <tag>_kernel_stats.csv             rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu-baseline <args>`
#   <tag>_pmc_fetch / _pmc_write       two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), kernel-trace only alongside
#   <tag>_pmc_hbm_traffic.json         profiles/make_pmc_traffic.py over the two
#   <tag>_pmc_valu.json                 VALU instructions per launch (a third --pmc pass: SQ_INSTS_VALU) for bench.py's issue_bound
#   <tag>_bench.json                   the plain `python bench.py <args>` line (reads the traffic file just made)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o $TAG -- python $R/bench.py --no-cpu-baseline --no-other-configs "$@" > $O/bench_stats.log 2>&1
SHORT="--steps 20 --warmup 5 --eager --no-cpu-baseline --no-other-configs --roofline-steps 2"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o fetch -- python $R/bench.py $SHORT "$@" > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o write -- python $R/bench.py $SHORT "$@" > $O/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/valu -o valu -- python $R/bench.py $SHORT "$@" > $O/valu.log 2>&1
cd $R
V=$(find $O/valu -name "*counter_collection.csv" | head -1)
head -1 $V > $O/${TAG}_pmc_valu_counter_collection.csv; grep vihds $V >> $O/${TAG}_pmc_valu_counter_collection.csv
python profiles/make_pmc_valu.py $V profiles/${TAG}_pmc_valu.json "rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace -- python bench.py $SHORT $*" > $O/valu_json.log 2>&1
cp profiles/${TAG}_pmc_valu.json $O/
S=$(find $O/stats -name "*kernel_stats.csv" | head -1)
F=$(find $O/fetch -name "*counter_collection.csv" | head -1); W=$(find $O/write -name "*counter_collection.csv" | head -1)
cp $S $O/${TAG}_kernel_stats.csv
head -1 $F > $O/${TAG}_pmc_fetch_counter_collection.csv; grep vihds $F >> $O/${TAG}_pmc_fetch_counter_collection.csv
head -1 $W > $O/${TAG}_pmc_write_counter_collection.csv; grep vihds $W >> $O/${TAG}_pmc_write_counter_collection.csv
python profiles/make_pmc_traffic.py $F $W profiles/${TAG}_pmc_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py $SHORT $*" > $O/traffic.log 2>&1
cp profiles/${TAG}_pmc_hbm_traffic.json $O/
python bench.py "$@" > $O/${TAG}_bench.json 2> $O/bench.err
rm -rf $O/stats $O/fetch $O/write $O/valu
tail -1 $O/${TAG}_bench.json | cut -c1-400
head -12 $O/${TAG}_kernel_stats.csv | cut -c1-160
