set -x
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/z; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r01z -- python $R/bench.py --no-cpu-baseline > $O/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o fetch -- python $R/bench.py --steps 20 --warmup 5 --eager --no-cpu-baseline --roofline-steps 2 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o write -- python $R/bench.py --steps 20 --warmup 5 --eager --no-cpu-baseline --roofline-steps 2 > $O/write.log 2>&1
cd $R
F=$(find $O/fetch -name "*counter_collection.csv" | head -1); W=$(find $O/write -name "*counter_collection.csv" | head -1)
python profiles/make_pmc_traffic.py $F $W profiles/r01_z_pmc_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 20 --warmup 5 --eager --no-cpu-baseline --roofline-steps 2" > $O/traffic.log 2>&1
cp profiles/r01_z_pmc_hbm_traffic.json $O/
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -1 $O/bench_default.json | cut -c1-300
ls -la $O $O/stats | head -30
