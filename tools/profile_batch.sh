set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/fin2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/r02_train_bench.json 2> $O/r02_train_bench.err
echo bench-done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 144 --warmup 72 --no-cpu-baseline > $O/kt.log 2>&1
echo kt-done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline > $O/pf.log 2>&1
echo pmc-fetch-done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline > $O/pw.log 2>&1
echo pmc-write-done
python3 $R/bench.py --mixed --no-cpu-baseline > $O/r02_train_mixed_bench.json 2>/dev/null
python3 $R/bench.py --mode policy --no-cpu-baseline > $O/r02_policy_bench.json 2>/dev/null
python3 $R/bench.py --mode group --learners 32 --no-cpu-baseline > $O/r02_group_bench.json 2>/dev/null
python3 $R/bench.py --envs 8192 --no-cpu-baseline > $O/r02_train_8192_bench.json 2>/dev/null
python3 $R/bench.py --envs 4096 --no-cpu-baseline > $O/r02_train_4096_bench.json 2>/dev/null
python3 $R/bench.py --envs 16384 --steps 144 --no-cpu-baseline > $O/r02_train_16384_bench.json 2>/dev/null
python3 $R/bench.py --envs 32768 --steps 144 --no-cpu-baseline > $O/r02_train_32768_bench.json 2>/dev/null
python3 $R/tools/update_forms.py > $O/r02_update_forms.json 2>/dev/null
echo all-done
find $O -name "*kernel_stats.csv" | head
