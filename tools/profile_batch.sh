# The round's evidence batch (run through gpurun from the repo root): bench lines, rocprofv3 kernel-trace summary, PMC passes.
#   gpurun --timeout 1200 -- 'bash tools/profile_batch.sh r04'
set -e
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}_final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/${TAG}_train_bench.json 2> $O/${TAG}_train_bench.err
echo bench-done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 144 --warmup 72 --no-cpu-baseline > $O/kt.log 2>&1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/${TAG}_train_kernel_stats.csv
echo kt-done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --mode train --steps 30 --warmup 5 --prewarm-s 0.2 --no-cpu-baseline > $O/pf.log 2>&1
echo pmc-fetch-done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --mode train --steps 30 --warmup 5 --prewarm-s 0.2 --no-cpu-baseline > $O/pw.log 2>&1
echo pmc-write-done
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc65536 -- python3 $R/bench.py --steps 144 --warmup 8 --prewarm-s 0.2 --no-cpu-baseline > $O/p1.log 2>&1
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc8192 -- python3 $R/bench.py --envs 8192 --steps 144 --warmup 8 --prewarm-s 0.2 --no-cpu-baseline > $O/p2.log 2>&1
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc4096 -- python3 $R/bench.py --envs 4096 --steps 144 --warmup 8 --prewarm-s 0.2 --no-cpu-baseline > $O/p3.log 2>&1
python3 $R/tools/pmc_mfma.py 65536_envs=$O/pmc65536 8192_envs=$O/pmc8192 4096_envs=$O/pmc4096 --out $O/${TAG}_pmc_mfma.csv
echo pmc-mfma-done
mkdir -p $O/prof && cd $R && python3 tools/pmc_summary.py --round=$TAG --outdir=$O/prof $O/pmc_fetch $O/pmc_write > $O/pmc_summary.log 2>&1 || cat $O/pmc_summary.log
cd /tmp
rm -rf $O/pmc65536 $O/pmc8192 $O/pmc4096 $O/pmc_fetch $O/pmc_write $O/kt
python3 $R/bench.py --mixed --no-cpu-baseline > $O/${TAG}_train_mixed_bench.json 2>/dev/null
python3 $R/bench.py --mode policy --no-cpu-baseline > $O/${TAG}_policy_bench.json 2>/dev/null
python3 $R/bench.py --mode group --learners 32 --no-cpu-baseline > $O/${TAG}_group_bench.json 2>/dev/null
python3 $R/bench.py --mode env --steps 720 > $O/${TAG}_env_mode_bench.json 2>/dev/null
for n in 4096 8192 16384 32768; do
  python3 $R/bench.py --envs $n --steps 288 --no-cpu-baseline > $O/${TAG}_train_${n}_bench.json 2>/dev/null
done
for n in 4096 8192; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$n -- python3 $R/bench.py --envs $n --steps 1440 --warmup 72 --no-cpu-baseline > $O/kt$n.log 2>&1
  cp $(find $O/kt$n -name "*kernel_stats.csv" | head -1) $O/${TAG}_train_${n}_kernel_stats.csv
  rm -rf $O/kt$n
done
python3 $R/tools/update_forms.py > $O/${TAG}_update_forms.json 2>/dev/null
python3 $R/tools/track_time.py > $O/${TAG}_track_time.txt 2>/dev/null
echo all-done
ls $O
