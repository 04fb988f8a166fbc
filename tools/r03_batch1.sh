set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1 || { tail -40 $O/gputest.log; exit 1; }
tail -3 $O/gputest.log
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
echo bench-done
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc65536 -- python3 $R/bench.py --steps 144 --warmup 8 --prewarm-s 0.2 --no-cpu-baseline > $O/p1.log 2>&1
echo pmc1-done
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc8192 -- python3 $R/bench.py --envs 8192 --steps 144 --warmup 8 --prewarm-s 0.2 --no-cpu-baseline > $O/p2.log 2>&1
echo pmc2-done
python3 $R/tools/pmc_mfma.py 65536_envs=$O/pmc65536 8192_envs=$O/pmc8192 --out $O/r03_pmc_mfma.csv
cat $O/r03_pmc_mfma.csv
rm -rf $O/pmc65536 $O/pmc8192
