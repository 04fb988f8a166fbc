# Evidence for the grouped update at the thesis protocol's width (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash tools/profile_group.sh r05 base'
# kernel-trace stats, MFMA-pipe counters and FETCH/WRITE passes of `bench.py --mode group --learners 400 --envs 51200 --mixed`.
set -e
# LEARNERS / ENVS (environment): another group shape (default 400 x 128 = 51200); GROUP_WINDOW=1: one remembered transition per update; KT_ONLY=1: kernel-trace stats only.
TAG=${1:-r05}
SUF=${2:-}
LEARNERS=${LEARNERS:-400}
ENVS=${ENVS:-51200}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}_group${LEARNERS}${SUF:+_$SUF}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--mode group --learners $LEARNERS --envs $ENVS --mixed --no-cpu-baseline ${GROUP_WINDOW:+--group-window $GROUP_WINDOW}"
python3 $R/bench.py $B --steps 144 --warmup 16 > $O/${TAG}_group${LEARNERS}_bench.json 2> $O/bench.err
echo bench-done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py $B --steps 72 --warmup 8 --prewarm-s 0.5 > $O/kt.log 2>&1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/${TAG}_group${LEARNERS}_kernel_stats.csv
rm -rf $O/kt
echo kt-done
if [ "${KT_ONLY:-0}" = "1" ]; then ls $O; exit 0; fi
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc -- python3 $R/bench.py $B --steps 24 --warmup 4 --prewarm-s 0.2 > $O/p1.log 2>&1
python3 $R/tools/pmc_mfma.py group${LEARNERS}=$O/pmc --out $O/${TAG}_group${LEARNERS}_pmc_mfma.csv
rm -rf $O/pmc
echo pmc-mfma-done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $B --steps 24 --warmup 4 --prewarm-s 0.2 > $O/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py $B --steps 24 --warmup 4 --prewarm-s 0.2 > $O/pw.log 2>&1
mkdir -p $O/prof && cd $R && python3 tools/pmc_summary.py --group=${LEARNERS},${ENVS} --round=${TAG}_group${LEARNERS} --outdir=$O/prof $O/pmc_fetch $O/pmc_write > $O/pmc_summary.log 2>&1 || cat $O/pmc_summary.log
rm -rf $O/pmc_fetch $O/pmc_write
echo all-done
ls $O $O/prof
