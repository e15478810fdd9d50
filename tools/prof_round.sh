#!/bin/bash
# Round profile on the GPU box: rocprofv3 PMC passes (separate runs, no tracing domains), kernel trace + stats, then the
# default bench line.  usage: gpurun -- 'bash tools/prof_round.sh r01g [extra bench args]'; copy gpurun_out/<tag>/*.md|json
# into profiles/ afterwards.
export TMPDIR=/tmp
TAG=${1:-prof}; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
S="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-graph $*"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- $S > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- $S > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o p -- $S > $O/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/pmc_sq -o p -- $S > $O/pmc_sq.log 2>&1
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $B > $O/trace.log 2>&1
python tools/pmc_summary.py $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_sq --json $O/pmc_traffic.json > $O/pmc_summary.md 2> $O/pmc_summary.err
if [ -z "$*" ]; then cp $O/pmc_traffic.json profiles/pmc_traffic.json; fi
python tools/stats_md.py $O/trace/t_kernel_stats.csv > $O/kernel_trace.md
python bench.py $* > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.err; tail -c 1500 $O/trace.log | grep '^{' | cut -c1-400
cut -c1-900 $O/bench.json
find $O -name "*.csv" -size +6M -delete
