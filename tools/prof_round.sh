#!/bin/bash
# Round profile on the GPU box: kernel trace + stats + one-iteration timeline, then (unless NOPMC=1) rocprofv3 PMC passes
# (separate runs, no tracing domains).  usage: gpurun -- 'bash tools/prof_round.sh <tag> [bench args for the workload]';
# copy gpurun_out/<tag>/*.md|json into profiles/ afterwards.
export TMPDIR=/tmp
TAG=${1:-prof}; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -c "from graphical_gan_amd import build; print(build.build_id())" > $O/build_id.txt
SPI=${SPI:-2}
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-variants $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $B > $O/trace.log 2>&1
python tools/stats_md.py $O/trace > $O/kernel_trace.md
python tools/timeline.py $O/trace --steps-per-iter $SPI --skip 5 > $O/timeline.md
# the captured iteration graph's own critical path at the durations of this trace (image scripts: one graph per iteration)
GGAN_GRAPH_DOT=$O/graph.dot python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-variants --repeats 0 $* > $O/dot.log 2>&1
[ -s $O/graph.dot ] && python tools/graph_critical_path.py $O/graph.dot $O/kernel_trace.md > $O/critical_path.md 2>> $O/dot.log
if [ -z "$NOPMC" ]; then
S="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-variants --no-graph $*"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- $S > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- $S > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o p -- $S > $O/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/pmc_sq -o p -- $S > $O/pmc_sq.log 2>&1
python tools/pmc_summary.py $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_sq --json $O/pmc_traffic.json > $O/pmc.md 2> $O/pmc_summary.err
fi
tail -c 1500 $O/trace.log | grep '^{' | cut -c1-300
find $O -name "*.csv" -size +4M -delete
find $O -name "*.db" -delete
