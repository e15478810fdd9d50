cd $GRAFT_REPO_ROOT
T=${T:-r06}
# the headline is the BASELINE metric's G+D+GP step (bench.py default: --mode wali-gp, 1 generator + 5 critic steps per iteration)
SPI=6 bash tools/prof_round.sh ${T}
bash tools/prof_round.sh ${T}_face --dataset face --mode ali
SPI=2 bash tools/prof_round.sh ${T}_ali --mode ali
bash tools/prof_round.sh ${T}_gmgan --mode local_ep --n-coms 10
SPI=2 bash tools/prof_round.sh ${T}_ssgan --dataset moving_mnist
SPI=2 bash tools/prof_round.sh ${T}_ssgan3d --dataset moving_mnist --ssgan-mode ali:3dcnn
# the static per-(kernel, grid) tables of THIS build first (the same call rebuilds them in the container from the merged gpurun_out/), so
# that the bench line below weighs the in-graph durations and counters of the build it runs (roofline.static_matches_build)
python tools/collect_profiles.py ${T} > /dev/null
python bench.py > gpurun_out/${T}_bench_line.log 2> gpurun_out/${T}_bench_line.err
cp gpurun_out/bench_full.json gpurun_out/${T}/bench.json
tail -c 400 gpurun_out/${T}_bench_line.log
