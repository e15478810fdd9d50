cd $GRAFT_REPO_ROOT
T=${T:-r05}
# the headline is the BASELINE metric's G+D+GP step (bench.py default: --mode wali-gp, 1 generator + 5 critic steps per iteration)
SPI=6 bash tools/prof_round.sh ${T}
bash tools/prof_round.sh ${T}_face --dataset face --mode ali
SPI=2 bash tools/prof_round.sh ${T}_ali --mode ali
bash tools/prof_round.sh ${T}_gmgan --mode local_ep --n-coms 10
SPI=2 bash tools/prof_round.sh ${T}_ssgan --dataset moving_mnist
SPI=2 bash tools/prof_round.sh ${T}_ssgan3d --dataset moving_mnist --ssgan-mode ali:3dcnn
python bench.py > gpurun_out/${T}_bench_line.log 2> gpurun_out/${T}_bench_line.err
cp gpurun_out/bench_full.json gpurun_out/${T}/bench.json
tail -c 400 gpurun_out/${T}_bench_line.log
