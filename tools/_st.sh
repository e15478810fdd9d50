cd /root/repo; out=gpurun_out/wgrad_split; mkdir -p $out
for m in "" "--mode ali" "--mode ali --dataset face" "--mode local_ep"; do echo "== $m"; bash tools/ab_env.sh "$m" GGAN_WGRAD_SPLIT=0 | head -4; done > $out/ab3.log 2>&1
cat $out/ab3.log
