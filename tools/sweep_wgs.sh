#!/bin/bash
# Whole-step sweep of the two workgroup-count targets (conv forward / data-gradient split selection, filter-gradient slabs).
# usage: gpurun -- 'bash tools/sweep_wgs.sh [bench args]'
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-kernel-profile "$@" 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f %.4f' % (d['value'], d['ms_per_step']))"; }
echo "default: $(run "$@")"
for t in 96 128 160 256 320 400 512; do echo "GGAN_TARGET_WGS=$t: $(GGAN_TARGET_WGS=$t run "$@")"; done
for t in 128 192 320 384 512 768; do echo "GGAN_WGRAD_WGS=$t: $(GGAN_WGRAD_WGS=$t run "$@")"; done
echo "default again: $(run "$@")"
