cd $GRAFT_REPO_ROOT
for e in gather xrun; do for z in "" "--zero"; do echo "== $e $z"; DZ_TUNE_SPCONV_ENGINE=$e timeout 200 python tools/bench_spconv.py --batch 16 --math f16x2 --reps 10 --only 32-32,64-64,128-128 $z 2>&1 | grep -E "^[xg] k27" | grep -v "+res" | sort -u | cut -c1-30,95-125; done; done
bash tools/gpu_x_diag.sh "0 128 144"
