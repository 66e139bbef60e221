cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_probes.py tests/test_gpu_renderer.py -m gpu -x -q -k "not full_sample and not full_frame and not soak and not timing and not headline" > $O/gputests5.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 $O/gputests5.log
[ $rc -ne 0 ] && exit 1
{
echo "== C5 instanced 2048x2048x16"; bash scripts/ab_variants.sh "base" --scene instanced --width 2048 --height 2048 --spp 16
echo "== terrain 1024x1024x32"; bash scripts/ab_variants.sh "base" --scene terrain --width 1024 --height 1024 --spp 32
echo "== C4 material-ball 1920x1080x32"; bash scripts/ab_variants.sh "base" --scene material-ball --width 1920 --height 1080 --spp 32
echo "== headline"; bash scripts/ab_variants.sh "base"
} 2>&1 | tee $O/inst_inline.txt
