cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
O=gpurun_out/r04
{
echo "== terrain 1024x1024x32"; bash scripts/ab_variants.sh "base pairs tri64 pairs64" --scene terrain --width 1024 --height 1024 --spp 32
echo "== C4 material-ball 1920x1080x32"; bash scripts/ab_variants.sh "base pairs tri64 pairs64" --scene material-ball --width 1920 --height 1080 --spp 32
echo "== C5 instanced 2048x2048x16"; bash scripts/ab_variants.sh "base pairs tri64 pairs64" --scene instanced --width 2048 --height 2048 --spp 16
echo "== headline"; bash scripts/ab_variants.sh "base tri64" 
} > $O/tri_variants_ab.txt 2>&1
cat $O/tri_variants_ab.txt
{
for m in 0 1 2 0 1 2; do
  echo -n "POLARIS_STREAM_PRIORITY=$m  "
  POLARIS_STREAM_PRIORITY=$m timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timers 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%.0f Mrays/s %.3f ms' % (d['value'], d['ms_per_frame']))"
done
} > $O/stream_priority_ab.txt 2>&1
cat $O/stream_priority_ab.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "full_sample or ordered_against or ring_and" > $O/new_tests.log 2>&1
echo "pytest rc=$?"; tail -8 $O/new_tests.log
timeout -k 10 900 python scripts/emulate_ranks.py r04 headline C3 C4 C5 > $O/emulate.log 2>&1
echo "emulate rc=$?"; tail -3 $O/emulate.log
