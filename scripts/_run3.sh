cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests3.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 $O/gputests3.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 bash scripts/profile_round.sh r04 > $O/profile_round.log 2>&1
echo "profile rc=$?"; tail -20 $O/profile_round.log
EMULATE_ROUNDS=8 timeout -k 10 600 python scripts/emulate_ranks.py r04b C4 C5 > $O/emulate8.log 2>&1
echo "emulate rc=$?"; tail -4 $O/emulate8.log
