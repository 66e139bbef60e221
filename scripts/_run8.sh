cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/gputests8.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 $O/gputests8.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 bash scripts/profile_round.sh r04 > $O/profile_round.log 2>&1
echo "profile rc=$?"; tail -16 $O/profile_round.log | cut -c1-300
