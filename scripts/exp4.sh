#!/bin/bash
# A/B of library variants with the traversal probes as the gate ("label|lib|bench args")
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  IFS='|' read -r label so args <<< "$lib"
  if [ "$so" != "-" ]; then
    POLARIS_HIP_LIB=$GRAFT_REPO_ROOT/polaris_amd/lib/exp/$so.so python -m pytest tests/test_gpu_probes.py tests/test_gpu_parity.py -m gpu -x -q -k "arbitrary or golden or exact_and_batched" > gpurun_out/e4_$label.log 2>&1 || { echo "$label: PARITY FAILED"; tail -5 gpurun_out/e4_$label.log; }
    tail -1 gpurun_out/e4_$label.log
  fi
done
bash scripts/abx.sh "$@" "$@" 2>&1 | grep -v "^ *[0-9]* \["
