#!/bin/bash
# experiment driver (inside gpurun): parity subset, A/B of library variants ("label|lib|bench args"), one SQ counter pass
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_probes.py -m gpu -x -q -k "golden or exact_and_batched or obj_scene or edge_shapes or probes or arbitrary" > gpurun_out/e3_parity.log 2>&1 || { tail -30 gpurun_out/e3_parity.log; exit 1; }
tail -2 gpurun_out/e3_parity.log
bash scripts/abx.sh "$@" 2>&1 | grep -v "^ *[0-9]* \[" > gpurun_out/e3_ab.txt
cat gpurun_out/e3_ab.txt
bash scripts/pmc.sh e3_pmc --opt overlap=1 > gpurun_out/e3_pmc.txt 2>&1
cat gpurun_out/e3_pmc.txt
