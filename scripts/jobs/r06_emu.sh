#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 scripts/emulate_ranks.py r06 headline C3 C4 C5 2>&1 | tee gpurun_out/r06_emulate.log
