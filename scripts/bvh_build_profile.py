#!/usr/bin/env python3
"""scripts/bvh_build_profile.py [scene] [algorithm] -- build one scene's BVH on the device a few times (for rocprofv3 --kernel-trace --stats:
which of the builder's kernels the build time goes to).  Prints the device build times."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from polaris_amd import bvh_build, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "terrain"
alg = sys.argv[2] if len(sys.argv) > 2 else "sah"
sc = scenes.SCENES[name]()
for i in range(4):
    _, info = bvh_build.rebuild_on_device(sc, max_leaf_tris=4, algorithm=alg)
    print(name, alg, info, flush=True)
