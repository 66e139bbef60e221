"""A long run of tests/test_gpu_api_sequences.py (GPU): seeds [first, last) of random call sequences against the model, then a
sensitivity check -- a model that forgets the Reset stage (tracer.go:208-213) must be caught.
    python tests/tools/api_sequence_sweep.py 12 1200"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))

import __graft_entry__ as g  # noqa: E402

g.build()
import test_gpu_api_sequences as S  # noqa: E402
from oracle import pybind as ob  # noqa: E402


def main():
    first, last = int(sys.argv[1]), int(sys.argv[2])
    orc = ob.Oracle("oracle")
    t0, bad = time.time(), []
    for seed in range(first, last):
        try:
            S.test_random_call_sequences_against_the_model(True, orc, seed)
        except BaseException as e:  # noqa: BLE001
            bad.append(seed)
            print("seed", seed, "FAILED:", repr(e)[:1500], flush=True)
    print(f"call-sequence sweep: seeds [{first}, {last}): {last - first - len(bad)} sequences equal to the model, {len(bad)} failing {bad[:20]}, {time.time() - t0:.0f} s")
    orig = S.Model.do_trace

    def forgets_the_reset(self, req, seeds):
        keep = self.frame.copy()
        st = orig(self, req, seeds)
        if keep.shape == self.frame.shape:
            self.frame[:] = keep
        return st

    S.Model.do_trace = forgets_the_reset
    caught = 0
    for seed in range(12):
        try:
            S.test_random_call_sequences_against_the_model(True, orc, seed)
        except AssertionError:
            caught += 1
    S.Model.do_trace = orig
    print(f"sensitivity: a model without the Reset stage is caught in {caught} of 12 sequences")
    return 1 if bad or caught == 0 else 0


if __name__ == "__main__":
    raise SystemExit(main())
