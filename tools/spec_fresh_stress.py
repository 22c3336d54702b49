#!/usr/bin/env python
"""fresh-process repeats of the 44-frame run per environment (tests/test_pipeline_gpu.py::_tracker_runs_under): which setting,
if any, is not reproducible from process to process"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import test_pipeline_gpu as T
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cases = [({"RAMP_SPEC_EDIT": "0"}, "False"), ({"RAMP_SPEC_EDIT": "0"}, "True"), ({"RAMP_SPEC_EDIT": "1"}, "False"),
         ({"RAMP_SPEC_EDIT": "1"}, "True"), ({"RAMP_SPEC_EDIT": "1", "RAMP_SPEC_STREAM": "own"}, "False"),
         ({"RAMP_SPEC_EDIT": "1", "RAMP_NO_FLAG_WAITS": "1"}, "False")]
ref = None
for env, ready in cases:
    runs = T._tracker_runs_under([env] * R, ready=ready)
    if ref is None:
        ref = runs[0]
    bad = [i for i, r in enumerate(runs) if any(not np.array_equal(ref[k], r[k]) for k in ref)]
    print(env, "ready=" + ready, "%d of %d fresh runs differ from the reference" % (len(bad), R), bad, flush=True)
    for i in bad[:2]:
        r = runs[i]
        print("   keys:", [k for k in ref if not np.array_equal(ref[k], r[k])], "first E mismatch at frame",
              next((t for t in range(len(ref["E"])) if ref["E"][t] != r["E"][t]), None), flush=True)
