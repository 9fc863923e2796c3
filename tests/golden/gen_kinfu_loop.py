#!/usr/bin/env python3
"""Writes tests/golden/kinfu_loop_96.npz: the trajectory of oracle/kinfu_loop.py (the reference's frame loop, kinfu.cpp:162-309,
restated over the scalar oracle) on the scenario of tests/test_oracle_loop.py.  Oracle-generated: a drift guard, not a
reference vector (the reference holds none).  Run from the repository root."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from loop_scenario import run_oracle, SCENARIOS  # noqa: E402

out = {}
for name in SCENARIOS:
    tr = run_oracle(name)
    out[name + "_R"] = np.stack([r for r, _ in tr["poses"]])
    out[name + "_t"] = np.stack([t for _, t in tr["poses"]])
    out[name + "_flags"] = np.array(tr["flags"], dtype=np.uint8)          # per frame: fused, integrated, reset
    out[name + "_weights_nonzero"] = np.array([tr["weights_nonzero"]], dtype=np.int64)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kinfu_loop_96.npz"), **out)
print({k: v.shape for k, v in out.items()})
