"""Drives the ASan+UBSan build of the C oracle through resets, several device counts and a marginal geometry."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))   # this file lives in tests/
import oracle.ct_oracle as co
co._LIB_PATH = sys.argv[1]            # the sanitizer build (tests/sanitize_cpu.sh)
co._lib = None
import numpy as np
from util import action_stream
from oracle.ct_oracle import CtOracle, default_config
for D, N, K, kw in ((2, 64, 200, {}), (4, 64, 120, {}), (16, 16, 60, {}), (32, 8, 40, {}),
                    (4, 64, 80, dict(positions=[(3.4167, 0.0), (0.0, 3.4168), (-5.5458, 0.0), (0.0, -2.0)]))):
    cfg = default_config(D, **kw)
    o = CtOracle(N, D, config=cfg, nthreads=2)
    dev, dur = action_stream(1, K, N, D)
    o.reset()
    for k in range(K):
        if k % 37 == 36: o.reset()
        o.step(dev[k], dur[k])
    for f in ("now", "wake", "counter", "qlen", "queue", "received", "rx_power", "flags", "n_tx"):
        o.get(f)
    print("ok", D, N, K, int(o.get("n_tx").sum()))
