#!/usr/bin/env python3
"""Longer PHY-grid parity runs than the pytest tier affords (the event-driven Python model is the slow side):
   python tests/soak_grid.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_grid as t

BIG = len(sys.argv) > 1 and sys.argv[1] == "big"
STATIC = ((4, 8, 1.0), (9, 8, 1.0), (16, 6, 1.0), (25, 3, 0.5))
MOBILE = ((4, 4, 0.6), (9, 4, 0.5), (16, 2, 0.4))
if BIG:
    STATIC = ((2, 32, 5.0), (4, 32, 4.0), (16, 32, 3.0), (20, 16, 2.0), (32, 8, 1.0), (49, 4, 0.6), (64, 4, 0.5))
    MOBILE = ((2, 16, 3.0), (4, 16, 2.0), (16, 8, 1.5), (32, 4, 0.5), (64, 2, 0.2))
t0 = time.time()
for n, N, T in STATIC:
    t.test_grid_kernel_matches_event_driven_oracle(n, N, T)
    print("static n=%d N=%d T=%.1f ok (%.0f s)" % (n, N, T, time.time() - t0), flush=True)
for n, N, T in MOBILE:
    t.test_mobile_grid_kernel_matches_event_driven_oracle(n, N, T)
    print("mobile n=%d N=%d T=%.1f ok (%.0f s)" % (n, N, T, time.time() - t0), flush=True)
print("grid soak ok")
