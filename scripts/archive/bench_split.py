#!/usr/bin/env python3
"""ShuffleSplit test sets: native restatement vs numpy.random.RandomState (the specification), same results."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strainscan_amd import l2
for n in (300_000, 2_000_000, 5_000_000):
    t0 = time.perf_counter(); a, _ = l2.shuffle_split_test_bits(n); t1 = time.perf_counter()
    b, _ = l2.shuffle_split_test_bits_numpy(n); t2 = time.perf_counter()
    print("n=%d native %.3f s numpy %.3f s equal=%s" % (n, t1 - t0, t2 - t1, np.array_equal(a, b)))
