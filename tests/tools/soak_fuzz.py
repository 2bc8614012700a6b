#!/usr/bin/env python3
"""One-off soak: the randomised parity sweep of tests/test_gpu_fuzz_parity.py over many more seeds (not part of the suite)."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import numericalnim_amd as nn
from oracle import oracle as O
import test_gpu_fuzz_parity as T
dev = torch.device("cuda:0")
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
bad = []
for s in range(120, 120 + n_seeds):
    try:
        T.test_random_case.__wrapped__(nn, O, dev, s) if hasattr(T.test_random_case, "__wrapped__") else T.test_random_case(nn, O, dev, s)
    except Exception as e:
        bad.append((s, repr(e)[:300]))
        if len(bad) <= 5:
            traceback.print_exc()
print(f"seeds {n_seeds}: failures {len(bad)}")
for b in bad[:20]:
    print(b)
