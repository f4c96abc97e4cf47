#!/usr/bin/env python3
"""cProfile of the host side of the training step (where does the enqueueing thread spend / lose its time):
python tests/tuning/host_profile.py 1024"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
wl = bench.MolVaeWorkload(B, "bf16", torch.device("cuda"), 0, None, 120, 35)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    wl.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
