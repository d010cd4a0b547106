#!/usr/bin/env python3
"""bench.py's kernels.corr_large alone: python3 tools/corr_large_bench.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    for name, v in bench.measure_corr_large(torch.device("cuda:0"), 8).items():
        print(name, json.dumps({k: (round(x, 3) if isinstance(x, float) else x) for k, x in v.items() if k != "note"}))
