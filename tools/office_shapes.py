#!/usr/bin/env python3
"""bench.py's kernels.office_shapes alone (configuration 5's in-scope chains at the network's shapes): python3 tools/office_shapes.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    res = bench.measure_office_shapes(torch.device("cuda:0"), 8)
    for name, v in res.items():
        print(name, json.dumps({k: (round(x, 3) if isinstance(x, float) else x) for k, x in v.items() if k != "note"}))
