"""The SURVEY 8d microbench shapes beyond the CIFAR sites, alone in one process (for rocprofv3: `-- python3
tools/roofline_shapes.py`): [128, 524288], [28, 802816], [28, 100352] site forward / backward, the weight quantiser on
[512, 512, 3, 3], the plain and packed quantiser on 2^26 elements.  Prints bench.py's `roofline_shapes` block as JSON."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
from alignq_amd import _lib as L  # noqa: E402
lib = L.load()
out = bench.measure_roofline_shapes(dev, 8)
n = 1 << 26
x, y, g = torch.randn(n, device=dev), torch.empty(n, device=dev), torch.randn(n, device=dev)
st, p = L.stream_ptr(), L.ptr
t_f = bench.time_call(lambda: lib.alignq_act_quant_fwd(p(x), p(y), None, n, 8, 2.0, 0, st), 20)
t_b = bench.time_call(lambda: lib.alignq_act_quant_bwd(p(g), p(x), p(y), n, 2.0, st), 20)
out["act_quant_2p26"] = {"fwd_us": t_f * 1e6, "fwd_hbm_gbs": 8.0 * n / t_f / 1e9, "bwd_us": t_b * 1e6, "bwd_hbm_gbs": 12.0 * n / t_b / 1e9}
print(json.dumps(out))
