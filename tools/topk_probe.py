"""Tuning aid: latency of mi_topk_batched per problem shape and score distribution (HIP events, back-to-back calls)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectron_pytorch_amd import synthetic as syn, topk  # noqa: E402
from tools.hot_path_bench import time_kernel  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.RandomState(0)
cases = []
for n, k in ((5000, 1000), (22400, 1000), (80000, 128), (201600, 1000), (201600, 2000)):
    cases.append(("normal n=%d k=%d" % (n, k), [rng.randn(n).astype(np.float32)], [k]))
sc = [syn.rpn_head_outputs(1, 3, h, w, seed=l)[0].reshape(-1) for l, h, w in ((2, 200, 336), (3, 100, 168), (4, 50, 84), (5, 25, 42), (6, 13, 21))]
cases.append(("rpn P2 only k=1000", [sc[0]], [1000]))
cases.append(("rpn P2..P6 k=1000", sc, [min(1000, len(s)) for s in sc]))
masked = np.full(80000, -np.inf, np.float32)
live = rng.choice(80000, 4000, replace=False)
masked[live] = rng.rand(4000).astype(np.float32)
cases.append(("masked 80000 (4000 live) k=128", [masked], [128]))
masked2 = np.full(80000, -np.inf, np.float32)
masked2[rng.choice(80000, 50, replace=False)] = 0.5
cases.append(("masked 80000 (50 live) k=128", [masked2], [128]))
for name, rows, ks in cases:
    t = [torch.from_numpy(r).to(dev) for r in rows]
    sec = time_kernel(lambda: topk.topk_flat(t, ks), 50, warmup=5)
    ref = time_kernel(lambda: [torch.topk(x, k) for x, k in zip(t, ks)], 20, warmup=3)
    print("%-36s %8.1f us   (torch.topk %8.1f us)" % (name, sec * 1e6, ref * 1e6), flush=True)
