#!/usr/bin/env python3
"""Why does the loss of the bench's fixed resident batch RISE over the first steps (BENCH_r02: 1.82 -> 2.38 in 25 steps)?
Loss trajectories of the same step at the bench's learning rate and at fractions of it, and which loss term moves.
usage: python tools/loss_probe.py [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
for scale in (1.0, 0.1, 0.01):
    work = bench.TrainHarness(dev, 0, 1, "f32", "eager")
    for g in work.opt.param_groups:
        g["lr"] *= scale
    traj, terms = [], []
    for i in range(steps):
        work.opt.zero_grad(set_to_none=True)
        loss, ret = work.forward_backward()
        work.opt.step()
        traj.append(round(float(loss), 4))
        if i in (0, steps - 1):
            terms.append({k: round(float(v), 4) for k, v in ret["losses"].items()})
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in work.net.parameters() if p.grad is not None)).item()
    print(json.dumps({"lr": work.opt.param_groups[0]["lr"], "loss": traj, "first_terms": terms[0], "last_terms": terms[-1],
                      "grad_norm_last": round(gn, 3)}), flush=True)
    del work
    torch.cuda.empty_cache()
