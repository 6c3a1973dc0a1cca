"""Probe: are plain torch expressions stable under hipGraph replay on this stack?  A training-step graph showed a
P2-level loss changing from replay to replay with constant weights and inputs; this reduces it to torch ops only (none of
this library's kernels).  usage: python tools/reduce_probe.py [backward] [junk] [dfirst] [clone]"""
import sys

import torch
import torch.nn.functional as F

flags = set(sys.argv[1:])
dev = torch.device("cuda", 0)
torch.manual_seed(0)
logits = torch.randn(2, 3, 200, 336, device=dev, requires_grad=True)
labels = (torch.rand(2, 3, 200, 336, device=dev) < 0.001).float()
weight = (torch.rand(2, 3, 200, 336, device=dev) < 0.002).float()


def variants():
    d = (logits.detach() * weight).sum() if "dfirst" in flags else None
    a = F.binary_cross_entropy_with_logits(logits, labels, weight, reduction="sum")
    b = F.binary_cross_entropy_with_logits(logits, labels, weight, reduction="none").view(-1, 336).sum(1).sum()
    c = logits.detach().sum()
    if d is None:
        d = (logits.detach() * weight).sum()
    e = (logits.detach().clone() * weight.clone()).sum() if "clone" in flags else d
    return a, b, c, d, e


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        eager = variants()
        if "backward" in flags:
            eager[0].backward()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print(sorted(flags), "eager ", [round(float(v), 4) for v in eager])
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    out = variants()
    if "backward" in flags:
        out[0].backward()
    if "junk" in flags:
        junk = [torch.randn(1 << 20, device=dev) for _ in range(4)]
        s2 = sum(j.sum() for j in junk)
vals = []
for i in range(6):
    g.replay()
    torch.cuda.synchronize()
    vals.append([round(float(v), 4) for v in out])
print("   replay 0", vals[0])
print("   replay 1", vals[1], "stable afterwards" if all(v == vals[1] for v in vals[1:]) else "keeps changing: %s" % vals[2:])
