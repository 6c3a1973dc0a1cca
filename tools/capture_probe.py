"""Tuning aid: which part of the training step refuses hipGraph capture?  Captures ONE piece per process
(a failed capture poisons the process).  usage: python tools/capture_probe.py <piece>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from detectron_pytorch_amd.rcnn import config, data as rdata, model as rmodel, targets, train as rtrain  # noqa: E402

piece = sys.argv[1]
dev = torch.device("cuda", 0)
cfg = config.mask_rcnn_r50_fpn()
torch.manual_seed(3)
net = rmodel.GeneralizedRCNN(cfg).to(dev).train()
batch = rdata.synthetic_minibatch(cfg, 2, seed=0)
data, im_info, roidb, rpn_t = rdata.to_device(batch, dev)
opt = rtrain.make_optimizer(net, cfg, lr=1e-3)
state = {}


def full():
    ret = net(data, im_info, roidb=roidb, rpn_targets=rpn_t)
    loss = sum(ret["losses"].values())
    loss.backward()
    state["losses"] = {k: v.detach() for k, v in ret["losses"].items()}
    state["n_fg"] = (ret["blobs"]["labels_int32"] > 0).sum()
    state["n_valid"] = ret["collected_valid"].sum()
    return loss


def checksums():
    w = sum(float(p.detach().double().sum()) for p in net.parameters())
    inp = float(data.double().sum()) + float(im_info.double().sum()) + sum(float(v.double().sum()) for v in roidb.values()) \
        + sum(float(v.double().sum()) for v in rpn_t.values())
    return "weights %.6f inputs %.6f" % (w, inp)


def prep():
    with torch.no_grad():
        blob = net.Conv_Body(data)
        rpn = net.RPN(blob)
    state["blob"], state["rpn"] = blob, rpn
    state["rois"], state["valid"] = net.proposals(rpn, im_info, static=True)


pieces = {
    "body_fwd": lambda: [b.sum() for b in net.Conv_Body(data)],
    "body_fwd_bwd": lambda: sum(b.sum() for b in net.Conv_Body(data)).backward(),
    "rpn_convs": lambda: net.RPN(state["blob"]),
    "proposals": lambda: net.proposals(state["rpn"], im_info, static=True),
    "labelling": lambda: targets.label_proposals(cfg, state["rois"], roidb["gt_boxes"], roidb["gt_classes"], roidb["gt_image"],
                                                 im_info[:, 2], torch.rand(16 + state["rois"].size(0), device=dev), 2,
                                                 net.iou_fn, roi_valid=state["valid"]),
    "fwd_only": lambda: net(data, im_info, roidb=roidb, rpn_targets=rpn_t),
    "fwd_bwd": full,
    "optimizer": lambda: opt.step(),
    "full_step": lambda: (full(), opt.step()),
}
fn = pieces[piece]
side = torch.cuda.Stream(dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    prep()
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        full()
        opt.step()
    fn()
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
opt.zero_grad(set_to_none=True)
if piece == "optimizer":
    full()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=(side if os.environ.get("CAPTURE_STREAM", "side") == "side" else None), capture_error_mode="relaxed"):
        out = fn()
    for i in range(int(os.environ.get("REPLAYS", "1"))):
        g.replay()
        torch.cuda.synchronize()
        if piece in ("fwd_bwd", "full_step") and i % 5 == 0:
            print("replay", i, "loss", float(out[0] if isinstance(out, tuple) else out), flush=True)
            print("   ", {k: round(float(v), 4) for k, v in state["losses"].items()}, "fg", int(state["n_fg"]), "valid",
                  int(state["n_valid"]), checksums(), flush=True)
    print("CAPTURE_OK", piece, flush=True)
except Exception as e:  # noqa: BLE001
    import traceback

    tb = traceback.format_exc()
    print("CAPTURE_FAIL", piece, repr(e)[:200], flush=True)
    print(tb[-1800:], flush=True)
    os._exit(1)
os._exit(0)
