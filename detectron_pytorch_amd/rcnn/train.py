"""One training iteration of the reference's loop (tools/train_net_step.py:262-307 optimizer groups, :411-432 step), one
process per GPU.

    zero_grad -> forward (losses) -> total = sum(losses) -> backward -> gradient all-reduce (RCCL) -> SGD step

The reference's `mynn.DataParallel` re-broadcasts all parameters every forward and reduces gradients to GPU 0
(nn/parallel/_functions.py:17,39); here parameters live on every rank and the only exchange is the averaged-gradient
all-reduce of `parallel.GradientAllReducer`, issued bucket by bucket from autograd hooks while the rest of the backward
still runs.
"""
import torch

from .model import total_loss


def parameter_groups(model, cfg):
    """train_net_step.py:262-300: weights (weight decay), biases (2x learning rate when SOLVER.BIAS_DOUBLE_LR, weight
    decay only when SOLVER.BIAS_WEIGHT_DECAY); frozen parameters are left out."""
    bias, nonbias = [], []
    for name, p in model.named_parameters():
        if p.requires_grad:
            (bias if "bias" in name else nonbias).append(p)
    return [{"params": nonbias, "lr": cfg.SOLVER.BASE_LR, "weight_decay": cfg.SOLVER.WEIGHT_DECAY},
            {"params": bias, "lr": cfg.SOLVER.BASE_LR * (cfg.SOLVER.BIAS_DOUBLE_LR + 1),
             "weight_decay": cfg.SOLVER.WEIGHT_DECAY if cfg.SOLVER.BIAS_WEIGHT_DECAY else 0}]


def make_optimizer(model, cfg, lr=None, fused=None):
    groups = parameter_groups(model, cfg)
    if lr is not None:
        groups[0]["lr"] = lr
        groups[1]["lr"] = lr * (cfg.SOLVER.BIAS_DOUBLE_LR + 1)
    # device parameters: the single-pass update (weight decay, momentum and step in one kernel over the parameter list) instead
    # of three multi-tensor passes -- 40.17 against 40.47 ms per step on the bench's model (A/B, one box); same expressions
    if fused is None:
        fused = all(p.is_cuda for g in groups for p in g["params"])
    return torch.optim.SGD(groups, momentum=cfg.SOLVER.MOMENTUM, **({"fused": True} if fused else {}))


def train_step(model, optimizer, data, im_info, roidb, rpn_targets, reducer=None, autocast_dtype=None, priority=None):
    """train_net_step.py:411-428 for iter_size 1.  Returns the forward's dict (losses are device scalars: reading them
    is the caller's synchronisation, not the step's)."""
    optimizer.zero_grad(set_to_none=True)
    if reducer is not None:
        reducer.begin_step()
    if autocast_dtype is not None:
        with torch.autocast("cuda", dtype=autocast_dtype):
            ret = model(data, im_info, roidb=roidb, rpn_targets=rpn_targets, priority=priority)
    else:
        ret = model(data, im_info, roidb=roidb, rpn_targets=rpn_targets, priority=priority)
    loss = total_loss(ret)
    loss.backward()
    if reducer is not None:
        reducer.finish_step()
    optimizer.step()
    ret["total_loss"] = loss.detach()
    return ret
