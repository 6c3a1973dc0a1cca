"""Weights formats against the reference's own code executed from /root/reference:
the Detectron blob-name mapping, loading a Detectron `.pkl`, and the checkpoint dict."""
import os
import pickle
import sys
import warnings

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import ref_model  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_model.available(), reason="needs /root/reference and oracle/_ref")


@pytest.fixture(scope="module")
def pair():
    warnings.filterwarnings("ignore")
    ref_model.configure("configs/baselines/e2e_mask_rcnn_R-50-FPN_1x.yaml", MODEL__LOAD_IMAGENET_PRETRAINED_WEIGHTS=False,
                        MODEL__NUM_CLASSES=81)
    from detectron_pytorch_amd.rcnn import config, model

    ref = ref_model.build_model(seed=1)
    cfg = config.mask_rcnn_r50_fpn()
    torch.manual_seed(2)                      # different weights on purpose
    return ref, model.GeneralizedRCNN(cfg), cfg


def test_detectron_name_mapping_equals_the_reference(pair):
    from detectron_pytorch_amd.rcnn import weights

    ref, mine, _ = pair
    want, want_orphans = ref.detectron_weight_mapping
    got, got_orphans = weights.detectron_weight_mapping(mine)
    assert got == want
    assert sorted(got_orphans) == sorted(want_orphans)
    assert set(got) == set(mine.state_dict())


def test_loading_a_detectron_pickle_equals_the_reference_loader(pair, tmp_path):
    from detectron_pytorch_amd.rcnn import weights

    ref, mine, _ = pair
    import utils.detectron_weight_helper as helper

    rng = np.random.RandomState(0)
    mapping, orphans = weights.detectron_weight_mapping(mine)
    blobs = {d: rng.randn(*mine.state_dict()[p].shape).astype(np.float32) for p, d in mapping.items() if d}
    for o in orphans:
        blobs[o] = rng.randn(3).astype(np.float32)
    path = str(tmp_path / "model_final.pkl")
    with open(path, "wb") as f:
        pickle.dump({"blobs": blobs}, f, protocol=2)
    helper.load_detectron_weight(ref, path)
    weights.load_detectron_weight(mine, path)
    a, b = ref.state_dict(), mine.state_dict()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    back = weights.to_detectron_blobs(mine)
    assert set(back) == {d for d in mapping.values() if d}
    assert all(np.array_equal(back[k], blobs[k]) for k in back)


def test_checkpoint_dict_round_trip_with_the_reference(pair, tmp_path):
    from detectron_pytorch_amd.rcnn import train as rtrain, weights

    ref, mine, cfg = pair
    import utils.net as net_utils

    opt = rtrain.make_optimizer(mine, cfg, lr=0.01)
    path = weights.save_ckpt(str(tmp_path / "model_step7.pth"), 7, 100, 16, mine, opt)
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ckpt) == {"step", "train_size", "batch_size", "model", "optimizer"} and ckpt["step"] == 7
    net_utils.load_ckpt(ref, ckpt["model"])                      # the reference reads what this package wrote
    for k, v in mine.state_dict().items():
        assert torch.equal(ref.state_dict()[k], v), k
    with torch.no_grad():
        for p in ref.parameters():
            p.add_(1.0)
    weights.load_ckpt(mine, ref.state_dict())                    # and the other way round
    for k, v in ref.state_dict().items():
        assert torch.equal(mine.state_dict()[k], v), k
    # SGD groups as tools/train_net_step.py:262-300: weights with decay, biases with 2x lr and no decay
    g = opt.param_groups
    assert g[0]["weight_decay"] == cfg.SOLVER.WEIGHT_DECAY and g[1]["weight_decay"] == 0 and g[1]["lr"] == 2 * g[0]["lr"]
