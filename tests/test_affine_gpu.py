"""mi_affine_channel_* against the torch expressions of the reference (lib/nn/modules/affine.py:15-17 followed by
ResNet.py:270-286): the fused pass must give the SAME BITS, forward and backward, in both layouts.  The reference chain is
plain torch fp32 on the same device (a floating-point kernel: tolerance 0)."""
import pytest
import torch
import torch.nn.functional as F

from detectron_pytorch_amd import affine_channel as ac

pytestmark = pytest.mark.gpu

SHAPES = [(2, 64, 40, 56), (2, 256, 25, 42), (1, 3, 5, 7), (3, 5, 25, 42), (1, 2048, 13, 21), (2, 8, 1, 1), (1, 4, 200, 336)]


def chain(x, w, b, residual, relu):
    c = w.numel()
    out = x * w.view(1, c, 1, 1) + b.view(1, c, 1, 1)
    if residual is not None:
        out = out + residual
    return F.relu(out) if relu else out


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("with_residual", [False, True])
@pytest.mark.parametrize("channels_last", [False, True])
def test_bit_identical_to_unfused_chain(shape, relu, with_residual, channels_last):
    if channels_last and shape[1] % 4:
        pytest.skip("channels-last kernel needs C % 4 == 0 (falls back to the torch chain)")
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(hash((shape, relu, with_residual)) % (1 << 31))
    fmt = torch.channels_last if channels_last else torch.contiguous_format
    x = torch.randn(shape, generator=g).to(dev).contiguous(memory_format=fmt)
    r = torch.randn(shape, generator=g).to(dev).contiguous(memory_format=fmt) if with_residual else None
    w = torch.rand(shape[1], generator=g).to(dev)
    b = torch.randn(shape[1], generator=g).to(dev)
    dy = torch.randn(shape, generator=g).to(dev).contiguous(memory_format=fmt)

    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ra = r.clone().requires_grad_(True) if r is not None else None
    rb = r.clone().requires_grad_(True) if r is not None else None
    assert ac.fused_supported(xa, w, b, ra)
    ya = ac.affine_channel(xa, w, b, ra, relu)
    yb = chain(xb, w, b, rb, relu)
    assert ya.stride() == yb.stride()
    assert torch.equal(ya, yb)
    ya.backward(dy)
    yb.backward(dy)
    assert torch.equal(xa.grad, xb.grad)
    if r is not None:
        assert torch.equal(ra.grad, rb.grad)


def test_nan_and_negative_zero_follow_torch():
    dev = torch.device("cuda", 0)
    x = torch.tensor([float("nan"), -0.0, 0.0, -1.0, 2.0, float("inf"), -float("inf"), 1e-30], device=dev).view(1, 8, 1, 1)
    x = x.expand(1, 8, 2, 2).contiguous()
    w, b = torch.ones(8, device=dev), torch.zeros(8, device=dev)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = ac.affine_channel(xa, w, b, None, True), chain(xb, w, b, None, True)
    assert torch.equal(torch.isnan(ya), torch.isnan(yb))
    assert torch.equal(torch.nan_to_num(ya, 7.0), torch.nan_to_num(yb, 7.0))
    ya.backward(torch.ones_like(ya))
    yb.backward(torch.ones_like(yb))
    assert torch.equal(xa.grad, xb.grad)


def test_unsupported_inputs_take_the_torch_chain_and_bad_calls_fail_loudly():
    dev = torch.device("cuda", 0)
    x = torch.randn(1, 8, 4, 4, device=dev)
    w = torch.rand(8, device=dev, requires_grad=True)      # trainable affine: not the frozen-BN case
    b = torch.zeros(8, device=dev)
    assert not ac.fused_supported(x, w, b)
    y = ac.affine_channel(x, w, b, None, True)
    y.sum().backward()
    assert w.grad is not None
    assert not ac.fused_supported(x.to(torch.bfloat16), w.detach(), b)
    from detectron_pytorch_amd import _lib
    rc = _lib.lib().mi_affine_channel_forward(x.data_ptr(), None, b.data_ptr(), None, x.data_ptr(), 1, 8, 4, 4, 1, 0, None)
    assert rc != 0 and b"null" in _lib.lib().mi_last_error()


def test_resnet_block_uses_the_fused_pass_and_matches_the_unfused_block():
    """Bottleneck through the HIP pass == the same block with the affine layers forced onto the torch chain."""
    from detectron_pytorch_amd.rcnn import config, resnet
    cfg = config.default_config()
    torch.manual_seed(5)
    stage, _ = resnet.make_stage(64, 256, 64, 2, cfg, 1, 1)
    stage = stage.to("cuda:0")
    stage.apply(lambda m: resnet.freeze_params(m) if isinstance(m, resnet.AffineChannel2d) else None)
    x = torch.randn(2, 64, 50, 84, device="cuda:0")
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = stage(xa)
    saved = ac.fused_supported
    ac.fused_supported = lambda *a, **k: False
    try:
        yb = stage(xb)
    finally:
        ac.fused_supported = saved
    assert torch.equal(ya, yb)
    dy = torch.randn_like(ya)
    ya.backward(dy)
    yb.backward(dy)
    # convolution backward is the same MIOpen call on identical inputs in both runs
    assert torch.allclose(xa.grad, xb.grad, rtol=0, atol=0) or (xa.grad - xb.grad).abs().max() < 1e-6


@pytest.mark.parametrize("kind", ["conv3x3", "conv1x1", "deconv2x2"])
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("with_residual", [False, True])
def test_conv_bias_act_equals_the_modules_own_forward_and_gradients(kind, relu, with_residual):
    """affine_channel.conv_bias_act (the bias, ReLU and FPN top-down sum of FPN.py:292-296,394 / mask_rcnn_heads.py:160-185
    behind a convolution, one pass of the AffineChannel kernel with weight 1): the forward is BIT-equal to
    `relu?(conv(x) (+ r))` of the module itself; the gradients of the input and the residual are equal (same
    masked output gradient into the same convolution backward), those of the weight and the bias to summation order."""
    dev = torch.device("cuda", 0)
    torch.manual_seed(5)
    if kind == "deconv2x2":
        conv = torch.nn.ConvTranspose2d(16, 24, 2, 2, 0).to(dev)
        x = torch.randn(3, 16, 14, 14, device=dev)
    else:
        k = 3 if kind == "conv3x3" else 1
        conv = torch.nn.Conv2d(16, 24, k, 1, k // 2).to(dev)
        x = torch.randn(2, 16, 25, 42, device=dev)
    torch.nn.init.normal_(conv.bias, std=0.5)
    ref = __import__("copy").deepcopy(conv)
    with torch.no_grad():
        shape = conv(x).shape
    r = torch.randn(shape, device=dev) if with_residual else None
    dy = torch.randn(shape, device=dev)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ra = r.clone().requires_grad_(True) if r is not None else None
    rb = r.clone().requires_grad_(True) if r is not None else None
    ya = ac.conv_bias_act(conv, xa, relu=relu, residual=ra)
    yb = ref(xb)
    if rb is not None:
        yb = yb + rb
    if relu:
        yb = F.relu(yb)
    assert torch.equal(ya, yb)
    assert ya.grad_fn is not None and "BiasAct" in type(ya.grad_fn).__name__, "the fused epilogue did not run"
    ya.backward(dy)
    yb.backward(dy)
    assert torch.equal(xa.grad, xb.grad)
    assert torch.allclose(conv.weight.grad, ref.weight.grad, rtol=1e-4, atol=1e-4)   # MIOpen's weight-gradient kernels add with atomics
    assert torch.allclose(conv.bias.grad, ref.bias.grad, rtol=1e-5, atol=1e-5)
    if r is not None:
        assert torch.equal(ra.grad, rb.grad)
    # what the kernel does not serve falls back to the module: CPU tensors, autocast
    cpu = __import__("copy").deepcopy(ref).cpu()
    assert torch.equal(ac.conv_bias_act(cpu, x.cpu(), relu=relu), F.relu(cpu(x.cpu())) if relu else cpu(x.cpu()))
