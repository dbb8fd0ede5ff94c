"""CPU: the oracle's restatement of the bf16 mode's dispatch rules (oracle/bf16.py) equals the library's planners over a shape
sweep (host-only entry points of the C ABI: no GPU needed), and its conv Function does what its docstring says."""
import itertools

import torch
import torch.nn.functional as F

from oracle import bf16 as OB
from oracle import ops as O


def test_bf16_rules_mirror_the_library_planners():
    from pesr_amd import _lib
    L = _lib.lib()
    n = 0
    for N, (H, W), Cin, Cout, mw in itertools.product((1, 2, 16), ((48, 48), (96, 96), (192, 192), (24, 24), (12, 12), (7, 48), (5, 100), (30, 36), (6, 144)),
                                                      (3, 32, 64, 96, 128, 256, 512), (64, 128, 192, 256, 384, 1024), (1, 64, 128)):
        assert OB.conv_score(N, H, W, Cin, Cout, mw) == L.pesr_conv3x3_bf16_score(N, H, W, Cin, Cout, mw), (N, H, W, Cin, Cout, mw)
        assert OB.conv_s2_score(N, H, W, Cin, Cout, mw) == L.pesr_conv3x3_bf16_s2_score(N, H, W, Cin, Cout, mw), (N, H, W, Cin, Cout, mw)
        assert OB.conv_s2_dgrad_score(N, H, W, Cin, Cout, mw) == L.pesr_conv3x3_bf16_s2_dgrad_score(N, H, W, Cin, Cout, mw), (N, H, W, Cin, Cout, mw)
        with OB.enabled(True, mw):
            lib_ok = W % 48 == 0 and Cin % 64 == 0 and Cout % 128 == 0 and L.pesr_conv3x3_wgrad_bf16_workspace_bytes(N, H, W, Cin, Cout) > 0 \
                and N * ((H + 1) // 2) * (W // 48) >= (96 if mw >= 64 else 1)
            assert OB.wgrad_eligible(N, H, W, Cin, Cout) == lib_ok, (N, H, W, Cin, Cout, mw)
        n += 1
    assert n > 3000


def test_round_bf16_is_round_to_nearest_even():
    x = torch.tensor([1.0, 1.00390625, 1.005859375, 1.01171875, -3.1415927, 1e-40, 65504.0, 3.3895314e38])
    r = O.round_bf16(x)
    bits = x.view(torch.int32)
    want = ((bits + 0x7FFF + ((bits >> 16) & 1)) >> 16 << 16).view(torch.float32)     # RNE on the fp32 bit pattern
    assert torch.equal(r, want)
    assert r[1].item() == 1.0 and r[2].item() == 1.0078125          # a tie goes to the even mantissa; just above it rounds up


def test_bf16_conv_function_rounds_what_it_says():
    g = torch.Generator().manual_seed(0)
    for Cin, all_bf16 in ((64, True), (32, False)):
        x = torch.randn(2, Cin, 48, 6, generator=g).permute(0, 1, 3, 2).contiguous()           # [2, Cin, 6, 48]
        w = (torch.randn(128, Cin, 3, 3, generator=g) * 0.1)
        b = torch.randn(128, generator=g)
        dy = torch.randn(2, 128, 6, 48, generator=g)
        xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
        with OB.enabled(True, 1):
            y = OB.conv3x3(xr, wr, br)
            y.backward(dy)
        assert torch.equal(y.detach(), F.conv2d(O.round_bf16(x), O.round_bf16(w), b, padding=1))     # the forward is covered either way
        assert torch.equal(br.grad, dy.sum(dim=(0, 2, 3)))
        dx32, dw32, _ = O.conv3x3_grads(x, w, dy)
        dxb, dwb, _ = O.conv3x3_bf16_grads(x, w, dy)
        if all_bf16:     # 64 input channels: the input gradient (128 -> 64) and the weight gradient are on the bf16 kernels too
            assert torch.allclose(xr.grad, dxb, rtol=0, atol=1e-5 * dxb.abs().max().item())
            assert torch.allclose(wr.grad, dwb, rtol=0, atol=1e-5 * dwb.abs().max().item())
            assert (wr.grad - dw32).abs().max() > 1e-4 * dw32.abs().max() and (xr.grad - dx32).abs().max() > 1e-4 * dx32.abs().max()
        else:            # 32 input channels: neither the 128 -> 32 input gradient nor the weight gradient is covered: fp32 on fp32 operands
            assert torch.allclose(xr.grad, dx32, rtol=0, atol=1e-5 * dx32.abs().max().item())
            assert torch.allclose(wr.grad, dw32, rtol=0, atol=1e-5 * dw32.abs().max().item())


def test_bf16_stride2_conv_rounds_forward_and_input_gradient():
    """A stride-2 conv the stride-2 forms of the kernel cover: forward and input gradient on rounded operands, weight gradient fp32."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 64, 24, 24, generator=g); w = torch.randn(128, 64, 3, 3, generator=g) * 0.1
    dy = torch.randn(1, 128, 12, 12, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    with OB.enabled(True, 1):
        assert OB.conv_eligible(1, 24, 24, 64, 128, 2) and not OB.conv_eligible(1, 24, 24, 48, 128, 2)
        y = OB.conv3x3(xr, wr, None, stride=2)
        y.backward(dy)
    assert torch.equal(y.detach(), F.conv2d(O.round_bf16(x), O.round_bf16(w), None, stride=2, padding=1))
    dx32, dw32, _ = O.conv3x3_grads(x, w, dy, 2)
    dxb = torch.nn.grad.conv2d_input(x.shape, O.round_bf16(w), O.round_bf16(dy), stride=2, padding=1)
    assert torch.allclose(xr.grad, dxb, rtol=0, atol=1e-5 * dxb.abs().max().item())
    assert (xr.grad - dx32).abs().max() > 1e-4 * dx32.abs().max()
    assert torch.allclose(wr.grad, dw32, rtol=0, atol=1e-5 * dw32.abs().max().item())
