"""Tensor-level wrappers over the C ABI (no autograd here; see functional.py).

All activation tensors handled here are fp32, on a GPU, NHWC-contiguous: shape [N, H, W, C].
(The nn.Module boundary works with logical NCHW tensors in torch.channels_last memory format; a
`.permute(0, 2, 3, 1)` of those is exactly this view - no copies.)
There is no CPU path: a CPU tensor raises.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2


def _chk(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise _lib.PesrHipError(f"{name}: pesr_amd ops run only on a GPU (HIP) tensor; there is no CPU fallback")
    if t.dtype != torch.float32:
        raise _lib.PesrHipError(f"{name}: expected float32, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.PesrHipError(f"{name}: expected a contiguous tensor (NHWC physical layout)")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


_workspaces = {}


def workspace(nbytes: int, device: torch.device) -> torch.Tensor:
    """Grow-only per-device scratch buffer (owned by torch's caching allocator)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


# ------------------------------------------------------------------------------------------------
# weight packing
# ------------------------------------------------------------------------------------------------
def pack_conv3x3(w: torch.Tensor, mode: int, ps: bool = False) -> torch.Tensor:
    """OIHW [O, I, 3, 3] -> packed [9, R/16, Nn, 16] (mode 0: forward, mode 1: dgrad)."""
    _chk(w, "pack_conv3x3.w")
    O, I = w.shape[0], w.shape[1]
    assert w.shape[2:] == (3, 3)
    out = torch.empty(9 * O * I, dtype=torch.float32, device=w.device)
    _lib.check(_lib.lib().pesr_pack_conv3x3(_p(w), _p(out), O, I, mode, int(ps), _stream()), "pesr_pack_conv3x3")
    return out


def pack_bias_ps(b: torch.Tensor) -> torch.Tensor:
    _chk(b, "pack_bias_ps.b")
    out = torch.empty_like(b)
    _lib.check(_lib.lib().pesr_pack_bias_ps(_p(b), _p(out), b.numel(), _stream()), "pesr_pack_bias_ps")
    return out


# ------------------------------------------------------------------------------------------------
# 3x3 conv family
# ------------------------------------------------------------------------------------------------
def conv3x3_fwd(x: torch.Tensor, wp: torch.Tensor, bias: Optional[torch.Tensor], cout: int, stride: int = 1,
                alpha: float = 1.0, act: int = ACT_NONE, slope: float = 0.0, skip: Optional[torch.Tensor] = None,
                mask: Optional[torch.Tensor] = None, ps_out: bool = False) -> torch.Tensor:
    _chk(x, "conv3x3_fwd.x")
    N, H, W, Cin = x.shape
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    if ps_out:
        y = torch.empty((N, 2 * OH, 2 * OW, cout // 4), dtype=torch.float32, device=x.device)
    else:
        y = torch.empty((N, OH, OW, cout), dtype=torch.float32, device=x.device)
    for t, n in ((skip, "skip"), (mask, "mask")):
        if t is not None:
            _chk(t, f"conv3x3_fwd.{n}")
            assert t.shape == y.shape, (t.shape, y.shape)
    rc = _lib.lib().pesr_conv3x3_fwd(_p(x), _p(wp), _p(bias), _p(skip), _p(mask), _p(y), N, H, W, Cin, cout, stride,
                                     alpha, act, slope, int(ps_out), _stream())
    _lib.check(rc, f"pesr_conv3x3_fwd[{N}x{H}x{W}x{Cin}->{cout},s{stride}]")
    return y


def conv3x3_dgrad(dy: torch.Tensor, wpd: torch.Tensor, in_shape, stride: int = 1, alpha: float = 1.0,
                  mask: Optional[torch.Tensor] = None, skip: Optional[torch.Tensor] = None,
                  ps_in: bool = False) -> torch.Tensor:
    """dx for a conv whose forward input had NHWC shape `in_shape`; dy is the (possibly shuffled) output grad."""
    _chk(dy, "conv3x3_dgrad.dy")
    N, H, W, Cin = in_shape
    cout = dy.shape[3] * (4 if ps_in else 1)
    dx = torch.empty((N, H, W, Cin), dtype=torch.float32, device=dy.device)
    for t, n in ((skip, "skip"), (mask, "mask")):
        if t is not None:
            _chk(t, f"conv3x3_dgrad.{n}")
            assert t.shape == dx.shape
    rc = _lib.lib().pesr_conv3x3_dgrad(_p(dy), _p(wpd), _p(mask), _p(skip), _p(dx), N, H, W, Cin, cout, stride, alpha,
                                       int(ps_in), _stream())
    _lib.check(rc, f"pesr_conv3x3_dgrad[{N}x{H}x{W}x{Cin}<-{cout},s{stride}]")
    return dx


def conv3x3_wgrad(x: torch.Tensor, dy: torch.Tensor, stride: int = 1, alpha: float = 1.0, want_bias: bool = True,
                  ps_in: bool = False):
    """(dw [O, I, 3, 3], db [O] | None)."""
    _chk(x, "conv3x3_wgrad.x")
    _chk(dy, "conv3x3_wgrad.dy")
    N, H, W, Cin = x.shape
    cout = dy.shape[3] * (4 if ps_in else 1)
    L = _lib.lib()
    nbytes = L.pesr_conv3x3_wgrad_workspace_bytes(N, H, W, Cin, cout, stride)
    if nbytes == 0:
        raise _lib.PesrHipError(f"pesr_conv3x3_wgrad: unsupported shape Cin={Cin} Cout={cout} stride={stride}")
    ws = workspace(nbytes, x.device)
    dw = torch.empty((cout, Cin, 3, 3), dtype=torch.float32, device=x.device)
    db = torch.empty((cout,), dtype=torch.float32, device=x.device) if want_bias else None
    rc = L.pesr_conv3x3_wgrad(_p(x), _p(dy), _p(dw), _p(db), N, H, W, Cin, cout, stride, alpha, int(ps_in), _p(ws),
                              ws.numel(), _stream())
    _lib.check(rc, f"pesr_conv3x3_wgrad[{N}x{H}x{W}x{Cin}->{cout},s{stride}]")
    return dw, db
