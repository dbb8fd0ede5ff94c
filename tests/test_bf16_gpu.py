"""GPU parity of the OPTIONAL bf16-operand mode: the bf16 MFMA conv kernels against THEIR oracle - the same
bf16 rounding of both operands, sums in float64 (oracle/ops.py conv3x3_bf16*) - to fp32-accumulation accuracy (1e-5 of the
maximum), and against the fp32 oracle at the accuracy bf16 operands give (a few 1e-3)."""
import pytest
import torch

from oracle import ops as O

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def _nchw(y):
    return y.permute(0, 3, 1, 2).cpu()


def _close(a, b, rel):
    scale = b.abs().max().item() + 1e-30
    err = (a - b).abs().max().item()
    assert err <= rel * scale, f"max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.3e} > {rel})"


BF16_CASES = [
    # N, H, W, Cin, Cout
    (2, 48, 48, 256, 256),     # G body
    (1, 7, 48, 64, 128),       # ragged rows, 128-channel n-tiles
    (2, 12, 24, 32, 128),      # 6 x 24 tiles, one chunk
    (1, 20, 100, 64, 256),     # ragged in x
    (3, 5, 16, 96, 128),       # 9 x 16 tiles, three chunks
    (1, 30, 36, 128, 384),     # three 128-channel n-tiles
    (2, 48, 48, 64, 64),       # 64-channel layers: the four-wave workgroups (vgg19 conv1_2)
    (1, 9, 16, 128, 192),      # ... three of them per pixel tile; the input gradient runs 192 -> 128
]


@pytest.mark.parametrize("N,H,W,Cin,Cout", BF16_CASES)
def test_conv3x3_bf16_kernel(N, H, W, Cin, Cout):
    """Forward with every fused epilogue and the input gradient on the bf16 kernel."""
    from pesr_amd import ops
    x = _rand(N, Cin, H, W, seed=1); w = _rand(Cout, Cin, 3, 3, seed=2, scale=0.1); b = _rand(Cout, seed=3)
    skip = _rand(N, Cout, H, W, seed=4); mk = _rand(N, Cout, H, W, seed=5)
    ref = O.conv3x3_bf16(x, w, b)
    wf = ops.pack_conv3x3_bf16(w.cuda(), 0)
    assert torch.equal(wf.t.cpu().view(9, Cin // 32, Cout, 32)[4, 0, :, :].float(), O.round_bf16(w[:, :32, 1, 1]))   # pack = RNE
    y = ops.conv3x3_fwd(_nhwc(x), wf, b.cuda(), Cout, act=ops.ACT_RELU)
    _close(_nchw(y), torch.relu(ref), 1e-5)
    _close(_nchw(y), torch.relu(O.conv3x3(x, w, b)), 1e-2)              # and it IS the conv, to bf16-operand accuracy
    y = ops.conv3x3_fwd(_nhwc(x), wf, b.cuda(), Cout, alpha=0.1, skip=_nhwc(skip), mask=_nhwc(mk))
    _close(_nchw(y), torch.where(mk > 0, ref * 0.1, torch.zeros_like(ref)) + skip, 1e-5)
    if Cin % 64 == 0 and Cout % 32 == 0:
        dy = _rand(N, Cout, H, W, seed=6)
        dx_ref, _, _ = O.conv3x3_bf16_grads(x, w, dy)
        dx = ops.conv3x3_dgrad(_nhwc(dy), ops.pack_conv3x3_bf16(w.cuda(), 1), (N, H, W, Cin), mask=_nhwc(x), skip=_nhwc(x))
        _close(_nchw(dx), torch.where(x > 0, dx_ref, torch.zeros_like(dx_ref)) + x, 1e-5)


def test_conv3x3_bf16_pixel_shuffle_fused():
    """The upsampler convs on the bf16 kernel: fused PixelShuffle store (forward), pixel-unshuffle load (input gradient)."""
    import torch.nn.functional as F
    from pesr_amd import ops
    N, H, W, C = 1, 12, 24, 256
    x = _rand(N, C, H, W, seed=1); w = _rand(4 * C, C, 3, 3, seed=2, scale=0.1); b = _rand(4 * C, seed=3)
    ref = F.pixel_shuffle(O.conv3x3_bf16(x, w, b), 2)
    y = ops.conv3x3_fwd(_nhwc(x), ops.pack_conv3x3_bf16(w.cuda(), 0, ps=True), ops.pack_bias_ps(b.cuda()), 4 * C, ps_out=True)
    _close(_nchw(y), ref, 1e-5)
    dys = _rand(N, C, 2 * H, 2 * W, seed=4)
    dx_ref, _, _ = O.conv3x3_bf16_grads(x, w, F.pixel_unshuffle(dys, 2))
    dx = ops.conv3x3_dgrad(_nhwc(dys), ops.pack_conv3x3_bf16(w.cuda(), 1, ps=True), (N, H, W, C), ps_in=True)
    _close(_nchw(dx), dx_ref, 1e-5)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 48, 48, 256, 256), (1, 5, 48, 64, 128), (2, 7, 96, 64, 128), (1, 12, 48, 128, 384),
                                             (3, 2, 144, 64, 128)])
def test_conv3x3_wgrad_bf16_kernel(N, H, W, Cin, Cout):
    """Weight + bias gradient on the bf16 kernel (transposed LDS reads) against its oracle, overwrite and accumulate forms;
    bit-reproducible (fixed-order split-K reduce)."""
    from pesr_amd import ops
    x = _rand(N, Cin, H, W, seed=900); w = _rand(Cout, Cin, 3, 3, seed=910, scale=0.1); dy = _rand(N, Cout, H, W, seed=920)
    _, dw_ref, db_ref = O.conv3x3_bf16_grads(x, w, dy)
    dw, db = ops.conv3x3_wgrad_bf16(_nhwc(x), _nhwc(dy), alpha=0.5)
    _close(dw.cpu(), 0.5 * dw_ref, 1e-5); _close(db.cpu(), 0.5 * db_ref, 1e-5)
    _, dw32, _ = O.conv3x3_grads(x, w, dy)
    _close(dw.cpu(), 0.5 * dw32, 1e-2)                                  # and it IS the weight gradient, to bf16-operand accuracy
    dw2, db2 = ops.conv3x3_wgrad_bf16(_nhwc(x), _nhwc(dy), alpha=0.5)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    ops.conv3x3_wgrad_bf16(_nhwc(x), _nhwc(dy), alpha=0.5, dw_out=dw2, db_out=db2, accumulate=True)
    _close(dw2.cpu(), dw_ref, 1e-5); _close(db2.cpu(), db_ref, 1e-5)


def test_conv3x3_wgrad_bf16_pixel_shuffle_fused():
    """Weight gradient of an upsampler conv (its output gradient arrives pixel-shuffled) on the bf16 kernel."""
    import torch.nn.functional as F
    from pesr_amd import ops
    N, H, W, C = 2, 6, 48, 128
    x = _rand(N, C, H, W, seed=1); w = _rand(4 * C, C, 3, 3, seed=2, scale=0.1)
    dys = _rand(N, C, 2 * H, 2 * W, seed=4)
    _, dw_ref, db_ref = O.conv3x3_bf16_grads(x, w, F.pixel_unshuffle(dys, 2))
    dw, db = ops.conv3x3_wgrad_bf16(_nhwc(x), _nhwc(dys), ps_in=True)
    _close(dw.cpu(), dw_ref, 1e-5); _close(db.cpu(), db_ref, 1e-5)


@pytest.fixture
def bf16_mode():
    """ops.PRECISION = "bf16" with the workgroup-count floor lowered, so that the small test networks take the bf16 kernels."""
    from pesr_amd import ops
    old = (ops.PRECISION, ops.BF16_MIN_WGS)
    ops.set_precision("bf16"); ops.BF16_MIN_WGS = 1
    yield
    ops.PRECISION, ops.BF16_MIN_WGS = old


def test_gan_step_in_bf16_mode_vs_its_oracle(bf16_mode):
    """One full GAN step (256-channel Generator with 2 blocks, Discriminator, VGG, 48 -> 192 patches) with --precision bf16
    against the oracle's restatement of that mode (oracle/bf16.py: the same convs round the same operands): losses, every
    Generator gradient through the post-Adam parameters, Discriminator parameters.  Also: the step is NOT the fp32 step
    (the bf16 kernels really ran), and the dispatch put the expected layers on them."""
    import sys, os, warnings
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import adam_close, dis_sd, gen_sd, vgg_sd
    from model import Discriminator, Generator, VGG
    from oracle import bf16 as OB, detrand, step as OS
    from pesr_amd import ops
    from pesr_amd.optim import FlatAdam
    from pesr_amd.step import Trainer
    warnings.filterwarnings("ignore", message=".*pretrained vgg19.*")
    C, depth, ps, B = 256, 2, 48, 2
    g_sd, d_sd, v_sd = gen_sd(C, depth), dis_sd(ps), vgg_sd()
    G = Generator({"num_channels": C, "depth": depth, "res_scale": 0.1}); G.load_state_dict(g_sd); G.cuda()
    D = Discriminator({"patch_size": ps, "spectral_norm": False}); D.load_state_dict(d_sd); D.cuda()
    V = VGG(); V.load_state_dict(v_sd); V.cuda()
    tr = Trainer(G, D, V, FlatAdam(G.parameters(), lr=5e-5), FlatAdam(D.parameters(), lr=5e-5))
    lr = detrand.image_batch((B, 3, ps, ps), 700); hr = detrand.image_batch((B, 3, 4 * ps, 4 * ps), 701)
    cfg = {"depth": depth, "res_scale": 0.1, "learning_rate": 5e-5}
    def oracle(b16, dtype):
        cv = lambda sd: {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        st_ = OS.TrainState(cv(g_sd), cv(d_sd), cv(v_sd), cfg)
        with OB.enabled(b16, 1):
            return st_, OS.gan_step(st_, lr.to(dtype), hr.to(dtype))
    st, ref = oracle(True, torch.float32)
    _, ref64 = oracle(True, torch.float64)           # the same rounding, float64 sums: the truth the mode defines
    _, ref32 = oracle(False, torch.float32)          # the reference's fp32 arithmetic
    ops.FLOPS.start()
    log = tr.gan_step(lr.cuda(), hr.cuda())
    fl = ops.FLOPS.stop()
    fam = fl["by_kernel_family"]
    assert fam["bf16"][0] > 0.6 * fl["algorithmic"], fam          # the bulk of the conv flops ran on the bf16 kernels
    # Rounding to bf16 turns an fp32-summation-order difference in an activation that sits on a rounding boundary into a 2^-8
    # relative jump of that operand, so exact-arithmetic-equivalent evaluations of the mode differ by up to ~1.5e-3 on the losses
    # that go through the Discriminator (g 0.6 - 1.9e-3, d 4e-4, vgg 3e-5, tv 1e-6 over the draws seen so far).  The bound is
    # MEASURED per loss, not a constant: the amplitude of that noise over five draws of the mode's own restatement - fp32 sums
    # instead of float64 sums, and every weight moved one fp32 ulp up / down (bench.py's noise floor), each against the float64-sum
    # truth resp. its own unperturbed value - and our error may be 3 x the largest draw, never tighter than 2e-5.
    def moved(direction, dtype):
        mv = lambda sd: {k: (torch.nextafter(v, torch.full_like(v, direction)).to(dtype) if v.is_floating_point() else v.clone())
                         for k, v in sd.items()}
        with OB.enabled(True, 1):
            return OS.gan_step(OS.TrainState(mv(g_sd), mv(d_sd), mv(v_sd), cfg), lr.to(dtype), hr.to(dtype))
    up64, dn64, up32, dn32 = moved(float("inf"), torch.float64), moved(-float("inf"), torch.float64), moved(float("inf"), torch.float32), moved(-float("inf"), torch.float32)
    for k in ("vgg", "g", "tv", "d"):
        draws = [abs(ref[k] - ref64[k]), abs(up64[k] - ref64[k]), abs(dn64[k] - ref64[k]), abs(up32[k] - ref64[k]), abs(dn32[k] - ref64[k])]
        own = max(draws) / abs(ref64[k])
        err = abs(float(log[k]) - ref64[k]) / abs(ref64[k])
        print(f"bf16 step loss {k}: error {err:.2e}, noise amplitude over 5 oracle draws {own:.2e}")
        assert err <= max(3.0 * own, 2e-5), (k, float(log[k]), ref[k], ref64[k], err, own)
    # the mode is visibly not fp32 where the bf16 rounding dominates that noise: the perceptual and TV losses sit on the bf16
    # oracle's values, several times closer than the fp32 oracle's are
    for k in ("vgg", "tv"):
        assert abs(float(log[k]) - ref64[k]) < 0.3 * abs(ref32[k] - ref64[k]), (k, float(log[k]), ref64[k], ref32[k])
    # Post-Adam parameters: the first Adam step moves every element by lr * sign(gradient), and the mode's noise floor (above) flips
    # the sign of more near-zero gradients than fp32's does (3.6 % of embed.weight here), so helpers.adam_close's 3 % flip budget
    # does not apply; what must hold: no element off by more than the sign-flip bound, and a mean error far below lr (a tensor
    # trained with a wrong gradient, or not at all, is off by >= lr on average).
    def params_close(got, want, what):
        err = (got.detach().cpu().double() - want.detach().double()).abs()
        assert err.max().item() <= 2.2 * 5e-5 + 2e-5 * want.abs().max().item(), (what, err.max().item() / 5e-5)
        assert err.mean().item() <= 0.25 * 5e-5, (what, err.mean().item() / 5e-5)
    for k, v in G.state_dict().items():
        params_close(v, st.g[k], "G." + k)
    for k, v in D.state_dict().items():
        if "running" in k or "num_batches" in k:
            continue
        params_close(v, st.d[k], "D." + k)


def test_generator_gradients_in_bf16_mode_vs_its_oracle(bf16_mode):
    """Generator forward + L1 backward (256 channels, 2 blocks, 48 -> 192) in the bf16 mode: output and every parameter gradient
    against the oracle's restatement with float64 sums; per tensor the error may be 3 x the error of the restatement with fp32
    sums (the mode's own noise floor, see the step test) or twice the worst such error in the network, never tighter than 2e-4
    of the gradient's maximum."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import gen_sd
    from model import Generator
    from oracle import bf16 as OB, detrand, model as OM
    from pesr_amd import functional as PF
    from pesr_amd.model.basic import nhwc
    C, depth, ps, B = 256, 2, 48, 2
    g_sd = gen_sd(C, depth)
    lr = detrand.image_batch((B, 3, ps, ps), 710); hr = detrand.image_batch((B, 3, 4 * ps, 4 * ps), 711)

    def oracle(dtype):
        sd = {k: v.clone().to(dtype).requires_grad_(True) for k, v in g_sd.items()}
        with OB.enabled(True, 1):
            sr = OM.generator_forward(sd, lr.to(dtype), depth, 0.1)
            (sr - hr.to(dtype)).abs().mean().backward()
        return sr.detach(), {k: v.grad for k, v in sd.items()}
    sr32, g32 = oracle(torch.float32)
    sr64, g64 = oracle(torch.float64)
    G = Generator({"num_channels": C, "depth": depth, "res_scale": 0.1}); G.load_state_dict(g_sd); G.cuda()
    sr = G(lr.cuda())
    own = (sr32.double() - sr64).abs().max().item()
    assert (sr.cpu().double() - sr64).abs().max().item() <= max(3.0 * own, 2e-3), own          # 0 .. 255 scale
    PF.l1_loss(nhwc(sr), nhwc(hr.cuda().contiguous(memory_format=torch.channels_last))).backward()
    owns = {k: (g32[k].double() - g64[k]).abs().max().item() / g64[k].abs().max().item() for k in g64}
    floor = max(owns.values())           # a tensor's own error is a one-sample estimate of noise: never demand better than twice
    for k, p in G.named_parameters():    # the worst error the restatement itself shows anywhere in the network (helpers.grads_vs_fp64)
        err = (p.grad.cpu().double() - g64[k]).abs().max().item() / g64[k].abs().max().item()
        assert err <= max(3.0 * owns[k], 2.0 * floor, 2e-4), (k, err, owns[k], floor)


def test_test_entrypoint_in_bf16_mode(tmp_path, monkeypatch, bf16_mode):
    """test.py --precision bf16 on a folder of PNGs (128-channel generator: its body convs take the bf16 kernel) against the
    oracle's plumbing with the bf16 restatement switched on: at most one grey level apart on < 1 % of the values."""
    import importlib.util, os, sys
    import numpy as np
    from PIL import Image
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import gen_sd
    from oracle import bf16 as OB, detrand, image as OI, model as OM
    from pesr_amd import ops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("entry_test_bf16", os.path.join(root, "test.py"))
    T = importlib.util.module_from_spec(spec); spec.loader.exec_module(T)
    monkeypatch.chdir(tmp_path)
    lr_dir = tmp_path / "data" / "origin" / "test" / "Toy" / "LR"
    lr_dir.mkdir(parents=True)
    arr = detrand.image_batch((24, 36, 3), 931).numpy().astype(np.uint8)
    Image.fromarray(arr).save(lr_dir / "a.png")
    sd = gen_sd(128, 2, seed=5)
    torch.save(sd, tmp_path / "perc.pt")
    ops.FLOPS.start()
    T.main(["--dataset", "Toy", "--perceptual_model", str(tmp_path / "perc.pt"), "--num_channels", "128", "--num_blocks", "2",
            "--alpha", "1", "--save_path", str(tmp_path / "out"), "--precision", "bf16"])
    fam = ops.FLOPS.stop()["by_kernel_family"]
    assert fam.get("bf16", [0.0])[0] > 0, fam
    got = np.asarray(Image.open(tmp_path / "out" / "Toy" / "a.png").convert("RGB")).astype(np.int32)
    x = torch.from_numpy(arr.transpose(2, 0, 1)[None].astype(np.float32))
    with torch.no_grad(), OB.enabled(True, 1):
        want = OI.tensor_to_img(OM.generator_forward(sd, x, 2, 0.1)).astype(np.int32)
    assert got.shape == want.shape == (96, 144, 3)
    assert np.abs(got - want).max() <= 1 and (got != want).mean() < 0.01, (np.abs(got - want).max(), (got != want).mean())


def test_bf16_random_shape_sweep():
    """Random shapes through the bf16 forward / input-gradient / weight-gradient kernels (every tile shape the planner can pick,
    ragged rows and columns, 64- / 128- / 256-channel workgroups) against the rounded-operand oracle."""
    import random
    from pesr_amd import ops
    rng = random.Random(29)
    for it in range(12):
        N = rng.choice([1, 2, 3]); H = rng.randint(1, 40); W = rng.choice([4, 12, 16, 20, 24, 36, 48, 50, 72, 96])
        Cin = rng.choice([32, 64, 96, 128]); Cout = rng.choice([64, 128, 192, 256])
        x = _rand(N, Cin, H, W, seed=1500 + it); w = _rand(Cout, Cin, 3, 3, seed=1600 + it, scale=0.1); b = _rand(Cout, seed=1700 + it)
        ref = O.conv3x3_bf16(x, w, b)
        y = ops.conv3x3_fwd(_nhwc(x), ops.pack_conv3x3_bf16(w.cuda(), 0), b.cuda(), Cout)
        _close(_nchw(y), ref, 1e-5)
        dy = _rand(N, Cout, H, W, seed=1800 + it)
        dx_ref, dw_ref, db_ref = O.conv3x3_bf16_grads(x, w, dy)
        if Cout % 32 == 0 and Cin % 64 == 0:
            dx = ops.conv3x3_dgrad(_nhwc(dy), ops.pack_conv3x3_bf16(w.cuda(), 1), (N, H, W, Cin))
            _close(_nchw(dx), dx_ref, 1e-5)
        if W % 48 == 0 and Cin % 64 == 0 and Cout % 128 == 0:
            dw, db = ops.conv3x3_wgrad_bf16(_nhwc(x), _nhwc(dy))
            _close(dw.cpu(), dw_ref, 1e-5); _close(db.cpu(), db_ref, 1e-5)


def test_bf16_determinism_race_screen():
    """The bf16 kernels hand staged data between waves by counted waits and barriers only, like the fp32 ones: a misplaced wait
    shows up as rare run-to-run differences.  Every sum has a fixed order, so repeated launches must give identical bits - the
    256- / 128- / 64-channel workgroups of the forward kernel, a fused-PixelShuffle shape and the weight gradient's split-K."""
    from pesr_amd import ops
    shapes = [(16, 48, 48, 256, 256, False), (2, 24, 24, 256, 1024, True), (4, 48, 48, 128, 128, False), (4, 96, 96, 64, 64, False)]
    for (N, H, W, Cin, Cout, ps) in shapes:
        x = _nhwc(_rand(N, Cin, H, W, seed=21)); w = _rand(Cout, Cin, 3, 3, seed=22, scale=0.05).cuda()
        dy = _nhwc(_rand(N, Cout // 4 if ps else Cout, 2 * H if ps else H, 2 * W if ps else W, seed=23))
        wf, wd = ops.pack_conv3x3_bf16(w, 0, ps=ps), ops.pack_conv3x3_bf16(w, 1, ps=ps)
        fwd = lambda: ops.conv3x3_fwd(x, wf, None, Cout, act=ops.ACT_RELU, ps_out=ps)
        dgr = lambda: ops.conv3x3_dgrad(dy, wd, (N, H, W, Cin), mask=x, ps_in=ps)
        wgr = (lambda: ops.conv3x3_wgrad_bf16(x, dy, ps_in=ps)) if (Cout % 128 == 0 and W % 48 == 0) else None
        y0, d0 = fwd(), dgr()
        g0 = wgr() if wgr else None
        for _ in range(12):
            assert torch.equal(fwd(), y0) and torch.equal(dgr(), d0)
            if wgr:
                g, b = wgr()
                assert torch.equal(g, g0[0]) and torch.equal(b, g0[1])


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 48, 48, 128, 128), (1, 24, 24, 256, 256), (2, 24, 24, 512, 512), (1, 13, 11, 32, 128),
                                             (1, 30, 70, 64, 64), (2, 96, 96, 64, 64), (1, 7, 300, 64, 128), (3, 2, 2, 32, 192)])
def test_conv3x3_bf16_stride2_forward(N, H, W, Cin, Cout):
    """The stride-2 form of the bf16 kernel (the Discriminator's down-sampling convs; de-interleaved halo columns): forward with
    the fused epilogues, odd sizes, the four-wave 64-channel workgroups; bit-reproducible."""
    from pesr_amd import ops
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = _rand(N, Cin, H, W, seed=31); w = _rand(Cout, Cin, 3, 3, seed=32, scale=0.1); b = _rand(Cout, seed=33)
    skip = _rand(N, Cout, OH, OW, seed=34); mk = _rand(N, Cout, OH, OW, seed=35)
    ref = O.conv3x3_bf16(x, w, b, stride=2)
    wf = ops.pack_conv3x3_bf16(w.cuda(), 0)
    y = ops.conv3x3_fwd(_nhwc(x), wf, b.cuda(), Cout, 2, act=ops.ACT_LRELU, slope=0.2)
    assert y.shape == (N, OH, OW, Cout)
    _close(_nchw(y), torch.nn.functional.leaky_relu(ref, 0.2), 1e-5)
    _close(_nchw(y), torch.nn.functional.leaky_relu(O.conv3x3(x, w, b, 2), 0.2), 1e-2)       # it IS the conv, to bf16-operand accuracy
    y2 = ops.conv3x3_fwd(_nhwc(x), wf, b.cuda(), Cout, 2, alpha=0.1, skip=_nhwc(skip), mask=_nhwc(mk))
    _close(_nchw(y2), torch.where(mk > 0, ref * 0.1, torch.zeros_like(ref)) + skip, 1e-5)
    for _ in range(4):
        assert torch.equal(ops.conv3x3_fwd(_nhwc(x), wf, b.cuda(), Cout, 2, act=ops.ACT_LRELU, slope=0.2), y)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 48, 48, 128, 128), (1, 24, 24, 256, 256), (2, 24, 24, 512, 512), (1, 13, 11, 64, 96),
                                             (1, 30, 70, 64, 64), (2, 96, 96, 64, 64), (1, 7, 300, 128, 64), (3, 2, 2, 64, 32), (1, 1, 5, 64, 32)])
def test_conv3x3_bf16_stride2_input_gradient(N, H, W, Cin, Cout):
    """The stride-2 input gradient on the bf16 kernel (four parity classes, one launch): odd sizes, mask / skip epilogue,
    bit-reproducible.  x [N, Cin, H, W] is the forward conv's input, the conv maps Cin -> Cout."""
    from pesr_amd import ops
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = _rand(N, Cin, H, W, seed=41); w = _rand(Cout, Cin, 3, 3, seed=42, scale=0.1); dy = _rand(N, Cout, OH, OW, seed=43)
    xr = O.round_bf16(x).double().requires_grad_(True)
    torch.nn.functional.conv2d(xr, O.round_bf16(w).double(), None, stride=2, padding=1).backward(O.round_bf16(dy).double())
    dx_ref = xr.grad.float()
    wd = ops.pack_conv3x3_bf16(w.cuda(), 1)
    dx = ops.conv3x3_dgrad(_nhwc(dy), wd, (N, H, W, Cin), 2)
    _close(_nchw(dx), dx_ref, 1e-5)
    dx32, _, _ = O.conv3x3_grads(x, w, dy, 2)
    _close(_nchw(dx), dx32, 1e-2)
    dx2 = ops.conv3x3_dgrad(_nhwc(dy), wd, (N, H, W, Cin), 2, alpha=0.5, mask=_nhwc(x), skip=_nhwc(x))
    _close(_nchw(dx2), torch.where(x > 0, 0.5 * dx_ref, torch.zeros_like(dx_ref)) + x, 1e-5)
    for _ in range(4):
        assert torch.equal(ops.conv3x3_dgrad(_nhwc(dy), wd, (N, H, W, Cin), 2), dx)
