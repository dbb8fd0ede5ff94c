"""CPU: plumbing of the kept entry points (utils.py, test.py x8 ensemble, data.py, train.py flags) - reference
utils.py known answers (golden GV9), x8 self-ensemble vs the oracle's numpy restatement (GV10)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from helpers import load_golden, x8_toy_model
from oracle import detrand
from oracle import image as OI

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location("entry_" + name, os.path.join(ROOT, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_utils_known_answers_gv9():
    U = _load("utils")
    g = load_golden("gv9_utils")
    a = detrand.image_batch((1, 3, 16, 20), 41)
    b = a + detrand.uniform((1, 3, 16, 20), 42, -20, 20)
    assert abs(U.compute_PSNR(a.clone(), b.clone()) - float(g["psnr"])) < 1e-9
    assert abs(OI.psnr_y(a, b) - float(g["psnr"])) < 1e-9
    [img] = U.tensors_to_imgs([b * 1.1 - 5])
    assert np.array_equal(img, g["img"]) and np.array_equal(OI.tensor_to_img(b * 1.1 - 5), g["img"])
    assert np.allclose(U.rgb2y(img.astype(np.float64)), g["y"], rtol=0, atol=1e-12)
    [t] = U.imgs_to_tensors([img], torch.device("cpu"))
    assert t.shape == (1, 3, 16, 20) and t.dtype == torch.float32 and torch.equal(t[0].permute(1, 2, 0).to(torch.uint8), torch.from_numpy(img))


def test_x8_self_ensemble_matches_reference_semantics_gv10():
    T = _load("test")
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(3, 3, 3, padding=1)          # asymmetric toy "model": not equivariant to flips

    def model(x):
        return torch.nn.functional.interpolate(conv(x), scale_factor=2, mode="nearest")

    img = detrand.image_batch((1, 3, 6, 9), 7)            # non-square: transposes change the shape
    with torch.no_grad():
        got = T.x8_forward(img, model)
        ref = OI.x8_forward(img, model)
    assert got.shape == (1, 3, 12, 18)
    assert torch.allclose(got, ref, rtol=0, atol=1e-5)
    # an equivariant model is a fixed point of the ensemble
    ident = lambda x: x * 2.0                              # noqa: E731
    assert torch.allclose(T.x8_forward(img, ident), img * 2.0)


def test_oracle_and_product_x8_pinned_by_the_reference_gv10():
    """GV10 = outputs of the reference's OWN x8_forward (reference test.py:45-74, imported by tests/golden/make_golden_plumbing.py)
    on a non-equivariant toy model with exact fp32 arithmetic: the oracle's restatement and the product's test.py reproduce them
    bit for bit."""
    T = _load("test")
    g = load_golden("gv10_x8")
    model = x8_toy_model(g["weight_seed"], g["bias_seed"])
    with torch.no_grad():
        for i, (shape, seed) in enumerate(zip(g["shapes"], g["seeds"])):
            img = detrand.image_batch(tuple(int(v) for v in shape), int(seed))
            want = torch.from_numpy(g[f"out{i}"])
            assert (want - model(img)).abs().max() > 1.0          # the ensemble differs from a plain forward: a wrong inverse shows
            assert torch.equal(OI.x8_forward(img, model), want), f"oracle x8, case {i}"
            assert torch.equal(T.x8_forward(img, model), want), f"product x8, case {i}"


def _gv12_cases():
    g = load_golden("gv12_crop_aug")
    images = {}
    for n, (ih, iw, ps, s_lr, s_hr, seed, is_random, y, x, aug) in enumerate(g["cases"].tolist()):
        if (s_lr, s_hr) not in images:
            mk = lambda h, w, s: detrand.image_batch((1, 3, h, w), s)[0].permute(1, 2, 0).contiguous().numpy().astype(np.uint8)  # noqa: E731
            images[(s_lr, s_hr)] = (mk(ih, iw, s_lr), mk(4 * ih, 4 * iw, s_hr))
        inp, lbl = images[(s_lr, s_hr)]
        yield n, inp, lbl, ps, seed, is_random, y, x, aug, torch.from_numpy(g[f"inp{n}"]).float(), torch.from_numpy(g[f"lbl{n}"]).float()


def test_oracle_and_product_crop_augment_pinned_by_the_reference_gv12():
    """GV12 = outputs of the reference's OWN SRDataset._crop / _aug_data / _to_tensor (reference data.py:79-126) under
    random.seed(s): all 8 aug_idx values, crops on every image edge, both crop types, image sizes that are multiples of nothing.
    The oracle's explicit-draw restatement and the product's data.py helpers reproduce them bit for bit."""
    import random
    D = _load("data")
    augs, n_cases = set(), 0
    for n, inp, lbl, ps, seed, is_random, y, x, aug, want_lr, want_hr in _gv12_cases():
        random.seed(seed)                                   # the stored draws ARE what this seed draws (the reference's call order)
        if is_random:
            assert (random.randint(0, inp.shape[0] - ps), random.randint(0, inp.shape[1] - ps)) == (y, x)
        assert random.randint(0, 7) == aug
        got_lr, got_hr = OI.crop_augment(inp, lbl, ps, y, x, aug)
        assert got_lr.dtype == torch.float32 and torch.equal(got_lr, want_lr) and torch.equal(got_hr, want_hr), f"oracle, case {n}"
        pl, ph = D.augment(inp[y:y + ps, x:x + ps], lbl[4 * y:4 * (y + ps), 4 * x:4 * (x + ps)], aug)
        assert torch.equal(D.to_tensor(pl), want_lr) and torch.equal(D.to_tensor(ph), want_hr), f"product data.py, case {n}"
        augs.add(aug); n_cases += 1
    assert augs == set(range(8)) and n_cases >= 40


def test_data_augment_and_shapes():
    D = _load("data")
    ds = D.SyntheticSRDataset(4, 12)
    lr, hr = ds[1]
    assert lr.shape == (3, 12, 12) and hr.shape == (3, 48, 48) and lr.min() >= 0 and lr.max() <= 255
    assert torch.equal(ds[1][0], lr)                       # deterministic per index
    a = np.arange(2 * 3 * 3).reshape(2, 3, 3)
    seen = set()
    for idx in range(8):
        l, h = D.augment(a, a, idx)
        seen.add(l.tobytes() + bytes(l.shape))
        assert np.array_equal(l, h)
    assert len(seen) == 8                                  # the 8 dihedral variants are distinct


def test_train_flags_match_reference_defaults():
    Tm = _load("train")
    a = Tm.build_parser().parse_args([])
    expect = dict(scale=4, num_channels=256, num_blocks=32, res_scale=0.1, phase="train", batch_size=16, learning_rate=5e-5,
                  lr_step=120, num_epochs=200, num_repeats=20, patch_size=24, snapshot_every=10, gan_type="RSGAN", GP=False,
                  spectral_norm=False, focal_loss=True, fl_gamma=1, alpha_vgg=50, alpha_gan=1, alpha_tv=1e-6, alpha_l1=0)
    for k, v in expect.items():
        assert getattr(a, k) == v, k
    assert Tm.build_parser().parse_args(["--focal_loss", "false"]).focal_loss is False


def test_product_focal_loss_matches_reference_values_and_q4_gradient():
    """pesr_amd.model.FocalLoss (scalar-sized torch ops, runs anywhere): forward vs the values the REFERENCE module gave
    (golden GV5), backward vs the torch-0.4-semantics closed form."""
    import pytest
    from oracle import step as OS
    from pesr_amd.model.focal_loss import FocalLoss
    g = load_golden("gv5_focal")
    x = torch.from_numpy(g["x"])
    for gamma in (0, 1, 2):
        for t in (0, 1):
            tt = torch.full_like(x, float(t))
            xr = x.clone().requires_grad_(True)
            loss = FocalLoss(gamma)(xr, tt)
            assert float(loss) == pytest.approx(float(g[f"mean_g{gamma}_t{t}"]), rel=1e-6)
            loss.backward()
            ref = OS.focal_loss_grad_closed_form(x, tt, gamma)
            assert torch.allclose(xr.grad, ref, rtol=1e-5, atol=1e-9)


def test_seeded_construction_matches_the_reference_when_it_is_available():
    """Build container only (skipped where /root/reference does not exist): torch.manual_seed(s) + construction gives
    bit-identical parameters and the same state_dict key order as the reference's modules."""
    import subprocess, sys
    import pytest
    if not os.path.isdir("/root/reference/model"):
        pytest.skip("reference not present (GPU box)")
    code = r'''
import sys, types, torch
sys.path.insert(0, %r)
from model import Generator, Discriminator
torch.manual_seed(3)
G = Generator({'depth': 2, 'num_channels': 16, 'res_scale': 0.1}); D = Discriminator({'patch_size': 8, 'spectral_norm': False})
mine = [(k, v.clone()) for k, v in list(G.state_dict().items()) + list(D.state_dict().items())]
for k in [k for k in list(sys.modules) if k == 'model' or k.startswith('model.')]:
    del sys.modules[k]
sys.path.remove(%r)
tv, tvm = types.ModuleType('torchvision'), types.ModuleType('torchvision.models')
tvm.vgg19 = lambda pretrained=False, **k: None; tv.models = tvm
sys.modules['torchvision'] = tv; sys.modules['torchvision.models'] = tvm
sys.path.insert(0, '/root/reference')
import model as R
torch.manual_seed(3)
Gr = R.Generator({'depth': 2, 'num_channels': 16, 'res_scale': 0.1}); Dr = R.Discriminator({'patch_size': 8, 'spectral_norm': False})
ref = list(Gr.state_dict().items()) + list(Dr.state_dict().items())
assert [k for k, _ in mine] == [k for k, _ in ref]
assert all(torch.equal(a, b) for (_, a), (_, b) in zip(mine, ref))
print("identical")
''' % (ROOT, ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "identical" in out.stdout, out.stderr[-2000:]


def test_lr_schedule_order_matches_reference_under_its_pinned_torch():
    """train.py steps its StepLR at the END of each epoch; on this torch that reproduces the reference's schedule under
    torch 0.4 (epoch e trains at lr*0.5^((e-1)//lr_step): epochs 1..lr_step at lr, the first halving at lr_step+1)."""
    import re
    import torch.optim.lr_scheduler as S
    from oracle import step as OS
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=5e-5)
    sch = S.StepLR(opt, step_size=3, gamma=0.5)
    seen = []
    for epoch in range(1, 9):
        seen.append(opt.param_groups[0]["lr"])      # the rate epoch `epoch` trains with
        opt.step()
        sch.step()
    assert seen == [pytest_approx(OS.step_lr(5e-5, e, 3)) for e in range(1, 9)]
    assert seen[2] == 5e-5 and seen[3] == 2.5e-5     # epochs lr_step and lr_step + 1
    # and train.py really has that order: the scheduler steps come after the epoch's validation block
    src = open(os.path.join(ROOT, "train.py")).read()
    body = src[src.index("for epoch in range(1, args.num_epochs + 1):"):]
    assert body.index("scheduler_G.step()") > body.index("Finish valid")


def pytest_approx(v):
    import pytest
    return pytest.approx(v, rel=1e-12)


def test_gpu_loader_epoch_composition_matches_the_reference():
    """train.py GpuLoader: every image num_repeats times per epoch (reference data.py:60-77 + shuffle=True, drop_last=True), dealt
    to the ranks without overlap; a new permutation each epoch."""
    Tm = _load("train")

    class _NoSampler:          # only the index plan is under test here (the device part needs a GPU)
        pass
    n, rep, batch, world = 7, 4, 3, 2
    loaders = [Tm.GpuLoader(_NoSampler(), batch, 24, n, rep, r, world) for r in range(world)]
    per_rank = [l.epoch_indices() for l in loaders]
    iters = (n * rep) // (batch * world)
    assert len(loaders[0]) == iters and all(len(p) == iters * batch for p in per_rank)
    used = sorted(per_rank[0] + per_rank[1])
    full = sorted(list(range(n)) * rep)
    assert len(used) == iters * batch * world and all(used.count(i) <= rep for i in range(n))
    assert sum((full.count(i) - used.count(i)) for i in range(n)) == n * rep - len(used)      # only the drop_last tail is missing
    e0 = list(per_rank[0])
    for l in loaders:
        l.set_epoch(1)
    assert loaders[0].epoch_indices() != e0                                                   # reshuffled per epoch
    assert loaders[0].epoch_indices() == Tm.GpuLoader(_NoSampler(), batch, 24, n, rep, 0, world).__class__.epoch_indices(loaders[0])


def test_train_refuses_a_gan_run_without_pretrained_vgg():
    """`python train.py` must not silently train the perceptual loss against random VGG19 features."""
    import pytest
    Tm = _load("train")
    args = Tm.build_parser().parse_args([])
    assert args.vgg_weights == "" and args.allow_random_vgg is False and args.synthetic == 0
    with pytest.raises(SystemExit, match="pretrained vgg19"):
        Tm.build_vgg(args, torch.device("cpu"), 0, 1)
    with pytest.raises(SystemExit, match="multiple of"):
        Tm.check_limits(Tm.build_parser().parse_args(["--batch_size", "10"]), 4)
    with pytest.raises(SystemExit, match="patch_size"):
        Tm.check_limits(Tm.build_parser().parse_args(["--patch_size", "22"]), 1)


def test_spectral_norm_discriminator_schema_matches_torch():
    """Discriminator(spectral_norm=True): state_dict keys / shapes / order and the seeded construction (RNG draws of the conv
    init, then u, then v) equal those of the module the reference's line evidently means - its BasicBlock with
    torch.nn.utils.spectral_norm around an nn.Conv2d (reference model/basic.py:19-31 with the missing import supplied)."""
    import torch.nn as nn
    from model import Discriminator

    def ref_block(cin, cout, stride):
        conv = torch.nn.utils.spectral_norm(nn.Conv2d(cin, cout, 3, padding=1, stride=stride, bias=False))
        return nn.Sequential(conv, nn.BatchNorm2d(cout), nn.LeakyReLU(0.2, True))
    torch.manual_seed(11)
    ours = Discriminator({"patch_size": 8, "spectral_norm": True})
    torch.manual_seed(11)
    plan = [(3, 64, 1), (64, 64, 2), (64, 128, 1), (128, 128, 2), (128, 256, 1), (256, 256, 2), (256, 512, 1), (512, 512, 2)]
    feats = nn.Sequential(*[ref_block(*p) for p in plan])
    cls = nn.Sequential(nn.Linear(512 * 2 * 2, 1024), nn.LeakyReLU(0.2, True), nn.Linear(1024, 1))
    ref = {**{"features." + k: v for k, v in feats.state_dict().items()}, **{"classifier." + k: v for k, v in cls.state_dict().items()}}
    mine = ours.state_dict()
    assert list(mine.keys()) == list(ref.keys())
    for k in ref:
        assert mine[k].shape == ref[k].shape and torch.equal(mine[k], ref[k]), k


def test_bench_refuses_stale_pmc_summary(tmp_path, monkeypatch):
    """bench.py quotes roofline.traffic from a committed rocprofv3 --pmc summary only if the summary's side-car says it was taken on
    the CURRENT source of that kernel (a stale CSV must be refused, not quoted)."""
    import hashlib
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
    prof = tmp_path / "profiles"; prof.mkdir()
    src = tmp_path / "pesr_amd" / "csrc"; os.makedirs(src)
    (src / "conv3x3_wino4.hip").write_text("// kernel v1\n")
    monkeypatch.setattr(B, "ROOT", str(tmp_path))
    monkeypatch.setattr(B, "PMC_SUMMARIES", ("rXX_k1_pmc_summary.csv",))
    (prof / "rXX_k1_pmc_summary.csv").write_text("kernel,counter,launches,mean_per_launch\n"
                                                 "\"void conv3x3_wino4_kernel<false, 12>(Wino4Args)\",FETCH_SIZE,12,1000.0\n"
                                                 "\"void conv3x3_wino4_kernel<false, 12>(Wino4Args)\",WRITE_SIZE,12,500.0\n")
    # no side-car: refused
    by, name, why = B.k1_hbm_traffic_bytes("conv3x3_wino4_kernel")
    assert by is None and name is None and "stale" in why
    h = hashlib.sha256((src / "conv3x3_wino4.hip").read_bytes()).hexdigest()
    (prof / "rXX_k1_pmc_summary.meta.json").write_text(json.dumps({"sources_sha256": {"conv3x3_wino4.hip": h}}))
    by, name, stamp = B.k1_hbm_traffic_bytes("conv3x3_wino4_kernel")
    assert by == int((2 * 1000.0 + 500.0) * 1024) and name == "rXX_k1_pmc_summary.csv" and stamp == h[:16]
    (src / "conv3x3_wino4.hip").write_text("// kernel v2\n")               # the kernel changed after the counters were taken
    by, name, why = B.k1_hbm_traffic_bytes("conv3x3_wino4_kernel")
    assert by is None and "stale" in why


def test_comm_torch_group_single_rank_gloo():
    """pesr_amd/comm.py on the CPU: the torch.distributed transport (what the gloo tests and shared-GPU rehearsals use) sums in
    place, agrees host scalars by MAX, and refuses a hipGraph capture; make_transport picks it for a gloo group."""
    import socket
    import torch.distributed as dist
    from pesr_amd import comm
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        tr = comm.get_transport(torch.device("cpu"))
        assert isinstance(tr, comm.TorchGroup) and tr.name == "torch.distributed[gloo]" and not tr.capturable and tr.world == 1
        assert comm.get_transport(torch.device("cpu")) is tr                      # one transport per group: shared by the optimizers
        t = torch.arange(6, dtype=torch.float32)
        tr.wait([tr.all_reduce_async(t)])
        assert torch.equal(t, torch.arange(6, dtype=torch.float32))
        assert tr.host_max([1.5, -2.0]) == [1.5, -2.0]
        with pytest.raises(comm.CommError, match="cannot be captured"):
            tr.begin_capture()
        with pytest.raises(ValueError):
            comm.make_transport(torch.device("cpu"), prefer="smoke-signals")
        with pytest.raises(comm.CommError, match="one GPU per rank"):      # the peer-memory transport has no host path (and no fallback)
            comm.make_transport(torch.device("cpu"), prefer="peer")
        ok, why = comm.probe_direct(torch.device("cpu"), kind="peer")       # its child-process rehearsal reports, it does not raise
        assert not ok and "no GPU visible" in why
    finally:
        comm.close_transports()
        dist.destroy_process_group()
