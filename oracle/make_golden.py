"""ORACLE tooling — dev-only, runs ONLY in the build container (needs /root/reference).

Imports the reference nndepth package (read-only, from /root/reference), loads the
deterministic weights of `nndepth_amd.weightgen` into it, runs the hot-path pieces and the
full BaseRAFTStereo forward, cross-checks `oracle/torch_ref.py` against them and writes the
golden vectors under tests/golden/.  Nothing of the reference is copied: only tensors
(inputs / expected outputs) are stored.

The reference imports loguru / wandb / cv2 / h5py / timm at module import time; none of them
is installed here and none is used on the inference path, so empty stand-in modules are
registered first (recipe recorded in SURVEY.md §8c/§9).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
"""
import os
import shutil
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")


def _install_standins():
    class _Quiet:
        def __getattr__(self, name):
            return lambda *a, **k: None

    lg = types.ModuleType("loguru")
    lg.logger = _Quiet()
    sys.modules["loguru"] = lg
    for name in ("wandb", "cv2", "h5py"):
        sys.modules[name] = types.ModuleType(name)
    timm = types.ModuleType("timm")
    tm = types.ModuleType("timm.models")
    tm.__path__ = []
    tmm = types.ModuleType("timm.models.mobilenetv3")
    tmm.tf_mobilenetv3_large_100 = None
    tml = types.ModuleType("timm.models.layers")
    tml.trunc_normal_ = torch.nn.init.trunc_normal_
    tml.DropPath = torch.nn.Identity
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.mobilenetv3": tmm,
                        "timm.models.layers": tml})
    sys.path.insert(0, REF)


def _np(t):
    return t.detach().cpu().numpy()


def main():
    from nndepth_amd import weightgen
    from oracle import torch_ref as R

    _install_standins()
    torch.manual_seed(0)
    from nndepth.models.raft_stereo.model import BaseRAFTStereo
    from nndepth.models.raft_stereo.cost_volume import CorrBlock1D
    from nndepth.blocks.update_block import BasicUpdateBlock

    os.makedirs(GOLD, exist_ok=True)
    report = {}

    # ---------------------------------------------------------------- state-dict spec
    model = BaseRAFTStereo(iters=32, context_dim=64).eval()
    ref_sd = model.state_dict()
    spec = R.raft_stereo_spec()
    assert [k for k, _ in spec].sort() == list(ref_sd.keys()).sort()
    assert {k: tuple(s) for k, s in spec} == {k: tuple(v.shape) for k, v in ref_sd.items()}, "spec mismatch"
    sd = weightgen.fill_state_dict(spec)
    model.load_state_dict(sd, strict=True)

    # ------------------------------------------------- a1-a4: corr build + lookup (small)
    cases = {}
    for name, (B, C, H, W) in {"c16_w21": (2, 16, 5, 21), "c256_w40": (1, 256, 6, 40), "c32_w39": (2, 32, 4, 39)}.items():
        n = B * C * H * W
        f1 = torch.from_numpy(weightgen.uniform01("f1" + name, n).reshape(B, C, H, W) * 2 - 1)
        f2 = torch.from_numpy(weightgen.uniform01("f2" + name, n).reshape(B, C, H, W) * 2 - 1)
        blk = CorrBlock1D(f1, f2, 4, 4)
        # coords: integers, half-integers, negatives, > W-1, generic
        u = torch.from_numpy(weightgen.uniform01("co" + name, B * H * W).reshape(B, 1, H, W))
        coords = (u * (W + 10) - 5).clone()
        coords[:, :, 0, :] = torch.arange(W).float()            # exact integers
        coords[:, :, 1, :] = torch.arange(W).float() + 0.5      # half integers
        coords[:, :, 2, 0] = -3.25
        coords[:, :, 2, 1] = W + 7.5
        coords[:, :, 2, 2] = float(W - 1)
        out = blk(coords)
        mine_pyr = R.corr1d_build(f1, f2, 4)
        mine = R.corr1d_lookup(mine_pyr, coords, 4, 4)
        d_b = max((a - b).abs().max().item() for a, b in zip(mine_pyr, blk.corr_pyramid))
        d_l = (mine - out).abs().max().item()
        report[f"corr/{name}"] = (d_b, d_l)
        cases[name + "_f1"] = _np(f1)
        cases[name + "_f2"] = _np(f2)
        cases[name + "_coords"] = _np(coords)
        cases[name + "_out"] = _np(out)
        for i, p in enumerate(blk.corr_pyramid):
            cases[f"{name}_pyr{i}"] = _np(p.reshape(B * H * W, -1))
    np.savez_compressed(os.path.join(GOLD, "corr1d.npz"), **cases)

    # ------------------------------------------------------ a5-a9: update block (small)
    cases = {}
    for name, (hid, ctx, cor_planes, fc, sps, B, H, W) in {
        "raft_h128_c64": (128, 64, 36, 1, 8, 1, 12, 20),
        "raft_h128_c128": (128, 128, 36, 1, 8, 2, 9, 14),
        "cre_h128_c128_f2": (128, 128, 36, 2, 8, 1, 8, 12),
        "igev_h64_c64_cp576": (64, 64, 576, 1, 4, 1, 8, 12),
    }.items():
        ub = BasicUpdateBlock(hidden_dim=hid, cor_planes=cor_planes, flow_channel=fc, context_dim=ctx,
                              spatial_scale=sps).eval()
        uspec = R.update_block_spec("ub." + name, hid, cor_planes, ctx, fc, sps)
        assert {k[len("ub." + name) + 1:]: s for k, s in uspec} == {k: tuple(v.shape) for k, v in ub.state_dict().items()}
        usd = weightgen.fill_state_dict(uspec)
        ub.load_state_dict({k[len("ub." + name) + 1:]: v for k, v in usd.items()})

        def rnd(tag, *shape, lo=-1.0, hi=1.0):
            n = int(np.prod(shape))
            return torch.from_numpy(weightgen.uniform01(tag + name, n).reshape(shape) * (hi - lo) + lo)

        net = torch.tanh(rnd("net", B, hid, H, W, lo=-2, hi=2))
        inp = torch.relu(rnd("inp", B, ctx, H, W))
        corr = rnd("corr", B, cor_planes, H, W, lo=-2, hi=2)
        flow = rnd("flow", B, fc, H, W, lo=-8, hi=8)
        with torch.no_grad():
            n2, m2, d2 = ub(net, inp, corr, flow)
            mf = ub.encoder(flow, corr)
            n3, m3, d3 = R.update_block(usd, "ub." + name, net, inp, corr, flow)
        report[f"update/{name}"] = tuple((a - b).abs().max().item() for a, b in ((n2, n3), (m2, m3), (d2, d3)))
        for k, v in (("net", net), ("inp", inp), ("corr", corr), ("flow", flow), ("motion", mf),
                     ("net_out", n2), ("mask_out", m2), ("delta_out", d2)):
            cases[f"{name}_{k}"] = _np(v)
    np.savez_compressed(os.path.join(GOLD, "update_block.npz"), **cases)

    # ------------------------------------- a12-a14: IGEV geometry-encoding volume (build, pyramids, lookup)
    from nndepth.models.igev_stereo.cost_volume import GeometryAwareCostVolume, CostVolumeFilterNetwork
    cases = {}
    for name, (B, C, H, W) in {"g8_c128": (1, 128, 8, 24), "g8_c64_b2": (2, 64, 8, 32)}.items():
        n = B * C * H * W
        f1 = torch.from_numpy(weightgen.uniform01("if1" + name, n).reshape(B, C, H, W) * 2 - 1)
        f2 = torch.from_numpy(weightgen.uniform01("if2" + name, n).reshape(B, C, H, W) * 2 - 1)
        guides = [torch.from_numpy(weightgen.uniform01(f"ig{j}" + name, B * c * (H >> (j + 1)) * (W >> (j + 1))
                                                       ).reshape(B, c, H >> (j + 1), W >> (j + 1)))
                  for j, c in enumerate((40, 80, 160))]
        reg = CostVolumeFilterNetwork(8, [40, 80, 160]).eval()
        weightgen.fill_module_(reg, "igev.cv_regularizer.")
        u = torch.from_numpy(weightgen.uniform01("ico" + name, B * H * W).reshape(B, 1, H, W))
        coords = u * (W + 8) - 4
        coords[:, :, 0, :] = torch.arange(W).float()
        with torch.no_grad():
            cv = GeometryAwareCostVolume(f1, f2, guides, reg, 4, 4, 8)
            out = cv(coords)
            fvol = R.group_corr_volume(f1, f2, 8)
            gvol = reg(fvol.clone().permute(0, 1, 4, 2, 3), guides)
            fp, gp = R.igev_pyramids(fvol, gvol, 4)
            mine = R.igev_lookup(fp, gp, coords, 8, 4, 4)
        report[f"igev/{name}"] = (max((a - b).abs().max().item() for a, b in zip(fp, cv.feat_corr_cv)),
                                  max((a - b).abs().max().item() for a, b in zip(gp, cv.geo_aware_cv)),
                                  (mine - out).abs().max().item())
        cases[name + "_f1"], cases[name + "_f2"], cases[name + "_coords"] = _np(f1), _np(f2), _np(coords)
        cases[name + "_out"] = _np(out)
        cases[name + "_geo0"] = _np(cv.geo_aware_cv[0][:, 0])
        for i in range(5):
            cases[f"{name}_feat{i}"] = _np(cv.feat_corr_cv[i][:, 0])
            if i:
                cases[f"{name}_geo{i}"] = _np(cv.geo_aware_cv[i][:, 0])
    np.savez_compressed(os.path.join(GOLD, "igev_volume.npz"), **cases)

    # ---------------------------------------------------------- a10: convex upsample
    cases = {}
    for name, (B, C, H, W, rate) in {"r8_c1": (2, 1, 6, 10, 8), "r4_c1": (1, 1, 7, 9, 4), "r8_c2": (1, 2, 5, 8, 8)}.items():
        flow = torch.from_numpy(weightgen.uniform01("uf" + name, B * C * H * W).reshape(B, C, H, W) * 20 - 10)
        mask = torch.from_numpy(weightgen.uniform01("um" + name, B * 9 * rate * rate * H * W).reshape(B, 9 * rate * rate, H, W) * 6 - 3)
        # reference method is written for C=1 (raft) / C=2 (cre); use the raft one per channel
        outs = [model.convex_upsample(flow[:, c:c + 1], mask, rate) for c in range(C)]
        out = torch.cat(outs, 1)
        mine = R.convex_upsample(flow, mask, rate)
        report[f"upsample/{name}"] = ((mine - out).abs().max().item(),)
        cases[name + "_flow"], cases[name + "_mask"], cases[name + "_out"] = _np(flow), _np(mask), _np(out)
    np.savez_compressed(os.path.join(GOLD, "upsample.npz"), **cases)

    # -------------------------------------------- a11: small full forward (synthetic frames)
    f1, f2 = weightgen.synthetic_frames(0, 1, 96, 160)
    small = BaseRAFTStereo(iters=6, context_dim=64).eval()
    small.load_state_dict(sd)
    with torch.no_grad():
        out = small(f1, f2)
        mine = R.raft_stereo_forward(sd, f1, f2, 6)
    report["forward/small96x160_it6"] = ((mine[-1] - out[-1]["up_disp"]).abs().max().item(),)
    np.savez_compressed(os.path.join(GOLD, "forward_small.npz"),
                        up_disp=np.stack([_np(o["up_disp"]) for o in out]))

    # ---------------------------------- a11: TartanAir sample @544x960, iters 1/4/12/32
    from PIL import Image
    sample = os.path.join(REF, "samples/tartanair/abandonedfactory/abandonedfactory/Easy/P000")
    for side in ("left", "right"):
        shutil.copyfile(os.path.join(sample, f"image_{side}/000000_{side}.png"),
                        os.path.join(GOLD, f"tartanair_000000_{side}.png"))
        os.chmod(os.path.join(GOLD, f"tartanair_000000_{side}.png"), 0o644)
    frames = []
    for side in ("left", "right"):
        img = np.asarray(Image.open(os.path.join(GOLD, f"tartanair_000000_{side}.png")).convert("RGB"))
        t = torch.from_numpy(img.copy()).permute(2, 0, 1).float().unsqueeze(0)
        # preprocessing of raft_stereo/scripts/inference.py:55-60
        t = torch.nn.functional.interpolate(t, (544, 960), mode="bilinear")
        frames.append((t - 127.5) / 127.5)
    lows = []
    orig_up = model.convex_upsample

    def spy(flow, mask, rate=8):
        lows.append(flow.clone())
        return orig_up(flow, mask, rate)

    model.convex_upsample = spy
    with torch.no_grad():
        t0 = time.time()
        out = model(frames[0], frames[1])
        t_ref = time.time() - t0
        t0 = time.time()
        mine, mine_low = R.raft_stereo_forward(sd, frames[0], frames[1], 32, return_lowres=True)
        t_mine = time.time() - t0
        torch.set_num_threads(1)
        out1 = model(frames[0], frames[1])
        torch.set_num_threads(8)
    keep = (1, 4, 12, 32)
    report["forward/tartanair_it32 (torch_ref vs ref)"] = tuple(
        (mine[i - 1] - out[i - 1]["up_disp"]).abs().max().item() for i in keep)
    report["forward/tartanair ref self-noise 1thr vs 8thr"] = tuple(
        (out1[i - 1]["up_disp"] - out[i - 1]["up_disp"]).abs().max().item() for i in keep)
    report["forward/tartanair |disp| max"] = (out[-1]["up_disp"].abs().max().item(),)
    report["forward/tartanair seconds (ref, torch_ref)"] = (t_ref, t_mine)
    # GT disparity for EPE parity (tartanair_dataset.py:28-30,239: disp = -0.25*320/depth... sign "negative")
    depth = np.load(os.path.join(sample, "depth_left/000000_left_depth.npy"))
    gt = torch.from_numpy((80.0 / depth).astype(np.float32))[None, None]
    gt = torch.nn.functional.interpolate(gt, (544, 960), mode="bilinear") * (960 / 640)
    gt = -gt
    report["forward/tartanair EPE ref"] = (R.epe(gt, out[-1]["up_disp"]),)
    np.savez_compressed(
        os.path.join(GOLD, "forward_tartanair.npz"),
        up_disp_it32=_np(out[-1]["up_disp"]).astype(np.float32),
        low_disp=np.stack([_np(lows[i - 1]) for i in keep]),
        low_iters=np.array(keep),
        up_disp_it1_row=_np(out[0]["up_disp"])[0, 0, ::8, ::8],
        gt_disp=_np(gt).astype(np.float16),
        epe_ref=np.array(R.epe(gt, out[-1]["up_disp"])),
    )

    print("\n== golden report (max-abs torch_ref vs imported reference) ==")
    for k, v in report.items():
        print(f"{k:55s} " + "  ".join(f"{x:.3e}" for x in v))
    with open(os.path.join(GOLD, "REPORT.txt"), "w") as f:
        f.write("golden vectors generated by oracle/make_golden.py from the imported reference\n")
        f.write(f"torch {torch.__version__}, numpy {np.__version__}, threads 8\n")
        for k, v in report.items():
            f.write(f"{k:55s} " + "  ".join(f"{x:.3e}" for x in v) + "\n")


if __name__ == "__main__":
    main()
