"""ORACLE tooling — dev-only, runs ONLY in the build container (needs /root/reference).

Real-image goldens that widen the parity gate of the split arithmetics (VERDICT r2 "Next round" item 2): the imported
reference is run, with the deterministic weights of nndepth_amd.weightgen, on the reference's own sample images

  kitti     samples/kitti-stereo-2015/training/image_2|image_3/000000_10.png (375x1242) -> (x - 127.5) / 127.5
            (dataloaders/disparity/kitti2015_disparity.py:31-37 without the resize) -> Padder(divis_by=32)
            (dataloaders/utils.py:5-21) -> BaseRAFTStereo(iters=32, context_dim=64)          [BASELINE.json configs[3] per pair]
  cre       the TartanAir sample pair at 544x960 (inference.py:55-60 preprocessing) -> CREStereoBase(iters=4)
  igev      the same pair -> IGEVStereoBase on the tiny backbone of tests/igev_double.py, iters=32  [configs[2] per sample]

and the outputs are stored under tests/golden/ (final full-resolution map, low-resolution state after 1 / 4 / 12 / 32
iterations where the model has one, EPE against the sample's ground truth).  The oracle restatements (oracle/torch_ref.py,
oracle/cre_ref.py) are checked against the same runs.  Only tensors and the reference's sample images (data) are stored.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_realdata.py [kitti] [cre] [igev]
"""
import os
import shutil
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
KEEP = (1, 4, 12, 32)


def _np(t):
    return t.detach().cpu().numpy()


def _png_frame(path):
    from PIL import Image
    img = np.asarray(Image.open(path).convert("RGB"))
    return torch.from_numpy(img.copy()).permute(2, 0, 1).float().unsqueeze(0)


def tartanair_frames():
    frames = []
    for side in ("left", "right"):
        t = _png_frame(os.path.join(GOLD, f"tartanair_000000_{side}.png"))
        t = torch.nn.functional.interpolate(t, (544, 960), mode="bilinear")  # raft_stereo/scripts/inference.py:55-60
        frames.append((t - 127.5) / 127.5)
    return frames


def kitti(report):
    from nndepth_amd import weightgen
    from oracle import torch_ref as R
    from nndepth.models.raft_stereo.model import BaseRAFTStereo
    from nndepth.data.dataloaders.utils import Padder
    from PIL import Image

    src = os.path.join(REF, "samples/kitti-stereo-2015/training")
    for cam, side in (("image_2", "left"), ("image_3", "right")):
        dst = os.path.join(GOLD, f"kitti_000000_10_{side}.png")
        shutil.copyfile(os.path.join(src, cam, "000000_10.png"), dst)
        os.chmod(dst, 0o644)
    frames = [(_png_frame(os.path.join(GOLD, f"kitti_000000_10_{s}.png")) - 127.5) / 127.5 for s in ("left", "right")]
    assert tuple(frames[0].shape) == (1, 3, 375, 1242)
    padder = Padder((375, 1242), divis_by=32)
    p1, p2 = padder.pad(*frames)
    assert tuple(p1.shape) == (1, 3, 384, 1248) and padder._pad == [3, 3, 0, 9]
    sd = weightgen.fill_state_dict(R.raft_stereo_spec())
    model = BaseRAFTStereo(iters=32, context_dim=64).eval()
    model.load_state_dict(sd, strict=True)
    lows, orig_up = [], model.convex_upsample

    def spy(flow, mask, rate=8):
        lows.append(flow.clone())
        return orig_up(flow, mask, rate)

    model.convex_upsample = spy
    with torch.no_grad():
        t0 = time.time()
        out = model(p1, p2)
        t_ref = time.time() - t0
        mine, mine_low = R.raft_stereo_forward(sd, p1, p2, 32, return_lowres=True)
        torch.set_num_threads(1)
        out1 = model(p1, p2)
        torch.set_num_threads(8)
    report["kitti/raft_it32 torch_ref vs ref (it 1,4,12,32)"] = tuple((mine[i - 1] - out[i - 1]["up_disp"]).abs().max().item() for i in KEEP)
    report["kitti/ref self-noise 1thr vs 8thr"] = tuple((out1[i - 1]["up_disp"] - out[i - 1]["up_disp"]).abs().max().item() for i in KEEP)
    report["kitti/|disp| max, seconds"] = (out[-1]["up_disp"].abs().max().item(), t_ref)
    # the spy collected the 1/8-resolution disparities of both runs: 32 of the 8-thread run, then 32 of the 1-thread run
    assert len(lows) == 64
    report["kitti/ref self-noise of the 1/8-resolution disparity (it 1,4,12,32)"] = tuple((lows[32 + i - 1] - lows[i - 1]).abs().max().item() for i in KEEP)
    # ground truth of the sample: uint16 / 256, 0 = no measurement (kitti_stereo_2015.py load_disp); sign "negative"
    gt = np.asarray(Image.open(os.path.join(src, "disp_occ_0/000000_10.png"))).astype(np.float32) / 256.0
    valid = gt > 0
    final = padder.unpad(out[-1]["up_disp"])[0, 0].numpy()
    epe_ref = float(np.abs(final - (-gt))[valid].mean())
    report["kitti/EPE of the reference output over the valid ground truth"] = (epe_ref,)
    np.savez_compressed(os.path.join(GOLD, "forward_kitti.npz"),
                        up_disp_it32=_np(out[-1]["up_disp"]).astype(np.float32),  # padded frame, 384x1248
                        low_disp=np.stack([_np(lows[i - 1]) for i in KEEP]), low_iters=np.array(KEEP),
                        gt_disp=(-gt).astype(np.float16), gt_valid=np.packbits(valid), epe_ref=np.array(epe_ref), pad=np.array(padder._pad),
                        ref_self_noise_up=np.array(report["kitti/ref self-noise 1thr vs 8thr"]),
                        ref_self_noise_low=np.array(report["kitti/ref self-noise of the 1/8-resolution disparity (it 1,4,12,32)"]),
                        # the full-resolution maps of the earlier iterations on every 4th pixel: the drift table of the GPU test
                        **{f"up_disp_sub4_it{i}": _np(out[i - 1]["up_disp"])[:, :, ::4, ::4].astype(np.float32).copy() for i in KEEP[:-1]})


def cre(report):
    from nndepth_amd import weightgen
    from oracle import cre_ref as C
    from nndepth.models.cre_stereo.model import CREStereoBase

    sd = weightgen.fill_state_dict(C.cre_stereo_spec())
    model = CREStereoBase(iters=4).eval()
    model.load_state_dict(sd)
    f1, f2 = tartanair_frames()
    with torch.no_grad():
        t0 = time.time()
        out = model(f1, f2)
        t_ref = time.time() - t0
        mine = C.cre_stereo_forward(sd, f1, f2, 4)
        torch.set_num_threads(1)
        out1 = model(f1, f2)
        torch.set_num_threads(8)
    assert len(out) == len(mine) == 8
    report["cre/544x960_it4 cre_ref vs ref (8 outputs)"] = tuple((o["up_disp"] - m).abs().max().item() for o, m in zip(out, mine))
    report["cre/ref self-noise 1thr vs 8thr (8 outputs)"] = tuple((a["up_disp"] - b["up_disp"]).abs().max().item() for a, b in zip(out1, out))
    report["cre/|flow| max, seconds"] = (out[-1]["up_disp"].abs().max().item(), t_ref)
    # the final map in full, the 7 earlier outputs of the cascade (1/4, 1/2 and full resolution) on every 4th pixel
    np.savez_compressed(os.path.join(GOLD, "forward_cre_tartanair.npz"), up_disp_final=_np(out[-1]["up_disp"]).astype(np.float32),
                        ref_self_noise_up=np.array(report["cre/ref self-noise 1thr vs 8thr (8 outputs)"]),
                        **{f"up_disp_sub4_{i}": _np(o["up_disp"])[:, :, ::4, ::4].copy() for i, o in enumerate(out[:-1])})


def igev(report):
    from nndepth_amd import weightgen
    from igev_double import make_igev
    from nndepth.models.igev_stereo.model import IGEVStereoBase
    from nndepth.models.igev_stereo.cost_volume import CostVolumeFilterNetwork

    model = make_igev(IGEVStereoBase, CostVolumeFilterNetwork, iters=32, hidden_dim=64, context_dim=64).eval()
    weightgen.fill_module_(model, "igev.")
    f1, f2 = tartanair_frames()
    lows, captured, orig_up, orig_reg = [], {}, model.convex_upsample, model.regress_disparity

    def spy(flow, mask, rate=4):
        lows.append(flow.clone())
        return orig_up(flow, mask, rate)

    def spy_reg(dist, width):
        captured["init"] = orig_reg(dist, width)
        return captured["init"]

    orig_sq = model.cv_squeezer.forward

    def spy_sq(x):
        captured["logits"] = orig_sq(x)
        return captured["logits"]

    model.convex_upsample, model.regress_disparity, model.cv_squeezer.forward = spy, spy_reg, spy_sq
    with torch.no_grad():
        t0 = time.time()
        out = model(f1, f2)
        t_ref = time.time() - t0
        lows8 = list(lows)
        del lows[:]
        torch.set_num_threads(1)
        out1 = model(f1, f2)
        torch.set_num_threads(8)
        lows1 = list(lows)
    W4 = lows8[0].shape[-1]
    rng = [(l.min().item(), l.max().item()) for l in (lows8[0], lows8[11], lows8[31])]
    report["igev/coords range at the 1/4 map (it 1, 12, 32), map width"] = tuple(x for r in rng for x in r) + (float(W4),)
    report["igev/init disparity range"] = (captured["init"].min().item(), captured["init"].max().item())
    report["igev/ref self-noise 1thr vs 8thr (it 1,4,12,32)"] = tuple((out1[i - 1]["up_disp"] - out[i - 1]["up_disp"]).abs().max().item() for i in KEEP)
    # how far the reference's own fp32 soft-argmin (softmax over 240 candidates, then -sum d * p_d) is from the float64 value of
    # the same expression on the same logits: the accuracy to which the initial disparity is DEFINED
    lg = captured["logits"].squeeze(1).double()
    init64 = -(torch.arange(lg.shape[1], dtype=torch.float64).view(1, -1, 1, 1) * torch.softmax(lg, 1)).sum(1, keepdim=True)
    init_err64 = (captured["init"].double() - init64).abs().max().item()
    report["igev/reference fp32 soft-argmin init vs float64 of the same logits (max-abs)"] = (init_err64,)
    noise_up = tuple((out1[i - 1]["up_disp"] - out[i - 1]["up_disp"]).abs().max().item() for i in KEEP)
    noise_low = tuple((lows1[i - 1] - lows8[i - 1]).abs().max().item() for i in KEEP)
    report["igev/ref self-noise of the 1/4-resolution coordinates (it 1,4,12,32)"] = noise_low
    report["igev/|up_disp| max, seconds"] = (out[-1]["up_disp"].abs().max().item(), t_ref)
    # up_disp = 4 x the absolute coordinate (quirk Q5): values up to ~530, one fp32 ulp there is 6.1e-5 — the reference's own
    # 1-thread vs 8-thread outputs differ by more than 1e-4 at full resolution; the self-noise is stored next to the outputs
    np.savez_compressed(os.path.join(GOLD, "forward_igev_tartanair.npz"), up_disp_it32=_np(out[-1]["up_disp"]).astype(np.float32),
                        low_coords=np.stack([_np(lows8[i - 1]) for i in KEEP]), low_iters=np.array(KEEP), init=_np(captured["init"]),
                        ref_self_noise_up=np.array(noise_up), ref_self_noise_low=np.array(noise_low), ref_init_err_vs_f64=np.array(init_err64))


def main():
    from oracle.make_golden import _install_standins
    _install_standins()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["kitti", "cre", "igev"]
    report = {}
    for name in which:
        {"kitti": kitti, "cre": cre, "igev": igev}[name](report)
        print(name, "done", flush=True)
    lines = [f"{k:70s} " + "  ".join(f"{x:.3e}" for x in v) for k, v in report.items()]
    print("\n".join(lines))
    path = os.path.join(GOLD, "REPORT_realdata.txt")
    old = {}
    if os.path.exists(path):  # keep the lines of the parts that were not regenerated
        for ln in open(path).read().splitlines()[2:]:
            old[ln[:70].rstrip()] = ln
    for k, ln in zip(report, lines):
        old[k] = ln
    with open(path, "w") as f:
        f.write("golden vectors generated by oracle/make_golden_realdata.py from the imported reference on its own sample images\n")
        f.write(f"torch {torch.__version__}, numpy {np.__version__}, threads 8\n")
        f.write("\n".join(old.values()) + "\n")


if __name__ == "__main__":
    main()
