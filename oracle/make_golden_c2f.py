"""ORACLE tooling — dev-only, runs ONLY in the build container (needs /root/reference).

Widening row: GroupCorrBlock1D and the cascade of Coarse2FineGroupRepViTRAFTStereo (nndepth/models/raft_stereo/cost_volume.py:64-128,
model.py:166-320).  Writes tests/golden/c2f.npz:
  (1) GroupCorrBlock1D fixtures: the reference class on small maps (pyramid levels, lookups at integer / fractional / out-of-range
      coordinates) — oracle/torch_ref.py's restatement is checked bit for bit against them;
  (2) the reference's own `Coarse2FineGroupRepViTRAFTStereo.forward` with its encoder side replaced by tests/c2f_double.py's tiny
      pyramid (the RepViT backbone is not on the hot path) and the deterministic weights of nndepth_amd.weightgen: the per-stage
      feature maps / cnets it fed the cascade with and every up_disp it returned;
  (3) a cross-check, not stored: the reference model with its REAL encoder side (RepViT, MobileOne, FeatureFusion) on one pair —
      oracle.torch_ref.coarse2fine_refine on the captured per-stage tensors must reproduce its outputs.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_c2f.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")


def _np(t):
    return t.detach().cpu().numpy()


def capture_stages(model, f1, f2):
    """Run the reference forward, recording what each cascade stage received: (fmap1, fmap2) through corr_fn, cnet through cnet_proj."""
    from nndepth.models.raft_stereo.cost_volume import GroupCorrBlock1D
    rec = {"fmaps": [], "cnets": []}

    def corr_spy(fmap1, fmap2, *a):
        rec["fmaps"].append((fmap1.clone(), fmap2.clone()))
        return GroupCorrBlock1D(fmap1, fmap2, *a)

    model.corr_fn = corr_spy
    hooks = [m.register_forward_hook(lambda mod, inp, out: rec["cnets"].append(out.clone())) for m in model.cnet_proj]
    with torch.no_grad():
        out = model(f1, f2)
    for h in hooks:
        h.remove()
    feats = [torch.cat([a, b], 0) for a, b in rec["fmaps"]]
    return feats, rec["cnets"], [o["up_disp"] for o in out]


def main():
    from nndepth_amd import weightgen
    from oracle import torch_ref as R
    from oracle.make_golden import _install_standins
    from c2f_double import make_c2f

    _install_standins()
    from nndepth.models.raft_stereo.cost_volume import GroupCorrBlock1D
    from nndepth.models.raft_stereo.model import Coarse2FineGroupRepViTRAFTStereo

    out = {}
    rep = []
    # ---- (1) GroupCorrBlock1D
    torch.manual_seed(21)
    for name, (B, C, H, W, L, r, G) in {"g4_c16_w20_l1": (2, 16, 3, 20, 1, 4, 4), "g4_c64_w33_l2": (1, 64, 2, 33, 2, 4, 4),
                                         "g2_c8_w12_l1_r2": (1, 8, 4, 12, 1, 2, 2)}.items():
        f1, f2 = torch.randn(B, C, H, W), torch.randn(B, C, H, W)
        coords = torch.rand(B, 1, H, W) * (W + 6) - 3
        coords[:, :, 0, :4] = torch.tensor([0.0, float(W - 1), 2.0, -1.5])  # exact integers, the last valid index, left of the row
        blk = GroupCorrBlock1D(f1, f2, L, r, G)
        samp = blk(coords)
        pyr = R.raft_group_corr_build(f1, f2, G, L)
        assert len(pyr) == len(blk.corr_pyramid) and all(torch.equal(a, b) for a, b in zip(pyr, blk.corr_pyramid)), name
        mine = R.raft_group_corr_lookup(pyr, coords, G, L, r)
        assert torch.equal(mine, samp), name
        out[name + "_cfg"] = np.array([B, C, H, W, L, r, G])
        out[name + "_f1"], out[name + "_f2"], out[name + "_coords"], out[name + "_out"] = _np(f1), _np(f2), _np(coords), _np(samp)
        for i, lv in enumerate(blk.corr_pyramid):
            out[f"{name}_pyr{i}"] = _np(lv)
        rep.append(f"GroupCorrBlock1D {name}: oracle == reference bit for bit (pyramid {len(pyr)} levels, lookup {tuple(samp.shape)})")

    # ---- (2) the cascade on the test double
    for name, (B, Hf, Wf, iters, seed) in {"c2f_b1_64x128_it3": (1, 64, 128, 3, 8), "c2f_b2_128x192_it2": (2, 128, 192, 2, 9)}.items():
        model = make_c2f(Coarse2FineGroupRepViTRAFTStereo, iters=iters, corr_levels=1).eval()
        weightgen.fill_module_(model, "c2f.")
        f1, f2 = weightgen.synthetic_frames(seed, B, Hf, Wf)
        feats, cnets, ups = capture_stages(model, f1, f2)
        sd = {k: v for k, v in model.state_dict().items() if k.startswith("update_block.")}
        mine = R.coarse2fine_refine(sd, feats, cnets, (Hf, Wf), iters, model.num_groups, model.corr_levels, model.corr_radius)
        err = max((a - b).abs().max().item() for a, b in zip(mine, ups))
        assert len(mine) == len(ups) == 3 * iters and err == 0.0, (name, err)
        out[name + "_cfg"] = np.array([B, Hf, Wf, iters, seed])
        for i, (ft, cn) in enumerate(zip(feats, cnets)):
            out[f"{name}_feat{i}"], out[f"{name}_cnet{i}"] = _np(ft), _np(cn)
        out[name + "_ups"] = np.stack([_np(u) for u in ups])
        rep.append(f"cascade {name}: stages {[tuple(f.shape) for f in feats]}, {len(ups)} outputs, |up| max {max(u.abs().max().item() for u in ups):.3f}; "
                   f"oracle.coarse2fine_refine == reference forward bit for bit")

    # ---- (3) the real encoder side, cross-check only
    torch.manual_seed(5)
    real = Coarse2FineGroupRepViTRAFTStereo(iters=2, corr_levels=1).eval()
    weightgen.fill_module_(real, "c2freal.")
    f1, f2 = weightgen.synthetic_frames(10, 1, 128, 256)
    feats, cnets, ups = capture_stages(real, f1, f2)
    sd = {k: v for k, v in real.state_dict().items() if k.startswith("update_block.")}
    mine = R.coarse2fine_refine(sd, feats, cnets, (128, 256), 2, real.num_groups, real.corr_levels, real.corr_radius)
    err = max((a - b).abs().max().item() for a, b in zip(mine, ups))
    assert err == 0.0, err
    rep.append(f"real Coarse2FineGroupRepViTRAFTStereo (RepViT, {sum(p.numel() for p in real.parameters()) / 1e6:.2f} M parameters) 128x256, 2 iterations: "
               f"oracle.coarse2fine_refine on its captured stage tensors == its forward bit for bit ({len(ups)} outputs)")

    np.savez_compressed(os.path.join(GOLD, "c2f.npz"), **out)
    with open(os.path.join(GOLD, "REPORT_c2f.txt"), "w") as f:
        f.write("golden vectors generated by oracle/make_golden_c2f.py from the imported reference (GroupCorrBlock1D, Coarse2FineGroupRepViTRAFTStereo)\n")
        f.write("\n".join(rep) + "\n")
    print("\n".join(rep))
    print("wrote", os.path.join(GOLD, "c2f.npz"), os.path.getsize(os.path.join(GOLD, "c2f.npz")), "bytes")


if __name__ == "__main__":
    main()
