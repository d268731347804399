"""CPU: the oracle (oracle/torch_ref.py) against the golden vectors produced by the imported reference
(oracle/make_golden.py).  This is what pins the oracle; the GPU tests then compare HIP vs oracle/golden."""
import numpy as np
import pytest
import torch

from conftest import t
from oracle import torch_ref as R


@pytest.mark.parametrize("name", ["c16_w21", "c256_w40", "c32_w39"])
def test_corr1d_build_and_lookup(gold, name):
    g = gold("corr1d.npz")
    f1, f2, coords = (t(g[f"{name}_{k}"]) for k in ("f1", "f2", "coords"))
    pyr = R.corr1d_build(f1, f2, 4)
    assert len(pyr) == 5  # one level more than is ever read
    for i, p in enumerate(pyr):
        assert np.array_equal(p[:, 0].numpy(), g[f"{name}_pyr{i}"]), f"level {i}"
    out = R.corr1d_lookup(pyr, coords, 4, 4)
    assert np.array_equal(out.numpy(), g[f"{name}_out"])


def test_linear_sampler_border_clamp():
    """Q2: coordinates outside [0, w-1] clamp to the border (not zero padding); integers return the sample."""
    row = torch.arange(10.0)[None] * 2.0
    x = torch.tensor([[-5.0, 0.0, 3.0, 3.5, 9.0, 20.0]])
    got = R.linear_sampler(row, x)
    assert torch.allclose(got, torch.tensor([[0.0, 0.0, 6.0, 7.0, 18.0, 18.0]]), atol=1e-5)


@pytest.mark.parametrize("name,hid,ctx,cp,fc,sps", [
    ("raft_h128_c64", 128, 64, 36, 1, 8), ("raft_h128_c128", 128, 128, 36, 1, 8),
    ("cre_h128_c128_f2", 128, 128, 36, 2, 8), ("igev_h64_c64_cp576", 64, 64, 576, 1, 4)])
def test_update_block(gold, name, hid, ctx, cp, fc, sps):
    from nndepth_amd import weightgen
    g = gold("update_block.npz")
    sd = weightgen.fill_state_dict(R.update_block_spec("ub." + name, hid, cp, ctx, fc, sps))
    net, inp, corr, flow = (t(g[f"{name}_{k}"]) for k in ("net", "inp", "corr", "flow"))
    with torch.no_grad():
        n, m, d = R.update_block(sd, "ub." + name, net, inp, corr, flow)
        mf = R.motion_encoder(sd, f"ub.{name}.encoder", flow, corr)
    # bit-exact in the container that made the goldens; allow last-bit drift across CPU ISAs (oneDNN kernels)
    for got, key in ((n, "net_out"), (m, "mask_out"), (d, "delta_out"), (mf, "motion")):
        assert np.abs(got.numpy() - g[f"{name}_{key}"]).max() <= 2e-5, key


@pytest.mark.parametrize("name,rate", [("r8_c1", 8), ("r4_c1", 4), ("r8_c2", 8)])
def test_convex_upsample(gold, name, rate):
    g = gold("upsample.npz")
    out = R.convex_upsample(t(g[name + "_flow"]), t(g[name + "_mask"]), rate)
    assert np.abs(out.numpy() - g[name + "_out"]).max() <= 1e-5


def test_forward_small(gold, raft_sd):
    from nndepth_amd import weightgen
    g = gold("forward_small.npz")
    f1, f2 = weightgen.synthetic_frames(0, 1, 96, 160)
    with torch.no_grad():
        ups = R.raft_stereo_forward(raft_sd, f1, f2, 6)
    for i in range(6):
        assert np.abs(ups[i].numpy() - g["up_disp"][i]).max() <= 1e-4, i


def test_spec_matches_module_and_weightgen_is_deterministic(raft_sd):
    """state_dict keys/shapes of the drop-in model == the spec verified against the reference;
    the generator is a pure function of (key, shape)."""
    from nndepth_amd import weightgen
    from nndepth_amd.raft_stereo import BaseRAFTStereo
    m = BaseRAFTStereo(iters=1, context_dim=64)
    sd = m.state_dict()
    spec = R.raft_stereo_spec()
    assert list(sd.keys()) == [k for k, _ in spec]
    assert all(tuple(sd[k].shape) == tuple(s) for k, s in spec)
    again = weightgen.fill_state_dict(spec)
    assert all(torch.equal(raft_sd[k], again[k]) for k in raft_sd)
    # the aliased shortcut norm carries identical values under both names (residual_block.py:51)
    assert torch.equal(raft_sd["fnet.layer1.0.norm3.weight"], raft_sd["fnet.layer1.0.downsample.1.weight"])
    m.load_state_dict(raft_sd, strict=True)


@pytest.mark.parametrize("name,B,H,W", [("g8_c128", 1, 8, 24), ("g8_c64_b2", 2, 8, 32)])
def test_igev_volume_oracle(gold, name, B, H, W):
    """a12-a14: group-wise volume (only the first 8 chunks of 8 channels, Q4), pyramids, combined lookup."""
    g = gold("igev_volume.npz")
    f1, f2, coords = (t(g[f"{name}_{k}"]) for k in ("f1", "f2", "coords"))
    fvol = R.group_corr_volume(f1, f2, 8)
    assert tuple(fvol.shape) == (B, 8, H, W, W)
    geo_vol = t(g[name + "_geo0"]).reshape(B, 8, H, W, W).permute(0, 1, 4, 2, 3)  # as the regulariser returns it
    fp, gp = R.igev_pyramids(fvol, geo_vol, 4)
    for i in range(5):
        assert np.array_equal(fp[i][:, 0].numpy(), g[f"{name}_feat{i}"]), i
        assert np.array_equal(gp[i][:, 0].numpy(), g[f"{name}_geo{i}"]), i
    out = R.igev_lookup(fp, gp, coords, 8, 4, 4)
    assert tuple(out.shape) == (B, 576, H, W)
    assert np.array_equal(out.numpy(), g[name + "_out"])


def test_igev_model_keys_and_init_disparity(gold):
    """a15/a16: the drop-in IGEVStereoBase + CostVolumeFilterNetwork register exactly the reference's state_dict keys
    (captured from the reference's IGEVStereoBase on the same tiny backbone), and the oracle's soft-argmin
    reproduces the reference's regress_disparity(softmax(.)) bit for bit."""
    from igev_double import make_igev
    from nndepth_amd.igev_stereo import IGEVStereoBase, CostVolumeFilterNetwork
    g = gold("igev_forward.npz")
    m = make_igev(IGEVStereoBase, CostVolumeFilterNetwork, iters=4, hidden_dim=64, context_dim=64)
    assert list(m.state_dict().keys()) == [str(k) for k in g["keys"]]
    assert np.array_equal(R.igev_init_disparity(t(g["logits"])).numpy(), g["init"])


def test_prepost_oracle(gold):
    """§8f-3: preprocess_frame, Padder, EvalCriterion restatements against the imported reference's outputs."""
    g = gold("prepost.npz")
    fr = t(g["pre_img_u8"].transpose(2, 0, 1).copy()).float()
    for name, HW in (("down", (68, 120)), ("odd", (77, 131)), ("up", (150, 333))):
        assert np.array_equal(R.preprocess_frame(fr, HW).numpy(), g[f"pre_{name}"])
    assert R.padder_pads((375, 1242), 32) == [3, 3, 0, 9] == list(g["pad_kitti32_pads"])
    for name, (H, W, div) in (("odd8", (37, 53, 8)), ("exact", (64, 96, 8))):
        pads = R.padder_pads((H, W), div)
        assert pads == list(g[f"pad_{name}_pads"])
        y = R.padder_pad(t(g[f"pad_{name}_x"]), pads)
        assert np.array_equal(y.numpy(), g[f"pad_{name}_y"]) and torch.equal(R.padder_unpad(y, pads), t(g[f"pad_{name}_x"]))
    for name in ("plain", "masked"):
        mask = t(g[f"ev_{name}_mask"]) if f"ev_{name}_mask" in g else None
        out = R.eval_criterion(t(g[f"ev_{name}_gt"]), t(g[f"ev_{name}_pred"]), mask, {"kitti-d1": 3.0, "d5": 5.0}, 1000)
        assert [out["epe"], out["kitti-d1"], out["d5"]] == pytest.approx(list(g[f"ev_{name}_out"]), abs=1e-7)


@pytest.mark.parametrize("name,hid,ctx,fc,sps", [("convgru_h128_c128", 128, 128, 1, 8), ("convgru_h64_c64_f2", 64, 64, 2, 4)])
def test_update_block_conv_gru(gold, name, hid, ctx, fc, sps):
    """a7: ConvGRU (single 3x3 pass, nndepth/blocks/gru.py:40-61) — the oracle's conv_gru and the whole update block with
    gru="conv_gru" against the imported reference's outputs (oracle/make_golden.py conv_gru)."""
    from nndepth_amd import weightgen
    g = gold("update_block_conv_gru.npz")
    sd = weightgen.fill_state_dict(R.update_block_spec("ub." + name, hid, 36, ctx, fc, sps, gru="conv_gru"))
    net, inp, corr, flow, x = (t(g[f"{name}_{k}"]) for k in ("net", "inp", "corr", "flow", "gru_x"))
    with torch.no_grad():
        h = R.conv_gru(sd, f"ub.{name}.gru", net, x)
        n, m, d = R.update_block(sd, "ub." + name, net, inp, corr, flow, gru="conv_gru")
    for got, key in ((h, "gru_out"), (n, "net_out"), (m, "mask_out"), (d, "delta_out")):
        assert np.abs(got.numpy() - g[f"{name}_{key}"]).max() <= 2e-5, key


@pytest.mark.parametrize("name,B,H,W", [("g8_c128", 1, 8, 24), ("g8_c64_b2", 2, 8, 32)])
def test_igev_regulariser_oracle(gold, name, B, H, W):
    """a15: the oracle's CostVolumeFilterNetwork restatement (Conv3d hourglass, BatchNorm3d eval, LeakyReLU, trilinear x2,
    feature gating) against the reference's regularised volume (igev_volume.npz: geo0), and its key list against the
    drop-in module's state_dict (which test_igev_model_keys_and_init_disparity ties to the reference's)."""
    from nndepth_amd import weightgen
    from nndepth_amd.igev_stereo import CostVolumeFilterNetwork
    g = gold("igev_volume.npz")
    spec = R.cost_volume_filter_spec("igev.cv_regularizer")
    assert [k[len("igev.cv_regularizer."):] for k, _ in spec] == list(CostVolumeFilterNetwork(8, [40, 80, 160]).state_dict().keys())
    sd = weightgen.fill_state_dict(spec)
    f1, f2 = t(g[name + "_f1"]), t(g[name + "_f2"])
    guides = [torch.from_numpy(weightgen.uniform01(f"ig{j}" + name, B * c * (H >> (j + 1)) * (W >> (j + 1))
                                                   ).reshape(B, c, H >> (j + 1), W >> (j + 1)))
              for j, c in enumerate((40, 80, 160))]
    with torch.no_grad():
        fvol = R.group_corr_volume(f1, f2, 8)
        geo = R.cost_volume_filter(sd, "igev.cv_regularizer", fvol.permute(0, 1, 4, 2, 3), guides)
    got = geo.permute(0, 1, 3, 4, 2).reshape(-1, W).numpy()
    assert np.abs(got - g[name + "_geo0"]).max() <= 2e-6


def test_raft_kitti_realdata_oracle(gold, raft_sd):
    """oracle/make_golden_realdata.py kitti: the oracle's RAFT-Stereo forward on the reference's KITTI sample pair
    (375x1242 -> Padder(32) -> 384x1248, 32 iterations) against the imported reference's output.  The same code gave 0.0 at
    generation time (tests/golden/REPORT_realdata.txt); across thread counts the reference itself moves by ~3e-5."""
    import os
    from PIL import Image
    g = gold("forward_kitti.npz")
    frames = []
    for side in ("left", "right"):
        img = np.asarray(Image.open(os.path.join(os.path.dirname(__file__), "golden", f"kitti_000000_10_{side}.png")).convert("RGB"))
        frames.append((torch.from_numpy(img.copy()).permute(2, 0, 1).float().unsqueeze(0) - 127.5) / 127.5)
    pads = R.padder_pads((375, 1242), 32)
    assert pads == list(g["pad"])
    p1, p2 = R.padder_pad(frames[0], pads), R.padder_pad(frames[1], pads)
    with torch.no_grad():
        ups, lows = R.raft_stereo_forward(raft_sd, p1, p2, 32, return_lowres=True)
    assert np.abs(ups[-1].numpy() - g["up_disp_it32"]).max() <= 1e-4
    for k, it in enumerate(g["low_iters"]):
        assert np.abs(lows[int(it) - 1].numpy() - g["low_disp"][k]).max() <= 1e-4, it


# ------------------------------------------------------------- GroupCorrBlock1D / Coarse2Fine cascade (widening)
@pytest.mark.parametrize("name", ["g4_c16_w20_l1", "g4_c64_w33_l2", "g2_c8_w12_l1_r2"])
def test_raft_group_corr_build_and_lookup(gold, name):
    """GroupCorrBlock1D (raft_stereo/cost_volume.py:64-128, quirks Q4 / Q6 kept) against the reference class' fixtures, bit for bit."""
    g = gold("c2f.npz")
    B, C, H, W, L, r, G = (int(v) for v in g[name + "_cfg"])
    f1, f2, coords = (t(g[f"{name}_{k}"]) for k in ("f1", "f2", "coords"))
    pyr = R.raft_group_corr_build(f1, f2, G, L)
    assert len(pyr) == L + 1
    for i, p in enumerate(pyr):
        assert p.shape[0] == B * G * H * W and np.array_equal(p.numpy(), g[f"{name}_pyr{i}"]), f"level {i}"
    assert np.array_equal(R.raft_group_corr_lookup(pyr, coords, G, L, r).numpy(), g[name + "_out"])


def test_raft_group_lookup_is_the_scrambled_view(gold):
    """Q6 spelled out: channel j of pixel p (flat index) is sample j % T of (group, pixel) row p*G + j // T — not group j // T of pixel p."""
    g = gold("c2f.npz")
    name = "g4_c16_w20_l1"
    B, C, H, W, L, r, G = (int(v) for v in g[name + "_cfg"])
    T = 2 * r + 1
    pyr0 = t(g[name + "_pyr0"]).reshape(B, G * H * W, W)
    coords, out = t(g[name + "_coords"]), t(g[name + "_out"])
    for (b, y, x, j) in ((0, 0, 0, 0), (1, 2, 7, 20), (0, 1, 19, 35), (1, 0, 3, 9)):
        row = (y * W + x) * G + j // T
        sp = row % (H * W)
        xs = torch.tensor([[float(j % T - r) + coords[b, 0, sp // W, sp % W].item()]])
        exp = R.linear_sampler(pyr0[b, row][None], xs)[0, 0]
        assert out[b, j, y, x] == exp, (b, y, x, j)


@pytest.mark.parametrize("name", ["c2f_b1_64x128_it3", "c2f_b2_128x192_it2"])
def test_coarse2fine_cascade(gold, name):
    """oracle.coarse2fine_refine (raft_stereo/model.py:280-320) on the stage tensors of the reference's forward == its outputs."""
    from nndepth_amd import weightgen
    g = gold("c2f.npz")
    B, Hf, Wf, iters, _ = (int(v) for v in g[name + "_cfg"])
    sd = weightgen.fill_state_dict([("c2f." + k, s) for k, s in R.update_block_spec("update_block", 128, 36, 128, 1, 4, gru="conv_gru")])
    sd = {k[len("c2f."):]: v for k, v in sd.items()}
    feats = [t(g[f"{name}_feat{i}"]) for i in range(3)]
    cnets = [t(g[f"{name}_cnet{i}"]) for i in range(3)]
    ups = R.coarse2fine_refine(sd, feats, cnets, (Hf, Wf), iters)
    assert len(ups) == 3 * iters
    for i, u in enumerate(ups):
        assert np.array_equal(u.numpy(), g[name + "_ups"][i]), i
