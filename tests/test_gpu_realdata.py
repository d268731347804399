"""GPU: real-image parity against the imported reference (goldens of oracle/make_golden_realdata.py), in the exact and in both
split arithmetics — the wider gate that licenses a split arithmetic as the default of the model classes:

  KITTI sample pair 375x1242 -> Padder(32) -> RAFT-Stereo base, 32 iterations          (BASELINE.json configs[3], per pair)
  TartanAir sample pair 544x960 -> CREStereo, iters = 4 (8 outputs of the 3-scale cascade)
  TartanAir sample pair 544x960 -> IGEV on the tiny backbone, 32 iterations            (configs[2], per sample)

Bar: <= 1e-4 max-abs on the final full-resolution map (north_star), drift printed at iterations 1 / 4 / 12 / 32.  IGEV needs
two qualifications, both measured on the reference itself and stored with the golden (tests/golden/REPORT_realdata.txt):
(1) its state is the ABSOLUTE coordinate (quirk Q5) and its full-resolution output 4 x that (values up to ~530, one fp32 ulp
= 6.1e-5): the reference's own 1-thread vs 8-thread outputs differ by 1.8e-4 there; (2) its initial disparity is a soft-argmin
over 240 candidates whose fp32 evaluation in the reference is itself 1.8e-4 from the float64 value of the same expression, so
"the reference's init" is only defined to that accuracy.  Hence: the LOOP is held to 1e-4 absolute on the 1/4-resolution
coordinates when it starts from the reference's own initial disparity, and the end-to-end forward (our soft-argmin kernel
included) to that init uncertainty propagated (x4 through the upsample) plus 8 ulp of the largest output."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ARITHS = ["fp32", "bf16x3", "fp16x2"]


def _png(name):
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(GOLD, name)).convert("RGB"))
    return torch.from_numpy(img.copy()).permute(2, 0, 1).float().unsqueeze(0)


@pytest.mark.parametrize("arith", ARITHS)
def test_raft_kitti_pair_32_iterations_vs_reference(gold, raft_sd, arith):
    from nndepth_amd.cost_volume import CorrBlock1D
    from nndepth_amd.prepost import Padder
    from nndepth_amd.raft_stereo import BaseRAFTStereo
    g = gold("forward_kitti.npz")
    frames = [((_png(f"kitti_000000_10_{s}.png") - 127.5) / 127.5).to(DEV) for s in ("left", "right")]
    padder = Padder((375, 1242), divis_by=32)
    p1, p2 = padder.pad(*frames)
    assert tuple(p1.shape) == (1, 3, 384, 1248) and list(g["pad"]) == list(padder._pad)
    m = BaseRAFTStereo(iters=32, context_dim=64, arithmetic=arith)
    m.load_state_dict(raft_sd, strict=True)
    m = m.to(DEV).eval()
    out = m(p1, p2)
    final = out[-1]["up_disp"].cpu().numpy()
    err32 = np.abs(final - g["up_disp_it32"]).max()
    print(f"\n[kitti {arith}] 384x1248 it32 max-abs vs reference = {err32:.3e}  (|disp| max {np.abs(g['up_disp_it32']).max():.2f})")
    # VERDICT r3 item 5 — drift of our path next to the reference's OWN 1-thread vs 8-thread drift on this pair (stored with the
    # golden): full-resolution map at iterations 1 / 4 / 12 (every 4th pixel) and 32 (all pixels)
    drift = [np.abs(out[int(it) - 1]["up_disp"].cpu().numpy()[:, :, ::4, ::4] - g[f"up_disp_sub4_it{int(it)}"]).max() for it in g["low_iters"][:-1]]
    drift.append(err32)
    print(f"[kitti {arith}] up_disp drift at it 1 / 4 / 12 / 32: " + " ".join(f"{e:.2e}" for e in drift) +
          "   reference 1 vs 8 threads: " + " ".join(f"{e:.2e}" for e in g["ref_self_noise_up"]))
    eng = m.update_block.sync_engine(DEV)
    fmap1, fmap2, cnet = m.forward_fnet(p1, p2)
    net, inp = torch.split(cnet, [128, 64], dim=1)
    net, inp = torch.tanh(net), torch.relu(inp)
    corr = CorrBlock1D(fmap1, fmap2, 4, 4)
    for k, it in enumerate(g["low_iters"]):
        _, low, _ = eng.refine(corr._pyr, 4, 4, net, inp, 8, int(it), keep_all=False)
        e = np.abs(low.cpu().numpy() - g["low_disp"][k]).max()
        print(f"[kitti {arith}] low-res disparity after {int(it):2d} iters: max-abs = {e:.3e}  (reference 1 vs 8 threads: {float(g['ref_self_noise_low'][k]):.2e})")
        assert e <= 1e-4
    # Bars.  Iterations 1 .. 12: the north-star's 1e-4 (measured <= 3.1e-5: at most one ulp beyond the reference's own noise).
    # Iteration 32: this pair is chaotic enough that the reference's own output moves by 9.6e-5 with its thread count; our exact
    # path lands 9.7e-5 from the 8-thread run (1.01 x that), the split arithmetics 4.3e-5 ... 9.6e-5 — inside the reference's
    # envelope, but a literal 1e-4 bar would sit 3 % above the reference's self-noise and trip on any reordering.  The bar for this
    # pair is therefore pinned to 1.25 x the stored self-noise (1.2e-4); the TartanAir pair keeps the literal 1e-4 (3.5e-5 measured).
    assert max(drift[:3]) <= 1e-4
    assert err32 <= max(1e-4, 1.25 * float(g["ref_self_noise_up"][-1]))
    # EPE parity on the sample's ground truth (valid pixels), through the device-side unpad
    valid = np.unpackbits(g["gt_valid"])[:375 * 1242].reshape(375, 1242).astype(bool)
    ours = padder.unpad(out[-1]["up_disp"])[0, 0].cpu().numpy()
    epe = float(np.abs(ours - g["gt_disp"].astype(np.float32))[valid].mean())
    print(f"[kitti {arith}] EPE ours {epe:.6f} vs reference forward {float(g['epe_ref']):.6f}")
    assert abs(epe - float(g["epe_ref"])) <= 1e-4


@pytest.mark.parametrize("arith", ARITHS)
def test_cre_tartanair_544x960_vs_reference(gold, cre_sd, tartanair_frames, arith):
    from nndepth_amd.cre_stereo import CREStereoBase
    g = gold("forward_cre_tartanair.npz")
    m = CREStereoBase(iters=4, arithmetic=arith)
    m.load_state_dict(cre_sd, strict=True)
    outs = m.to(DEV).eval()(tartanair_frames[0].to(DEV), tartanair_frames[1].to(DEV))
    assert len(outs) == 8
    errs = [np.abs(o["up_disp"].cpu().numpy()[:, :, ::4, ::4] - g[f"up_disp_sub4_{i}"]).max() for i, o in enumerate(outs[:-1])]
    errs.append(np.abs(outs[-1]["up_disp"].cpu().numpy() - g["up_disp_final"]).max())
    print(f"\n[cre {arith}] 544x960 iters=4 max-abs per output: " + " ".join(f"{e:.2e}" for e in errs) +
          f"  (|flow| max {np.abs(g['up_disp_final']).max():.1f})")
    assert max(errs) <= 1e-4


@pytest.mark.parametrize("arith", ARITHS)
def test_igev_tartanair_544x960_32_iterations_vs_reference(gold, tartanair_frames, arith):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from igev_double import make_igev
    from nndepth_amd import weightgen
    from nndepth_amd.igev_stereo import IGEVStereoBase, CostVolumeFilterNetwork
    g = gold("forward_igev_tartanair.npz")
    m = make_igev(IGEVStereoBase, CostVolumeFilterNetwork, iters=32, hidden_dim=64, context_dim=64, arithmetic=arith)
    weightgen.fill_module_(m, "igev.")
    m = m.to(DEV).eval()
    f1, f2 = tartanair_frames[0].to(DEV), tartanair_frames[1].to(DEV)
    outs = m(f1, f2)
    final = outs[-1]["up_disp"].cpu().numpy()
    err_up = np.abs(final - g["up_disp_it32"]).max()
    e_init_ref = float(g["ref_init_err_vs_f64"])
    ulp_up = float(np.spacing(np.float32(np.abs(g["up_disp_it32"]).max())))
    # round 4: the soft-argmin kernel evaluates in ATen's order (csrc/corr1d.hip; profiles/r04_igev_init_order_study.txt) — the 1/4-
    # resolution coordinates now meet the north-star bar literally end to end (measured 6.9e-5; rounds 2-3: 2.8e-4 against a
    # bar of 4.5e-4), the full-resolution map (4 x the coordinate, up to 530: one ulp = 6.1e-5) is held to 5 ulp (measured 3 - 3.5)
    tol_low_e2e = 1e-4
    tol_up = 5.0 * ulp_up
    e_low_e2e = np.abs(m.last_low_coords.cpu().numpy() - g["low_coords"][-1]).max()
    print(f"\n[igev {arith}] end to end, 544x960 it32: up_disp max-abs vs reference = {err_up:.3e} (|4 x coords| max {np.abs(g['up_disp_it32']).max():.1f}, "
          f"tolerance {tol_up:.2e}), 1/4-res coordinates {e_low_e2e:.3e} (tolerance {tol_low_e2e:.2e}; the reference's fp32 init is "
          f"{e_init_ref:.2e} from float64, its 1-thread vs 8-thread outputs differ by {float(g['ref_self_noise_up'].max()):.2e})")
    assert err_up <= tol_up and e_low_e2e <= tol_low_e2e
    # the loop alone, from the reference's own initial disparity: 1/4-resolution coordinates after 1 / 4 / 12 / 32 iterations
    fmap1, fmap2, cnet1, guides = m.forward_fnet(f1, f2)
    net, inp = torch.tanh(cnet1[:, :64]).contiguous(), torch.relu(cnet1[:, 64:]).contiguous()
    corr = m.corr_fn(fmap1.float(), fmap2.float(), guides, m.cv_regularizer, 4, 4, 8)
    eng = m.update_block.sync_engine(DEV)
    init = torch.from_numpy(g["init"]).to(DEV)
    for k, it in enumerate(g["low_iters"]):
        up, low, _ = eng.refine_igev(corr._feat, corr._geo, 8, 4, 4, net, inp, 4, int(it), disp_init=init, keep_all=False,
                                     interleaved=corr.interleaved())
        e = np.abs(low.cpu().numpy() - g["low_coords"][k]).max()
        print(f"[igev {arith}] loop from the reference's init, 1/4-res coordinates after {int(it):2d} iters: max-abs = {e:.3e}  "
              f"(reference self-noise {float(g['ref_self_noise_low'][k]):.2e})")
        assert e <= 1e-4
    e_up_loop = np.abs(up[0].cpu().numpy() - g["up_disp_it32"]).max()
    print(f"[igev {arith}] loop from the reference's init, up_disp it32: max-abs = {e_up_loop:.3e} ({e_up_loop / ulp_up:.1f} ulp of the largest value)")
    assert e_up_loop <= max(1e-4, 8.0 * ulp_up)
