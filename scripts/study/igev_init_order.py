"""Study for VERDICT r3 "Next round" item 4 — dev-only, build container only (imports the reference from /root/reference).

IGEV's initial disparity is regress_disparity(softmax(cv_squeezer(geo), dim=1)) (nndepth/models/igev_stereo/model.py:92-95,144-146)
over 240 candidates.  Our fused kernel (csrc/corr1d.hip: igev_squeeze_softargmin_kernel) ends 2.8e-4 from the reference's fp32
value on the TartanAir pair.  Question: would evaluating the soft-argmin in the reference's own order (max-subtract, exp, sum,
divide, then sum_d d * p_d) bring the kernel closer, or is the distance set by something no order of that reduction can remove?

The script takes the reference's own fp32 logits on that pair and evaluates the expectation in several fp32 orders, and it
perturbs the logits by the rounding a different (equally valid) summation order of the 216-term Conv3d produces.
    PYTHONDONTWRITEBYTECODE=1 python scripts/study/igev_init_order.py > profiles/r04_igev_init_order_study.txt
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    from oracle.make_golden import _install_standins
    from oracle.make_golden_realdata import tartanair_frames
    _install_standins()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from nndepth_amd import weightgen
    from igev_double import make_igev
    from nndepth.models.igev_stereo.model import IGEVStereoBase
    from nndepth.models.igev_stereo.cost_volume import CostVolumeFilterNetwork

    model = make_igev(IGEVStereoBase, CostVolumeFilterNetwork, iters=1, hidden_dim=64, context_dim=64).eval()
    weightgen.fill_module_(model, "igev.")
    f1, f2 = tartanair_frames()
    cap = {}
    orig_reg, orig_sq = model.regress_disparity, model.cv_squeezer.forward

    def spy_reg(dist, width):
        cap["init"] = orig_reg(dist, width)
        return cap["init"]

    def spy_sq(x):
        cap["geo"] = x
        cap["logits"] = orig_sq(x)
        return cap["logits"]

    model.regress_disparity, model.cv_squeezer.forward = spy_reg, spy_sq
    with torch.no_grad():
        model(f1, f2)
    lg = cap["logits"].squeeze(1).float()  # (B, D, H, W)
    ref = cap["init"].float()
    D = lg.shape[1]
    dd = torch.arange(D, dtype=torch.float32).view(1, D, 1, 1)
    x64 = lg.double()
    exact = -(dd.double() * torch.softmax(x64, 1)).sum(1, keepdim=True)

    def report(name, v):
        print(f"{name:86s} vs reference fp32 {float((v.double() - ref.double()).abs().max()):.3e}   vs float64 {float((v.double() - exact).abs().max()):.3e}")

    print(f"logits {tuple(lg.shape)}, |init| max {float(ref.abs().max()):.1f}; one fp32 ulp at 128: {float(np.spacing(np.float32(128))):.2e}")
    report("reference's own ops again: -(d * softmax(logits, 1)).sum(1)", -(dd * torch.softmax(lg, 1)).sum(1, keepdim=True))
    # sequential fp32 evaluation in the reference's formula order: p_d = exp(x_d - max) / sum_e exp(x_e - max); acc += d * p_d
    mx = lg.max(1, keepdim=True).values
    e = torch.exp(lg - mx)
    s_seq = torch.zeros_like(mx)
    for d in range(D):
        s_seq = s_seq + e[:, d:d + 1]
    acc = torch.zeros_like(mx)
    for d in range(D):
        acc = acc + float(d) * (e[:, d:d + 1] / s_seq)
    report("softmax then expectation, both sums strictly sequential in d (fp32)", -acc)
    acc2 = torch.zeros_like(mx)
    for d in range(D):
        acc2 = acc2 + float(d) * (e[:, d:d + 1] / e.sum(1, keepdim=True))
    report("softmax with torch's sum, expectation strictly sequential (fp32)", -acc2)
    # our kernel's formulation: (sum_d d * e_d) / (sum_d e_d), pairwise sums
    num = (dd * e).numpy().astype(np.float32)
    den = e.numpy().astype(np.float32)

    def tree(a):  # pairwise over axis 1
        a = a.copy()
        n = a.shape[1]
        while n > 1:
            h = (n + 1) // 2
            b = a[:, :h].copy()
            b[:, :n - h] += a[:, h:n]
            a, n = b, h
        return a
    ours = -torch.from_numpy(tree(num) / tree(den))
    report("our kernel's form: (sum d * e_d) / (sum e_d), pairwise sums, one division (fp32)", ours)
    # what a different, equally valid summation order of the squeezer's 216 products does to the logits: the same Conv3d
    # evaluated in float64 and rounded once = the best any fp32 order can do; the reference's oneDNN order is one sample of the rest
    sq = model.cv_squeezer
    w64, b64 = sq.weight.double(), (sq.bias.double() if sq.bias is not None else None)
    with torch.no_grad():
        lg_best = torch.nn.functional.conv3d(cap["geo"].double(), w64, b64, padding=1).squeeze(1).float()
    print(f"{'logits: reference fp32 Conv3d vs the float64 Conv3d rounded once':86s} max-abs {float((lg_best - lg).abs().max()):.3e} (|logit| max {float(lg.abs().max()):.2f})")
    report("reference's soft-argmin ops on the once-rounded float64 logits", -(dd * torch.softmax(lg_best, 1)).sum(1, keepdim=True))
    report("float64 soft-argmin on the once-rounded float64 logits", (-(dd.double() * torch.softmax(lg_best.double(), 1)).sum(1, keepdim=True)).float())
    # ATen's own order (CPU, fp32): softmax over a non-innermost dim accumulates exp(x - max) strictly in order of d
    # (SoftMaxKernel.cpp, vec_softmax); sum(1) of the products is cascade_sum / multi_row_sum (SumKernel.cpp): blocks of 16
    # consecutive rows summed in order, the block sums summed in order (levels of 2^4; 240 rows never reach the third level)
    def aten_order(x):
        m_ = x.max(1, keepdim=True).values
        e_ = torch.exp(x - m_)
        s_ = torch.zeros_like(m_)
        for d in range(D):
            s_ = s_ + e_[:, d:d + 1]
        prod = dd * (e_ / s_)
        lvl1 = torch.zeros_like(m_)
        d = 0
        while d + 16 <= D:
            blk = torch.zeros_like(m_)
            for j in range(16):
                blk = blk + prod[:, d + j:d + j + 1]
            lvl1 = lvl1 + blk
            d += 16
        tail = torch.zeros_like(m_)
        for j in range(d, D):
            tail = tail + prod[:, j:j + 1]
        return -(tail + lvl1)
    report("ATen's order: sequential softmax sum; expectation in blocks of 16, block sums in order", aten_order(lg))
    report("the same order on the once-rounded float64 logits (what a kernel with other conv rounding gets)", aten_order(lg_best.detach()))


if __name__ == "__main__":
    main()
