"""ORACLE tooling — dev-only, runs ONLY in the build container (needs /root/reference).

CREStereo rows: imports the reference package read-only, loads the deterministic weights of
`nndepth_amd.weightgen`, runs AGCL (both modes / window shapes), the bilinear sampler and a small 3-scale cascade,
cross-checks `oracle/cre_ref.py` against them and writes tests/golden/cre_*.npz.  Only tensors are stored.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_cre.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def _np(t):
    return t.detach().cpu().numpy()


def main():
    from nndepth_amd import weightgen
    from oracle import cre_ref as C
    from oracle.make_golden import _install_standins

    _install_standins()
    torch.manual_seed(0)
    from nndepth.models.cre_stereo.model import CREStereoBase
    from nndepth.models.cre_stereo.cost_volume import AGCL
    from nndepth.models.cre_stereo.utils import bilinear_sampler

    report = {}

    def rnd(tag, *shape, lo=-1.0, hi=1.0):
        n = int(np.prod(shape))
        return torch.from_numpy(weightgen.uniform01(tag, n).reshape(shape) * (hi - lo) + lo)

    # ---------------------------------------------------------------- model + spec
    model = CREStereoBase(iters=4).eval()
    spec = C.cre_stereo_spec()
    ref_shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert [k for k, _ in spec] == list(ref_shapes.keys()), "state-dict key order differs"
    assert dict(spec) == ref_shapes
    sd = weightgen.fill_state_dict(spec)
    model.load_state_dict(sd)

    # ---------------------------------------------------------------- a19: sampler
    cases = {}
    for name, (N, Cc, H, W, Hg, Wg) in {"s_small": (2, 6, 7, 9, 5, 11), "s_wide": (1, 3, 4, 33, 8, 16)}.items():
        img = rnd("simg" + name, N, Cc, H, W)
        xs = rnd("sx" + name, N, Hg, Wg, lo=-3.0, hi=W + 2.0)
        ys = rnd("sy" + name, N, Hg, Wg, lo=-3.0, hi=H + 2.0)
        xs[:, 0, :] = torch.round(xs[:, 0, :])          # exact integers
        ys[:, :, 0] = torch.round(ys[:, :, 0])
        xs[:, 1, 0], ys[:, 1, 0] = 0.0, 0.0             # corners
        xs[:, 1, 1], ys[:, 1, 1] = W - 1.0, H - 1.0
        xs[:, 1, 2], ys[:, 1, 2] = -1.0, -1.0
        xs[:, 1, 3], ys[:, 1, 3] = float(W), float(H)
        coords = torch.stack([xs, ys], -1)
        ref = bilinear_sampler(img, coords)
        mine = C.bilinear_sampler(img, coords)
        report[f"cre/sampler/{name}"] = ((ref - mine).abs().max().item(),)
        cases[name + "_img"], cases[name + "_coords"], cases[name + "_out"] = _np(img), _np(coords), _np(ref)
    np.savez_compressed(os.path.join(GOLD, "cre_sampler.npz"), **cases)

    # ---------------------------------------------------------------- a17/a18: AGCL
    cases = {}
    for name, (N, Cc, H, W, famp) in {"c32": (2, 32, 9, 14, 3.0), "c256": (1, 256, 6, 10, 5.0), "c64_big": (1, 64, 5, 8, 20.0)}.items():
        f1 = rnd("af1" + name, N, Cc, H, W)
        f2 = rnd("af2" + name, N, Cc, H, W)
        flow = rnd("afl" + name, N, 2, H, W, lo=-famp, hi=famp)
        flow[:, :, 0, :] = torch.round(flow[:, :, 0, :])
        off = rnd("aof" + name, N, 18, H, W)
        agcl = AGCL(f1, f2)
        cases[name + "_f1"], cases[name + "_f2"], cases[name + "_flow"], cases[name + "_off"] = map(_np, (f1, f2, flow, off))
        for sp in (False, True):
            with torch.no_grad():
                r_it = agcl(flow, None, small_patch=sp, iter_mode=True)
                r_of = agcl(flow, off, small_patch=sp)
            m_it = C.agcl_corr_iter(f1, f2, flow, sp)
            m_of = C.agcl_corr_att_offset(f1, f2, flow, off, sp)
            report[f"cre/agcl/{name}/small={int(sp)}"] = ((r_it - m_it).abs().max().item(), (r_of - m_of).abs().max().item())
            cases[f"{name}_iter_sp{int(sp)}"] = _np(r_it)
            cases[f"{name}_off_sp{int(sp)}"] = _np(r_of)
    # with the cross-attention in front (the 1/32 stage): tokens of width 256
    f1, f2 = rnd("att_f1", 1, 256, 5, 8), rnd("att_f2", 1, 256, 5, 8)
    flow, off = rnd("att_fl", 1, 2, 5, 8, lo=-2, hi=2), rnd("att_of", 1, 18, 5, 8)
    with torch.no_grad():
        ref = AGCL(f1, f2, att=model.cross_att_fn)(flow, off, small_patch=False)
    mine = C.agcl_corr_att_offset(f1, f2, flow, off, False,
                                  att=lambda a, b: C.feature_transformer(sd, "cross_att_fn", "cross", a, b))
    report["cre/agcl/att"] = ((ref - mine).abs().max().item(),)
    for k, v in (("f1", f1), ("f2", f2), ("flow", flow), ("off", off), ("out", ref)):
        cases["att_" + k] = _np(v)
    np.savez_compressed(os.path.join(GOLD, "cre_agcl.npz"), **cases)

    # ---------------------------------------------------------------- a20: small cascade
    fr1, fr2 = weightgen.synthetic_frames(3, 1, 128, 192)
    with torch.no_grad():
        out = model(fr1, fr2)
        mine = C.cre_stereo_forward(sd, fr1, fr2, 4)
    assert len(out) == len(mine) == 8
    report["cre/forward/128x192_it4"] = tuple((o["up_disp"] - m).abs().max().item() for o, m in zip(out, mine))
    init = rnd("cre_init", 1, 2, 64, 96, lo=-4, hi=4)
    model2 = CREStereoBase(iters=2).eval()
    model2.load_state_dict(sd)
    with torch.no_grad():
        out2 = model2(fr1, fr2, flow_init=init)
        mine2 = C.cre_stereo_forward(sd, fr1, fr2, 2, flow_init=init)
    assert len(out2) == len(mine2) == 2
    report["cre/forward/flow_init_it2"] = tuple((o["up_disp"] - m).abs().max().item() for o, m in zip(out2, mine2))
    np.savez_compressed(os.path.join(GOLD, "cre_forward.npz"),
                        flow_init=_np(init), up_disp_init=np.stack([_np(o["up_disp"]) for o in out2]),
                        **{f"up_disp_{i}": _np(o["up_disp"]) for i, o in enumerate(out)})

    print("\n== CRE golden report (max-abs oracle/cre_ref.py vs imported reference) ==")
    lines = [f"{k:45s} " + "  ".join(f"{x:.3e}" for x in v) for k, v in report.items()]
    print("\n".join(lines))
    with open(os.path.join(GOLD, "REPORT_cre.txt"), "w") as f:
        f.write("golden vectors generated by oracle/make_golden_cre.py from the imported reference\n")
        f.write(f"torch {torch.__version__}, numpy {np.__version__}\n")
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
