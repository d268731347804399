"""CPU: the CREStereo oracle (oracle/cre_ref.py) against the golden vectors produced by the imported reference
(oracle/make_golden_cre.py) — SURVEY §8 rows a17-a20.  Pins the oracle the GPU tests compare against."""
import numpy as np
import pytest
import torch

from conftest import t
from oracle import cre_ref as C


@pytest.mark.parametrize("name", ["s_small", "s_wide"])
def test_bilinear_sampler(gold, name):
    g = gold("cre_sampler.npz")
    out = C.bilinear_sampler(t(g[name + "_img"]), t(g[name + "_coords"]))
    assert np.array_equal(out.numpy(), g[name + "_out"])


def test_bilinear_sampler_zero_border():
    """Taps outside the image read zero; integer coordinates return the pixel itself."""
    img = torch.arange(12.0).reshape(1, 1, 3, 4) + 1
    pts = torch.tensor([[[[0.0, 0.0], [3.0, 2.0], [-1.0, 0.0], [4.0, 2.0], [-0.5, 0.0], [1.0, -5.0]]]])
    out = C.bilinear_sampler(img, pts)[0, 0, 0]
    assert out.tolist() == pytest.approx([1.0, 12.0, 0.0, 0.0, 0.5, 0.0], abs=1e-6)  # the [-1,1] round trip is inexact


@pytest.mark.parametrize("name", ["c32", "c256", "c64_big"])
@pytest.mark.parametrize("sp", [0, 1])
def test_agcl_both_modes(gold, name, sp):
    g = gold("cre_agcl.npz")
    f1, f2, flow, off = (t(g[f"{name}_{k}"]) for k in ("f1", "f2", "flow", "off"))
    it = C.agcl_corr_iter(f1, f2, flow, bool(sp))
    assert it.shape[1] == 36
    assert np.abs(it.numpy() - g[f"{name}_iter_sp{sp}"]).max() <= 1e-7
    of = C.agcl_corr_att_offset(f1, f2, flow, off, bool(sp))
    assert np.abs(of.numpy() - g[f"{name}_off_sp{sp}"]).max() <= 1e-7


def test_agcl_with_cross_attention(gold, cre_sd):
    g = gold("cre_agcl.npz")
    f1, f2, flow, off = (t(g["att_" + k]) for k in ("f1", "f2", "flow", "off"))
    out = C.agcl_corr_att_offset(f1, f2, flow, off, False,
                                 att=lambda a, b: C.feature_transformer(cre_sd, "cross_att_fn", "cross", a, b))
    assert np.abs(out.numpy() - g["att_out"]).max() <= 2e-6


def test_cascade_small(gold, cre_sd):
    from nndepth_amd import weightgen
    g = gold("cre_forward.npz")
    fr1, fr2 = weightgen.synthetic_frames(3, 1, 128, 192)
    outs = C.cre_stereo_forward(cre_sd, fr1, fr2, 4)
    assert [tuple(o.shape) for o in outs] == [(1, 2, 32, 48)] * 2 + [(1, 2, 64, 96)] * 2 + [(1, 2, 128, 192)] * 4
    for i, o in enumerate(outs):
        assert np.abs(o.numpy() - g[f"up_disp_{i}"]).max() <= 2e-5, i
    outs = C.cre_stereo_forward(cre_sd, fr1, fr2, 2, flow_init=t(g["flow_init"]))
    for i, o in enumerate(outs):
        assert np.abs(o.numpy() - g["up_disp_init"][i]).max() <= 2e-5, i


def test_cre_model_state_dict_matches_spec(cre_sd):
    """The drop-in CREStereoBase registers the reference's state_dict keys/shapes in the reference's order
    (spec checked against the imported reference in oracle/make_golden_cre.py) and loads strictly."""
    from nndepth_amd.cre_stereo import CREStereoBase
    m = CREStereoBase(iters=4)
    got = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert got == [(k, tuple(s)) for k, s in C.cre_stereo_spec()]
    m.load_state_dict(cre_sd, strict=True)
    with pytest.raises(ValueError):
        CREStereoBase(hidden_dim=128, context_dim=64)


def test_cascade_tartanair_544x960_realdata(gold, cre_sd, tartanair_frames):
    """oracle/make_golden_realdata.py cre: the oracle's cascade on the TartanAir sample pair at 544x960, iters = 4, against the
    imported reference's 8 outputs (<= 4.8e-6 at generation time, tests/golden/REPORT_realdata.txt)."""
    g = gold("forward_cre_tartanair.npz")
    with torch.no_grad():
        outs = C.cre_stereo_forward(cre_sd, tartanair_frames[0], tartanair_frames[1], 4)
    assert len(outs) == 8
    for i, o in enumerate(outs[:-1]):
        assert np.abs(o.numpy()[:, :, ::4, ::4] - g[f"up_disp_sub4_{i}"]).max() <= 2e-5, i
    assert np.abs(outs[-1].numpy() - g["up_disp_final"]).max() <= 2e-5
