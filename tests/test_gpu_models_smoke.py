"""Mirror of the reference's own model tests (tests/models/test_stereo.py: construct each model with its defaults, run one forward on
`torch.rand` frames of the same sizes, `assert isinstance(outputs, list)`) on the HIP path.  Differences, on purpose: the models are put
in eval mode (the HIP path is inference-only and says so in training mode), the two models whose backbone is not part of the hot path
(IGEV: timm's MobileNetV3, unavailable offline; Coarse2Fine: RepViT) take the test doubles' encoder side, and the outputs are also
checked for count, shape and finiteness."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _check(outputs, n, hw):
    assert isinstance(outputs, list) and len(outputs) == n
    for o in outputs:
        assert o["up_disp"].shape[-2:] == hw and torch.isfinite(o["up_disp"]).all()


def test_BaseRAFTStereo():
    from nndepth_amd.raft_stereo import BaseRAFTStereo
    model = BaseRAFTStereo().to(DEV).eval()
    left, right = torch.rand((1, 3, 480, 640), device=DEV), torch.rand((1, 3, 480, 640), device=DEV)
    _check(model(left, right), model.iters, (480, 640))
    with pytest.raises(Exception):
        model.train()(left, right)  # inference-only: training mode is refused, not silently run


def test_Coarse2FineGroupRepViTRAFTStereo():
    from c2f_double import make_c2f
    from nndepth_amd.raft_stereo import Coarse2FineRAFTStereoBase
    model = make_c2f(Coarse2FineRAFTStereoBase, corr_levels=1).to(DEV).eval()
    left, right = torch.rand((1, 3, 384, 512), device=DEV), torch.rand((1, 3, 384, 512), device=DEV)
    _check(model(left, right), 3 * model.iters, (384, 512))


def test_CREStereoBase():
    from nndepth_amd.cre_stereo import CREStereoBase
    model = CREStereoBase().to(DEV).eval()
    left, right = torch.rand((1, 3, 480, 640), device=DEV), torch.rand((1, 3, 480, 640), device=DEV)
    outputs = model(left, right)
    assert isinstance(outputs, list) and len(outputs) == 2 * model.iters
    # 2-channel flow; the 1/16 and 1/8 stages emit at 1/4 and 1/2 of the frame size, the last stage at frame size (cre_stereo/model.py:198-283)
    assert all(o["up_disp"].shape[:2] == (1, 2) and torch.isfinite(o["up_disp"]).all() for o in outputs)
    assert [tuple(o["up_disp"].shape[-2:]) for o in outputs[::6]] == [(120, 160), (240, 320), (480, 640), (480, 640)]


def test_IGEVStereo():
    from igev_double import make_igev
    from nndepth_amd.igev_stereo import IGEVStereoBase, CostVolumeFilterNetwork
    model = make_igev(IGEVStereoBase, CostVolumeFilterNetwork).to(DEV).eval()
    left, right = torch.rand((1, 3, 480, 640), device=DEV), torch.rand((1, 3, 480, 640), device=DEV)
    _check(model(left, right), model.iters, (480, 640))
