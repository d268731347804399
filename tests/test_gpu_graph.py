"""GPU: a whole forward captured into a HIP graph and replayed (nndepth_amd/graph.py) — ONE host-side launch per pair — must
produce the outputs of the directly launched forward bit for bit: same kernels, same order, no arithmetic involved."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def test_graphed_raft_forward_equals_direct(raft_sd):
    from nndepth_amd import weightgen
    from nndepth_amd.graph import GraphedForward
    from nndepth_amd.raft_stereo import BaseRAFTStereo
    m = BaseRAFTStereo(iters=6, context_dim=64)
    m.load_state_dict(raft_sd, strict=True)
    m = m.to(DEV).eval()
    fwd = GraphedForward(m)
    for seed in (0, 1, 2):  # the second and third pair go through the replay only
        f1, f2 = (x.to(DEV) for x in weightgen.synthetic_frames(seed, 1, 96, 160))
        direct = [o["up_disp"].clone() for o in m(f1, f2)]
        replay = fwd(f1, f2)
        assert len(replay) == 6
        for a, b in zip(direct, replay):
            assert torch.equal(a, b["up_disp"])
    # another shape: its own graph
    f1, f2 = (x.to(DEV) for x in weightgen.synthetic_frames(3, 2, 64, 128))
    direct = m(f1, f2)[-1]["up_disp"].clone()
    assert torch.equal(direct, fwd(f1, f2)[-1]["up_disp"])
    assert len(fwd._graphs) == 2


def test_graphed_cre_forward_equals_direct(cre_sd):
    from nndepth_amd import weightgen
    from nndepth_amd.cre_stereo import CREStereoBase
    from nndepth_amd.graph import GraphedForward
    m = CREStereoBase(iters=4)
    m.load_state_dict(cre_sd, strict=True)
    m = m.to(DEV).eval()
    fwd = GraphedForward(m)
    for seed in (3, 4):
        f1, f2 = (x.to(DEV) for x in weightgen.synthetic_frames(seed, 1, 128, 192))
        direct = [o["up_disp"].clone() for o in m(f1, f2)]
        replay = fwd(f1, f2)
        assert len(replay) == len(direct) == 8
        for a, b in zip(direct, replay):
            assert torch.equal(a, b["up_disp"])
