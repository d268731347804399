"""CPU: the C-ABI library loads without a GPU, exports every symbol include/nndepth_amd.h declares,
its host-side functions (layout, packing, sizing, argument validation) behave, and the Python seams
refuse to compute anywhere but on the HIP device."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "nndepth_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nnd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from nndepth_amd import _lib
    syms = _declared_symbols()
    assert len(syms) >= 15
    raw = C.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/nndepth_amd.h but not exported"
    assert set(_lib.SIGNATURES) == set(syms), "ctypes table and header out of sync"
    assert _lib.lib.nnd_version() == 104
    assert _lib.lib.nnd_device_count() >= 0  # must not raise / abort on a CPU-only host


def test_pyramid_layout_host():
    from nndepth_amd import ops
    offs, widths, total = ops.pyramid_layout(2, 5, 21, 4)
    assert widths == [21, 10, 5, 2, 1]  # floor on odd widths; 5 levels stored, 4 read (Q1)
    rows = 2 * 5 * 21
    assert offs == [0, rows * 21, rows * 31, rows * 36, rows * 38] and total == rows * 39
    from nndepth_amd._lib import NndError
    with pytest.raises(NndError):
        ops.pyramid_layout(0, 5, 21, 4)


def _unpack(blob, Cout, Cin, KH, KW):
    """Inverse of the documented A-fragment order [cb][chunk][tap][q][lane][j]."""
    NT = KH * KW
    CI_T = 128 if (KH == 1 and KW == 1 and Cin >= 128) else 32
    if KH * KW == 5 and Cin >= 256 and Cout >= 256:
        CI_T = 64  # wide GRU gate convs (csrc/conv_mfma.hip: conv_ci_t)
    NQ, nch, ncb = CI_T // 8, -(-Cin // CI_T), -(-Cout // 32)
    n = ncb * nch * NT * NQ * 256
    Wp = blob[:n].reshape(ncb, nch, NT, NQ, 2, 32, 4)  # cb, chunk, tap, q, h2, i, j
    rec = np.zeros((ncb, 32, nch, CI_T, NT), np.float32)
    for q in range(NQ):
        for j in range(4):
            for h2 in range(2):
                rec[:, :, :, (q * 4 + j) * 2 + h2, :] = Wp[:, :, :, q, h2, :, j].transpose(0, 3, 1, 2)
    return rec.reshape(ncb * 32, nch * CI_T, NT), blob[n:n + ncb * 32]


@pytest.mark.parametrize("shape", [(192, 256, 3, 3), (127, 256, 3, 3), (576, 256, 1, 1), (256, 36, 1, 1),
                                   (256, 320, 1, 5), (128, 320, 5, 1), (1, 128, 3, 3)])
def test_conv2d_pack_host(shape):
    from nndepth_amd import ops
    Cout, Cin, KH, KW = shape
    g = torch.Generator().manual_seed(0)
    w, b = torch.randn(shape, generator=g), torch.randn(Cout, generator=g)
    conv = ops.Conv2d(w, b, device=None)  # host-only: pack, never touches a GPU
    rec, bias = _unpack(conv.packed_host.numpy(), Cout, Cin, KH, KW)
    assert np.array_equal(rec[:Cout, :Cin], w.numpy().reshape(Cout, Cin, KH * KW))
    assert not rec[Cout:].any() and not rec[:, Cin:].any(), "padding must be zero"
    assert np.array_equal(bias[:Cout], b.numpy()) and not bias[Cout:].any()


def test_update_block_pack_and_sizes_host():
    from nndepth_amd import ops, weightgen
    from nndepth_amd._lib import lib, NndError, UpdateBlockDesc
    from oracle import torch_ref as R
    eng = ops.UpdateBlockEngine(128, 64, 36, 1, 576, "sep_conv")
    assert lib.nnd_update_block_num_tensors(C.byref(eng.desc)) == 30 == len(ops.update_block_keys("sep_conv"))
    assert len(ops.update_block_keys("conv_gru")) == 24
    sd = weightgen.fill_state_dict(R.update_block_spec("u", 128, 36, 64, 1, 8))
    blob = eng.pack_host(sd, "u.")
    assert blob.numel() == eng.packed_floats and torch.isfinite(blob).all()
    # first layer of the blob is encoder.convc1 (1x1, 36 -> 256, CI_T 32)
    rec, bias = _unpack(blob.numpy(), 256, 36, 1, 1)
    assert np.array_equal(rec[:256, :36, 0], sd["u.encoder.convc1.weight"].numpy()[:, :, 0, 0])
    assert np.array_equal(bias[:256], sd["u.encoder.convc1.bias"].numpy())
    assert lib.nnd_update_block_workspace_floats(C.byref(eng.desc), 1, 68, 120) > 2000 * 68 * 120
    bad = UpdateBlockDesc(100, 64, 36, 1, 576, 0, 0, 0, 0)  # hidden_dim not a multiple of 32
    assert lib.nnd_update_block_packed_floats(C.byref(bad)) < 0
    assert b"hidden_dim" in lib.nnd_last_error()
    with pytest.raises(NndError):
        ops.UpdateBlockEngine(128, 64, 36, 3, 576)  # flow_channels must be 1 or 2


def test_descriptors_carry_their_size_and_fp16x2_layers_their_scale_slot():
    """Round 4: (i) every descriptor starts with struct_size and an entry point refuses another size or unknown flags; (ii) which
    layers take the split arithmetic is the descriptor's split_layers (blob layout = function of the descriptor alone); (iii) a
    packed fp16x2 layer carries the 4-float slot {2^-(s+xs), 2^xs, record, 2^-s} behind its bias with xs = 2 until calibrated
    (csrc/split_arith.h: SPLIT_TAIL_*, include/nndepth_amd.h "fp16x2 activation range")."""
    from nndepth_amd import ops, weightgen
    from nndepth_amd._lib import lib, UpdateBlockDesc, EncoderDesc, Conv3dDesc
    from oracle import torch_ref as R
    eng = ops.UpdateBlockEngine(128, 64, 36, 1, 576, "sep_conv", "fp16x2")
    assert eng.desc.struct_size == C.sizeof(UpdateBlockDesc) == 40
    n_full = lib.nnd_update_block_packed_floats(C.byref(eng.desc))
    assert n_full > 0
    eng.desc.struct_size = 28  # a caller compiled against round 3's header
    assert lib.nnd_update_block_packed_floats(C.byref(eng.desc)) < 0 and b"struct_size" in lib.nnd_last_error()
    eng.desc.struct_size = C.sizeof(UpdateBlockDesc)
    eng.desc.flags = 8
    assert lib.nnd_update_block_packed_floats(C.byref(eng.desc)) < 0 and b"flags" in lib.nnd_last_error()
    eng.desc.flags = 0
    for D in (EncoderDesc(256, 1, 192, 2, 0), Conv3dDesc(16, 8, 0, 1, 2, 0)):
        assert D.struct_size == C.sizeof(type(D))
    bad = EncoderDesc(256, 1, 192, 2, 0)
    bad.struct_size = 16
    assert lib.nnd_encoder_packed_floats(C.byref(bad)) < 0 and b"struct_size" in lib.nnd_last_error()
    # split_layers: only encoder.convc2 (bit 1) in the split arithmetic -> another blob size, same for pack and forward by construction
    sub = UpdateBlockDesc(128, 64, 36, 1, 576, 0, 2, 1 << 1, 0)
    n_sub = lib.nnd_update_block_packed_floats(C.byref(sub))
    assert 0 < n_sub != n_full
    # scale slots: one per fp16x2 convolution, -1 for the others (convc1's 36 planes stay exact fp32)
    n = lib.nnd_update_block_scale_slots(C.byref(eng.desc), None, 0)
    offs = (C.c_int64 * n)()
    assert lib.nnd_update_block_scale_slots(C.byref(eng.desc), offs, n) == n
    names = [lib.nnd_conv_name(C.byref(eng.desc), i).decode() for i in range(lib.nnd_num_convs(C.byref(eng.desc)))]
    assert offs[names.index("encoder.convc1")] == -1 and offs[names.index("encoder.convc2")] > 0
    sd = weightgen.fill_state_dict(R.update_block_spec("u", 128, 36, 64, 1, 8))
    blob = eng.pack_host(sd, "u.").numpy()
    slots = [o for o in offs if o >= 0]
    assert len(slots) >= 12 and len(set(slots)) == len(slots)
    for o in slots:
        osc, xsc, rec, wsinv = blob[o:o + 4]
        assert xsc == 4.0 and rec == 0.0 and osc == wsinv / xsc
        assert np.log2(wsinv) == np.round(np.log2(wsinv))  # an exact power of two
    wmax = np.abs(sd["u.encoder.convc2.weight"].numpy()).max()
    wsinv = blob[offs[names.index("encoder.convc2")] + 3]
    assert 2.0 ** 13 <= wmax / wsinv < 2.0 ** 14  # weights scaled into the top of fp16's range


def test_null_and_shape_errors_do_not_crash():
    from nndepth_amd._lib import lib
    assert lib.nnd_corr1d_build(None, None, None, 1, 8, 4, 8, 4, None) == -1
    assert b"null" in lib.nnd_last_error()
    assert lib.nnd_convex_upsample(None, None, None, 1, 1, 4, 8, 8, None) == -1
    assert lib.nnd_conv2d_packed_floats(8, 8, 7, 7) < 0  # kernel size not built


def test_seams_refuse_cpu_tensors():
    """No silent CPU/eager fallback: every op raises on non-HIP tensors."""
    from nndepth_amd import ops
    from nndepth_amd._lib import NndError
    from nndepth_amd.blocks import BasicUpdateBlock
    from nndepth_amd.cost_volume import CorrBlock1D
    from nndepth_amd.upsample import convex_upsample
    with pytest.raises(NndError):
        CorrBlock1D(torch.zeros(1, 8, 4, 16), torch.zeros(1, 8, 4, 16), 4, 4)
    with pytest.raises(NndError):
        convex_upsample(torch.zeros(1, 1, 4, 8), torch.zeros(1, 576, 4, 8), 8)
    ub = BasicUpdateBlock(hidden_dim=128, cor_planes=36, context_dim=64, flow_channel=1, spatial_scale=8)
    with pytest.raises(NndError):
        ub(torch.zeros(1, 128, 4, 8), torch.zeros(1, 64, 4, 8), torch.zeros(1, 36, 4, 8), torch.zeros(1, 1, 4, 8))
    with pytest.raises(NndError):
        ops.corr1d_lookup(torch.zeros(10), torch.zeros(1, 1, 2, 2), 4, 4)


def test_update_block_state_dict_keys_match_reference_order():
    from nndepth_amd.blocks import BasicUpdateBlock
    from nndepth_amd import ops
    ub = BasicUpdateBlock(hidden_dim=128, cor_planes=36, context_dim=64, flow_channel=1, spatial_scale=8)
    assert list(ub.state_dict().keys()) == ops.update_block_keys("sep_conv")
    ub2 = BasicUpdateBlock(hidden_dim=128, cor_planes=36, context_dim=128, gru="conv_gru", flow_channel=2)
    assert list(ub2.state_dict().keys()) == ops.update_block_keys("conv_gru")
    assert tuple(ub2.state_dict()["mask.2.weight"].shape) == (576, 256, 1, 1)


def test_conv_norm_pack_folds_batchnorm_on_host():
    """nnd_conv_pack (HOST): the eval-mode BatchNorm is folded into scale = gamma / sqrt(var + eps) and
    shift = (bias - mean) * scale + beta, stored behind the fragment-ordered weights."""
    from nndepth_amd._lib import lib, ConvDesc
    Cout, Cin = 40, 24
    g = torch.Generator().manual_seed(3)
    w, b = torch.randn(Cout, Cin, 3, 3, generator=g), torch.randn(Cout, generator=g)
    gamma, beta, mean = (torch.randn(Cout, generator=g) for _ in range(3))
    var = torch.rand(Cout, generator=g) + 0.5
    for stride, ci_t in ((1, 32), (2, 16)):
        d = ConvDesc(Cout, Cin, 3, 3, stride)
        n = lib.nnd_conv_packed_floats(C.byref(d))
        ncb, nchunks = 2, -(-Cin // ci_t)
        assert n == ncb * nchunks * 9 * ci_t * 32 + 2 * ncb * 32
        blob = torch.zeros(n)
        P = lambda t: C.c_void_p(t.data_ptr())
        assert lib.nnd_conv_pack(C.byref(d), P(w), P(b), P(gamma), P(beta), P(mean), P(var), 1e-5, P(blob)) == 0
        scale = gamma.double() / torch.sqrt(var.double() + 1e-5)
        shift = (b.double() - mean.double()) * scale + beta.double()
        wf = ncb * nchunks * 9 * ci_t * 32
        assert torch.allclose(blob[wf:wf + Cout].double(), shift, atol=1e-6)
        assert torch.allclose(blob[wf + 64:wf + 64 + Cout].double(), scale, atol=1e-6)
        assert not blob[wf + Cout:wf + 64].any()  # padded channels: shift 0
        # no norm: scale 1, shift = bias
        assert lib.nnd_conv_pack(C.byref(d), P(w), P(b), None, None, None, None, 1e-5, P(blob)) == 0
        assert torch.equal(blob[wf:wf + Cout], b) and bool((blob[wf + 64:wf + 64 + Cout] == 1).all())
    assert lib.nnd_conv_packed_floats(C.byref(ConvDesc(8, 8, 1, 5, 2))) < 0  # 1x5 at stride 2 is not built


def test_encoder_pack_host(raft_sd):
    from nndepth_amd import ops
    from nndepth_amd._lib import lib, NndError
    eng = ops.EncoderEngine(256, "batch", 192)
    assert lib.nnd_encoder_num_tensors(C.byref(eng.desc)) == 6 * (1 + 18 + 1 + 1)
    enc_sd = {k[len("fnet."):]: v for k, v in raft_sd.items() if k.startswith("fnet.")}
    cnet_sd = {k[len("cnet_proj."):]: v for k, v in raft_sd.items() if k.startswith("cnet_proj.")}
    eng.load(enc_sd, cnet_sd, device="cpu")  # host-only packing
    assert eng.packed.numel() == eng.packed_floats and torch.isfinite(eng.packed).all()
    # stem weights: transposed to [k = c*49 + dy*7 + dx][co] with one zero row (K = 148) for the MFMA stem kernel
    stem = eng.packed[:148 * 64].view(148, 64)
    assert torch.equal(stem[:147].t().reshape(64, 3, 7, 7), enc_sd["conv1.weight"]) and not stem[147].any()
    assert lib.nnd_encoder_workspace_floats(C.byref(eng.desc), 2, 544, 960) == 4 * 2 * 64 * 272 * 480
    with pytest.raises(NndError):
        ops.EncoderEngine(256, "group", 0)
    with pytest.raises(NndError):
        ops.EncoderEngine(256, "batch", 192).load(enc_sd, None, device="cpu")


def test_new_entry_points_reject_bad_arguments_without_a_gpu():
    """Argument validation happens before any HIP call: the CRE / encoder / conv entry points must return a negative
    status (never crash) on a CPU-only host."""
    from nndepth_amd._lib import lib, ConvDesc, EncoderDesc, UpdateBlockDesc
    assert lib.nnd_bilinear_sample(None, None, None, 1, 4, 8, 8, 4, 4, None) < 0
    assert lib.nnd_agcl_corr_iter(None, None, None, None, None, 1, 32, 8, 8, 0, None) < 0
    assert lib.nnd_agcl_corr_offset(None, None, None, None, None, 1, 32, 8, 8, 0, None) < 0
    assert lib.nnd_softargmin_disparity(None, None, 1, 8, 4, 4, None) < 0
    assert lib.nnd_igev_init_disparity(None, None, None, None, 1, 8, 4, 4, 4, None) < 0
    assert lib.nnd_igev_interleave_pyramids(None, None, None, 1, 8, 4, 4, 4, None) < 0
    # round 4 (GroupCorrBlock1D / Coarse2Fine cascade)
    assert lib.nnd_group_corr_build_scaled(None, None, None, 1, 16, 4, 8, 4, 4, 1, 4.0, None) < 0
    assert lib.nnd_group_corr1d_lookup(None, None, None, 1, 4, 4, 8, 1, 4, None) < 0
    ug = UpdateBlockDesc(128, 128, 36, 1, 144, 1)
    assert lib.nnd_raft_stereo_group_refine(C.byref(ug), None, None, 4, 1, 4, None, None, None, None, 0, None, None, None, 1, 4, 8, 4, 2,
                                            None) < 0
    assert lib.nnd_igev_interleaved_floats(1, 8, 4, 8, 2) == 4 * 8 * (8 + 4) * 16
    d = ConvDesc(16, 16, 3, 3, 3)
    assert lib.nnd_conv_packed_floats(C.byref(d)) < 0 and b"stride" in lib.nnd_last_error()
    assert lib.nnd_conv_forward(C.byref(ConvDesc(16, 16, 3, 3, 1)), None, None, None, None, 1, 8, 8, 0, 0, None) < 0
    e = EncoderDesc(256, 3, 0)  # group norm is not built
    assert lib.nnd_encoder_packed_floats(C.byref(e)) < 0
    assert lib.nnd_encoder_forward(C.byref(EncoderDesc(256, 1, 0)), None, None, None, None, 0, None, 2, 64, 64, None) < 0
    u = UpdateBlockDesc(128, 128, 36, 2, 576, 0)
    assert lib.nnd_cre_stereo_refine(C.byref(u), None, None, None, 256, None, None, 0, None, None, None, None, 0, None, None, None,
                                     1, 8, 8, 8, 2, None) < 0


def test_loftr_pack_host(cre_sd):
    from nndepth_amd import ops
    from nndepth_amd._lib import lib, NndError
    eng = ops.LoftrEngine(256, 8).load(cre_sd, "self_att_fn.layers.0.", device="cpu")
    assert eng.packed.numel() == eng.packed_floats and torch.isfinite(eng.packed).all()
    C_ = 256
    assert torch.equal(eng.packed[-4 * C_:-3 * C_], cre_sd["self_att_fn.layers.0.norm1.weight"])
    assert torch.equal(eng.packed[-C_:], cre_sd["self_att_fn.layers.0.norm2.bias"])
    assert lib.nnd_loftr_workspace_floats(256, 8, 1, 33, 60) > 6 * 256 * 33 * 60
    with pytest.raises(NndError):
        ops.LoftrEngine(256, 4)  # 64 channels per head: the attention kernels are built for 32
    assert lib.nnd_loftr_layer_forward(256, 8, None, None, None, None, None, 1, 8, 8, None) < 0


def test_models_are_inference_only_and_name_their_fallbacks(raft_sd):
    """No silent PyTorch fallback: training mode raises, an encoder the HIP path does not build raises and
    names the explicit opt-in, and the refine wrappers validate the shapes the C-ABI cannot see."""
    from nndepth_amd._lib import NndError
    from nndepth_amd.raft_stereo import BaseRAFTStereo
    from nndepth_amd.igev_stereo import CostVolumeFilterNetwork
    m = BaseRAFTStereo(iters=1, context_dim=64)
    x = torch.zeros(1, 3, 32, 64)
    with pytest.raises(NndError, match="inference-only"):
        m.train()(x, x)
    m.eval()
    m.fnet.norm_fn = "group"
    with pytest.raises(NndError, match="hip_encoder=False"):
        m(x, x)
    reg = CostVolumeFilterNetwork(8, [40, 80, 160])
    with pytest.raises(NndError, match="inference-only"):
        reg.train()(torch.zeros(1, 8, 8, 8, 8), [])
    with pytest.raises(NndError):  # hip = True and a CPU volume: no CPU path exists
        reg.eval()(torch.zeros(1, 8, 8, 8, 8), [torch.zeros(1, 40, 4, 4), torch.zeros(1, 80, 2, 2), torch.zeros(1, 160, 1, 1)])


@pytest.mark.skipif(not os.path.isdir("/root/reference/nndepth"), reason="needs the reference checkout (build container only)")
def test_patch_swaps_the_seams_of_the_imported_reference_model(raft_sd):
    """`patch()` on an UNMODIFIED instance of the reference's BaseRAFTStereo — the three seams are replaced, the
    update block keeps the reference's state_dict keys (strict load both ways), and the patched seams refuse CPU tensors
    (nothing computes without the HIP device).  Build container only: the reference never travels to the GPU box."""
    import sys
    from oracle.make_golden import _install_standins
    saved = dict(sys.modules), list(sys.path)
    try:
        _install_standins()
        from nndepth.models.raft_stereo.model import BaseRAFTStereo as RefModel
        from nndepth_amd._lib import NndError
        from nndepth_amd.blocks import BasicUpdateBlock
        from nndepth_amd.cost_volume import CorrBlock1D
        from nndepth_amd.raft_stereo import patch
        ref = RefModel(iters=2, context_dim=64).eval()
        ref.load_state_dict(raft_sd, strict=True)
        keys = list(ref.state_dict().keys())
        patch(ref)
        assert isinstance(ref.update_block, BasicUpdateBlock) and ref.corr_fn is CorrBlock1D
        assert list(ref.state_dict().keys()) == keys                      # same keys, same order
        ref.load_state_dict(raft_sd, strict=True)                         # the reference's load_weights path still works
        assert all(torch.equal(ref.state_dict()[k], raft_sd[k]) for k in keys)
        with pytest.raises(NndError):                                     # the reference's own forward now reaches our seams
            ref(torch.zeros(1, 3, 32, 64), torch.zeros(1, 3, 32, 64))
        with pytest.raises(NndError):
            ref.convex_upsample(torch.zeros(1, 1, 4, 8), torch.zeros(1, 576, 4, 8), 8)
    finally:
        for k in set(sys.modules) - set(saved[0]):
            del sys.modules[k]
        sys.path[:] = saved[1]


def test_no_packed_fp32_fma_in_the_update_blocks_valu_convolutions(tmp_path):
    """Round-3 finding (profiles/r03_flow_branch_coresidency.txt): the fused flow-branch kernel and the 2-channel
    flow_head.conv2 kernel returned wrong values when another stream's fp16-MFMA waves shared their CU, and only in builds in
    which the compiler had packed their FMAs (v_pk_fma_f32 with a swapped / broadcast source half).  Round 4: the stand-alone
    reproducer of that instruction pair does not miscompute (scripts/ubench/pk_fma_hazard.hip: 0 of 2 M results for every
    half-selection form, victim and aggressor sharing the SIMDs), so the instruction itself is not the mechanism and the other
    kernels of the library keep their packed code; the two kernels that did fail now issue their FMAs as explicit scalar
    v_fmac_f32 (common.h: fmac_scalar) — a property of the source — and this test holds the built library to it:
    (i) those kernels contain no packed-fp32 arithmetic at all; (ii) the half-selecting form — `op_sel:[...]` or an
    `op_sel_hi` other than [1,1,1] (the round-3 regex missed the second) — is counted per kernel and must be absent from them."""
    import re
    import shutil
    import subprocess
    from nndepth_amd._lib import LIB_PATH
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    lib = tmp_path / "lib.so"
    shutil.copy(LIB_PATH, lib)
    subprocess.run([objdump, "--offloading", str(lib)], cwd=tmp_path, check=True, capture_output=True)
    objs = [f for f in os.listdir(tmp_path) if f.endswith("gfx950")]
    assert objs, "no gfx950 code object in the library"
    seen = set()
    halfsel = {}
    for f in objs:
        dis = subprocess.run([objdump, "-d", str(tmp_path / f)], check=True, capture_output=True, text=True).stdout
        for m in re.finditer(r"^[0-9a-f]+ <(\S+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", dis, re.S | re.M):
            name, body = m.group(1), m.group(2)
            if "flow_branch" in name or "flow_head2" in name:
                seen.add(name)
                packed = re.findall(r"v_pk_(?:fma|mul|add)_f32", body)
                assert not packed, f"{name}: {len(packed)} packed-fp32 instructions"
            bcast = [ln for ln in re.findall(r"v_pk_fma_f32[^\n]*", body)
                     if re.search(r"op_sel:\[", ln) or re.search(r"op_sel_hi:\[(?!1,1,1\])", ln)]
            if bcast:
                halfsel[name] = len(bcast)
            assert not (bcast and ("flow_branch" in name or "flow_head2" in name)), f"{name}: {len(bcast)} half-selecting packed FMAs"
    assert any("flow_branch_kernel" in n for n in seen) and any("flow_head2_kernel" in n for n in seen) and \
        any("flow_branch_lookup_kernel" in n for n in seen), sorted(seen)
    print("kernels with half-selecting v_pk_fma_f32 (accepted outside the update block's VALU convolutions):", halfsel)
