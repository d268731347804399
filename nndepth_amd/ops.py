"""Tensor-level wrappers over the C-ABI (include/nndepth_amd.h).

PyTorch is plumbing only: it owns device memory and the HIP stream; every computation is a
call into libnndepth_amd.so.  All ops require fp32 tensors on a HIP ("cuda") device and
raise otherwise — there is no CPU or eager fallback.
"""
import contextlib
import ctypes as C
import os
import threading
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from ._lib import NND_FLAG_CALIBRATE, Conv3dDesc, ConvDesc, EncoderDesc, NndError, UpdateBlockDesc, check, lib


def _dev(*tensors: torch.Tensor) -> torch.device:
    d = tensors[0].device
    for t in tensors:
        if t.device.type != "cuda":
            raise NndError("nndepth_amd ops run on the HIP device only; got a tensor on "
                           f"{t.device} (no CPU fallback exists)")
        if t.dtype != torch.float32:
            raise NndError(f"nndepth_amd ops are fp32; got {t.dtype}")
        if t.device != d:
            raise NndError("all tensors must live on the same device")
    return d


def _stream(device: torch.device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _p(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


# ------------------------------------------------------------- fp16x2 activation-range calibration
# include/nndepth_amd.h "fp16x2 activation range", csrc/calib.hip.  Inside `with calibration() as c:` every engine call made by
# this thread carries NND_FLAG_CALIBRATE (the layers record the largest |activation| they stage); leaving the block fixes the
# per-layer activation scales of every engine that ran (`nnd_*_calibration_finish`, on the stream, no synchronisation) —
# reading `c.status` afterwards synchronises: bit 0 = a layer saw inf / NaN at its old scale (run the block again), bit 1 = some
# fp16x2 layer of an engine was not on the path.
_calib_tls = threading.local()


class _Calibration:
    def __init__(self):
        self.engines: List[object] = []
        self._status: Optional[torch.Tensor] = None

    def flags(self, engine) -> int:
        """Called by an engine for each C-ABI call it makes: registers the engine, returns the flag word of the descriptor."""
        if all(e is not engine for e in self.engines):
            self.engines.append(engine)
        return NND_FLAG_CALIBRATE

    @property
    def status(self) -> int:
        return 0 if self._status is None else int(self._status.item())


class NeedsCalibration(NndError):
    """Raised inside `with require_calibrated()` by an fp16x2 engine whose activation scales are still the defaults."""


def _calib_flags(engine) -> int:
    c = getattr(_calib_tls, "active", None)
    if c is not None:
        return c.flags(engine)
    if getattr(_calib_tls, "require", False) and not engine.calibrated:
        raise NeedsCalibration(f"{type(engine).__name__}: fp16x2 activation scales not calibrated")
    return 0


@contextlib.contextmanager
def require_calibrated():
    """Inside the block an fp16x2 engine that has not been calibrated since its parameters were packed raises NeedsCalibration
    instead of running with the default activation scales (the model classes use it to calibrate on their first forward)."""
    old = getattr(_calib_tls, "require", False)
    _calib_tls.require = True
    try:
        yield
    finally:
        _calib_tls.require = old


@contextlib.contextmanager
def calibration():
    if getattr(_calib_tls, "active", None) is not None:
        raise NndError("calibration(): already active on this thread")
    c = _Calibration()
    _calib_tls.active = c
    try:
        yield c
    finally:
        _calib_tls.active = None
    for e in c.engines:
        d = e.packed.device
        if c._status is None:
            c._status = torch.zeros(1, dtype=torch.int32, device=d)
        with torch.cuda.device(d):
            e._calibration_finish(c._status)
        e.calibrated = True


# ---------------------------------------------------------------------------- correlation
def pyramid_layout(B: int, H: int, W: int, num_levels: int) -> Tuple[List[int], List[int], int]:
    offs = (C.c_int64 * (num_levels + 1))()
    wid = (C.c_int32 * (num_levels + 1))()
    tot = C.c_int64()
    check(lib.nnd_corr1d_pyramid_layout(B, H, W, num_levels, offs, wid, C.byref(tot)), "corr1d_pyramid_layout")
    return list(offs), list(wid), tot.value


def corr1d_build(fmap1: torch.Tensor, fmap2: torch.Tensor, num_levels: int) -> torch.Tensor:
    """-> flat fp32 pyramid buffer (see pyramid_layout)."""
    d = _dev(fmap1, fmap2)
    if fmap1.shape != fmap2.shape or fmap1.dim() != 4:
        raise NndError(f"corr1d_build: fmap shapes {tuple(fmap1.shape)} vs {tuple(fmap2.shape)}")
    fmap1, fmap2 = fmap1.contiguous(), fmap2.contiguous()
    B, Cc, H, W = fmap1.shape
    _, _, total = pyramid_layout(B, H, W, num_levels)
    pyr = torch.empty(total, dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_corr1d_build(_p(fmap1), _p(fmap2), _p(pyr), B, Cc, H, W, num_levels, _stream(d)), "corr1d_build")
    return pyr


def corr1d_lookup(pyr: torch.Tensor, coords: torch.Tensor, num_levels: int, radius: int) -> torch.Tensor:
    d = _dev(pyr, coords)
    coords = coords.contiguous()
    B, one, H, W = coords.shape
    if one != 1:
        raise NndError("corr1d_lookup: coords must be (B,1,H,W)")
    out = torch.empty((B, num_levels * (2 * radius + 1), H, W), dtype=torch.float32, device=d)
    if out.numel() == 0:
        return out
    with torch.cuda.device(d):
        check(lib.nnd_corr1d_lookup(_p(pyr), _p(coords), _p(out), B, H, W, num_levels, radius, _stream(d)), "corr1d_lookup")
    return out


def convex_upsample(flow: torch.Tensor, mask: torch.Tensor, rate: int) -> torch.Tensor:
    d = _dev(flow, mask)
    flow, mask = flow.contiguous(), mask.contiguous()
    B, Cf, H, W = flow.shape
    if tuple(mask.shape) != (B, 9 * rate * rate, H, W):
        raise NndError(f"convex_upsample: mask shape {tuple(mask.shape)} != {(B, 9 * rate * rate, H, W)}")
    out = torch.empty((B, Cf, rate * H, rate * W), dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_convex_upsample(_p(flow), _p(mask), _p(out), B, Cf, H, W, rate, _stream(d)), "convex_upsample")
    return out


# ------------------------------------------------------------------------ generic conv2d
class Conv2d:
    """One packed stride-1 "same" convolution (1x1, 3x3, 1x5, 5x1): arithmetic "fp32" = the exact fp32-MFMA kernel,
    "bf16x3" = fp32 operands as 3 bf16 pieces on the bf16 MFMA (csrc/conv_split.hip, Cin % 16 == 0)."""

    def __init__(self, weight: torch.Tensor, bias: torch.Tensor, device="cuda", arithmetic: str = "fp32"):
        self.Cout, self.Cin, self.KH, self.KW = (int(s) for s in weight.shape)
        self.arith = UpdateBlockEngine.ARITHMETIC[arithmetic]
        n = int(lib.nnd_conv2d_packed_floats_ex(self.Cout, self.Cin, self.KH, self.KW, self.arith))
        if n <= 0:
            check(n, "conv2d_packed_floats")
        w = weight.detach().to("cpu", torch.float32).contiguous()
        b = bias.detach().to("cpu", torch.float32).contiguous()
        blob = torch.empty(n, dtype=torch.float32)
        check(lib.nnd_conv2d_pack_ex(_p(w), _p(b), self.Cout, self.Cin, self.KH, self.KW, self.arith, _p(blob)), "conv2d_pack")
        self.packed_host = blob
        self.packed = blob.to(device) if device is not None else None
        self.calibrated = False

    def calibrate(self, x: torch.Tensor) -> "Conv2d":
        """fp16x2: set the layer's activation scale from the largest |x| of this input (include/nndepth_amd.h
        "fp16x2 activation range"); the other arithmetics have no range to calibrate."""
        d = _dev(x, self.packed)
        x = x.contiguous()
        B, Cin, H, W = x.shape
        if Cin != self.Cin:
            raise NndError(f"conv2d: input has {Cin} channels, weights expect {self.Cin}")
        with torch.cuda.device(d):
            check(lib.nnd_conv2d_calibrate_ex(_p(self.packed), _p(x), B, Cin, H, W, self.Cout, self.KH, self.KW, self.arith, None,
                                              _stream(d)), "conv2d_calibrate")
        self.calibrated = True
        return self

    def activation_range(self) -> float:
        """Largest |x| the layer represents (fp16x2: 65504 / its activation scale; inf otherwise)."""
        if self.arith != 2:
            return float("inf")
        ncb = (self.Cout + 31) // 32
        n = self.packed.numel()
        return 65504.0 / float(self.packed[n - 4 + 1].item()) if ncb else float("inf")

    def __call__(self, x: torch.Tensor, relu: bool = False) -> torch.Tensor:
        d = _dev(x, self.packed)
        x = x.contiguous()
        B, Cin, H, W = x.shape
        if Cin != self.Cin:
            raise NndError(f"conv2d: input has {Cin} channels, weights expect {self.Cin}")
        if self.arith == 2:
            if getattr(_calib_tls, "active", None) is not None:  # inside `with calibration()`: a single layer calibrates on the spot
                self.calibrate(x)
            else:
                _calib_flags(self)  # raises inside `with require_calibrated()` if the scale is still the default
        y = torch.empty((B, self.Cout, H, W), dtype=torch.float32, device=d)
        with torch.cuda.device(d):
            check(lib.nnd_conv2d_forward_ex(_p(self.packed), _p(x), _p(y), B, Cin, H, W, self.Cout, self.KH, self.KW,
                                            int(relu), self.arith, _stream(d)), "conv2d_forward")
        return y


def conv2d_offset(conv: "Conv2d", x: torch.Tensor, rng: float) -> torch.Tensor:
    """rng * (sigmoid(conv(x)) - 0.5) * 2 in the conv's epilogue (CREStereo search offsets, cre_stereo/model.py:158-159)."""
    d = _dev(x, conv.packed)
    if conv.arith != 0:
        raise NndError("conv2d_offset: exact fp32 packing expected")
    x = x.contiguous()
    B, Cin, H, W = x.shape
    if Cin != conv.Cin:
        raise NndError(f"conv2d_offset: input has {Cin} channels, weights expect {conv.Cin}")
    y = torch.empty((B, conv.Cout, H, W), dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_conv2d_offset_forward(_p(conv.packed), _p(x), _p(y), B, Cin, H, W, conv.Cout, conv.KH, conv.KW, float(rng),
                                            _stream(d)), "conv2d_offset_forward")
    return y


def split_tanh_relu(x: torch.Tensor, c_net: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """net = tanh(x[:, :c_net]), inp = relu(x[:, c_net:]) in one kernel (raft_stereo/model.py:119-122)."""
    d = _dev(x)
    x = x.contiguous()
    B, Cc, H, W = x.shape
    if not 0 < c_net < Cc:
        raise NndError(f"split_tanh_relu: cannot split {Cc} channels at {c_net}")
    net = torch.empty((B, c_net, H, W), dtype=torch.float32, device=d)
    inp = torch.empty((B, Cc - c_net, H, W), dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_split_tanh_relu(_p(x), _p(net), _p(inp), B, c_net, Cc - c_net, H, W, _stream(d)), "split_tanh_relu")
    return net, inp


def pos_enc_sine_add(x0: torch.Tensor, x1: Optional[torch.Tensor] = None, temp_bug_fix: bool = False):
    """x + PositionEncodingSine table (nndepth/blocks/pos_enc.py:22-42, incl. its `/ d_model // 2` precedence quirk unless
    temp_bug_fix), generated on the fly by the kernel; a second map of the same shape gets the same table in the same launch."""
    d = _dev(x0) if x1 is None else _dev(x0, x1)
    x0 = x0.contiguous()
    N, Cc, H, W = x0.shape
    y0 = torch.empty_like(x0)
    y1 = None
    if x1 is not None:
        x1 = x1.contiguous()
        if tuple(x1.shape) != tuple(x0.shape):
            raise NndError(f"pos_enc_sine_add: second map {tuple(x1.shape)} != {tuple(x0.shape)}")
        y1 = torch.empty_like(x1)
    with torch.cuda.device(d):
        check(lib.nnd_pos_enc_sine_add(_p(x0), _p(x1), _p(y0), _p(y1), N, Cc, H, W, int(temp_bug_fix), _stream(d)), "pos_enc_sine_add")
    return y0 if x1 is None else (y0, y1)


def avg_pool_2x_4x(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(F.avg_pool2d(x, 2, stride=2), F.avg_pool2d(x, 4, stride=4)) in one pass (cre_stereo/model.py:154-177)."""
    d = _dev(x)
    x = x.contiguous()
    N, Cc, H, W = x.shape
    o2 = torch.empty((N, Cc, H // 2, W // 2), dtype=torch.float32, device=d)
    o4 = torch.empty((N, Cc, H // 4, W // 4), dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_avg_pool_2x_4x(_p(x), _p(o2), _p(o4), N, Cc, H, W, _stream(d)), "avg_pool_2x_4x")
    return o2, o4


def resize_bilinear_ac(x: torch.Tensor, size: Tuple[int, int], mul: float = 1.0) -> torch.Tensor:
    """mul * F.interpolate(x, size, mode="bilinear", align_corners=True) (cre_stereo/model.py:235-241)."""
    d = _dev(x)
    x = x.contiguous()
    N, Cc, h, w = x.shape
    H, W = int(size[0]), int(size[1])
    y = torch.empty((N, Cc, H, W), dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_resize_bilinear_ac(_p(x), _p(y), N, Cc, h, w, H, W, float(mul), _stream(d)), "resize_bilinear_ac")
    return y


def mask_upsample(conv: "Conv2d", x: torch.Tensor, flow: torch.Tensor, rate: int) -> torch.Tensor:
    """convex_upsample(flow, 0.25 * conv1x1(x)) in one kernel; `conv` = Conv2d packed from mask.2's (9r^2,Cin,1,1)."""
    d = _dev(x, flow, conv.packed)
    x, flow = x.contiguous(), flow.contiguous()
    B, Cin, H, W = x.shape
    out = torch.empty((B, 1, rate * H, rate * W), dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_mask_upsample_forward(_p(conv.packed), _p(x), _p(flow), _p(out), B, Cin, H, W, rate, _stream(d)),
              "mask_upsample_forward")
    return out


# --------------------------------------------------------------------------- update block
# order of the reference module's state_dict (nndepth/blocks/update_block.py:39-55,68-101)
def update_block_keys(gru: str = "sep_conv") -> List[str]:
    names = ["encoder.convc1", "encoder.convc2", "encoder.convf1", "encoder.convf2", "encoder.conv",
             "gru.convz1", "gru.convr1", "gru.convq1"]
    if gru == "sep_conv":
        names += ["gru.convz2", "gru.convr2", "gru.convq2"]
    names += ["flow_head.conv1", "flow_head.conv2", "mask.0", "mask.2"]
    return [f"{n}.{s}" for n in names for s in ("weight", "bias")]


class UpdateBlockEngine:
    """Packed parameters + workspace for one BasicUpdateBlock configuration."""

    ARITHMETIC = {"fp32": 0, "bf16x3": 3, "fp16x2": 2}

    def __init__(self, hidden_dim: int, context_dim: int, cor_planes: int, flow_channels: int,
                 mask_channels: int, gru: str = "sep_conv", arithmetic: str = "fp32"):
        if gru not in ("sep_conv", "conv_gru"):
            raise NndError(f"unknown gru kind {gru!r}")
        if arithmetic not in self.ARITHMETIC:
            raise NndError(f"unknown arithmetic {arithmetic!r} (fp32 = exact fp32 MFMA, bf16x3 = 3-piece split on the bf16 MFMA, "
                           "fp16x2 = 2-piece range-scaled split on the fp16 MFMA)")
        self.gru = gru
        self.arithmetic = arithmetic
        # split_layers: diagnostic subset of the convolutions that take the split arithmetic (bit = index in conv_names();
        # part of the blob layout, so it lives in the descriptor) — NND_SPLIT_MASK is read here, once, by the bisecting scripts
        self.desc = UpdateBlockDesc(hidden_dim, context_dim, cor_planes, flow_channels, mask_channels,
                                    0 if gru == "sep_conv" else 1, self.ARITHMETIC[arithmetic],
                                    int(os.environ.get("NND_SPLIT_MASK", "0"), 0) & 0x7fffffff, 0)
        self.calibrated = False
        n = lib.nnd_update_block_packed_floats(C.byref(self.desc))
        if n <= 0:
            check(int(n), "update_block_packed_floats")
        self.packed_floats = int(n)
        self.packed: Optional[torch.Tensor] = None
        self._ws: Optional[torch.Tensor] = None

    # ---- parameters
    def pack_host(self, state: dict, prefix: str = "") -> torch.Tensor:
        """state: {key: tensor}; returns the packed CPU blob (pure host work, no GPU)."""
        keys = update_block_keys(self.gru)
        assert len(keys) == lib.nnd_update_block_num_tensors(C.byref(self.desc))
        hold = [state[prefix + k].detach().to("cpu", torch.float32).contiguous() for k in keys]
        ptrs = (C.c_void_p * len(hold))(*[t.data_ptr() for t in hold])
        out = torch.empty(self.packed_floats, dtype=torch.float32)
        check(lib.nnd_update_block_pack(C.byref(self.desc), ptrs, _p(out)), "update_block_pack")
        return out

    def load(self, state: dict, prefix: str = "", device="cuda") -> "UpdateBlockEngine":
        self.packed = self.pack_host(state, prefix).to(device)
        self.calibrated = False  # a fresh blob carries the default activation scales
        return self

    # ---- fp16x2 activation range (include/nndepth_amd.h "fp16x2 activation range")
    def _desc(self):
        """The descriptor for one C-ABI call: carries NND_FLAG_CALIBRATE inside `with ops.calibration()`."""
        self.desc.flags = _calib_flags(self) if self.arithmetic == "fp16x2" else 0
        return C.byref(self.desc)

    def _calibration_finish(self, status: Optional[torch.Tensor]) -> None:
        self.desc.flags = 0
        check(lib.nnd_update_block_calibration_finish(C.byref(self.desc), _p(self.packed), _p(status), _stream(self.packed.device)),
              "update_block_calibration_finish")

    def activation_ranges(self) -> Dict[str, float]:
        """{convolution: largest |activation| its fp16x2 operands represent} (65504 / the layer's activation scale)."""
        self.desc.flags = 0
        n = int(lib.nnd_update_block_scale_slots(C.byref(self.desc), None, 0))
        offs = (C.c_int64 * n)()
        check(min(0, int(lib.nnd_update_block_scale_slots(C.byref(self.desc), offs, n))), "update_block_scale_slots")
        blob = self.packed.cpu() if self.packed is not None else None
        out = {}
        for i in range(n):
            if offs[i] >= 0 and blob is not None:
                name = lib.nnd_conv_name(C.byref(self.desc), i).decode()
                out[name or f"conv{i}"] = 65504.0 / float(blob[offs[i] + 1])
        return out

    # ---- workspace
    def workspace(self, B: int, H: int, W: int, device) -> torch.Tensor:
        n = int(lib.nnd_update_block_workspace_floats(C.byref(self.desc), B, H, W))
        if n <= 0:
            check(n, "update_block_workspace_floats")
        if self._ws is None or self._ws.numel() < n or self._ws.device != torch.device(device):
            self._ws = torch.zeros(n, dtype=torch.float32, device=device)
        return self._ws

    def _check_state(self, what: str, net, inp, init, init_channels: int):
        """Shapes the C-ABI cannot see (it receives raw pointers): a wrong one would be an out-of-bounds device access."""
        ds = self.desc
        B, _, H, W = net.shape
        if tuple(net.shape) != (B, ds.hidden_dim, H, W):
            raise NndError(f"{what}: net shape {tuple(net.shape)} != {(B, ds.hidden_dim, H, W)}")
        if tuple(inp.shape) != (B, ds.context_dim, H, W):
            raise NndError(f"{what}: inp shape {tuple(inp.shape)} != {(B, ds.context_dim, H, W)}")
        if init is not None and tuple(init.shape) != (B, init_channels, H, W):
            raise NndError(f"{what}: initial disparity / flow shape {tuple(init.shape)} != {(B, init_channels, H, W)}")

    # ---- ops
    def forward(self, net, inp, corr, flow, want_mask: bool = True):
        if self.packed is None:
            raise NndError("UpdateBlockEngine: parameters not loaded")
        d = _dev(net, inp, corr, flow, self.packed)
        net, inp, corr, flow = (t.contiguous() for t in (net, inp, corr, flow))
        B, _, H, W = net.shape
        ds = self.desc
        exp = {"net": (B, ds.hidden_dim, H, W), "inp": (B, ds.context_dim, H, W),
               "corr": (B, ds.cor_planes, H, W), "flow": (B, ds.flow_channels, H, W)}
        for name, t in (("net", net), ("inp", inp), ("corr", corr), ("flow", flow)):
            if tuple(t.shape) != exp[name]:
                raise NndError(f"update_block: {name} shape {tuple(t.shape)} != {exp[name]}")
        net_out = torch.empty_like(net)
        mask = torch.empty((B, ds.mask_channels, H, W), dtype=torch.float32, device=d) if want_mask else None
        delta = torch.empty_like(flow)
        ws = self.workspace(B, H, W, d)
        with torch.cuda.device(d):
            check(lib.nnd_update_block_forward(self._desc(), _p(self.packed), _p(net), _p(inp), _p(corr), _p(flow),
                                               _p(net_out), _p(mask), _p(delta), _p(ws), B, H, W, _stream(d)),
                  "update_block_forward")
        return net_out, mask, delta

    def refine(self, pyr, num_levels: int, radius: int, net, inp, rate: int, iters: int,
               disp_init=None, keep_all: bool = True):
        """Fused loop -> (up (iters or 1, B,1,rate*H,rate*W), low (B,1,H,W), net (B,hid,H,W))."""
        if self.packed is None:
            raise NndError("UpdateBlockEngine: parameters not loaded")
        d = _dev(pyr, net, inp, self.packed)
        net, inp = net.contiguous(), inp.contiguous()
        B, _, H, W = net.shape
        self._check_state("refine", net, inp, disp_init, 1)
        if pyr.numel() != pyramid_layout(B, H, W, num_levels)[2]:
            raise NndError(f"refine: pyramid holds {pyr.numel()} floats, a {B}x{H}x{W} pyramid of {num_levels} levels has "
                           f"{pyramid_layout(B, H, W, num_levels)[2]}")
        if disp_init is not None:
            _dev(disp_init)
        n_up = iters if keep_all else 1
        up = torch.empty((n_up, B, 1, rate * H, rate * W), dtype=torch.float32, device=d)
        low = torch.empty((B, 1, H, W), dtype=torch.float32, device=d)
        net_out = torch.empty_like(net)
        ws = self.workspace(B, H, W, d)
        stride = up[0].numel() if keep_all else 0
        if disp_init is not None:
            disp_init = disp_init.contiguous()
        with torch.cuda.device(d):
            check(lib.nnd_raft_stereo_refine(self._desc(), _p(self.packed), _p(pyr), num_levels, radius,
                                             _p(net), _p(inp), _p(disp_init), _p(up), stride, _p(low), _p(net_out),
                                             _p(ws), B, H, W, rate, iters, _stream(d)), "raft_stereo_refine")
        return up, low, net_out

    def refine_group(self, group_pyr, num_groups: int, num_levels: int, radius: int, net, inp, rate: int, iters: int,
                     disp_init=None, keep_all: bool = True):
        """One cascade stage of Coarse2FineGroupRepViTRAFTStereo (raft_stereo/model.py:297-311): refine() with GroupCorrBlock1D's
        lookup over the pyramid of raft_group_corr_build -> (up, low, net)."""
        if self.packed is None:
            raise NndError("UpdateBlockEngine: parameters not loaded")
        d = _dev(group_pyr, net, inp, self.packed)
        net, inp = net.contiguous(), inp.contiguous()
        B, _, H, W = net.shape
        self._check_state("refine_group", net, inp, disp_init, 1)
        need = pyramid_layout(B * num_groups, H, W, num_levels)[2]
        if group_pyr.numel() != need:
            raise NndError(f"refine_group: pyramid holds {group_pyr.numel()} floats, expected {need} for B*G={B * num_groups}, "
                           f"{H}x{W}, {num_levels} levels")
        n_up = iters if keep_all else 1
        up = torch.empty((n_up, B, 1, rate * H, rate * W), dtype=torch.float32, device=d)
        low = torch.empty((B, 1, H, W), dtype=torch.float32, device=d)
        net_out = torch.empty_like(net)
        ws = self.workspace(B, H, W, d)
        stride = up[0].numel() if keep_all else 0
        if disp_init is not None:
            _dev(disp_init)
            disp_init = disp_init.contiguous()
        with torch.cuda.device(d):
            check(lib.nnd_raft_stereo_group_refine(self._desc(), _p(self.packed), _p(group_pyr), num_groups, num_levels, radius,
                                                   _p(net), _p(inp), _p(disp_init), _p(up), stride, _p(low), _p(net_out),
                                                   _p(ws), B, H, W, rate, iters, _stream(d)), "raft_stereo_group_refine")
        return up, low, net_out

    def refine_igev(self, feat_pyr, geo_pyr, num_groups: int, num_levels: int, radius: int, net, inp, rate: int,
                    iters: int, disp_init=None, keep_all: bool = True, interleaved=None):
        """IGEV loop (absolute coordinates, combined lookup) -> (up, low, net) like refine().
        interleaved: optional igev_interleave_pyramids(feat_pyr, geo_pyr, ...) — the loop then gathers from it."""
        if self.packed is None:
            raise NndError("UpdateBlockEngine: parameters not loaded")
        d = _dev(feat_pyr, geo_pyr, net, inp, self.packed)
        net, inp = net.contiguous(), inp.contiguous()
        B, _, H, W = net.shape
        self._check_state("refine_igev", net, inp, disp_init, 1)
        need = pyramid_layout(B * num_groups, H, W, num_levels)[2]
        if feat_pyr.numel() != need or geo_pyr.numel() != need:
            raise NndError(f"refine_igev: pyramids hold {feat_pyr.numel()} / {geo_pyr.numel()} floats, expected {need} "
                           f"for B*G={B * num_groups}, {H}x{W}, {num_levels} levels")
        if interleaved is not None:
            _dev(interleaved)
            need_il = int(lib.nnd_igev_interleaved_floats(B, num_groups, H, W, num_levels))
            if interleaved.numel() != need_il:
                raise NndError(f"refine_igev: interleaved copy holds {interleaved.numel()} floats, expected {need_il}")
        if disp_init is not None:
            _dev(disp_init)
        n_up = iters if keep_all else 1
        up = torch.empty((n_up, B, 1, rate * H, rate * W), dtype=torch.float32, device=d)
        low = torch.empty((B, 1, H, W), dtype=torch.float32, device=d)
        net_out = torch.empty_like(net)
        ws = self.workspace(B, H, W, d)
        stride = up[0].numel() if keep_all else 0
        if disp_init is not None:
            disp_init = disp_init.contiguous()
        with torch.cuda.device(d):
            check(lib.nnd_igev_stereo_refine(self._desc(), _p(self.packed), _p(feat_pyr), _p(geo_pyr), _p(interleaved), num_groups,
                                             num_levels, radius, _p(net), _p(inp), _p(disp_init), _p(up), stride, _p(low),
                                             _p(net_out), _p(ws), B, H, W, rate, iters, _stream(d)), "igev_stereo_refine")
        return up, low, net_out

    def refine_cre(self, fmap1, fmap2, net, inp, rate: int, iters: int, flow_init=None, extra_offset=None,
                   scratch=None, keep_all: bool = True):
        """One CREStereo cascade stage (AGCL -> update block -> flow += delta -> 2-channel upsample, `iters` times)
        -> (up (iters or 1, B,2,rate*H,rate*W), flow (B,2,H,W), net).  extra_offset=None: iter mode."""
        if self.packed is None:
            raise NndError("UpdateBlockEngine: parameters not loaded")
        d = _dev(fmap1, fmap2, net, inp, self.packed)
        fmap1, fmap2, net, inp = (t.contiguous() for t in (fmap1, fmap2, net, inp))
        B, Cf, H, W = fmap1.shape
        if fmap2.shape != fmap1.shape or tuple(net.shape[2:]) != (H, W) or net.shape[0] != B:
            raise NndError(f"refine_cre: shapes fmap {tuple(fmap1.shape)} / {tuple(fmap2.shape)}, net {tuple(net.shape)}")
        self._check_state("refine_cre", net, inp, flow_init, 2)
        if extra_offset is not None and extra_offset.numel() != B * 18 * H * W:
            raise NndError(f"refine_cre: extra_offset shape {tuple(extra_offset.shape)} != {(B, 18, H, W)}")
        n_up = iters if keep_all else 1
        up = torch.empty((n_up, B, 2, rate * H, rate * W), dtype=torch.float32, device=d)
        low = torch.empty((B, 2, H, W), dtype=torch.float32, device=d)
        net_out = torch.empty_like(net)
        ws = self.workspace(B, H, W, d)
        stride = up[0].numel() if keep_all else 0
        if flow_init is not None:
            flow_init = flow_init.contiguous()
            _dev(flow_init)
        need = fmap2.numel() if extra_offset is None else 2 * fmap2.numel()  # warped map / the two channels-last copies
        if extra_offset is not None:
            extra_offset = extra_offset.contiguous()
            _dev(extra_offset)
        if scratch is None or scratch.numel() < need:
            scratch = torch.empty(need, dtype=torch.float32, device=d)
        with torch.cuda.device(d):
            check(lib.nnd_cre_stereo_refine(self._desc(), _p(self.packed), _p(fmap1), _p(fmap2), Cf, _p(extra_offset),
                                            _p(scratch), scratch.numel(), _p(net), _p(inp), _p(flow_init), _p(up), stride, _p(low),
                                            _p(net_out), _p(ws), B, H, W, rate, iters, _stream(d)), "cre_stereo_refine")
        return up, low, net_out

    # ---- profiling (bench.py roofline)
    def conv_names(self) -> List[str]:
        n = lib.nnd_num_convs(C.byref(self.desc))
        names = [lib.nnd_conv_name(C.byref(self.desc), i).decode() for i in range(n)]
        if self.gru != "sep_conv":
            names = [x for x in names if not x.endswith("2+convr2") and not x.endswith("convq2")]
        return names

    def profile_conv(self, which: int, B: int, H: int, W: int, reps: int, device) -> Tuple[float, float]:
        """-> (avg ms per launch measured with hipEvents on the launch stream, algorithmic FLOPs)."""
        ws = self.workspace(B, H, W, device)
        ms, fl = C.c_float(), C.c_double()
        d = torch.device(device)
        with torch.cuda.device(d):
            check(lib.nnd_profile_conv(C.byref(self.desc), _p(self.packed), _p(ws), B, H, W, which, reps, _stream(d),
                                       C.byref(ms), C.byref(fl)), "profile_conv")
        return ms.value, fl.value


    def profile_loop_conv(self, which: int, pyr, num_levels: int, radius: int, net, inp, rate: int, iters: int,
                          event_pair_only: bool = False) -> float:
        """-> avg ms of conv `which` INSIDE the fused RAFT-Stereo loop (hipEvents on the launch stream in every iteration).
        event_pair_only: the calibration run — both events in front of the conv, nothing between them."""
        d = _dev(pyr, net, inp, self.packed)
        net, inp = net.contiguous(), inp.contiguous()
        B, _, H, W = net.shape
        self._check_state("profile_loop_conv", net, inp, None, 1)
        up = torch.empty((B, 1, rate * H, rate * W), dtype=torch.float32, device=d)
        ws = self.workspace(B, H, W, d)
        ms = C.c_float()
        with torch.cuda.device(d):
            fn = lib.nnd_profile_loop_event_pair if event_pair_only else lib.nnd_profile_loop_conv
            check(fn(C.byref(self.desc), _p(self.packed), _p(pyr), num_levels, radius, _p(net), _p(inp),
                     _p(up), _p(ws), B, H, W, rate, iters, which, _stream(d), C.byref(ms)), "profile_loop_conv")
        return ms.value


# ------------------------------------------------------------------ IGEV geometry-encoding volume
def group_corr_build(fmap1: torch.Tensor, fmap2: torch.Tensor, num_groups: int, group_channels: int,
                     num_levels: int, pooled: bool = True) -> torch.Tensor:
    """Group-wise 1-D correlation pyramid; layout = pyramid_layout(B*num_groups, H, W, num_levels).
    pooled=False: the buffer has that layout but only level 0 is written (pyramid_pool_levels_ fills the rest on demand)."""
    d = _dev(fmap1, fmap2)
    fmap1, fmap2 = fmap1.contiguous(), fmap2.contiguous()
    B, Ctot, H, W = fmap1.shape
    _, _, total = pyramid_layout(B * num_groups, H, W, num_levels)
    pyr = torch.empty(total, dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_group_corr_build(_p(fmap1), _p(fmap2), _p(pyr), B, Ctot, H, W, num_groups, group_channels,
                                       num_levels if pooled else 0, _stream(d)), "group_corr_build")
    return pyr


def raft_group_corr_build(fmap1: torch.Tensor, fmap2: torch.Tensor, num_groups: int, num_levels: int) -> torch.Tensor:
    """GroupCorrBlock1D.corr + pyramid (raft_stereo/cost_volume.py:84-92,115-128): the first num_groups chunks of num_groups channels,
    divided by sqrt(C_total) (Q4); rows ordered (b,g,h,w1); layout = pyramid_layout(B*num_groups, H, W, num_levels)."""
    d = _dev(fmap1, fmap2)
    fmap1, fmap2 = fmap1.contiguous(), fmap2.contiguous()
    B, Ctot, H, W = fmap1.shape
    if num_groups * num_groups > Ctot:
        raise NndError(f"raft_group_corr_build: {num_groups} chunks of {num_groups} channels exceed the {Ctot} channels of the maps")
    _, _, total = pyramid_layout(B * num_groups, H, W, num_levels)
    pyr = torch.empty(total, dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_group_corr_build_scaled(_p(fmap1), _p(fmap2), _p(pyr), B, Ctot, H, W, num_groups, num_groups, num_levels,
                                              float(Ctot) ** 0.5, _stream(d)), "group_corr_build_scaled")
    return pyr


def group_corr1d_lookup(pyr: torch.Tensor, coords: torch.Tensor, num_groups: int, num_levels: int, radius: int) -> torch.Tensor:
    """GroupCorrBlock1D.__call__ (raft_stereo/cost_volume.py:94-113, view without the group permute included) ->
    (B, num_levels*num_groups*(2r+1), H, W)."""
    d = _dev(pyr, coords)
    coords = coords.contiguous()
    B, one, H, W = coords.shape
    if one != 1:
        raise NndError("group_corr1d_lookup: coords must be (B,1,H,W)")
    if pyr.numel() != pyramid_layout(B * num_groups, H, W, num_levels)[2]:
        raise NndError(f"group_corr1d_lookup: pyramid holds {pyr.numel()} floats, expected "
                       f"{pyramid_layout(B * num_groups, H, W, num_levels)[2]} for B*G={B * num_groups}, {H}x{W}, {num_levels} levels")
    out = torch.empty((B, num_levels * num_groups * (2 * radius + 1), H, W), dtype=torch.float32, device=d)
    if out.numel() == 0:
        return out
    with torch.cuda.device(d):
        check(lib.nnd_group_corr1d_lookup(_p(pyr), _p(coords), _p(out), B, num_groups, H, W, num_levels, radius, _stream(d)),
              "group_corr1d_lookup")
    return out


def pyramid_from_level0(level0: torch.Tensor, B: int, H: int, W: int, num_levels: int) -> torch.Tensor:
    """level0: (B*H*W, W) rows -> full avg-pool pyramid buffer (levels 1..num_levels built on the device)."""
    d = _dev(level0)
    offs, widths, total = pyramid_layout(B, H, W, num_levels)
    pyr = torch.empty(total, dtype=torch.float32, device=d)
    pyr[:B * H * W * W].copy_(level0.reshape(-1))
    with torch.cuda.device(d):
        check(lib.nnd_pyramid_from_level0(_p(pyr), B, H, W, num_levels, _stream(d)), "pyramid_from_level0")
    return pyr


def pyramid_pool_levels_(pyr: torch.Tensor, B: int, H: int, W: int, num_levels: int) -> torch.Tensor:
    """Levels 1..num_levels of a pyramid buffer whose level 0 is already in place (in-place variant of pyramid_from_level0)."""
    d = _dev(pyr)
    with torch.cuda.device(d):
        check(lib.nnd_pyramid_from_level0(_p(pyr), B, H, W, num_levels, _stream(d)), "pyramid_from_level0")
    return pyr


def igev_lookup(feat_pyr: torch.Tensor, geo_pyr: torch.Tensor, coords: torch.Tensor, num_groups: int,
                num_levels: int, radius: int) -> torch.Tensor:
    d = _dev(feat_pyr, geo_pyr, coords)
    coords = coords.contiguous()
    B, one, H, W = coords.shape
    out = torch.empty((B, num_levels * 2 * num_groups * (2 * radius + 1), H, W), dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_igev_lookup(_p(feat_pyr), _p(geo_pyr), _p(coords), _p(out), B, num_groups, H, W, num_levels,
                                  radius, _stream(d)), "igev_lookup")
    return out


def igev_interleave_pyramids(feat_pyr: torch.Tensor, geo_pyr: torch.Tensor, B: int, num_groups: int, H: int, W: int,
                             num_levels: int) -> torch.Tensor:
    """Group-interleaved copy of both pyramids (levels 0..num_levels-1) for the refinement loop's gathers."""
    d = _dev(feat_pyr, geo_pyr)
    out = torch.empty(lib.nnd_igev_interleaved_floats(B, num_groups, H, W, num_levels), dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_igev_interleave_pyramids(_p(feat_pyr), _p(geo_pyr), _p(out), B, num_groups, H, W, num_levels,
                                               _stream(d)), "igev_interleave_pyramids")
    return out


def igev_refine_reads_interleaved(num_groups: int, num_levels: int, radius: int) -> bool:
    """True when nnd_igev_stereo_refine, given an interleaved copy, never reads the plain pyramids (their pooled levels
    need not exist)."""
    return bool(lib.nnd_igev_refine_reads_interleaved(num_groups, num_levels, radius))


def igev_interleave_level0_supported(num_groups: int, W: int, num_levels: int) -> bool:
    return bool(lib.nnd_igev_interleave_level0_supported(num_groups, W, num_levels))


def igev_interleave_level0(feat_level0: torch.Tensor, geo_level0: torch.Tensor, B: int, num_groups: int, H: int, W: int,
                           num_levels: int) -> torch.Tensor:
    """The interleaved levels 0..num_levels-1 from the two level-0 volumes (pooling in LDS; same values as pooling both
    pyramids and igev_interleave_pyramids).  The tensors may be whole pyramid buffers: only their level 0 is read."""
    d = _dev(feat_level0, geo_level0)
    n0 = B * num_groups * H * W * W
    assert feat_level0.numel() >= n0 and geo_level0.numel() >= n0 and feat_level0.is_contiguous() and geo_level0.is_contiguous()
    out = torch.empty(lib.nnd_igev_interleaved_floats(B, num_groups, H, W, num_levels), dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_igev_interleave_level0(_p(feat_level0), _p(geo_level0), _p(out), B, num_groups, H, W, num_levels,
                                             _stream(d)), "igev_interleave_level0")
    return out


# ------------------------------------------------------------------ CREStereo AGCL (include/nndepth_amd.h)
def bilinear_sample(img: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """img (N,C,H,W), coords (N,Hg,Wg,2) pixel (x,y) -> (N,C,Hg,Wg); zero outside (cre_stereo/utils.py:5-20)."""
    d = _dev(img, coords)
    img, coords = img.contiguous(), coords.contiguous()
    N, C, H, W = img.shape
    if coords.dim() != 4 or coords.shape[0] != N or coords.shape[3] != 2:
        raise NndError(f"bilinear_sample: coords {tuple(coords.shape)} must be (N={N}, Hg, Wg, 2)")
    Hg, Wg = coords.shape[1], coords.shape[2]
    out = torch.empty((N, C, Hg, Wg), dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_bilinear_sample(_p(img), _p(coords), _p(out), N, C, H, W, Hg, Wg, _stream(d)), "bilinear_sample")
    return out


def agcl_corr_iter(fmap1: torch.Tensor, fmap2: torch.Tensor, flow: torch.Tensor, small_patch: bool,
                   scratch: torch.Tensor = None) -> torch.Tensor:
    """cre_stereo/cost_volume.py:51-79.  scratch: optional (N,C,H,W) buffer for the warped right features."""
    d = _dev(fmap1, fmap2, flow)
    fmap1, fmap2, flow = fmap1.contiguous(), fmap2.contiguous(), flow.contiguous()
    N, C, H, W = fmap1.shape
    if fmap2.shape != fmap1.shape or tuple(flow.shape) != (N, 2, H, W):
        raise NndError(f"agcl_corr_iter: shapes {tuple(fmap1.shape)}, {tuple(fmap2.shape)}, flow {tuple(flow.shape)}")
    if scratch is None or scratch.numel() < fmap2.numel():
        scratch = torch.empty_like(fmap2)
    out = torch.empty((N, 36, H, W), dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_agcl_corr_iter(_p(fmap1), _p(fmap2), _p(flow), _p(scratch), _p(out), N, C, H, W,
                                     int(bool(small_patch)), _stream(d)), "agcl_corr_iter")
    return out


def nchw_to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """(N,C,H,W) -> contiguous (N,H,W,C) copy (the channels-last maps of agcl_corr_offset)."""
    d = _dev(x)
    x = x.contiguous()
    N, C, H, W = x.shape
    out = torch.empty((N, H, W, C), dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_nchw_to_nhwc(_p(x), _p(out), N, C, H, W, _stream(d)), "nchw_to_nhwc")
    return out


def agcl_corr_offset(fmap1: torch.Tensor, fmap2: torch.Tensor, flow: torch.Tensor, extra_offset: torch.Tensor,
                     small_patch: bool, channels_last: bool = False) -> torch.Tensor:
    """cre_stereo/cost_volume.py:81-154 after the optional attention.  channels_last: fmap1 / fmap2 are (N,H,W,C) copies
    made by nchw_to_nhwc (C = 256): the line-per-tap kernel."""
    d = _dev(fmap1, fmap2, flow, extra_offset)
    fmap1, fmap2, flow, extra_offset = (t.contiguous() for t in (fmap1, fmap2, flow, extra_offset))
    if channels_last:
        N, H, W, C = fmap1.shape
    else:
        N, C, H, W = fmap1.shape
    if fmap2.shape != fmap1.shape or tuple(flow.shape) != (N, 2, H, W) or extra_offset.numel() != N * 18 * H * W:
        raise NndError(f"agcl_corr_offset: shapes {tuple(fmap1.shape)}, {tuple(fmap2.shape)}, flow {tuple(flow.shape)}, "
                       f"extra_offset {tuple(extra_offset.shape)}")
    out = torch.empty((N, 36, H, W), dtype=torch.float32, device=d)
    fn = lib.nnd_agcl_corr_offset_nhwc if channels_last else lib.nnd_agcl_corr_offset
    with torch.cuda.device(d):
        check(fn(_p(fmap1), _p(fmap2), _p(flow), _p(extra_offset), _p(out), N, C, H, W, int(bool(small_patch)), _stream(d)),
              "agcl_corr_offset")
    return out


# ------------------------------------------------------------------ conv + folded norm, encoder (include/nndepth_amd.h)
def _host(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.detach().to("cpu", torch.float32).contiguous()


class ConvNorm:
    """nn.Conv2d [+ BatchNorm2d(eval)] [+ ReLU] [+ residual add + ReLU] as one MFMA convolution (stride 1 or 2,
    1x1 / 3x3; 1x5 / 5x1 at stride 1).  `bn` = (weight, bias, running_mean, running_var) or None."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int = 1, bn=None, eps: float = 1e-5,
                 device="cuda"):
        Cout, Cin, KH, KW = (int(v) for v in weight.shape)
        self.desc = ConvDesc(Cout, Cin, KH, KW, int(stride))
        n = int(lib.nnd_conv_packed_floats(C.byref(self.desc)))
        if n <= 0:
            check(n, "conv_packed_floats")
        w, b = _host(weight), _host(bias)
        g, be, m, v = (_host(t) for t in bn) if bn is not None else (None, None, None, None)
        blob = torch.empty(n, dtype=torch.float32)
        check(lib.nnd_conv_pack(C.byref(self.desc), _p(w), _p(b), _p(g), _p(be), _p(m), _p(v), float(eps), _p(blob)), "conv_pack")
        self.packed = blob.to(device)

    def __call__(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None, relu: bool = False,
                 relu_after_residual: bool = False) -> torch.Tensor:
        d = _dev(x, self.packed)
        x = x.contiguous()
        B, Cin, H, W = x.shape
        if Cin != self.desc.Cin:
            raise NndError(f"conv: input has {Cin} channels, weights expect {self.desc.Cin}")
        st = self.desc.stride
        Ho, Wo = (H + st - 1) // st, (W + st - 1) // st
        y = torch.empty((B, self.desc.Cout, Ho, Wo), dtype=torch.float32, device=d)
        if residual is not None:
            residual = residual.contiguous()
            _dev(residual)
            if residual.shape != y.shape:
                raise NndError(f"conv: residual {tuple(residual.shape)} != output {tuple(y.shape)}")
        with torch.cuda.device(d):
            check(lib.nnd_conv_forward(C.byref(self.desc), _p(self.packed), _p(x), _p(residual), _p(y), B, H, W,
                                       int(relu), int(relu_after_residual), _stream(d)), "conv_forward")
        return y


ENCODER_BLOCKS = ("layer1.0", "layer1.1", "layer2.0", "layer2.1", "layer3.0", "layer3.1")


class EncoderEngine:
    """BasicEncoder (+ optional cnet_proj) on the HIP encoder (csrc/encoder.hip).  Parameters come from state-dict
    style mappings: `enc_sd` with the reference's BasicEncoder keys (conv1.weight, norm1.running_mean,
    layer1.0.conv1.weight, layer1.0.downsample.0.weight, layer1.0.norm3.weight, ..., conv2.bias) and, optionally,
    `cnet_sd` with `0.weight` / `0.bias` of the cnet_proj Sequential."""

    def __init__(self, output_dim: int, norm: str = "batch", cnet_dim: int = 0, arithmetic: str = "fp32"):
        if norm not in ("batch", "none", "instance"):
            raise NndError(f"EncoderEngine: norm_fn '{norm}' is not built in HIP (batch in eval mode, instance, none)")
        self.norm = norm
        self.arithmetic = arithmetic
        self.desc = EncoderDesc(int(output_dim), {"none": 0, "batch": 1, "instance": 2}[norm], int(cnet_dim),
                                UpdateBlockEngine.ARITHMETIC[arithmetic], 0)
        self.calibrated = False
        n = int(lib.nnd_encoder_packed_floats(C.byref(self.desc)))
        if n <= 0:
            check(n, "encoder_packed_floats")
        self.packed_floats = n
        self.packed = None
        self._ws = None

    def _units(self, enc_sd, cnet_sd):
        def unit(conv: str, norm: Optional[str]):
            t = [enc_sd[conv + ".weight"], enc_sd[conv + ".bias"]]
            if norm is not None and self.norm == "batch":
                t += [enc_sd[norm + ".weight"], enc_sd[norm + ".bias"], enc_sd[norm + ".running_mean"], enc_sd[norm + ".running_var"]]
            else:
                t += [None] * 4
            return t
        units = [unit("conv1", "norm1")]
        for b in ENCODER_BLOCKS:
            units += [unit(f"{b}.conv1", f"{b}.norm1"), unit(f"{b}.conv2", f"{b}.norm2"), unit(f"{b}.downsample.0", f"{b}.norm3")]
        units.append(unit("conv2", None))
        if self.desc.cnet_dim > 0:
            units.append([cnet_sd["0.weight"], cnet_sd["0.bias"], None, None, None, None])
        return units

    def load(self, enc_sd, cnet_sd=None, eps: float = 1e-5, device="cuda") -> "EncoderEngine":
        if self.desc.cnet_dim > 0 and cnet_sd is None:
            raise NndError("EncoderEngine: cnet_dim > 0 needs the cnet_proj parameters")
        host = [_host(t) for u in self._units(enc_sd, cnet_sd) for t in u]
        n = int(lib.nnd_encoder_num_tensors(C.byref(self.desc)))
        if n != len(host):
            raise NndError(f"EncoderEngine: {len(host)} tensors, the library expects {n}")
        arr = (C.c_void_p * n)(*[0 if t is None else t.data_ptr() for t in host])
        blob = torch.empty(self.packed_floats, dtype=torch.float32)
        check(lib.nnd_encoder_pack(C.byref(self.desc), arr, float(eps), _p(blob)), "encoder_pack")
        self.packed = blob.to(device)
        self.calibrated = False
        return self

    def _calibration_finish(self, status: Optional[torch.Tensor]) -> None:
        self.desc.flags = 0
        check(lib.nnd_encoder_calibration_finish(C.byref(self.desc), _p(self.packed), _p(status), _stream(self.packed.device)),
              "encoder_calibration_finish")

    def forward(self, frames: torch.Tensor, n_cnet: int = 0, frames_b: Optional[torch.Tensor] = None):
        """frames (N,3,H,W) -> (fmap (N,output_dim,H/8,W/8), cnet (n_cnet,cnet_dim,H/8,W/8) or None).  With `frames_b` (same shape)
        the batch is [frames | frames_b] read where the two tensors lie (no torch.cat copy): fmap has 2N samples."""
        if self.packed is None:
            raise NndError("EncoderEngine: parameters not loaded")
        d = _dev(frames, self.packed)
        frames = frames.contiguous()
        N, c, H, W = frames.shape
        if c != 3:
            raise NndError(f"encoder: frames have {c} channels, expected 3")
        nsplit = N
        if frames_b is not None:
            _dev(frames_b, self.packed)
            frames_b = frames_b.contiguous()
            if tuple(frames_b.shape) != tuple(frames.shape):
                raise NndError(f"encoder: second frame tensor {tuple(frames_b.shape)} != {tuple(frames.shape)}")
            N = 2 * N
        h8, w8 = H, W
        for _ in range(3):
            h8, w8 = (h8 + 1) // 2, (w8 + 1) // 2
        fmap = torch.empty((N, self.desc.output_dim, h8, w8), dtype=torch.float32, device=d)
        cnet = torch.empty((n_cnet, self.desc.cnet_dim, h8, w8), dtype=torch.float32, device=d) if n_cnet > 0 else None
        need = int(lib.nnd_encoder_workspace_floats(C.byref(self.desc), N, H, W))
        if self._ws is None or self._ws.numel() < need or self._ws.device != d:
            self._ws = torch.empty(need, dtype=torch.float32, device=d)
        self.desc.flags = _calib_flags(self) if self.arithmetic == "fp16x2" else 0
        with torch.cuda.device(d):
            check(lib.nnd_encoder_forward2(C.byref(self.desc), _p(self.packed), _p(frames), _p(frames_b), nsplit, _p(fmap), _p(cnet), n_cnet,
                                           _p(self._ws), N, H, W, _stream(d)), "encoder_forward")
        return fmap, cnet


def softargmin_disparity(logits: torch.Tensor) -> torch.Tensor:
    """IGEV initial disparity: logits (B,D,H,W) -> -sum_d d * softmax_d (B,1,H,W) (igev_stereo/model.py:92-95,145-146)."""
    d = _dev(logits)
    logits = logits.contiguous()
    B, D, H, W = logits.shape
    out = torch.empty((B, 1, H, W), dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_softargmin_disparity(_p(logits), _p(out), B, D, H, W, _stream(d)), "softargmin_disparity")
    return out


def igev_init_disparity_supported(num_groups: int, D: int) -> bool:
    return num_groups <= 8 and D <= 512


def igev_init_disparity(geo_level0: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], B: int, G: int, H: int,
                        W: int, D: int) -> torch.Tensor:
    """cv_squeezer Conv3d(G,1,3,1,1) + soft-argmin in one kernel (igev_stereo/model.py:144-146): geo_level0 = rows
    (b,g,h,w1) of D candidates on the device, weight (1,G,3,3,3) / bias (1) = the squeezer's parameters -> (B,1,H,W)."""
    d = _dev(geo_level0)
    assert geo_level0.numel() == B * G * H * W * D and tuple(weight.shape) == (1, G, 3, 3, 3)
    # 27*G floats passed by value to the kernel: hand in host tensors (a device tensor costs a synchronising copy here)
    wh = weight.detach().to("cpu", torch.float32).contiguous()
    bh = None if bias is None else bias.detach().to("cpu", torch.float32).contiguous()
    out = torch.empty((B, 1, H, W), dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_igev_init_disparity(_p(geo_level0.contiguous()), _p(wh), _p(bh), _p(out), B, G, H, W, D, _stream(d)),
              "igev_init_disparity")
    return out


LOFTR_KEYS = ("q_proj.weight", "k_proj.weight", "v_proj.weight", "merge.weight", "mlp.0.weight", "mlp.2.weight",
              "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")


class LoftrEngine:
    """One LoFTR encoder layer (linear attention) on (N, d_model, H, W) maps (csrc/loftr.hip)."""

    def __init__(self, d_model: int, nhead: int):
        self.d_model, self.nhead = int(d_model), int(nhead)
        n = int(lib.nnd_loftr_packed_floats(self.d_model, self.nhead))
        if n <= 0:
            check(n, "loftr_packed_floats")
        self.packed_floats = n
        self.packed = None
        self._ws = None

    def load(self, sd, prefix: str = "", device="cuda") -> "LoftrEngine":
        host = [_host(sd[prefix + k]) for k in LOFTR_KEYS]
        arr = (C.c_void_p * len(host))(*[t.data_ptr() for t in host])
        blob = torch.empty(self.packed_floats, dtype=torch.float32)
        check(lib.nnd_loftr_pack(self.d_model, self.nhead, arr, _p(blob)), "loftr_pack")
        self.packed = blob.to(device)
        return self

    def forward(self, x: torch.Tensor, source: torch.Tensor) -> torch.Tensor:
        if self.packed is None:
            raise NndError("LoftrEngine: parameters not loaded")
        d = _dev(x, source, self.packed)
        x, source = x.contiguous(), source.contiguous()
        N, Cc, H, W = x.shape
        if Cc != self.d_model or source.shape != x.shape:
            raise NndError(f"loftr: x {tuple(x.shape)} / source {tuple(source.shape)} must both be (N, {self.d_model}, H, W)")
        need = int(lib.nnd_loftr_workspace_floats(self.d_model, self.nhead, N, H, W))
        if self._ws is None or self._ws.numel() < need or self._ws.device != d:
            self._ws = torch.empty(need, dtype=torch.float32, device=d)
        out = torch.empty_like(x)
        with torch.cuda.device(d):
            check(lib.nnd_loftr_layer_forward(self.d_model, self.nhead, _p(self.packed), _p(x), _p(source), _p(out), _p(self._ws),
                                              N, H, W, _stream(d)), "loftr_layer_forward")
        return out


# ------------------------------------------------------------------ Conv3d on depth-major volumes (IGEV regulariser, a15)
def volume_to_depth_major(x: torch.Tensor) -> torch.Tensor:
    """(N,C,D,H,W) -> (N,D+2,C,H,W) with zero end slices."""
    d = _dev(x)
    x = x.contiguous()
    N, Cc, D, H, W = x.shape
    y = torch.empty((N, D + 2, Cc, H, W), dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_volume_to_depth_major(_p(x), _p(y), N, Cc, D, H, W, _stream(d)), "volume_to_depth_major")
    return y


def depth_major_to_volume(x: torch.Tensor) -> torch.Tensor:
    """(N,D+2,C,H,W) -> (N,C,D,H,W)."""
    d = _dev(x)
    x = x.contiguous()
    N, Dp, Cc, H, W = x.shape
    y = torch.empty((N, Cc, Dp - 2, H, W), dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_depth_major_to_volume(_p(x), _p(y), N, Cc, Dp - 2, H, W, _stream(d)), "depth_major_to_volume")
    return y


def volume_rows_to_depth_major(x: torch.Tensor) -> torch.Tensor:
    """(N,C,H,W,D) (candidate axis contiguous: the pyramid rows) -> (N,D+2,C,H,W) with zero end slices."""
    d = _dev(x)
    if not x.is_contiguous():
        raise NndError("volume_rows_to_depth_major: contiguous (N,C,H,W,D) expected")
    N, Cc, H, W, D = x.shape
    y = torch.empty((N, D + 2, Cc, H, W), dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_volume_rows_to_depth_major(_p(x), _p(y), N, Cc, D, H, W, _stream(d)), "volume_rows_to_depth_major")
    return y


def depth_major_to_volume_rows(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(N,D+2,C,H,W) -> (N,C,H,W,D), optionally into `out` (e.g. level 0 of a pyramid buffer)."""
    d = _dev(x)
    x = x.contiguous()
    N, Dp, Cc, H, W = x.shape
    if out is None:
        out = torch.empty((N, Cc, H, W, Dp - 2), dtype=torch.float32, device=d)
    if not out.is_contiguous() or out.numel() != N * Cc * H * W * (Dp - 2) or out.dtype != torch.float32 or out.device != d:
        raise NndError("depth_major_to_volume_rows: `out` must be a contiguous fp32 (N,C,H,W,D) tensor on the same device")
    with torch.cuda.device(d):
        check(lib.nnd_depth_major_to_volume_rows(_p(x), _p(out), N, Cc, Dp - 2, H, W, _stream(d)), "depth_major_to_volume_rows")
    return out


class Conv3dNorm:
    """nn.Conv3d(k=3, padding=1, stride 1|2) [+ BatchNorm3d(eval)] [+ LeakyReLU] on depth-major volumes; the input may be the
    channel concat of two volumes.  `bn` = (weight, bias, running_mean, running_var) or None."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int = 1, bn=None, eps: float = 1e-5,
                 leaky_slope: float = 1.0, split: int = 0, device="cuda", arithmetic: str = "fp32"):
        Cout, Cin = int(weight.shape[0]), int(weight.shape[1])
        if tuple(weight.shape[2:]) != (3, 3, 3):
            raise NndError("Conv3dNorm: only 3x3x3 kernels")
        cin0 = split if split > 0 else Cin
        self.arithmetic = arithmetic
        self.desc = Conv3dDesc(Cout, cin0, Cin - cin0, int(stride), UpdateBlockEngine.ARITHMETIC[arithmetic], 0)
        self.calibrated = False
        self.leaky = float(leaky_slope)
        n = int(lib.nnd_conv3d_packed_floats(C.byref(self.desc)))
        if n <= 0:
            check(n, "conv3d_packed_floats")
        w, b = _host(weight), _host(bias)
        g, be, m, v = (_host(t) for t in bn) if bn is not None else (None, None, None, None)
        blob = torch.empty(n, dtype=torch.float32)
        check(lib.nnd_conv3d_pack(C.byref(self.desc), _p(w), _p(b), _p(g), _p(be), _p(m), _p(v), float(eps), _p(blob)), "conv3d_pack")
        self.packed = blob.to(device)

    def _calibration_finish(self, status: Optional[torch.Tensor]) -> None:
        self.desc.flags = 0
        check(lib.nnd_conv3d_calibration_finish(C.byref(self.desc), _p(self.packed), _p(status), _stream(self.packed.device)),
              "conv3d_calibration_finish")

    def __call__(self, x0: torch.Tensor, x1: Optional[torch.Tensor] = None) -> torch.Tensor:
        d = _dev(x0, self.packed)
        x0 = x0.contiguous()
        N, Dp, c0, H, W = x0.shape
        if c0 != self.desc.Cin0 or (self.desc.Cin1 > 0) != (x1 is not None):
            raise NndError(f"conv3d: inputs do not match the layer ({c0} vs {self.desc.Cin0} channels, second input {x1 is not None})")
        if x1 is not None:
            x1 = x1.contiguous()
            _dev(x1)
            if tuple(x1.shape) != (N, Dp, self.desc.Cin1, H, W):
                raise NndError(f"conv3d: second input {tuple(x1.shape)} != {(N, Dp, self.desc.Cin1, H, W)}")
        st, D = self.desc.stride, Dp - 2
        Do, Ho, Wo = (D + st - 1) // st, (H + st - 1) // st, (W + st - 1) // st
        y = torch.empty((N, Do + 2, self.desc.Cout, Ho, Wo), dtype=torch.float32, device=d)
        self.desc.flags = _calib_flags(self) if self.arithmetic == "fp16x2" else 0
        with torch.cuda.device(d):
            check(lib.nnd_conv3d_forward(C.byref(self.desc), _p(self.packed), _p(x0), _p(x1), _p(y), N, D, H, W, self.leaky,
                                         _stream(d)), "conv3d_forward")
        return y


def volume_upsample2x(x: torch.Tensor) -> torch.Tensor:
    """Trilinear x2 (align_corners=True) of a depth-major volume (N,D+2,C,H,W) -> (N,2D+2,C,2H,2W)."""
    d = _dev(x)
    x = x.contiguous()
    N, Dp, Cc, H, W = x.shape
    y = torch.empty((N, 2 * (Dp - 2) + 2, Cc, 2 * H, 2 * W), dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_volume_upsample2x(_p(x), _p(y), N, Cc, Dp - 2, H, W, _stream(d)), "volume_upsample2x")
    return y


def volume_gate_(vol: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
    """vol (N,D+2,C,H,W) *= sigmoid(logits (N,C,H,W)) in place (FeatureGuidedBlock)."""
    d = _dev(vol, logits)
    logits = logits.contiguous()
    N, Dp, Cc, H, W = vol.shape
    if not vol.is_contiguous() or tuple(logits.shape) != (N, Cc, H, W):
        raise NndError(f"volume_gate: vol {tuple(vol.shape)} / logits {tuple(logits.shape)}")
    with torch.cuda.device(d):
        check(lib.nnd_volume_gate(_p(vol), _p(logits), N, Cc, Dp - 2, H, W, _stream(d)), "volume_gate")
    return vol
