"""Device-side pre- and post-processing: drop-ins for `preprocess_frame`
(nndepth/models/raft_stereo/scripts/inference.py:55-60), `Padder` (nndepth/data/dataloaders/utils.py:5-21) and
`EvalCriterion` (nndepth/models/raft_stereo/scripts/evaluate.py:29-83) that keep the frames and the disparity on the GPU."""
import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from ._lib import NndError, check, lib
from .ops import _dev, _p, _stream


def preprocess_frame(frame: torch.Tensor, HW: Tuple[int, int]) -> torch.Tensor:
    """frame: float (3,h,w) / (B,3,h,w), or the decoded uint8 image (h,w,3) / (B,h,w,3), on the GPU
    -> (B,3,H,W) float: bilinear resize to HW, then (x - 127.5) / 127.5."""
    if frame.device.type != "cuda":
        raise NndError("preprocess_frame: the frame must be on the HIP device (upload the decoded image, then call)")
    u8 = frame.dtype == torch.uint8
    if frame.dim() == 3:
        frame = frame.unsqueeze(0)
    frame = frame.contiguous()
    if u8:
        B, h, w, Cc = frame.shape
    else:
        _dev(frame)
        B, Cc, h, w = frame.shape
    H, W = int(HW[0]), int(HW[1])
    out = torch.empty((B, Cc, H, W), dtype=torch.float32, device=frame.device)
    with torch.cuda.device(frame.device):
        check(lib.nnd_resize_normalize(_p(frame), int(u8), _p(out), B, Cc, h, w, H, W, 127.5, 127.5, _stream(frame.device)),
              "resize_normalize")
    return out


def replicate_pad(x: torch.Tensor, pad) -> torch.Tensor:
    """F.pad(x, (left, right, top, bottom), mode="replicate") for (B,C,H,W) fp32 on the GPU; negative entries crop."""
    d = _dev(x)
    x = x.contiguous()
    B, Cc, H, W = x.shape
    left, right, top, bottom = (int(v) for v in pad)
    out = torch.empty((B, Cc, H + top + bottom, W + left + right), dtype=torch.float32, device=d)
    with torch.cuda.device(d):
        check(lib.nnd_replicate_pad(_p(x), _p(out), B, Cc, H, W, left, right, top, bottom, _stream(d)), "replicate_pad")
    return out


class Padder:
    """Pads images such that dimensions are divisible by `divis_by` (same arithmetic as the reference's Padder)."""

    def __init__(self, HW: Tuple[int, int], divis_by: int = 8):
        self.ht, self.wd = HW
        pad_ht = (((self.ht // divis_by) + 1) * divis_by - self.ht) % divis_by
        pad_wd = (((self.wd // divis_by) + 1) * divis_by - self.wd) % divis_by
        self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, 0, pad_ht]

    def pad(self, *inputs):
        assert all((x.ndim == 4) for x in inputs)
        return [replicate_pad(x, self._pad) for x in inputs]

    def unpad(self, x: torch.Tensor) -> torch.Tensor:
        assert x.ndim == 4
        return replicate_pad(x, [-self._pad[0], -self._pad[1], -self._pad[2], -self._pad[3]])


class EvalCriterion:
    """EPE and dX metrics computed on the device; returns the same dict of floats as the reference (one D2H of a few
    floats instead of the disparity maps).  Inputs of different sizes are first brought to the prediction's size the way
    the reference does it (max-pool of the negated GT, nearest resize) with PyTorch ops on the device."""

    def __init__(self, d_threshold: Optional[Dict[str, float]] = None, max_flow: int = 1000):
        self.d_threshold = d_threshold
        self.max_flow = max_flow

    def __call__(self, disp_gt: torch.Tensor, disp_pred: torch.Tensor, valid_mask: Optional[torch.Tensor] = None):
        d = _dev(disp_gt, disp_pred)
        if disp_pred.shape[-2:] != disp_gt.shape[-2:]:
            scale = disp_gt.shape[-1] // disp_pred.shape[-1]
            gt = -torch.nn.functional.max_pool2d(-disp_gt, kernel_size=scale) / scale
            gt = torch.nn.functional.interpolate(gt, size=disp_pred.shape[-2:])
        else:
            gt = disp_gt
        gt, pred = gt.contiguous(), disp_pred.contiguous()
        B, Cc, H, W = pred.shape
        keys = list(self.d_threshold.keys()) if self.d_threshold else []
        if len(keys) > 4:
            raise NndError("EvalCriterion: at most 4 thresholds per call")
        thr = (C.c_float * max(1, len(keys)))(*[float(self.d_threshold[k]) for k in keys])
        mask = None
        if valid_mask is not None:
            mask = valid_mask.to(device=d, dtype=torch.uint8).contiguous()
            if mask.numel() != B * H * W:
                raise NndError(f"EvalCriterion: valid_mask has {mask.numel()} elements, expected {B * H * W}")
        ws = torch.empty(int(lib.nnd_epe_metrics_workspace_bytes()), dtype=torch.uint8, device=d)
        out = torch.empty(2 + len(keys), dtype=torch.float32, device=d)
        with torch.cuda.device(d):
            check(lib.nnd_epe_metrics(_p(gt), _p(pred), _p(mask), B, Cc, H, W, float(self.max_flow), thr, len(keys), _p(ws), _p(out),
                                      _stream(d)), "epe_metrics")
        vals = out.cpu().tolist()
        metrics = {"epe": vals[0]}
        for i, k in enumerate(keys):
            metrics[k] = vals[2 + i]
        return metrics
