"""HIP-graph replay of a whole forward pass: ONE host-side launch per stereo pair.

Every kernel of the RAFT-Stereo / IGEV / CREStereo forwards is enqueued through the C-ABI on torch's current stream with no
host synchronisation and no allocation inside the library, so a forward can be captured into a HIP graph once per input shape
and replayed.  What that buys is independence from the host: a CREStereo 1080x1920 pair is ~600 kernel launches (8-10 ms of
host-side launch work, three C-ABI calls for the cascade stages plus the encoder / attention calls) in front of ~28 ms of GPU
work — on a busy or slow host core the GPU starves (round 2 saw medians of 60-100 ms on such boxes); the
replay is one hipGraphLaunch.  On an idle host the direct launches already keep the GPU fed (launch gaps ~2 % of a RAFT-Stereo
pair), so the replay is not faster there; it is never slower.

    fwd = GraphedForward(model)              # model: BaseRAFTStereo / CREStereoBase / IGEVStereoBase ... in eval mode, on a GPU
    outs = fwd(frame1, frame2)               # first call per input shape: warm-up + capture; then: copy inputs, replay
    outs[-1]["up_disp"]                      # tensors owned by the graph: valid until the next call with the same shape

The captured forward reads the model's parameters where they lie (packed blobs on the device): after changing parameters call
`fwd.reset()`.  Nothing here touches the arithmetic: the same kernels run in the same order (tests/test_gpu_graph.py requires
bit-identical outputs).
"""
from typing import Callable, Dict, Tuple

import torch


class GraphedForward:
    def __init__(self, forward: Callable, warmup: int = 2):
        self.forward = forward
        self.warmup = int(warmup)
        self._graphs: Dict[Tuple, Tuple] = {}

    def reset(self) -> None:
        self._graphs.clear()

    def _key(self, args) -> Tuple:
        return tuple((tuple(a.shape), a.dtype, str(a.device)) for a in args)

    def __call__(self, *args: torch.Tensor):
        for a in args:
            if not (torch.is_tensor(a) and a.is_cuda):
                raise TypeError("GraphedForward takes device tensors (the frames of a pair already resident in HBM)")
        key = self._key(args)
        entry = self._graphs.get(key)
        if entry is None:
            static_in = [a.clone() for a in args]
            for _ in range(self.warmup):  # packs weights, raises LDS limits, fills the workspace caches: nothing of that may be captured
                self.forward(*static_in)
            torch.cuda.synchronize()
            side = torch.cuda.Stream(device=args[0].device)
            side.wait_stream(torch.cuda.current_stream(args[0].device))
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    static_out = self.forward(*static_in)
            torch.cuda.current_stream(args[0].device).wait_stream(side)
            entry = (graph, static_in, static_out)
            self._graphs[key] = entry
        graph, static_in, static_out = entry
        for s, a in zip(static_in, args):
            s.copy_(a, non_blocking=True)
        graph.replay()
        return static_out
