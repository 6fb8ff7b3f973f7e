"""Drop-in ``nn.Module`` for the reference's separation network.

``MossFormer2_SS(args).model`` in the reference is a ``MossFormer`` module
(reference: models/mossformer2/mossformer2.py:648-757) whose callers do
``model.to(device); reload_for_eval(model, dir, use_cuda); model.eval();
out_list = model(inputs)`` (reference: inference.py:21-26, app.py:175-181,
utils/decode.py:62,73).  ``MossFormerHIP`` offers the same surface:

* ``state_dict()`` / ``load_state_dict()`` with exactly the reference's key
  names, shapes and order (a skeleton of empty sub-modules carries the
  parameters; it has no forward of its own);
* ``__call__(x[B, T]) -> list of num_spks tensors [B, T]`` on the device and
  dtype of ``x``;
* the compute always runs on the module's MI355X (``compute_device``) through
  libmsf2 -- a CPU input (the Flask app pins ``device='cpu'``, reference:
  app.py:166-170) is copied over and the result copied back.

The forward is re-entrant: packed weights are read-only and every call takes
its own workspace from torch's caching allocator (the reference's Flask server
is threaded and shares one model object, reference: app.py:256-259).

``use_graphs=True`` replays the forward of a given input shape from a captured
HIP graph (the library call allocates nothing and never synchronises, so the
whole launch sequence -- about 1 000 kernels -- is capturable).  That is for the
callers' own shape, one utterance per call (reference: inference.py:35-41,
app.py:269), where the eager launch sequence is host-bound.  A graph owns its
input, output and workspace buffers, so replays of one shape are serialised by
a lock; the outputs returned are copies.
"""
import threading

import torch
from torch import nn

from . import _lib
from .config import MossFormer2Config
from .packing import PackedWeights
from .schema import state_dict_schema
from .synth import make_state_dict

_BUFFERS = ("pos_enc.inv_freq",)
_FROZEN = ("rotary_pos_emb.freqs",)


def _attach(root: nn.Module, key: str, tensor: torch.Tensor):
    parts = key.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, nn.Module())
        mod = mod._modules[p]
    if any(key.endswith(b) for b in _BUFFERS):
        mod.register_buffer(parts[-1], tensor)
    else:
        frozen = any(key.endswith(f) for f in _FROZEN)
        mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=not frozen))


class MossFormerHIP(nn.Module):
    MAX_GRAPHS = 8          # captured shapes kept per model (each holds its own workspace)

    def __init__(self, cfg: MossFormer2Config, compute_device=None, precision=_lib.PREC_FP32, init_seed=0, use_graphs=False):
        super().__init__()
        self.cfg = cfg
        self.num_spks = cfg.num_spks
        self.precision = precision
        self._compute_device = compute_device
        # deterministic, non-degenerate initial weights (a checkpoint normally replaces them)
        init = make_state_dict(cfg, init_seed)
        for key, shape in state_dict_schema(cfg).items():
            _attach(self, key, torch.from_numpy(init[key]).reshape(shape).clone())
        self._packed = None
        self._pack_lock = threading.Lock()
        self._lib = None
        self.use_graphs = bool(use_graphs)
        self._graphs = {}
        self._graph_lock = threading.Lock()

    # -- weight life-cycle ------------------------------------------------------
    def _invalidate(self):
        self._packed = None
        self._graphs = {}           # captured graphs hold pointers into the old packed weights

    def load_state_dict(self, state_dict, strict=True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self._invalidate()
        return r

    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        self._invalidate()
        return r

    @property
    def compute_device(self) -> torch.device:
        if self._compute_device is not None:
            return torch.device(self._compute_device)
        p = next(self.parameters())
        if p.device.type == "cuda":
            return p.device
        return torch.device("cuda", torch.cuda.current_device())

    def packed(self) -> PackedWeights:
        pk = self._packed
        if pk is None:
            with self._pack_lock:
                pk = self._packed
                if pk is None:
                    if not torch.cuda.is_available():
                        raise _lib.Msf2Error("no MI355X visible: the separation forward has no CPU path")
                    self._lib = _lib.load()
                    pk = PackedWeights(super().state_dict(), self.cfg, self.compute_device, self.precision)
                    self._packed = pk
        return pk

    # -- forward ------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x: torch.Tensor, taps: dict = None):
        if x.dim() != 2:
            raise ValueError("expected input of shape [batch, samples]")
        pk = self.packed()
        lib = self._lib
        dev = pk.device
        src_device, src_dtype = x.device, x.dtype
        B, T = x.shape
        if T < self.cfg.enc_kernel:
            raise ValueError(f"input shorter than the encoder kernel ({T} < {self.cfg.enc_kernel})")
        # One library call addresses its widest activation with 32-bit byte offsets (about 3.9 M samples per call for the
        # 512-channel model).  Utterances are independent, so a larger batch is separated in equal-sized pieces.
        if taps is None and B > 1 and lib.msf2_workspace_bytes(pk.c_cfg, B, T) == 0 and lib.msf2_workspace_bytes(pk.c_cfg, 1, T) != 0:
            step = B
            while step > 1 and lib.msf2_workspace_bytes(pk.c_cfg, step, T) == 0:
                step = (step + 1) // 2
            parts = [self.forward(x[i:i + step]) for i in range(0, B, step)]
            return [torch.cat([p[s] for p in parts], 0) for s in range(self.num_spks)]
        if self.use_graphs and taps is None:
            out = self._forward_graph(pk, x, B, T)
            outs = [out[s] for s in range(self.num_spks)]
            if src_device != dev or src_dtype != torch.float32:
                outs = [o.to(device=src_device, dtype=src_dtype) for o in outs]
            return outs
        with torch.cuda.device(dev):
            xin = x.detach().to(device=dev, dtype=torch.float32, non_blocking=True).contiguous()
            out = torch.empty((self.num_spks, B, T), device=dev, dtype=torch.float32)
            nbytes = lib.msf2_workspace_bytes(pk.c_cfg, B, T)
            if nbytes == 0:
                raise _lib.Msf2Error("msf2_workspace_bytes rejected the configuration")
            ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            tp = None
            tap_bufs = {}
            if taps is not None:
                S = lib.msf2_num_frames(T)
                tp = _lib.Taps()
                for name, _ in _lib.Taps._fields_:
                    tap_bufs[name] = torch.zeros((B, S, self.cfg.dim), device=dev, dtype=torch.float32)
                    setattr(tp, name, tap_bufs[name].data_ptr())
            stream = torch.cuda.current_stream(dev).cuda_stream
            rc = lib.msf2_forward(pk.c_cfg, pk.struct, xin.data_ptr(), out.data_ptr(), B, T,
                                  ws.data_ptr(), nbytes, tp, stream)
            _lib.check(rc, "msf2_forward")
            # the workspace must outlive the kernels enqueued on `stream`
            ws.record_stream(torch.cuda.current_stream(dev))
            if taps is not None:
                taps.update(tap_bufs)
        outs = [out[s] for s in range(self.num_spks)]
        if src_device != dev or src_dtype != torch.float32:
            outs = [o.to(device=src_device, dtype=src_dtype) for o in outs]
        return outs


    def _forward_graph(self, pk, x, B, T):
        lib, dev = self._lib, pk.device
        with self._graph_lock, torch.cuda.device(dev):
            ent = self._graphs.get((B, T))
            if ent is not None and ent[5] is not pk:
                ent = None                      # captured against packed weights that have been replaced since
            if ent is None:
                nbytes = lib.msf2_workspace_bytes(pk.c_cfg, B, T)
                if nbytes == 0:
                    raise _lib.Msf2Error("msf2_workspace_bytes rejected the configuration")
                self._graphs.pop((B, T), None)
                if len(self._graphs) >= self.MAX_GRAPHS:
                    self._graphs.pop(next(iter(self._graphs)))
                graph = torch.cuda.CUDAGraph()
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    # the graph's buffers belong to the stream every replay runs on (allocated under it, so the
                    # caching allocator orders their reuse after that stream's work when an entry is evicted)
                    xin = torch.zeros((B, T), device=dev, dtype=torch.float32)
                    out = torch.empty((self.num_spks, B, T), device=dev, dtype=torch.float32)
                    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
                    # one eager pass first: a launch failure surfaces here, not inside the capture
                    _lib.check(lib.msf2_forward(pk.c_cfg, pk.struct, xin.data_ptr(), out.data_ptr(), B, T,
                                                ws.data_ptr(), nbytes, None, side.cuda_stream), "msf2_forward")
                    side.synchronize()
                    with torch.cuda.graph(graph, stream=side):
                        rc = lib.msf2_forward(pk.c_cfg, pk.struct, xin.data_ptr(), out.data_ptr(), B, T,
                                              ws.data_ptr(), nbytes, None, torch.cuda.current_stream(dev).cuda_stream)
                    _lib.check(rc, "msf2_forward (graph capture)")
                torch.cuda.current_stream(dev).wait_stream(side)
                ent = (graph, xin, out, ws, side, pk)       # pk: the graph holds pointers into these packed weights
                self._graphs[(B, T)] = ent
            graph, xin, out, _, side, _ = ent
            # The graph is always launched on the stream it was captured on, ordered against the caller's stream by
            # events on both sides (launched on the null stream, replays were not ordered against the copies around
            # them on ROCm 7.2).  Replays of one shape therefore run one after the other.
            cur = torch.cuda.current_stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                xin.copy_(x.detach().to(dtype=torch.float32), non_blocking=True)
                graph.replay()
                res = out.clone()
            cur.wait_stream(side)
            res.record_stream(cur)
            return res


class MossFormer2_SS(nn.Module):
    """Model wrapper with the reference's constructor contract: reads the
    hyper-parameters off ``args`` and exposes the network as ``.model``
    (reference: models/mossformer2/mossformer2.py:739-757)."""

    def __init__(self, args):
        super().__init__()
        self.model = MossFormerHIP(MossFormer2Config.from_args(args),
                                   precision=int(getattr(args, "msf2_precision", _lib.PREC_FP32)),
                                   use_graphs=bool(getattr(args, "msf2_use_graphs", False)))

    def forward(self, x):
        return self.model(x)
