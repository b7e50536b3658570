"""Tensor-level entry points over the C ABI (device pointers taken from torch tensors).

PyTorch is plumbing here: it owns device memory and the stream; every FLOP is executed by
``libaid_hip.so``.  All functions enqueue on ``torch.cuda.current_stream()`` and never
synchronise.  Tensors must live on a HIP device and be fp16 / bf16 (the fast kernels) or fp32 (the reference's SD1.x storage type,
correctness-first kernels) — anything else raises.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import threading
import weakref
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import (DTYPE_BF16, DTYPE_F16, DTYPE_F32, IP_NONE, IP_PLAIN, IP_SAME, MODE_INNER, MODE_OUTER, MODE_PLAIN, AidAttnArgs,
                   AidGemmProblem, AidProcessorArgs)

MODES = {"plain": MODE_PLAIN, "inner": MODE_INNER, "outer": MODE_OUTER}
IP_MODES = {"same": IP_SAME, "plain": IP_PLAIN}
SUPPORTED_HEAD_DIMS = (40, 64, 80, 160)


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float16:
        return DTYPE_F16
    if t.dtype == torch.bfloat16:
        return DTYPE_BF16
    if t.dtype == torch.float32:        # the reference's SD1.x default (gradio_src/app.py:62): fp32 tensors, fp32 matrix-pipe arithmetic
        return DTYPE_F32
    raise TypeError(f"the HIP path computes in float16 / bfloat16 / float32; got {t.dtype}")


def _require_16bit(t: torch.Tensor, what: str) -> None:
    if t.dtype == torch.float32:
        raise TypeError(f"{what} is implemented for float16 / bfloat16 storage (the float32 path covers the projections, the attention "
                        "core and the processor call; run the LayerNorm in torch there)")


def _require_gpu(*ts: Optional[torch.Tensor]) -> torch.device:
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("attention-interpolation-diffusion_amd has no CPU path: tensors must be on "
                               f"an MI355X (HIP) device, got device={t.device}")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


@contextlib.contextmanager
def _on(dev: Optional[torch.device]):
    """Make the tensors' device the current one for the launch: the library enqueues on the CURRENT device's stream and
    keeps its per-kernel attributes per device, so a process that drives several GPUs (or calls with tensors of a
    non-current device) must not launch on the wrong device's stream."""
    if dev is None or dev.index is None or dev.index == torch.cuda.current_device():
        yield
    else:
        with torch.cuda.device(dev):
            yield


def executed_segments(mode: str, fused: bool, coef_vals: Optional[Sequence[float]], n_frames: int,
                      own_rows: Optional[Sequence[int]] = None, begin: int = 0, end: int = -1) -> int:
    """(frame, key segment) passes the attention kernel really runs for one launch — the bookkeeping behind
    ``AidProfileEntry.flops_executed`` (csrc/aid_attn.hip): a PLAIN rider (negative coefficient) and a fused END-POINT
    frame take one pass over their own keys; a coefficient of exactly 0 / 1 drops the zero-weighted side of OUTER."""
    if mode == "plain" or coef_vals is None:
        return n_frames
    own = list(range(n_frames)) if own_rows is None else list(own_rows)
    n_rows = max(own) + 1
    begin, end = begin % n_rows, end % n_rows
    total = 0
    for i in range(n_frames):
        c = coef_vals[i]
        if c < 0 or (fused and ((c == 0 and own[i] == begin) or (c == 1 and own[i] == end))):
            total += 1
        elif mode == "inner":
            total += 2 if fused else 1
        else:
            total += (1 if fused else 0) + (1 if c != 1 else 0) + (1 if c != 0 else 0)
    return total


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ---- scratch memory --------------------------------------------------------------------------------------------------------
# One workspace per (device, stream): kernels on one stream run in order, so reuse is safe.  Stream captures get their OWN workspace,
# keyed by the capture's id (aid_stream_capture_id) and never handed to eager calls or to another capture, even when torch recycles the
# stream handle — a captured graph keeps the address for as long as it can be replayed.  Who keeps the memory alive:
#   * captures of THIS package (``_PassGraphs``, ``AidDenoiseLoop``) run under a ``WorkspaceOwner``: the capture ADOPTS the workspace
#     the eager warm-up on the capture stream left behind (nothing is allocated inside the capture), the owner records the key and
#     drops it when the object that holds the graphs is released or garbage-collected (the adopted workspaces used to
#     live for the life of the process, ~150 MB per captured SDXL pass kind);
#   * a FOREIGN capture (``with torch.cuda.graph(g): unet(...)`` around processor calls) allocates its workspace inside the capture,
#     i.e. from the graph's private pool: the memory belongs to the graph and goes when the graph goes.  The dictionary entry of a
#     finished foreign capture is dropped at the next call on that stream.  (A capture torch's allocator does not know about
#     cannot allocate: it adopts the workspace of an eager warm-up on the capture stream — kept for the life of the process — or
#     gets a RuntimeError that asks for that warm-up.)
_eager_ws: Dict[Tuple[int, int], torch.Tensor] = {}
_capture_ws: Dict[Tuple[int, int, int], torch.Tensor] = {}
_pool_keys: set = set()                      # keys of _capture_ws whose tensor lives in a foreign graph's private pool
_ws_tls = threading.local()


def _drop_capture_keys(keys: set) -> None:
    for k in list(keys):
        _capture_ws.pop(k, None)
    keys.clear()


class WorkspaceOwner:
    """Ties the capture workspaces adopted inside ``with owner:`` blocks to the life of the object that holds the captured graphs:
    ``release()`` — or the garbage collection of the owner — drops them.  Release only when the graphs are gone."""

    def __init__(self):
        self._keys: set = set()
        self._fin = weakref.finalize(self, _drop_capture_keys, self._keys)

    def __enter__(self):
        stack = getattr(_ws_tls, "owners", None)
        if stack is None:
            stack = _ws_tls.owners = []
        stack.append(self)
        return self

    def __exit__(self, *exc):
        _ws_tls.owners.pop()
        return False

    def release(self) -> None:
        _drop_capture_keys(self._keys)

    def __len__(self) -> int:
        return len(self._keys)


def _current_owner() -> Optional[WorkspaceOwner]:
    stack = getattr(_ws_tls, "owners", None)
    return stack[-1] if stack else None


def _capture_id(stream: int) -> int:
    cid = C.c_ulonglong(0)
    _lib.check(_lib.load().aid_stream_capture_id(stream, C.byref(cid)), "aid_stream_capture_id")
    return int(cid.value)


def workspace(nbytes: int, device: torch.device) -> torch.Tensor:
    """Scratch buffer of at least ``nbytes`` for a library call on the current stream of ``device`` (policy above)."""
    dev = device.index if device.index is not None else torch.cuda.current_device()
    stream = _stream()
    cid = _capture_id(stream) if torch.cuda.is_current_stream_capturing() else 0
    for k in [k for k in _pool_keys if k[0] == dev and k[1] == stream and k[2] != cid]:
        _pool_keys.discard(k)                                   # a foreign capture on this stream has ended: its pool owns the memory
        _capture_ws.pop(k, None)
    if cid:
        key = (dev, stream, cid)
        ws = _capture_ws.get(key)
        if ws is not None and ws.numel() >= nbytes:
            return ws
        owner = _current_owner()
        if owner is None and (ws is None or key in _pool_keys):  # foreign capture: the graph's private pool owns its workspace
            # (a later, larger request of the same capture allocates again; the smaller block goes back to that pool, stream-ordered)
            try:
                ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
                _capture_ws[key] = ws
                _pool_keys.add(key)
                return ws
            except RuntimeError:                                # a capture torch's allocator does not know about
                ws = None
        if ws is None:
            ws = _eager_ws.pop((dev, stream), None)             # adopt: eager calls on this stream allocate afresh from now on
            if ws is not None:
                _capture_ws[key] = ws
                if owner is not None:
                    owner._keys.add(key)
        if ws is None or ws.numel() < nbytes:
            raise RuntimeError(
                f"a library call inside a stream capture needs {nbytes} bytes of workspace on a stream that has "
                f"{'none' if ws is None else str(ws.numel()) + ' bytes'}: run the same calls once EAGERLY on the capture stream first "
                "(torch.cuda.graph(g, stream=s) after a warm-up under torch.cuda.stream(s)) — the capture then adopts the warm-up's "
                "workspace")
        return ws
    key = (dev, stream)
    ws = _eager_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _eager_ws[key] = ws
    return ws


def release_workspaces() -> None:
    """Drop every cached workspace (also the ones captured graphs still point into: call it when those graphs are gone)."""
    _eager_ws.clear()
    _capture_ws.clear()
    _pool_keys.clear()


# ---- per-call hint: launch streams that share the device ---------------------------------------------------------------------
_tls = threading.local()


@contextlib.contextmanager
def cu_share(n: int):
    """Inside the block every library call of THIS host thread carries ``cu_share = n`` (AidGemmProblem / AidProcessorArgs, ABI v7):
    n independent launch streams run side by side — the conditional and the unconditional UNet call of a step on two streams — so
    a launch plans with 1 / n of the CUs.  A hint: results never depend on it.  Thread-local: two host threads driving two GPUs (or
    two streams) can hold different values at the same time."""
    n = int(n)
    if not 0 <= n <= 8:
        raise ValueError("cu_share must be in 0 .. 8")
    prev = getattr(_tls, "cu_share", 0)
    _tls.cu_share = n
    try:
        yield
    finally:
        _tls.cu_share = prev


def current_cu_share() -> int:
    return int(getattr(_tls, "cu_share", 0))


# ---------------------------------------------------------------------------------------------
def gemm_nt(problems: Sequence[dict]) -> None:
    """Grouped C = A @ B^T (+ bias) (+ residual, added after the rounding).  Each problem: dict(a=[.., m, k],
    b=[n, k] or [batch, n, k], c=out tensor, bias=None|[n], residual=None|like c, batch=1, m, n, k, lda, ldb, ldc,
    stride_a, stride_b, stride_c)."""
    lib = _lib.load()
    n = len(problems)
    arr = (AidGemmProblem * n)()
    dt = None
    for i, p in enumerate(problems):
        _require_gpu(p["a"], p["b"], p["c"], p.get("bias"), p.get("residual"))
        code = _dtype_code(p["a"])
        if dt is None:
            dt = code
        if code != dt or _dtype_code(p["b"]) != dt or _dtype_code(p["c"]) != dt:
            raise TypeError("all GEMM operands must share one dtype")
        q = arr[i]
        q.a, q.b, q.c = p["a"].data_ptr(), p["b"].data_ptr(), p["c"].data_ptr()
        q.bias = _ptr(p.get("bias"))
        q.residual = _ptr(p.get("residual"))
        q.m, q.n, q.k = p["m"], p["n"], p["k"]
        q.lda, q.ldb, q.ldc = p["lda"], p["ldb"], p["ldc"]
        q.batch = p.get("batch", 1)
        q.scale = float(p.get("scale", 0.0))            # 0 = 1 (zero-initialised structs)
        q.stride_a, q.stride_b, q.stride_c = p.get("stride_a", 0), p.get("stride_b", 0), p.get("stride_c", 0)
        q.trans_rows = int(p.get("trans_rows", 0))      # C transposed per frame of that many rows (aid_hip.h)
        q.cu_share = current_cu_share()
        if p.get("ln_stats") is not None:       # folded LayerNorm: dict(ln_stats=, ln_colsum=, ln_shift=, ln_side=1|2[, stride_stats=])
            for t_ in (p["ln_stats"], p["ln_colsum"], p["ln_shift"]):
                _require_gpu(t_)
                if t_.dtype != torch.float32 or not t_.is_contiguous():
                    raise TypeError("ln_stats / ln_colsum / ln_shift are contiguous float32 tensors")
            q.ln_stats, q.ln_colsum, q.ln_shift = p["ln_stats"].data_ptr(), p["ln_colsum"].data_ptr(), p["ln_shift"].data_ptr()
            q.ln_side, q.stride_stats = int(p["ln_side"]), int(p.get("stride_stats", 0))
    with _on(problems[0]["a"].device):
        _lib.check(lib.aid_gemm_nt(arr, n, dt, _stream()), "aid_gemm_nt")


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None,
           out: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x @ w.T + bias (+ residual) on the HIP GEMM (x [..., k] contiguous, w [n, k] contiguous)."""
    assert x.is_contiguous() and w.is_contiguous()
    k = x.shape[-1]
    n = w.shape[0]
    m = x.numel() // k
    if out is None:
        out = torch.empty(*x.shape[:-1], n, dtype=x.dtype, device=x.device)
    if residual is not None and (residual.shape != out.shape or residual.dtype != out.dtype or not residual.is_contiguous()):
        raise ValueError("residual must be a contiguous tensor shaped like the output")
    gemm_nt([dict(a=x, b=w, c=out, bias=bias, residual=residual, m=m, n=n, k=k, lda=k, ldb=k, ldc=n)])
    return out


def layernorm(x: torch.Tensor, gamma: Optional[torch.Tensor] = None, beta: Optional[torch.Tensor] = None,
              eps: float = 1e-5, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """LayerNorm over the last dimension on the HIP kernel (fp32 statistics, one rounding)."""
    _require_gpu(x, gamma, beta, out)
    _require_16bit(x, "layernorm")
    if not x.is_contiguous():
        raise ValueError("operands must be contiguous")
    for t_ in (gamma, beta):
        if t_ is not None and (t_.dtype != x.dtype or t_.numel() != x.shape[-1] or not t_.is_contiguous()):
            raise ValueError("gamma / beta must be contiguous [c] tensors of the activation dtype")
    if out is None:
        out = torch.empty_like(x)
    c = x.shape[-1]
    with _on(x.device):
        _lib.check(_lib.load().aid_layernorm(x.data_ptr(), _ptr(gamma), _ptr(beta), out.data_ptr(), x.numel() // c, c,
                                             float(eps), _dtype_code(x), _stream()), "aid_layernorm")
    return out


def ln_stats(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """(mean, rstd) per row of x over the last dimension, fp32 [rows, 2] — what the folded LayerNorm needs of x."""
    _require_gpu(x)
    _require_16bit(x, "ln_stats")
    if not x.is_contiguous():
        raise ValueError("operands must be contiguous")
    c = x.shape[-1]
    st = torch.empty(x.numel() // c, 2, dtype=torch.float32, device=x.device)
    with _on(x.device):
        _lib.check(_lib.load().aid_ln_stats(x.data_ptr(), st.data_ptr(), x.numel() // c, c, float(eps), _dtype_code(x),
                                            _stream()), "aid_ln_stats")
    return st


def ln_fold(w: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor]):
    """Fold a LayerNorm's affine part into the Linear that consumes it: returns (w * gamma in the storage dtype,
    colsum fp32 [rows], shift fp32 [rows]) — see AidGemmProblem.ln_* in include/aid_hip.h."""
    _require_gpu(w, gamma, beta)
    _require_16bit(w, "ln_fold")
    if w.ndim != 2 or not w.is_contiguous():
        raise ValueError("w must be a contiguous [rows, c] weight")
    for t_ in (gamma, beta):
        if t_ is not None and (t_.dtype != w.dtype or t_.numel() != w.shape[1] or not t_.is_contiguous()):
            raise ValueError("gamma / beta must be contiguous [c] tensors of the weight dtype")
    wf = torch.empty_like(w)
    cs = torch.empty(w.shape[0], dtype=torch.float32, device=w.device)
    sh = torch.empty_like(cs)
    with _on(w.device):
        _lib.check(_lib.load().aid_ln_fold(w.data_ptr(), _ptr(gamma), _ptr(beta), wf.data_ptr(), cs.data_ptr(), sh.data_ptr(),
                                           w.shape[0], w.shape[1], _dtype_code(w), _stream()), "aid_ln_fold")
    return wf, cs, sh


def project_kv(e: torch.Tensor, wk: torch.Tensor, wv: torch.Tensor, extra_rows: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """k = e @ wk.T  [F, L, C]  and  V^T = wv @ e^T  [F, C, Lp]  (Lp = L rounded up to 8) in one launch.
    ``extra_rows``: allocate that many more (uninitialised) frame rows behind the F projected ones — room for the
    end-point frames' keys / values a rank receives from their owners (dist.EndpointExchange)."""
    f, l, cc = e.shape
    c = wk.shape[0]
    lp = (l + 7) // 8 * 8
    k = torch.empty(f + extra_rows, l, c, dtype=e.dtype, device=e.device)
    vt = torch.empty(f + extra_rows, c, lp, dtype=e.dtype, device=e.device)
    if l % 8 == 0:          # flat value projection, transposed epilogue (tile count of the key projection: whole CU rounds)
        pv = dict(a=e, b=wv, c=vt, m=f * l, n=c, k=cc, lda=cc, ldb=cc, ldc=lp, stride_c=c * lp, trans_rows=l)
    else:                   # V^T[f] = Wv E_f^T, one batch entry per frame (pad columns l .. lp of V^T are written with zeros)
        pv = dict(a=wv, b=e, c=vt, m=c, n=l, k=cc, lda=cc, ldb=cc, ldc=lp, batch=f, stride_a=0, stride_b=l * cc, stride_c=c * lp)
    gemm_nt([dict(a=e, b=wk, c=k, m=f * l, n=c, k=cc, lda=cc, ldb=cc, ldc=c), pv])
    return k, vt


def score_bias_layout(bias: torch.Tensor, n: int, heads: int, s: int, l: int, dtype: torch.dtype) -> Tuple[int, int, int]:
    """(frame stride, head stride, row stride) in elements of an additive score bias (AidAttnArgs.bias, ABI v8) given as diffusers
    hands it around: ``[N * H, 1 | S, L]`` (``Attention.prepare_attention_mask``), ``[N, 1 | S, L]`` or ``[N, 1 | H, 1 | S, L]``."""
    if bias.dtype != dtype:
        raise TypeError(f"the score bias must have the activation dtype ({dtype}), got {bias.dtype}")
    if bias.ndim == 3:
        b0, r, ll = bias.shape
        if b0 == n * heads and heads > 1:
            fs, hs = heads * bias.stride(0), bias.stride(0)
        elif b0 == n:
            fs, hs = bias.stride(0), 0
        else:
            raise ValueError(f"score bias of {b0} rows for {n} frames x {heads} heads")
        rs = bias.stride(1)
    elif bias.ndim == 4:
        b0, hx, r, ll = bias.shape
        if b0 != n or hx not in (1, heads):
            raise ValueError(f"score bias must be [N, 1 | H, 1 | S, L]; got {tuple(bias.shape)} for N = {n}, H = {heads}")
        fs, hs, rs = bias.stride(0), (bias.stride(1) if hx > 1 else 0), bias.stride(2)
    else:
        raise ValueError("score bias must be a 3-D or 4-D tensor")
    if ll != l or r not in (1, s) or (ll > 1 and bias.stride(-1) != 1):
        raise ValueError(f"score bias must be [..., 1 | {s}, {l}] with contiguous rows; got {tuple(bias.shape)}")
    return int(fs), int(hs), int(rs if r > 1 else 0)


def attn_fwd(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, *, l: int,
             mode: str = "plain", fused: bool = False, coef: Optional[torch.Tensor] = None,
             begin: int = 0, end: int = -1, out: Optional[torch.Tensor] = None,
             accumulate: bool = False, out_scale: float = 1.0,
             frame_scale: Optional[torch.Tensor] = None, kv_map: Optional[torch.Tensor] = None,
             softmax_scale: Optional[float] = None, n_plain: int = 0, seg_executed: int = 0,
             q_prescaled: bool = False, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Interpolated attention core (see AidAttnArgs in include/aid_hip.h).
    q [N, S, C], k [F, L, C], vt [F, C, Lp] contiguous; coef / frame_scale fp32 device [N].  ``bias``: additive score bias
    (diffusers' prepared attention_mask, ``score_bias_layout``); not with ``fused``."""
    lib = _lib.load()
    _require_gpu(q, k, vt, out, coef, frame_scale, kv_map, bias)
    dt = _dtype_code(q)
    if _dtype_code(k) != dt or _dtype_code(vt) != dt:
        raise TypeError("q, k, vt must share one dtype")
    assert q.is_contiguous() and k.is_contiguous() and vt.is_contiguous()
    n, s, c = q.shape
    f = k.shape[0]
    d = c // heads
    if out is None:
        if accumulate:
            raise ValueError("accumulate=True needs an existing `out`")
        out = torch.empty_like(q)
    for t_, nm in ((coef, "coef"), (frame_scale, "frame_scale")):
        if t_ is not None and (t_.dtype != torch.float32 or t_.numel() != n):
            raise ValueError(f"{nm} must be a float32 device tensor with one entry per frame ({n})")
    if kv_map is not None and (kv_map.dtype != torch.int32 or kv_map.numel() != n):
        raise ValueError("kv_map must be an int32 device tensor with one entry per frame")
    a = AidAttnArgs()
    a.q, a.k, a.vt, a.out = q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr()
    a.coef, a.frame_scale, a.kv_map = _ptr(coef), _ptr(frame_scale), _ptr(kv_map)
    if mode == "inner":
        # interpolated K / V^T of the interior frames (one streaming launch), read by the attention kernel
        if kv_map is not None or f < n:
            raise ValueError("inner mode needs one key/value row per frame (no kv_map)")
        k2, vt2 = torch.empty_like(k[:n]), torch.empty_like(vt[:n])
        with _on(q.device):
            _lib.check(lib.aid_lerp_kv(k.data_ptr(), vt.data_ptr(), k2.data_ptr(), vt2.data_ptr(), coef.data_ptr(), n,
                                       begin % f, end % f, k.shape[1] * k.shape[2], vt.shape[1] * vt.shape[2], dt,
                                       _stream()), "aid_lerp_kv")
        a.k2, a.vt2 = k2.data_ptr(), vt2.data_ptr()
    a.n_frames, a.n_kv = n, f
    a.s, a.l, a.heads, a.d = s, l, heads, d
    a.ldq, a.ldk, a.ldvt, a.ldo = c, k.shape[2], vt.shape[2], out.shape[2]
    a.q_fs, a.k_fs, a.vt_fs, a.o_fs = s * c, k.shape[1] * k.shape[2], vt.shape[1] * vt.shape[2], s * out.shape[2]
    a.mode, a.fused = MODES[mode], int(bool(fused))
    a.begin, a.end = begin % f, end % f
    a.accumulate, a.dtype = int(bool(accumulate)), dt
    a.softmax_scale = float(d ** -0.5 if softmax_scale is None else softmax_scale)
    a.out_scale = float(out_scale)
    a.n_plain = int(n_plain)
    a.seg_executed = int(seg_executed)
    if bias is not None:
        a.bias = bias.data_ptr()
        a.bias_fs, a.bias_hs, a.bias_rs = score_bias_layout(bias, n, heads, s, l, q.dtype)
    a.q_prescaled = int(bool(q_prescaled))              # q already holds q * softmax_scale * log2(e) (the processor path's q projection)
    with _on(q.device):
        _lib.check(lib.aid_attn_fwd(C.byref(a), _stream()), "aid_attn_fwd")
    return out


def set_tuning(name: str, value: int = -1) -> None:
    """Development knob of the library (``aid_set_tuning``): ``value < 0`` hands the choice back to the launch heuristics."""
    _lib.check(_lib.load().aid_set_tuning(name.encode(), int(value)), f"aid_set_tuning({name})")


def get_tuning(name: str) -> int:
    """Current value of a development knob (``aid_get_tuning``); -1 = the launch heuristics decide."""
    import ctypes
    v = ctypes.c_int(0)
    _lib.check(_lib.load().aid_get_tuning(name.encode(), ctypes.byref(v)), f"aid_get_tuning({name})")
    return int(v.value)


def last_attn_variant() -> str:
    return _lib.load().aid_last_attn_variant().decode()


def last_gemm_variant() -> str:
    return _lib.load().aid_last_gemm_variant().decode()


def processor_fwd(x: torch.Tensor, ctx: Optional[torch.Tensor], wq: torch.Tensor, wk: torch.Tensor,
                  wv: torch.Tensor, wo: torch.Tensor, bo: Optional[torch.Tensor], heads: int, *,
                  mode: str = "plain", fused: bool = False, coef: Optional[torch.Tensor] = None,
                  begin: int = 0, end: int = -1, ctx_map: Optional[torch.Tensor] = None,
                  out: Optional[torch.Tensor] = None, n_plain: int = 0,
                  ln: Optional[Tuple[Optional[torch.Tensor], Optional[torch.Tensor], float]] = None,
                  residual: Optional[torch.Tensor] = None, seg_executed: int = 0,
                  ip: Optional[dict] = None, ln_folded: Optional[tuple] = None,
                  kv_cached: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                  attn_bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One whole processor call: y = to_out(AID-attention(to_q(x), to_k(ctx), to_v(ctx)))
    in three launches (grouped q/k/V^T GEMM, attention core, out-proj GEMM).
    ``ln = (gamma, beta, eps)`` computes on LayerNorm(x); ``residual`` is added to the result (the transformer
    block's norm in front of the call and its residual add after it).
    ``ip`` = the IP-Adapter image branch (AidProcessorArgs.ip_*): dict(tokens=[R, T, Cc] tensor whose rows may be a
    strided view, wk=to_k_ip weight, wv=to_v_ip weight, mode="same"|"plain", scale=float, map=int32 device [N] or None,
    frame_scale=fp32 device [N] or None, begin=int, end=int).
    ``ln_folded`` = (wq', wk', wv', const) with ``ln``: the LayerNorm is folded into the projections (wq' etc. from
    ``ln_fold``, wk' / wv' None for cross-attention, const fp32 [6, C] = colsum_q, shift_q, colsum_k, shift_k, colsum_v,
    shift_v): only the row statistics of x are computed, LayerNorm(x) is never written.
    ``kv_cached`` = (k, vt) from ``project_kv(ctx, wk, wv)``: the step-invariant keys / values of a cross-attention layer,
    projected once by the caller; the call then projects the queries only.
    ``attn_bias``: additive score bias of the (text) attention — diffusers' prepared attention_mask (AidProcessorArgs.attn_bias,
    ``score_bias_layout``); the library refuses it together with ``fused`` or ``ip`` (the reference fails there, aid_hip.h)."""
    lib = _lib.load()
    ipt = ip or {}
    dev = _require_gpu(x, ctx, wq, wk, wv, wo, bo, coef, ctx_map, out, residual, attn_bias, *(ln[:2] if ln else ()),
                       ipt.get("tokens"), ipt.get("wk"), ipt.get("wv"), ipt.get("map"), ipt.get("frame_scale"))
    dt = _dtype_code(x)
    for t_ in (ctx, wq, wk, wv, wo, bo):
        if t_ is not None and t_.dtype != x.dtype:
            raise TypeError(f"dtype mismatch: hidden states are {x.dtype}, another operand is {t_.dtype}")
    for t_ in (x, ctx, wq, wk, wv, wo, bo):
        if t_ is not None and not t_.is_contiguous():
            raise ValueError("operands must be contiguous")
    n, s, c = x.shape
    if out is None:
        out = torch.empty_like(x)
    a = AidProcessorArgs()
    a.x, a.ctx = x.data_ptr(), _ptr(ctx)
    a.wq, a.wk, a.wv, a.wo, a.bo = wq.data_ptr(), wk.data_ptr(), wv.data_ptr(), wo.data_ptr(), _ptr(bo)
    a.y, a.coef, a.ctx_map = out.data_ptr(), _ptr(coef), _ptr(ctx_map)
    a.n_frames, a.s, a.c, a.heads = n, s, c, heads
    if ctx is not None:
        a.n_ctx, a.l, a.cc = ctx.shape
    else:
        a.n_ctx, a.l, a.cc = n, s, c
    nkv = a.n_ctx
    a.mode, a.fused = MODES[mode], int(bool(fused))
    a.begin, a.end = begin % nkv, end % nkv
    a.dtype = dt
    a.n_plain = int(n_plain)
    a.seg_executed = int(seg_executed)
    if ip is not None:
        tok = ip["tokens"]
        if ctx is None:
            raise ValueError("the image branch belongs to a cross-attention call (encoder_hidden_states = (text, [ip]))")
        if tok.ndim != 3 or tok.shape[2] != ctx.shape[2] or tok.dtype != x.dtype or tok.stride(2) != 1 \
                or tok.stride(1) != tok.shape[2]:
            raise ValueError("image tokens must be [rows, T, Cc] in the activation dtype with contiguous [T, Cc] rows")
        for t_ in (ip["wk"], ip["wv"]):
            if t_.dtype != x.dtype or not t_.is_contiguous() or tuple(t_.shape) != (c, ctx.shape[2]):
                raise ValueError("to_k_ip / to_v_ip weights must be contiguous [C, Cc] tensors of the activation dtype")
        a.ip, a.wk_ip, a.wv_ip = tok.data_ptr(), ip["wk"].data_ptr(), ip["wv"].data_ptr()
        a.ip_map, a.ip_frame_scale = _ptr(ip.get("map")), _ptr(ip.get("frame_scale"))
        a.n_ip, a.t_ip = tok.shape[0], tok.shape[1]
        a.ip_stride = tok.stride(0) if tok.shape[0] > 1 else tok.shape[1] * tok.shape[2]
        a.ip_mode, a.ip_scale = IP_MODES[ip["mode"]], float(ip.get("scale", 1.0))
        a.ip_begin, a.ip_end = int(ip.get("begin", 0)) % a.n_ip, int(ip.get("end", -1)) % a.n_ip
    if ln is not None:
        _require_16bit(x, "the LayerNorm fusion of processor_fwd")
        g_, b_, eps = ln
        if not eps > 0:
            raise ValueError("LayerNorm eps must be > 0")
        for t_ in (g_, b_):
            if t_ is not None and (t_.dtype != x.dtype or t_.numel() != c or not t_.is_contiguous()):
                raise ValueError("LayerNorm gamma / beta must be contiguous [c] tensors of the activation dtype")
        a.ln_gamma, a.ln_beta, a.ln_eps = _ptr(g_), _ptr(b_), float(eps)
        if ln_folded is not None:
            fq, fk, fv, cst = ln_folded
            _require_gpu(fq, fk, fv, cst)
            if cst.dtype != torch.float32 or tuple(cst.shape) != (6, c) or not cst.is_contiguous():
                raise ValueError("ln_folded constants must be a contiguous float32 [6, C] tensor")
            for t_, ref in ((fq, wq), (fk, wk), (fv, wv)):
                if t_ is not None and (t_.dtype != x.dtype or t_.shape != ref.shape or not t_.is_contiguous()):
                    raise ValueError("folded weights must look like the weights they replace")
            if ctx is None and (fk is None or fv is None):
                raise ValueError("self-attention needs the folded to_k / to_v weights too")
            a.ln_wq, a.ln_wk, a.ln_wv, a.ln_const = fq.data_ptr(), _ptr(fk), _ptr(fv), cst.data_ptr()
    elif ln_folded is not None:
        raise ValueError("ln_folded goes with ln=(gamma, beta, eps)")
    if residual is not None:
        if residual.shape != x.shape or residual.dtype != x.dtype or not residual.is_contiguous():
            raise ValueError("residual must be a contiguous tensor shaped like the hidden states")
        a.residual = residual.data_ptr()
    if kv_cached is not None:
        if ctx is None:
            raise ValueError("cached keys / values belong to a cross-attention call")
        kc, vc = kv_cached
        _require_gpu(kc, vc)
        lp = (ctx.shape[1] + 7) // 8 * 8
        ok = kc.dtype == x.dtype and vc.dtype == x.dtype and kc.is_contiguous() and vc.is_contiguous() \
            and kc.shape[0] >= ctx.shape[0] and vc.shape[0] >= ctx.shape[0] \
            and tuple(kc.shape[1:]) == (ctx.shape[1], c) and tuple(vc.shape[1:]) == (c, lp)
        if not ok:
            raise ValueError("kv_cached must be (k [n_ctx, L, C], vt [n_ctx, C, round_up(L, 8)]) as project_kv returns them")
        a.k_cached, a.vt_cached = kc.data_ptr(), vc.data_ptr()
    if attn_bias is not None:
        a.attn_bias = attn_bias.data_ptr()
        a.attn_bias_fs, a.attn_bias_hs, a.attn_bias_rs = score_bias_layout(attn_bias, n, heads, s, a.l, x.dtype)
    a.cu_share = current_cu_share()
    nbytes = lib.aid_processor_workspace_bytes(C.byref(a))
    with _on(dev):
        if nbytes == 0:
            # let the library say why
            a.workspace, a.workspace_bytes = None, 0
            _lib.check(lib.aid_processor_fwd(C.byref(a), _stream()), "aid_processor_fwd")
            raise RuntimeError("aid_processor_workspace_bytes returned 0")
        ws = workspace(nbytes, dev)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        _lib.check(lib.aid_processor_fwd(C.byref(a), _stream()), "aid_processor_fwd")
    return out
