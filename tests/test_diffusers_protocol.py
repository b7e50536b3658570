"""The drop-in boundary against the diffusers PROTOCOL (tests/diffusers_double.py: Attention.forward's signature-filtered dispatch,
set_processor / attn_processors key naming, BasicTransformerBlock, tuple encoder_hidden_states of an IP-Adapter UNet) — what the
reference's load_aid / load_aid_ip_adapter / activate_aid / deactivate_aid rely on (pipeline_interpolated_sd.py:950-1020) and what
its processors are called through (interpolation.py:573-580).  diffusers itself is on neither box; tests/test_hip_diffusers_attention.py
runs against the real class wherever it is importable.  CPU part: installation and dispatch; GPU part: numbers."""
import threading
import warnings

import numpy as np
import pytest
import torch

from diffusers_double import Attention, IPAdapterAttnProcessor2_0, UNetDouble
from oracle import aid_oracle as O
from util import TOL, rel_l2, to_np64

import aid_amd
from aid_amd import ops
from aid_amd.processors import (HipAttnProcessor, HipIPAdapterAttnProcessor, InnerInterpolatedAttnProcessor,
                                OuterInterpolatedAttnProcessor, OuterInterpolatedIPAttnProcessor, activate_aid, deactivate_aid,
                                load_aid, load_aid_ip_adapter)


class _Recorder:
    """A processor with the reference's call signature that only records how it was called."""

    def __init__(self):
        self.calls = []

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, ctx_index=None):
        self.calls.append(dict(ehs=encoder_hidden_states, mask=attention_mask, ctx_index=ctx_index))
        return hidden_states


# ---- CPU: installation and dispatch ------------------------------------------------------------------------------------------
def test_processor_keys_follow_diffusers_naming_and_load_aid_wraps_every_layer():
    unet = UNetDouble()
    unet.set_attn_processor(_Recorder())
    keys = list(unet.attn_processors.keys())
    assert len(keys) == 2 * (2 + 2 + 1 + 3 + 3) and len(set(keys)) == len(keys)
    assert "down_blocks.1.attentions.0.transformer_blocks.0.attn1.processor" in keys
    assert "mid_block.attentions.0.transformer_blocks.0.attn2.processor" in keys
    assert "up_blocks.1.attentions.2.transformer_blocks.0.attn2.processor" in keys
    load_aid(unet, t=0.5, is_fused=True, atype="fused_outer", keep_original=True)      # the reference's form: wraps what was installed
    procs = unet.attn_processors
    assert list(procs.keys()) == keys                                                  # same names, same order
    assert all(isinstance(p, OuterInterpolatedAttnProcessor) and isinstance(p.original_attn, _Recorder) for p in procs.values())
    assert all(p.activated and p.size == 3 and p.coef.tolist() == [0.0, 0.5, 1.0] for p in procs.values())   # attn1 AND attn2 (App. D2)
    activate_aid(unet, 0.25)
    assert all(abs(float(p.coef[1]) - 0.25) < 1e-7 and p.activated for p in unet.attn_processors.values())
    deactivate_aid(unet)
    assert not any(p.activated for p in unet.attn_processors.values())
    load_aid(unet, t=0.3, atype="fused_inner")                                          # default: the HIP plain processor as fallback
    assert all(isinstance(p, InnerInterpolatedAttnProcessor) and isinstance(p.original_attn, HipAttnProcessor)
               for p in unet.attn_processors.values())
    with pytest.raises(ValueError, match="does not match the number of attention layers"):
        unet.set_attn_processor({"only.one.processor": _Recorder()})


def test_attention_forward_filters_kwargs_by_the_processor_signature():
    """diffusers drops every cross_attention_kwargs entry the processor's __call__ does not NAME.  ``ctx_index`` therefore reaches the
    AID processors (they name it) and a de-activated processor's wrapped original; an unknown keyword is dropped with a warning and
    never raises."""
    attn = Attention(80, None, heads=2, dim_head=40)
    rec = _Recorder()
    attn.set_processor(rec)
    x = torch.zeros(3, 4, 80)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        attn(x, encoder_hidden_states=None, ctx_index=[0, 1, 1], scale=0.5, no_such_keyword=1)
    assert rec.calls[-1]["ctx_index"] == [0, 1, 1]
    assert any("no_such_keyword" in str(m.message) and "scale" in str(m.message) for m in w)
    import inspect
    for cls in (OuterInterpolatedAttnProcessor, InnerInterpolatedAttnProcessor, HipAttnProcessor):
        names = set(inspect.signature(cls.__call__).parameters)
        assert {"attn", "hidden_states", "encoder_hidden_states", "attention_mask", "temb", "ctx_index"} <= names, cls
    names = set(inspect.signature(HipIPAdapterAttnProcessor.__call__).parameters)       # diffusers' IP processor signature
    assert {"attn", "hidden_states", "encoder_hidden_states", "attention_mask", "temb", "scale", "ip_adapter_masks"} <= names
    # a de-activated AID processor hands the call (and the map, when given) to what it wraps
    p = OuterInterpolatedAttnProcessor(t=0.5, is_fused=True, original_attn=rec)
    p.deactivate()
    attn.set_processor(p)
    attn(x, ctx_index=[0, 0, 0])
    assert rec.calls[-1]["ctx_index"] == [0, 0, 0]


def test_module_processors_are_registered_and_follow_the_unet():
    """The reference's processors subclass nn.Module so that the IP-Adapter weights they share follow ``unet.to(...)``
    (interpolation.py:10, 70-74): installed through set_processor they appear in the UNet's parameter tree."""
    unet = UNetDouble(cross_dim=96)
    for name, m in unet.named_modules():
        if name.endswith("attn2"):
            m.set_processor(IPAdapterAttnProcessor2_0(m.inner_dim, 96, num_tokens=(4,), scale=0.7))
        elif name.endswith("attn1"):
            m.set_processor(_Recorder())
    load_aid_ip_adapter(unet, t=0.5, is_fused=True, early="fused_outer", keep_original=True)
    names = dict(unet.named_parameters())
    assert "down_blocks.0.attentions.0.transformer_blocks.0.attn2.processor.ip_attn.to_k_ip.0.weight" in names
    unet.to(torch.float64)
    assert names["mid_block.attentions.0.transformer_blocks.0.attn2.processor.ip_attn.to_v_ip.0.weight"].dtype == torch.float64
    p2 = unet.attn_processors["mid_block.attentions.0.transformer_blocks.0.attn2.processor"]
    assert isinstance(p2, OuterInterpolatedIPAttnProcessor) and p2.scale is p2.ip_attn.scale and p2.num_tokens == (4,)
    unet.set_attn_processor({k: p.ip_attn for k, p in unet.attn_processors.items()})    # back to what diffusers' load_ip_adapter installed
    load_aid_ip_adapter(unet, t=0.5, early="scale_control")                              # default: HIP fallbacks sharing the adapter weights
    p2 = unet.attn_processors["mid_block.attentions.0.transformer_blocks.0.attn2.processor"]
    assert isinstance(p2.ip_attn, HipIPAdapterAttnProcessor)
    p1 = unet.attn_processors["mid_block.attentions.0.transformer_blocks.0.attn1.processor"]
    assert isinstance(p1.ip_attn, HipAttnProcessor)


# ---- GPU: numbers through the double ---------------------------------------------------------------------------------------------
DEV = "cuda:0"


def _oracle_unet(unet, streams, ctx, mode, fused, coef, ctx_rows=None):
    """The UNetDouble forward in fp64: per block  h += A(LN1(h));  h += A(LN2(h), ctx)."""
    out = {}
    blocks = list(unet.down_blocks) + [unet.mid_block] + list(unet.up_blocks)
    for (nm, *_), blk in zip(unet.LEVELS, blocks):
        h = to_np64(streams[nm])
        for tr in blk.attentions:
            for b in tr.transformer_blocks:
                for norm, attn, c in ((b.norm1, b.attn1, None), (b.norm2, b.attn2, ctx)):
                    w = O.AttnWeights(*(to_np64(t) for t in (attn.to_q.weight, attn.to_k.weight, attn.to_v.weight,
                                                              attn.to_out[0].weight, attn.to_out[0].bias)), attn.heads)
                    hn = to_np64(torch.from_numpy(O.layer_norm(h, to_np64(norm.weight), to_np64(norm.bias), norm.eps)).to(streams[nm].dtype))
                    cc = None if c is None else (to_np64(c) if ctx_rows is None else to_np64(c)[ctx_rows])
                    if mode == "plain":
                        a = O.plain_attention(hn, cc, w)
                    else:
                        a = (O.outer_attention if mode == "outer" else O.inner_attention)(hn, cc, w, coef, fused)
                    h = to_np64(torch.from_numpy(a).to(streams[nm].dtype)) + h
                    h = to_np64(torch.from_numpy(h).to(streams[nm].dtype))
        out[nm] = h
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("atype", ["fused_outer", "fused_inner"])
def test_unet_double_pass_activated_and_deactivated_vs_oracle(atype):
    dtype = torch.float16
    torch.manual_seed(5)
    unet = UNetDouble(cross_dim=96, dtype=dtype, device=DEV)
    with torch.no_grad():
        for p in unet.parameters():
            if p.ndim == 2:
                p.copy_(torch.randn_like(p, dtype=torch.float32) / p.shape[1] ** 0.5)
    g = torch.Generator().manual_seed(6)
    streams = unet.streams(3, g, dtype, DEV)
    ctx = torch.randn(3, 77, 96, generator=g).to(dtype).to(DEV)
    load_aid(unet, t=0.5, is_fused=True, atype=atype)
    activate_aid(unet, 0.3)
    mode = "outer" if atype == "fused_outer" else "inner"
    coef = torch.tensor([0.0, 0.3, 1.0]).to(dtype).float().numpy()
    got = unet(streams, ctx)
    ref = _oracle_unet(unet, streams, ctx, mode, True, coef)
    for nm in got:
        assert rel_l2(to_np64(got[nm]), ref[nm]) < 2 * TOL[dtype], nm        # five attention calls deep on a residual stream
    deactivate_aid(unet)
    got = unet(streams, ctx)
    ref = _oracle_unet(unet, streams, ctx, "plain", False, None)
    for nm in got:
        assert rel_l2(to_np64(got[nm]), ref[nm]) < 2 * TOL[dtype], nm


@pytest.mark.gpu
def test_ctx_index_through_cross_attention_kwargs_equals_repeated_contexts():
    """PAID guide prompt: the distinct contexts + a frame -> row map handed over as ``cross_attention_kwargs={"ctx_index": ...}`` —
    diffusers forwards the keyword because the processors name it — give the bits of the repeated-rows call."""
    dtype, n = torch.bfloat16, 5
    torch.manual_seed(7)
    unet = UNetDouble(cross_dim=96, dtype=dtype, device=DEV)
    g = torch.Generator().manual_seed(8)
    streams = unet.streams(n, g, dtype, DEV)
    distinct = torch.randn(3, 77, 96, generator=g).to(dtype).to(DEV)
    idx = [0, 1, 1, 1, 2]
    load_aid(unet, t=None, size=n, is_fused=True, atype="fused_outer", alpha=4, beta=4)
    a = unet(streams, distinct, cross_attention_kwargs={"ctx_index": idx})
    b = unet(streams, distinct[idx].contiguous())
    deactivate_aid(unet)
    c = unet(streams, distinct, cross_attention_kwargs={"ctx_index": idx})
    d = unet(streams, distinct[idx].contiguous())
    for nm in a:
        assert torch.equal(a[nm], b[nm]) and torch.equal(c[nm], d[nm]) and not torch.equal(a[nm], c[nm]), nm


@pytest.mark.gpu
def test_ip_adapter_unet_tuple_contexts_hip_fallback_equals_the_wrapped_diffusers_processor():
    """An IP-Adapter UNet hands every attention layer ``(text, [image_embeds])``.  De-activated, the reference calls the wrapped
    IPAdapterAttnProcessor2_0 (interpolation.py:248-251): ``keep_original=True`` runs the double's torch restatement of it, the
    default runs the HIP equivalent on the shared weights — same numbers within the storage tolerance.  Activated, the three IP
    variants run through the same dispatch."""
    dtype = torch.float16
    torch.manual_seed(9)
    unet = UNetDouble(cross_dim=96, dtype=dtype, device=DEV)
    for name, m in unet.named_modules():
        if name.endswith("attn2"):
            m.set_processor(IPAdapterAttnProcessor2_0(m.inner_dim, 96, num_tokens=(4,), scale=0.6, dtype=dtype, device=DEV))
        elif name.endswith("attn1"):
            m.set_processor(HipAttnProcessor())
    g = torch.Generator().manual_seed(10)
    streams = unet.streams(3, g, dtype, DEV)
    text = torch.randn(3, 77, 96, generator=g).to(dtype).to(DEV)
    ip = torch.randn(9, 1, 4, 96, generator=g).to(dtype).to(DEV)                       # 3 copies x {start, target, end}

    class Tuple_UNet(torch.nn.Module):                                                  # attn1 gets None, attn2 the tuple
        def __init__(self, u):
            super().__init__()
            self.u = u

        def forward(self, s):
            return self.u(s, (text, [ip]))
    # the double's BasicTransformerBlock passes encoder_hidden_states=None to attn1, the tuple to attn2
    load_aid_ip_adapter(unet, t=0.4, is_fused=True, early="fused_outer", keep_original=True)
    deactivate_aid(unet)
    ref = Tuple_UNet(unet)(streams)                                                    # wrapped diffusers processors (torch fp32 inside)
    originals = {k: p.ip_attn for k, p in unet.attn_processors.items()}
    unet.set_attn_processor(dict(originals))
    load_aid_ip_adapter(unet, t=0.4, is_fused=True, early="fused_outer")               # HIP fallbacks
    deactivate_aid(unet)
    got = Tuple_UNet(unet)(streams)
    for nm in got:
        assert rel_l2(to_np64(got[nm]), to_np64(ref[nm])) < 2 * TOL[dtype], nm
    for early in ("fused_outer", "fused_inner", "scale_control"):
        unet.set_attn_processor(dict(originals))
        load_aid_ip_adapter(unet, t=0.4, is_fused=True, early=early)
        out = Tuple_UNet(unet)(streams)
        assert all(torch.isfinite(v.float()).all() and not torch.equal(v, got[k]) for k, v in out.items()), early


@pytest.mark.gpu
def test_two_host_threads_two_streams_different_cu_share_are_bit_stable():
    """the library is re-entrant for distinct streams + workspaces.  Two host threads, each on its own stream with
    its own per-call cu_share hint (thread-local in ops), drive the projection shapes of a two-stream SDXL step and a processor
    call concurrently, 40 rounds: every result equals the single-threaded one bit for bit."""
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(17)
    shapes = ((7168, 1280, 1280), (28672, 640, 640), (3584, 1280, 1280))
    data = [((torch.randn(m, k, generator=g) * 0.5).to(dtype).to(DEV), (torch.randn(n, k, generator=g) * k ** -0.5).to(dtype).to(DEV),
             torch.randn(n, generator=g).to(dtype).to(DEV)) for m, n, k in shapes]
    attn = aid_amd.AttnShim(640, 10, dtype=dtype, device=DEV)
    xs = torch.randn(7, 1024, 640, generator=g).to(dtype).to(DEV)
    proc = OuterInterpolatedAttnProcessor(size=7, is_fused=True, alpha=5, beta=5)

    def work(share):
        with ops.cu_share(share):
            outs = [ops.linear(x, w, b) for x, w, b in data]
            outs.append(proc(attn, xs))
        return outs
    ref = work(0)
    torch.cuda.synchronize()
    errors, results = [], {}

    def runner(tag, share):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(40):
                    assert ops.current_cu_share() == 0
                    outs = work(share)
                    st.synchronize()
                    for o, r in zip(outs, ref):
                        if not torch.equal(o, r):
                            raise AssertionError(f"thread {tag}: result differs")
            results[tag] = True
        except Exception as e:          # noqa: BLE001
            errors.append((tag, repr(e)))
    ts = [threading.Thread(target=runner, args=("a", 2)), threading.Thread(target=runner, args=("b", 4))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors and results == {"a": True, "b": True}, errors
    assert ops.current_cu_share() == 0                                                  # the hint never leaks out of its block / thread


@pytest.mark.gpu
def test_capture_workspaces_belong_to_their_graphs():
    """ops.workspace: (1) a FOREIGN capture — the standard ``with torch.cuda.graph(g):`` idiom around processor calls, warmed up
    or not — gets its scratch from the graph's private pool and the dictionary forgets it once the capture has ended; (2) a capture of
    this package (under a WorkspaceOwner) ADOPTS the scratch of the eager warm-up on its stream, keeps it for itself, and drops it when the
    owner is released / collected — nothing lives for the life of the process any more."""
    import gc
    dtype = torch.float16
    attn = aid_amd.AttnShim(320, 8, dtype=dtype, device=DEV)
    x = torch.randn(3, 256, 320, device=DEV).to(dtype)
    proc = HipAttnProcessor()
    y = proc(attn, x).clone()
    # (1) foreign capture on a stream that never ran the library eagerly
    cold = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=cold):
        yc = proc(attn, x)
    assert any(k[1] == cold.cuda_stream for k in ops._pool_keys)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(yc, y)
    with torch.cuda.stream(cold):
        proc(attn, x)                                                                   # the next call on that stream: the entry is gone
    torch.cuda.synchronize()
    assert not any(k[1] == cold.cuda_stream for k in ops._pool_keys)
    assert not any(k[1] == cold.cuda_stream for k in ops._capture_ws)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(yc, y)                                                           # the pool still owns the scratch
    # (2) an owned capture adopts the warm-up's workspace
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        proc(attn, x)                                                                   # eager warm-up on the capture stream
    torch.cuda.synchronize()
    key = (torch.cuda.current_device(), st.cuda_stream)
    warm = ops._eager_ws[key]
    owner = ops.WorkspaceOwner()
    graph2 = torch.cuda.CUDAGraph()
    with owner, torch.cuda.graph(graph2, stream=st):
        yg = proc(attn, x)
    assert key not in ops._eager_ws and len(owner) == 1                                 # adopted: eager calls on this stream allocate afresh
    owned = [v for k, v in ops._capture_ws.items() if k[:2] == key]
    assert any(v.data_ptr() == warm.data_ptr() for v in owned)
    with torch.cuda.stream(st):
        proc(attn, x)                                                                   # an eager call on the SAME stream handle afterwards ...
    torch.cuda.synchronize()
    assert ops._eager_ws[key].data_ptr() != warm.data_ptr()                              # ... never shares the graph's scratch
    graph2.replay()
    torch.cuda.synchronize()
    assert torch.equal(yg, y)
    n_before = len(ops._capture_ws)
    del graph2, owned, warm
    del owner
    gc.collect()
    assert len(ops._capture_ws) == n_before - 1                                         # the owner's finalizer dropped the entry


@pytest.mark.gpu
def test_pipeline_runs_do_not_strand_capture_workspaces():
    """A long-lived process (the reference is a gradio app) calls the pipeline again and again: every call builds its own graphs and
    their workspaces must go with them."""
    import gc
    from aid_amd.loop import AidDenoiseLoop, install_sequence_processors
    dtype = torch.float16
    unet = aid_amd.AttnStackUNet("sd15", dtype=dtype, device=DEV, scale_down=16)
    n = 3
    install_sequence_processors(unet, n, "fused_inner")
    sample = {(s_, c_): torch.randn(n, s_, c_, device=DEV).to(dtype) for (s_, c_, _, _) in set(unet.shapes)}
    cond = torch.randn(n, 77, unet.cross_dim, device=DEV).to(dtype)
    base = None
    for it in range(4):
        loop = AidDenoiseLoop(unet, sample, cond, cond.clone(), num_inference_steps=4)
        for i in range(4):
            loop.step(i)
        torch.cuda.synchronize()
        assert len(loop._ws) >= 1
        del loop
        gc.collect()
        if base is None:
            base = len(ops._capture_ws)
        assert len(ops._capture_ws) == base, (it, len(ops._capture_ws), base)
