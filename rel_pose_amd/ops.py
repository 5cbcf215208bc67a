"""Host-side op layer: thin, typed wrappers over the C ABI (include/relpose_hip.h) plus the
torch.autograd.Function classes that stitch the HIP kernels into the reference's modules.

Granularity: one autograd Function per reference module on the hot path --
  TokensFn        src/model.py:136-141,170-171          (token layout + pos_embed)
  BlockFn         vision_transformer.py:349-354 (Block = LN, Attention :321-333, LN, Mlp mlp.py:20-26)
  CrossBlockFn    vision_transformer.py:285-296 (CrossBlock = LN, CrossAttention/EMM :188-238, LN, Mlp)
  HeadFn          src/model.py:178,189,91-98,145-159    (final LN, regressor, quaternion normalise)
so every residual add, bias, GELU/ReLU and their derivatives is fused into a kernel epilogue and autograd never
inserts an elementwise kernel of its own.  PyTorch is used for memory, streams and the autograd graph only.
All tensors are fp32, contiguous, on the GPU; anything else raises (there is no CPU path for the hot-path ops; only the CNN
front-end's bn_act / maxpool3x3s2 wrappers hand CPU tensors to the stock torch modules, for the fixture generator).
"""
import ctypes
import itertools
import math
import os
import threading

import torch

from . import _lib

N_TOK = 576
DIM = 192
HEADS = 3
LN_EPS = 1e-6
XW = 96          # padded width of the EMM's augmented value rows (64 + 6 -> 96)
GW = 224         # padded K of proj_fundamental (3*70 = 210 -> 224)
NWG = 6          # workgroup partials per (image, head) in rp_emm_apply


# ------------------------------------------------------------------------------------------------
# plumbing
# ------------------------------------------------------------------------------------------------
def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("rel_pose_amd ops need contiguous fp32 GPU tensors (no CPU fallback exists); got "
                               "%s %s contiguous=%s" % (t.device, t.dtype, t.is_contiguous()))


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _chk_act(*ts):
    """activation tensors of the CNN front-end kernels: contiguous, on the GPU, all fp32 or all bf16 (the bf16 configuration keeps the
    tensors between MIOpen's bf16 convolutions in bf16).  Returns 1 for bf16, 0 for fp32."""
    dt = None
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16)):
            raise RuntimeError("rel_pose_amd CNN ops need contiguous fp32 / bf16 GPU tensors; got %s %s contiguous=%s"
                               % (t.device, t.dtype, t.is_contiguous()))
        if dt is not None and t.dtype != dt:
            raise RuntimeError("rel_pose_amd CNN ops: mixed activation dtypes %s / %s" % (dt, t.dtype))
        dt = t.dtype
    return 1 if dt == torch.bfloat16 else 0


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _empty(*shape, like):
    return torch.empty(shape, device=like.device, dtype=torch.float32)


# how rp_gemm multiplies its fp32 operands (RpGemm.precision): 0 exact fp32 MFMA (default), 3 split-bf16 (3 limbs,
# fp32-grade, 1.45x on isolated GEMMs but only +1-2 % on the step: DESIGN.md section 4), 1 bf16 operands.
# Process-wide default; bench.py / tests switch it through set_gemm_precision().
GEMM_PRECISION = int(os.environ.get("RP_GEMM_PRECISION", "0"))


def set_gemm_precision(p):
    global GEMM_PRECISION
    if p not in (0, 1, 3):
        raise ValueError("precision must be 0 (fp32 MFMA), 3 (split-bf16, fp32-grade) or 1 (bf16 operands)")
    GEMM_PRECISION = p


# bf16 STORAGE of the MLP's hidden tensors in the bf16 configuration (operand precision 1): h = GELU(fc1), its pre-activation and the
# gradient of the pre-activation are [tokens, 768] -- four times the model width and 60 % of a Block's HBM traffic; the Linear kernels are
# HBM-bound there (profiles/r2_bf16_128pairs_full_step_summary.txt), and their MFMA operands are rounded to bf16 anyway.  fc1 writes them as
# bf16, fc2 / the weight-gradient / input-gradient GEMMs read bf16 (RpGemm.io_bf16).  Never in the fp32 configurations.
ACT_BF16 = os.environ.get("RP_ACT_BF16", "1") == "1"


def _act_bf16():
    return ACT_BF16 and GEMM_PRECISION == 1


# operand precision of the attention / EMM contractions (rp_attn_*, rp_emm_*: the `bf16` argument): 0 exact fp32 MFMA (default,
# the parity path), 1 bf16 operands on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- BASELINE.json configs[4]
ATTN_BF16 = 1 if os.environ.get("RP_ATTN_BF16", "0") == "1" else 0


def set_attention_precision(bf16):
    global ATTN_BF16
    ATTN_BF16 = 1 if bf16 else 0


# The bf16 DATA PATH of the bf16 configuration (BASELINE.json configs[4]; operand precision 1 + bf16 attention): q | k | v, the attention
# output and their gradients live in HBM as bf16 and the attention runs on csrc/attention_bf16.hip (LDS-DMA tiles, no conversions in the
# loops, recompute-form backward).  The residual stream, LayerNorm statistics and every accumulation stay fp32.  RP_BF16_PATH=0 keeps
# the round-3 form (fp32 storage, operands rounded per MFMA) for A/B.
BF16_PATH = os.environ.get("RP_BF16_PATH", "1") == "1"


def _bf16_path():
    return BF16_PATH and GEMM_PRECISION == 1 and ATTN_BF16 == 1


def gemm_tile(M, N, a_layout, b_layout, reads_mn=False, precision=None):
    """(TM, TN) that rp_gemm picks (mirrors csrc/gemm.hip)."""
    precision = GEMM_PRECISION if precision is None else precision
    if M <= 64:
        return 1, (1 if N <= 64 else 2)
    if precision:
        return (1, 3) if (a_layout == 1 and b_layout == 1 and N % 192 == 0 and not reads_mn) else (2, 1)
    if a_layout == 0 and b_layout == 0:
        tm, tn = 2, 1
    elif N % 192 == 0 and not reads_mn:
        tm, tn = 1, 3
    else:
        tm, tn = 2, 1
    return tm, tn


# ------------------------------------------------------------------------------------------------
# two-stream backward: for every Linear the input gradient (dX, on the critical path) and the parameter gradients
# (dW split-K GEMM, bias column sums) are independent, as are the two passes of the attention backward.  Issuing the
# off-critical-path work on a second HIP stream lets its kernels fill the ramp-up / tail of the critical-path kernels
# (each ~150 us GEMM otherwise spends ~10 % of its time with a partly empty chip) and hides the small column-sum / reduce
# launches entirely.  Fork = side waits for main; join = main waits for side (always before a backward returns).
# ------------------------------------------------------------------------------------------------
_SIDE = {}
USE_SIDE_STREAM = os.environ.get("RP_SIDE_STREAM", "0") == "1"   # measured: -0.7 ms on the 25 ms hot step, neutral on the
# full step, and it makes per-kernel profiles overlap -> opt-in


_PAD_GEN = 0


def invalidate_pad_cache():
    """Forget every cached alignment pad.  Needed after anything that rewrites parameters WITHOUT bumping Tensor._version:
    a HIP-graph replay of the captured optimizer step (graph.GraphedTrainStep) is the one such writer in this package."""
    global _PAD_GEN
    _PAD_GEN += 1


def _padded(t, pad):
    """F.pad(t, pad).contiguous() of a weight / bias, cached ON the tensor (attribute `_rp_pads`: the cache lives and dies with the
    parameter -- no id() reuse, nothing pinned after the model is deleted) until the tensor is next modified in place (optimizer step,
    load_state_dict, DDP broadcast all bump Tensor._version) or invalidate_pad_cache() is called.  Under stream capture nothing is
    cached: the pad is captured into the graph and re-executed by every replay, so a replayed forward sees the replayed weights."""
    if t.is_cuda and torch.cuda.is_current_stream_capturing():
        return torch.nn.functional.pad(t.detach(), pad).contiguous()
    pads = getattr(t, "_rp_pads", None)
    if pads is None:
        pads = {}
        try:
            t._rp_pads = pads
        except AttributeError:          # a tensor type without a __dict__: no caching
            return torch.nn.functional.pad(t.detach(), pad).contiguous()
    key = tuple(pad)
    hit = pads.get(key)
    if hit is not None and hit[0] == t._version and hit[1] == t.data_ptr() and hit[2] == _PAD_GEN:
        return hit[3]
    out = torch.nn.functional.pad(t.detach(), pad).contiguous()
    pads[key] = (t._version, t.data_ptr(), _PAD_GEN, out)
    return out


_WS_CACHE = {}


def _workspace(nbytes, device):
    """split-K slab buffer, one per (device, stream), grown on demand: launches on one stream are ordered, so consecutive
    split-K GEMMs can share it (a fresh torch.empty per launch was ~30 allocator calls per step)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS_CACHE.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = _WS_CACHE[key] = torch.empty(max(nbytes // 4, 1 << 20), device=device, dtype=torch.float32)
    return ws


class _Fork:
    def __init__(self, device):
        self.enabled = USE_SIDE_STREAM
        if self.enabled:
            key = str(device)
            if key not in _SIDE:
                _SIDE[key] = torch.cuda.Stream(device=device)
            self.side = _SIDE[key]
            self.main = torch.cuda.current_stream(device)

    def sync_side(self):
        """side stream waits for everything issued so far on the main stream"""
        if self.enabled:
            self.side.wait_stream(self.main)

    def on_side(self, fn, *a, **k):
        if not self.enabled:
            return fn(*a, **k)
        with torch.cuda.stream(self.side):
            return fn(*a, **k)

    def sync_main(self):
        """main stream waits for the side stream (required before the results are consumed / the backward returns)"""
        if self.enabled:
            self.main.wait_stream(self.side)


def pick_split_k(M, N, K, a_layout=0, b_layout=0, target_wgs=768, min_ktiles=8, max_split=128):
    tm, tn = gemm_tile(M, N, a_layout, b_layout)
    tiles = -(-M // (64 * tm)) * -(-N // (64 * tn))
    ktiles = -(-K // 32)
    if tiles >= 256 or ktiles < 2 * min_ktiles:
        return 1
    # ~3 workgroups per CU keeps >=2 waves per SIMD resident (measured: 0.76 waves/SIMD at 288 WGs left the MFMA pipe
    # 44 % busy); more than 128 partial slabs makes the reduce pass visible
    # floor, not ceil: tiles * split must not spill past the resident slots (9 tiles x 86 splits = 774 workgroups on 768 slots
    # ran a second, nearly empty round: 173 vs 137 us); whole groups of 8 because split z is pinned to XCD z % 8
    sk = max(1, min(ktiles // min_ktiles, target_wgs // tiles, max_split))
    if sk < 8:
        # split z runs on XCD z % 8 only (csrc/gemm.hip: the sharers of a K chunk meet in one L2), so fewer than 8 splits leave whole
        # XCDs idle: the regressor's [64,512] x [512,26880] input gradient took 74 us with 2 splits, 23.5 us with none
        return 1
    return sk - sk % 8 if sk >= 16 else sk


SURVEY = "*"      # KernelTimer instance that times EVERY tagged launch (and every rp_gemm instance), kept per tag


class KernelTimer:
    """HIP-event timing of every launch of ONE kernel on torch's current stream (bench.py's `roofline` object).  Off unless bench.py
    installs one.  `instance` is either a gemm_kernel<a_layout,b_layout,TM,TN> tuple (events recorded by rp_gemm itself around the
    main kernel, so a split-K launch's reduce is outside the span) or a string tag of one of the other MFMA kernels (`timed(tag, ...)`
    call sites below: the events bracket the one C-ABI call, on the stream it launches on).
    instance = SURVEY ("*"): every tagged call site and every rp_gemm instance is timed and kept PER TAG (`per_tag`, read with
    survey()): bench.py runs a few untimed steps this way to find which own kernel the step spends most time in, and then times
    THAT one over the judged steps (the roofline object must follow the dominant kernel, not a hard-coded symbol)."""

    def __init__(self, instance):
        self.instance = instance if isinstance(instance, str) else tuple(instance)
        self.events = []
        self.tevents = []         # torch.cuda.Event pairs of tagged (non-rp_gemm) launches
        self.flops = 0.0
        self.bytes = 0.0          # compulsory bytes: A + B + C (+ [M,N] epilogue operands), each once
        self.enabled = False
        self.per_tag = {}         # SURVEY: tag (str, or the rp_gemm instance tuple) -> [event pairs, flops, bytes]

    def reset(self):
        """forget the recorded launches; their hipEvent handles go back to the free list"""
        if not hasattr(self, "_pool"):
            self._pool = []
        for s, e in self.events:
            self._pool += [s, e]
        self.events, self.tevents, self.flops, self.bytes, self.per_tag = [], [], 0.0, 0.0, {}

    def wants(self, tag):
        return self.enabled and (self.instance == SURVEY or self.instance == tag)

    def add(self, tag, pair, flops, nbytes):
        """one timed launch (torch event pair)"""
        if self.instance == SURVEY:
            ent = self.per_tag.setdefault(tag, [[], 0.0, 0.0])
            ent[0].append(pair)
            ent[1] += flops
            ent[2] += nbytes
        else:
            self.tevents.append(pair)
            self.flops += flops
            self.bytes += nbytes

    def survey(self):
        """-> [(tag, launches, total seconds, total algorithmic flops, total algorithmic bytes)], largest total time first; call
        after a device sync."""
        rows = []
        for tag, (pairs, fl, by) in self.per_tag.items():
            rows.append((tag, len(pairs), sum(s.elapsed_time(e) for s, e in pairs) * 1e-3, fl, by))
        return sorted(rows, key=lambda r: -r[2])

    def summary(self):
        """-> (launches, mean seconds per launch, total algorithmic flops); call after a device sync."""
        lib = _lib.load()
        n = len(self.events) + len(self.tevents)
        tot = sum(lib.rp_event_elapsed_ms(s, e) for s, e in self.events) * 1e-3
        tot += sum(s.elapsed_time(e) for s, e in self.tevents) * 1e-3
        return n, (tot / n if n else 0.0), self.flops

    def new_pair(self):
        """two hipEvent handles from a free list (events of earlier summaries are reused after reset())"""
        lib = _lib.load()
        if not hasattr(self, "_pool"):
            self._pool = []
        if not self._pool:
            self._pool = [ctypes.c_void_p(lib.rp_event_create()) for _ in range(64)]
        return self._pool.pop(), self._pool.pop()


class timed:
    """`with timed("attn_fwd", flops, nbytes): lib.rp_...(...)` -- a no-op unless bench.py's timer targets this tag."""

    def __init__(self, tag, flops, nbytes):
        tm = TIMER
        self.tm = tm if (tm is not None and tm.wants(tag)) else None
        self.tag, self.flops, self.nbytes = tag, flops, nbytes

    def __enter__(self):
        if self.tm is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, et, ev, tb):
        if self.tm is not None and et is None:
            self.e1.record()
            self.tm.add(self.tag, (self.e0, self.e1), self.flops, self.nbytes)
        return False


TIMER = None

# ------------------------------------------------------------------------------------------------
# deferred split-K reduces: inside `with splitk_batch():` a split-K product that ASKS for it (gemm(..., defer=True): the weight-gradient
# GEMMs of linear_dw) launches only its main kernel into a private slab region and the block ends with ONE rp_splitk_reduce_multi
# (bit-identical to the per-GEMM reduce).  Those outputs are only FILLED at exit: return them, do not compute with them inside the
# block.  Every other split-K gemm() inside the block reduces immediately as usual.  The batch state is per THREAD (autograd runs one
# backward thread per device) and is honoured only on the stream it was opened on; a block opened inside another one (a re-entrant
# backward: torch.autograd.grad or a checkpoint recompute inside a BlockFn.backward) does not share the outer block's slab arena --
# its products reduce immediately, and the outer block resumes deferring when it closes.
# ------------------------------------------------------------------------------------------------
# Kernel-path selectors WITHOUT an environment switch (round 6: every alternative below lost its interleaved A/B two or more rounds
# ago; the RP_* variables that used to flip them are retired).  They stay module attributes because the alternatives are real code --
# the fallbacks other shapes and variants take -- and tests/test_gpu_entrypoints.py flips each one on a whole training step so that no
# fallback rots: SPLITK_BATCHING, ROWS_LINEAR, ROWS_DX, FUSE_LN_BWD, DX_LNBWD_BF16, COLSUM_BATCHING, QKV_BIAS_FROM_PRODUCERS,
# EMM_STATS_ONE_PASS, FUSE_MLP, FUSE_MLP_TRAIN, FUSE_MLP_BWD, MLP_W2_CHUNK_MAJOR, STEM_CONV, STEM_STATS, FUSE_STEM_POOL.
SPLITK_BATCHING = True
_TLS = threading.local()
_ARENA = {}


def _sk_batch():
    return getattr(_TLS, "batch", None)


def _arena_take(nbytes, device, state):
    """nbytes (rounded to 256) of a grow-only per-(device, stream) arena; `state` = [offset] of the enclosing batch."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    n4 = (nbytes + 255) // 256 * 64
    buf = _ARENA.get(key)
    if buf is None or buf.numel() < state[0] + n4:
        # (an arena outgrown mid-batch is replaced; the slabs already handed out stay alive through the views the tasks hold)
        buf = _ARENA[key] = torch.empty(max(2 * (state[0] + n4), 16 << 20), device=device, dtype=torch.float32)
        state[0] = 0
    out = buf[state[0]:state[0] + n4]
    state[0] += n4
    return out


class splitk_batch:
    def __enter__(self):
        st = torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else None
        self.prev = _sk_batch()
        nested = getattr(_TLS, "depth", 0) > 0
        _TLS.depth = getattr(_TLS, "depth", 0) + 1
        _TLS.batch = ([], [0], st) if (SPLITK_BATCHING and not nested) else None
        return self

    def __exit__(self, et, ev, tb):
        cur, _TLS.batch = _sk_batch(), self.prev
        _TLS.depth -= 1
        if et is None and cur and cur[0]:
            tasks = cur[0]
            lib = _lib.load()
            for i in range(0, len(tasks), _lib.RP_SPLITK_MAX):
                chunk = tasks[i:i + _lib.RP_SPLITK_MAX]
                arr = (_lib.RpSplitkTask * len(chunk))()
                for a, (ws, out, M, N, ldc, sk, tr) in zip(arr, chunk):
                    a.ws, a.C, a.M, a.N, a.ldc, a.split_k, a.trans_c = ws.data_ptr(), out.data_ptr(), M, N, ldc, sk, 1 if tr else 0
                _lib.check(lib.rp_splitk_reduce_multi(arr, len(chunk), _st()), "rp_splitk_reduce_multi")
        return False


def gemm_instance(M, N, a_layout, b_layout, reads_mn=False):
    """(a_layout, b_layout, TM, TN) that rp_gemm dispatches to (mirrors the selection in csrc/gemm.hip)."""
    return (a_layout, b_layout) + gemm_tile(M, N, a_layout, b_layout, reads_mn)


def gemm(A, B, M, N, K, *, a_layout=0, b_layout=0, lda=None, ldb=None, out=None, ldc=None, bias=None, act=0,
         pre_out=None, dact=0, aux=None, residual=None, split_k=None, batch=1, strides=(0, 0, 0), trans_c=False,
         precision=None, want_colsum=False, ln=None, out_dtype=None, defer=False):
    """C[M,N] = epilogue(op(A) op(B)); see RpGemm in include/relpose_hip.h.
    ln = (x, mean, rstd, gamma, part): LayerNorm backward fused into the epilogue (RpGemm.ln_*).
    bf16 storage (RpGemm.io_bf16, operand precision 1 only): A and aux may be bf16 tensors; out_dtype=torch.bfloat16 makes C (and
    pre_out) bf16."""
    lib = _lib.load()
    _chk(B, bias, residual)
    io = 0
    bfd = torch.bfloat16
    if A.dtype == bfd:
        io |= 1
    if (out_dtype == bfd) or (out is not None and out.dtype == bfd):
        io |= 2
    if aux is not None and aux.dtype == bfd:
        io |= 4
    for t, bit in ((A, 1), (out, 2), (pre_out, 2), (aux, 4)):
        if t is None:
            continue
        want = bfd if io & bit else torch.float32
        if not (t.is_cuda and t.is_contiguous() and t.dtype == want):
            raise RuntimeError("rp_gemm operand: expected a contiguous %s GPU tensor, got %s %s" % (want, t.device, t.dtype))
    if io and (GEMM_PRECISION if precision is None else precision) != 1:
        raise RuntimeError("bf16-stored GEMM operands need operand precision 1 (the bf16 configuration)")
    if io & 6:
        # mirrors rp_gemm: a bf16 C / pre_out / aux is only honoured by the LDS-staged epilogue modes, never by the generic
        # per-element epilogue or the split-K reduce
        if dact == 0:
            staged = ((act == 0 and pre_out is None) or (act == 1 and bias is not None and residual is None)
                      or (act == 2 and bias is not None and residual is None and pre_out is None))
        else:
            staged = bias is None and pre_out is None and residual is None and act == 0 and aux is not None
        if not staged or (a_layout == 1 and b_layout == 1):
            raise RuntimeError("bf16-stored C / pre_out / aux: unsupported epilogue (act=%d dact=%d bias=%s pre_out=%s residual=%s) "
                               "or layout" % (act, dact, bias is not None, pre_out is not None, residual is not None))
        if (io & 4) and split_k is not None and split_k > 1:
            raise RuntimeError("bf16-stored aux cannot be combined with split-K")
        if split_k is None:
            split_k = 1              # (the staged modes are forward / input-gradient shapes; never auto-split a bf16-output product)
    if lda is None:
        lda = K if a_layout == 0 else M
    if ldb is None:
        ldb = K if b_layout == 0 else N
    if out is None:
        out = torch.empty((batch, M, N) if batch > 1 else (M, N), device=A.device, dtype=bfd if io & 2 else torch.float32)
    if ldc is None:
        ldc = N
    if split_k is None:
        split_k = pick_split_k(M, N, K, a_layout, b_layout) if batch == 1 and N % 4 == 0 else 1
    g = _lib.RpGemm()
    g.A, g.B, g.C = A.data_ptr(), B.data_ptr(), out.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = lda, ldb, ldc
    g.a_layout, g.b_layout, g.batch = a_layout, b_layout, batch
    g.stride_a, g.stride_b, g.stride_c = strides
    g.split_k = split_k
    ws = None
    defer = None
    if split_k > 1:
        nbytes = lib.rp_gemm_workspace_bytes(M, N, split_k)
        plain = bias is None and pre_out is None and aux is None and residual is None and act == 0 and dact == 0 and ln is None
        if (defer and _sk_batch() is not None and plain and not want_colsum
                and torch.cuda.current_stream(A.device).cuda_stream == _sk_batch()[2]):       # (not under fork.on_side)
            ws = _arena_take(nbytes, A.device, _sk_batch()[1])
            g.defer_reduce = 1
            defer = (ws, out, M, N, ldc, split_k, trans_c)
        else:
            ws = _workspace(nbytes, A.device)
        g.workspace, g.workspace_bytes = ws.data_ptr(), nbytes
    g.bias = None if bias is None else bias.data_ptr()
    g.pre_out = None if pre_out is None else pre_out.data_ptr()
    g.act, g.dact = act, dact
    g.aux = None if aux is None else aux.data_ptr()
    g.residual = None if residual is None else residual.data_ptr()
    g.trans_c = 1 if trans_c else 0
    g.precision = GEMM_PRECISION if precision is None else precision
    g.io_bf16 = io
    if ln is not None:
        _chk(*ln)
        g.ln_x, g.ln_mean, g.ln_rstd, g.ln_gamma, g.ln_part = (t.data_ptr() for t in ln)
        split_k = g.split_k = 1
    cpart = None
    if want_colsum:      # column sums of the stored values, folded into the epilogue (see RpGemm.colsum_part)
        tm_, tn_ = gemm_tile(M, N, a_layout, b_layout, aux is not None or residual is not None, g.precision)
        if tn_ > 2 or split_k != 1 or batch != 1:
            raise RuntimeError("want_colsum needs a TN <= 2 tile without split-K (pass aux/residual GEMMs)")
        cpart = _empty(2 * (-(-M // (64 * tm_))), N, like=A)
        g.colsum_part = cpart.data_ptr()
    tm = TIMER
    if tm is not None and tm.enabled and tm.instance == SURVEY:
        # survey: the whole call (a split-K launch's reduce included) between two torch events, kept under the instance tuple
        ob = 2.0 if (out_dtype == torch.bfloat16) else 4.0
        nby = batch * (M * K * float(A.element_size()) + 4.0 * N * K + M * N * (ob * (1 + (pre_out is not None))
                       + (float(aux.element_size()) if aux is not None else 0.0) + (4.0 if residual is not None else 0.0)))
        with timed(gemm_instance(M, N, a_layout, b_layout, aux is not None or residual is not None), 2.0 * M * N * K * batch, nby):
            _lib.check(lib.rp_gemm(ctypes.byref(g), _st()), "rp_gemm")
        if defer is not None and g.split_k > 1:
            _sk_batch()[0].append(defer)
        return (out, colsum(cpart)) if want_colsum else out
    if (tm is not None and tm.enabled and batch == 1 and ln is None and not isinstance(tm.instance, str) and
            gemm_instance(M, N, a_layout, b_layout, aux is not None or residual is not None) == tm.instance):
        # events are recorded by rp_gemm itself around the MAIN kernel (a split-K launch's reduce is a separate kernel)
        e0, e1 = tm.new_pair()
        g.ev_start, g.ev_stop = e0, e1
        _lib.check(lib.rp_gemm(ctypes.byref(g), _st()), "rp_gemm")
        if defer is not None and g.split_k > 1:
            _sk_batch()[0].append(defer)
        tm.events.append((e0, e1))
        tm.flops += 2.0 * M * N * K * batch
        ob = 2.0 if (out_dtype == torch.bfloat16) else 4.0                     # bf16-stored operands (RpGemm.io_bf16) count 2 bytes
        tm.bytes += batch * (M * K * float(A.element_size()) + 4.0 * N * K + M * N * (ob * (1 + (pre_out is not None))
                             + (float(aux.element_size()) if aux is not None else 0.0) + (4.0 if residual is not None else 0.0)))
        return (out, colsum(cpart)) if want_colsum else out
    _lib.check(lib.rp_gemm(ctypes.byref(g), _st()), "rp_gemm")
    if defer is not None and g.split_k > 1:
        _sk_batch()[0].append(defer)
    return (out, colsum(cpart)) if want_colsum else out


# ------------------------------------------------------------------------------------------------
# W^T copies for the row-resident input-gradient kernels (rp_linear_rows192 on the transposed weight, rp_mlp_fused_bwd): every weight
# that will need one registers at forward time; the first request of a step transposes ALL registered weights whose copy is stale in ONE
# rp_transpose_multi launch (round 2: 17 separate `.t().contiguous()` copies per step), later requests hit the cache.
# ------------------------------------------------------------------------------------------------
_T_REGISTRY = []      # weakrefs of the registered weights


def register_transposed(*ws):
    import weakref
    for w in ws:
        if getattr(w, "_rp_treg", False):
            continue
        try:
            w._rp_treg = True
        except AttributeError:
            continue
        _T_REGISTRY.append(weakref.ref(w))


def _t_fresh(w):
    c = getattr(w, "_rp_t", None)
    return c is not None and c[0] == w._version and c[1] == w.data_ptr() and c[2] == _PAD_GEN


def transposed(w):
    """contiguous W^T of a 2-D fp32 GPU weight, cached on the tensor until it is modified (see _padded for the invalidation rules)"""
    if _t_fresh(w):
        return w._rp_t[3]
    if (w.is_cuda and torch.cuda.is_current_stream_capturing()) or not w.is_cuda or w.dim() != 2:
        return w.detach().t().contiguous()
    todo, alive = [], []
    for r in _T_REGISTRY:
        t = r()
        if t is None:
            continue
        alive.append(r)
        if t is not w and t.is_cuda and t.device == w.device and not _t_fresh(t):
            todo.append(t)
    _T_REGISTRY[:] = alive
    todo = [w] + todo[:_lib.RP_TRANSPOSE_MAX - 1]
    lib = _lib.load()
    arr = (_lib.RpTransposeTask * len(todo))()
    outs = []
    for a, t in zip(arr, todo):
        src = t.detach()
        if not src.is_contiguous():
            src = src.contiguous()
        o = torch.empty(t.shape[1], t.shape[0], device=t.device, dtype=torch.float32)
        a.src, a.dst, a.rows, a.cols = src.data_ptr(), o.data_ptr(), t.shape[0], t.shape[1]
        outs.append((t, src, o))
    _lib.check(lib.rp_transpose_multi(arr, len(todo), _st()), "rp_transpose_multi")
    for t, _, o in outs:
        try:
            t._rp_t = (t._version, t.data_ptr(), _PAD_GEN, o)
        except AttributeError:
            pass
    return outs[0][2]


def bf16_weight(w):
    """bf16 copy (round to nearest even) of a weight for the bf16 configuration's row-resident Linear kernel, cached on the tensor
    until it is modified (same invalidation rules as transposed())."""
    c = getattr(w, "_rp_b", None)
    if c is not None and c[0] == w._version and c[1] == w.data_ptr() and c[2] == _PAD_GEN:
        return c[3]
    o = w.detach().contiguous().to(torch.bfloat16)
    if not (w.is_cuda and torch.cuda.is_current_stream_capturing()):
        try:
            w._rp_b = (w._version, w.data_ptr(), _PAD_GEN, o)
        except AttributeError:
            pass
    return o


# K = 192 Linear layers on the row-resident kernel (csrc/linear_rows.hip) instead of the generic LDS-DMA GEMM: exact fp32, and the
# bf16 configuration (operand precision 1) on the same kernel with v_mfma_f32_16x16x32_bf16
ROWS_LINEAR = True


ROWS_DX = True      # input-gradient GEMMs that contract over 192 (fc2, proj) likewise


def _rows_ok(x, W):
    return (ROWS_LINEAR and GEMM_PRECISION in (0, 1) and (x.dtype == torch.float32 or (x.dtype == torch.bfloat16 and GEMM_PRECISION == 1))
            and x.shape[1] == DIM and W.shape[1] == DIM and W.shape[0] % 32 == 0 and W.shape[0] <= 1024 and x.is_contiguous() and W.is_contiguous())


def linear_rows(x, W, b=None, act=0, want_pre=False, residual=None, ln=None, want_ln_out=False, dact_aux=None, want_colsum=False,
                out_dtype=None, xn_dtype=None):
    """rp_linear_rows192: y = act(LN?(x) W^T + b) (+ residual) for K = 192.  ln = (gamma, beta) fuses the LayerNorm;
    want_ln_out additionally returns (xn, mean, rstd).  dact_aux [M,N]: y *= GELU'(aux); want_colsum: also the column sums of y
    (from per-tile partials).  Returns y [, pre] [, xn, mean, rstd] [, colsum].  At operand precision 1 (the bf16 configuration)
    dact_aux may be a bf16 tensor, x may be a bf16 tensor (no LayerNorm), out_dtype=torch.bfloat16 stores y (and pre) as bf16 and
    xn_dtype=torch.bfloat16 the normalised rows."""
    lib = _lib.load()
    _chk(W, b, residual)
    _chk_act(x)
    _chk_act(dact_aux)
    M, K = x.shape
    N = W.shape[0]
    obf = out_dtype == torch.bfloat16
    xnbf = xn_dtype == torch.bfloat16 and ln is not None and want_ln_out
    io = ((1 if x.dtype == torch.bfloat16 else 0) | (2 if obf else 0) | (4 if (dact_aux is not None and dact_aux.dtype == torch.bfloat16) else 0)
          | (8 if xnbf else 0))
    if (io & 1) and ln is not None:
        raise RuntimeError("the fused LayerNorm reads the fp32 residual stream, not bf16 rows")
    if io and GEMM_PRECISION != 1:
        raise RuntimeError("bf16-stored operands need operand precision 1 (the bf16 configuration)")
    y = torch.empty(M, N, device=x.device, dtype=torch.bfloat16) if obf else _empty(M, N, like=x)
    pre = (torch.empty(M, N, device=x.device, dtype=torch.bfloat16) if obf else _empty(M, N, like=x)) if want_pre else None
    xn = mean = rstd = None
    g = be = None
    if ln is not None:
        g, be = ln
        _chk(g, be)
        if want_ln_out:
            xn = torch.empty(M, K, device=x.device, dtype=torch.bfloat16 if xnbf else torch.float32)
            mean, rstd = _empty(M, like=x), _empty(M, like=x)
    part = _empty(-(-M // lib.rp_linear_rows192_tile_rows()), N, like=x) if want_colsum else None
    nmn = 1 + (pre is not None) + (residual is not None) + (dact_aux is not None)
    with timed("linear_rows_ln" if ln is not None else "linear_rows", 2.0 * M * N * K,
               4.0 * (M * K * (1 + (xn is not None)) + N * K + M * N * nmn)):
        wk = bf16_weight(W) if GEMM_PRECISION == 1 else W
        _lib.check(lib.rp_linear_rows192(_p(x), _p(wk), _p(b), _p(residual), _p(g), _p(be), LN_EPS, _p(y), _p(pre), _p(xn), _p(mean),
                                         _p(rstd), _p(dact_aux), _p(part), M, N, K, act, GEMM_PRECISION, io, _st()),
                   "rp_linear_rows192")
    out = ((y,) + ((pre,) if want_pre else ()) + ((xn, mean, rstd) if (ln is not None and want_ln_out) else ())
           + ((colsum(part),) if want_colsum else ()))
    return out[0] if len(out) == 1 else out


def ln_linear(x, gamma, beta, W, b, act=0, want_pre=False, train=True, out_dtype=None, xn_dtype=None):
    """(y [, pre], xn, mean, rstd) of  act(LayerNorm(x) W^T + b): one kernel when the row-resident path applies (xn / stats are
    None at inference), LayerNorm kernel + GEMM otherwise."""
    if _rows_ok(x, W):
        r = linear_rows(x, W, b, act=act, want_pre=want_pre, ln=(gamma, beta), want_ln_out=train, out_dtype=out_dtype, xn_dtype=xn_dtype)
        r = r if isinstance(r, tuple) else (r,)
        return r if train else r + (None, None, None)
    xn, m, rs = layernorm_fwd(x, gamma, beta)
    r = linear(xn, W, b, act=act, want_pre=want_pre, out_dtype=out_dtype)
    return (r if isinstance(r, tuple) else (r,)) + (xn, m, rs)


def linear(x, W, b=None, act=0, want_pre=False, residual=None, out_dtype=None):
    """y = act(x W^T + b) (+ residual); x [M,K], W [N,K].  out_dtype=torch.bfloat16: y (and the pre-activation) stored as bf16
    (bf16 configuration only)."""
    if _rows_ok(x, W) and (residual is None or residual.is_contiguous()) and (out_dtype is None or residual is None):
        return linear_rows(x, W, b, act=act, want_pre=want_pre, residual=residual, out_dtype=out_dtype)
    M, K = x.shape
    N = W.shape[0]
    pre = torch.empty(M, N, device=x.device, dtype=out_dtype or torch.float32) if want_pre else None
    y = gemm(x, W, M, N, K, bias=b, act=act, pre_out=pre, residual=residual, out_dtype=out_dtype)
    return (y, pre) if want_pre else y


def linear_dx(dy, W, dact=0, aux=None, want_colsum=False, out_dtype=None):
    """dx = (dy W) o act'(aux); dy [M,N], W [N,K] -> [M,K].  want_colsum: also sum_m dx[m][:] (the bias gradient of the
    layer below when dx is its pre-activation gradient), from the epilogue."""
    M, N = dy.shape
    K = W.shape[1]
    if (ROWS_DX and N == DIM and K % 32 == 0 and K <= 1024 and GEMM_PRECISION in (0, 1) and dy.is_contiguous() and dact in (0, 1)
            and (dy.dtype == torch.float32 or GEMM_PRECISION == 1)):
        # contraction over the layer's 192 outputs: the row-resident kernel on the transposed weight (a 0.1-0.6 MB copy)
        return linear_rows(dy, transposed(W), dact_aux=aux if dact else None, want_colsum=want_colsum, out_dtype=out_dtype)
    return gemm(dy, W, M, K, N, b_layout=1, dact=dact, aux=aux, want_colsum=want_colsum, out_dtype=out_dtype)


# LayerNorm backward fused into the epilogue of the input-gradient GEMM that feeds it (RpGemm.ln_*): the exact-fp32 GEMM only
FUSE_LN_BWD = True
DX_LNBWD_BF16 = True      # A/B aid


def linear_dx_lnbwd(dy, W, x, gamma, mean, rstd, add=None):
    """The pair  dxn = dy W ; (dx, dgamma, dbeta[, colsum(add)]) = layernorm_bwd(dxn, x, gamma, mean, rstd, add)  as ONE
    GEMM: dxn (the gradient of the LayerNorm output, [M,192]) never goes to memory.  dy [M,N], W [N,192]."""
    if not FUSE_LN_BWD or W.shape[1] != DIM:
        return layernorm_bwd(linear_dx(dy, W), x, gamma, mean, rstd, add=add)
    M, N = dy.shape
    if DX_LNBWD_BF16 and dy.dtype == torch.bfloat16 and GEMM_PRECISION == 1 and N == 3 * DIM and dy.is_contiguous():
        # the bf16 data path: output-resident kernel, dY read once as MFMA operands (csrc/dx_lnbwd_bf16.hip)
        lib = _lib.load()
        _chk(x, gamma, mean, rstd, add)
        np_ = 3 if add is not None else 2
        part = _empty(-(-M // lib.rp_dx_lnbwd_bf16_tile_rows()), np_ * DIM, like=x)
        dx = torch.empty_like(x)
        wt = bf16_weight(transposed(W))
        with timed("dx_lnbwd_bf16", 2.0 * M * N * DIM, M * (2.0 * N + 4.0 * DIM * (3 if add is not None else 2))):
            _lib.check(lib.rp_dx_lnbwd_bf16(_p(dy), _p(wt), _p(x), _p(gamma), _p(mean), _p(rstd), _p(add), _p(dx), _p(part), M, N, _st()),
                       "rp_dx_lnbwd_bf16")
        sums = colsum(part)
        if add is None:
            return dx, sums[:DIM], sums[DIM:]
        return dx, sums[:DIM], sums[DIM:2 * DIM], sums[2 * DIM:]
    np_ = 3 if add is not None else 2
    part = _empty(-(-M // 64), np_ * DIM, like=dy)
    dx = gemm(dy, W, M, DIM, N, b_layout=1, residual=add, ln=(x, mean, rstd, gamma, part))
    sums = colsum(part)
    if add is None:
        return dx, sums[:DIM], sums[DIM:]
    return dx, sums[:DIM], sums[DIM:2 * DIM], sums[2 * DIM:]


DW192 = os.environ.get("RP_DW192", "1") == "1"      # A/B aid: the streaming bf16 weight-gradient kernel of the bf16 data path


DW192_F32 = os.environ.get("RP_DW192_F32", "1") == "1"      # A/B aid: the output-stationary exact-fp32 weight-gradient kernel
# OPT-IN (the split-bf16x3 gate): the same weight gradients on the bf16 matrix pipe from on-chip 3-limb splits of the
# fp32 operands (csrc/dw192_split3.hip; error vs fp64 <= the fp32 MFMA kernel's).  The default stays exact fp32 MFMA.
DW_SPLIT3 = os.environ.get("RP_DW_SPLIT3", "0") == "1"


def _dw192(a, b, out, trans):
    """slabs of a^T b by rp_dw192_bf16 (a [M,N] bf16, b [M,192] bf16 / fp32) or rp_dw192_f32 (both fp32, exact) + the fixed-order
    split-K reduce into `out` ([N,192], or [192,N] when trans) -- deferred into the enclosing splitk_batch like every other
    weight gradient."""
    lib = _lib.load()
    M, N = a.shape
    f32 = a.dtype == torch.float32
    sk = (lib.rp_dw192_f32_splits if f32 else lib.rp_dw192_bf16_splits)(M, N)
    nbytes = (lib.rp_dw192_f32_workspace_bytes if f32 else lib.rp_dw192_bf16_workspace_bytes)(M, N)
    deferred = (_sk_batch() is not None and torch.cuda.current_stream(a.device).cuda_stream == _sk_batch()[2])
    ws = _arena_take(nbytes, a.device, _sk_batch()[1]) if deferred else _workspace(nbytes, a.device)
    if f32 and DW_SPLIT3:
        with timed("dw192_split3", 2.0 * M * N * DIM, 4.0 * (M * (N + DIM) + sk * N * DIM)):
            _lib.check(lib.rp_dw192_split3(_p(a), N, _p(b), M, N, _p(ws), nbytes, _st()), "rp_dw192_split3")
    elif f32:
        with timed("dw192_f32", 2.0 * M * N * DIM, 4.0 * (M * (N + DIM) + sk * N * DIM)):
            _lib.check(lib.rp_dw192_f32(_p(a), N, _p(b), M, N, _p(ws), nbytes, _st()), "rp_dw192_f32")
    else:
        with timed("dw192_bf16" if b.dtype == torch.bfloat16 else "dw192_bf16_f32b", 2.0 * M * N * DIM,
                   M * (2.0 * N + b.element_size() * DIM) + 4.0 * sk * N * DIM):
            _lib.check(lib.rp_dw192_bf16(_p(a), N, _p(b), 1 if b.dtype == torch.float32 else 0, M, N, _p(ws), nbytes, _st()), "rp_dw192_bf16")
    task = (ws, out, N, DIM, N if trans else DIM, sk, trans)
    if deferred:
        _sk_batch()[0].append(task)
    else:
        arr = (_lib.RpSplitkTask * 1)()
        arr[0].ws, arr[0].C, arr[0].M, arr[0].N, arr[0].ldc, arr[0].split_k, arr[0].trans_c = ws.data_ptr(), out.data_ptr(), N, DIM, task[4], sk, 1 if trans else 0
        _lib.check(lib.rp_splitk_reduce_multi(arr, 1, _st()), "rp_splitk_reduce_multi")
    return out


def linear_dw(dy, x):
    """dW = dy^T x; dy [M,N], x [M,K] -> [N,K]  (reduction over the M token rows, split-K)."""
    M, N = dy.shape
    K = x.shape[1]
    if DW192 and GEMM_PRECISION == 1 and M % 64 == 0 and M >= 4096 and dy.is_contiguous() and x.is_contiguous():
        bfd = torch.bfloat16
        # one operand bf16 with a width that is a multiple of 192 (the streamed "A"), the other [M,192] bf16 or fp32
        if dy.dtype == bfd and N % DIM == 0 and K == DIM and (N >= K or x.dtype != bfd):
            return _dw192(dy, x, torch.empty(N, K, device=dy.device, dtype=torch.float32), False)
        if x.dtype == bfd and K % DIM == 0 and N == DIM:
            return _dw192(x, dy, torch.empty(N, K, device=dy.device, dtype=torch.float32), True)
    if (DW192_F32 and GEMM_PRECISION == 0 and M % 32 == 0 and M >= 4096 and dy.dtype == torch.float32 and x.dtype == torch.float32
            and dy.is_contiguous() and x.is_contiguous()):
        # exact fp32: the wider operand streams as "A" of the output-stationary kernel, the 192-wide one as "B"
        if N % DIM == 0 and K == DIM and N >= K:
            return _dw192(dy, x, torch.empty(N, K, device=dy.device, dtype=torch.float32), False)
        if K % DIM == 0 and N == DIM:
            return _dw192(x, dy, torch.empty(N, K, device=dy.device, dtype=torch.float32), True)
    if x.dtype != torch.float32 and not (K > N and M >= 4096):
        x = x.float()              # (only the A operand of rp_gemm may be bf16-stored; small launches are not worth a second form)
    if dy.dtype != torch.float32 and K > N and M >= 4096:
        dy = dy.float()
    if K > N and M >= 4096:
        # wide-K' weight (fc2: [192,768]): contract as (x^T dy) so the long extent is the row-panel dimension, and let
        # the split-K reduce write the transpose
        sk = max(2, pick_split_k(K, N, M, 1, 1))
        out = _empty(N, K, like=dy)
        return gemm(x, dy, K, N, M, a_layout=1, b_layout=1, out=out, ldc=K, split_k=sk, trans_c=True, defer=True)
    return gemm(dy, x, N, K, M, a_layout=1, b_layout=1, defer=True)


COLSUM_BATCHING = True      # A/B aid
_COLSUM_BATCH = None      # inside `with colsum_batch():` the (input, output) pairs collected so far
_COLSUM_STREAM = None     # ... and the stream the block was entered on = the stream rp_colsum_multi will run on at exit


class colsum_batch:
    """Collect the column sums requested inside the block and run them as one rp_colsum_multi (two launches for all of them
    instead of two each) when the block exits.  The tensors colsum() returns inside the block are only FILLED at exit: return
    them (or views of them), do not compute with them inside the block."""

    def __enter__(self):
        global _COLSUM_BATCH
        self.prev, _COLSUM_BATCH = _COLSUM_BATCH, ([] if COLSUM_BATCHING else None)
        global _COLSUM_STREAM
        self.prev_stream, _COLSUM_STREAM = _COLSUM_STREAM, (torch.cuda.current_stream() if torch.cuda.is_available() else None)
        return self

    def __exit__(self, et, ev, tb):
        global _COLSUM_BATCH
        global _COLSUM_STREAM
        tasks, _COLSUM_BATCH = _COLSUM_BATCH, self.prev
        _COLSUM_STREAM = self.prev_stream
        if et is None and tasks:
            for i in range(0, len(tasks), _lib.RP_COLSUM_MAX):
                _colsum_multi(tasks[i:i + _lib.RP_COLSUM_MAX])
        return False


def _colsum_multi(pairs):
    lib = _lib.load()
    arr = (_lib.RpColsumTask * len(pairs))()
    for a, (t, o) in zip(arr, pairs):
        a.in_, a.rows, a.cols, a.ld, a.out = t.data_ptr(), t.shape[0], t.shape[1], t.shape[1], o.data_ptr()
    nbytes = lib.rp_colsum_multi_workspace_bytes(arr, len(pairs))
    ws = torch.empty(max(nbytes // 4, 1), device=pairs[0][0].device, dtype=torch.float32)
    _lib.check(lib.rp_colsum_multi(arr, len(pairs), _p(ws), nbytes, _st()), "rp_colsum_multi")


def colsum(t2d):
    lib = _lib.load()
    _chk(t2d)
    rows, cols = t2d.shape
    out = _empty(cols, like=t2d)
    if _COLSUM_BATCH is not None:
        _COLSUM_BATCH.append((t2d, out))          # filled when the enclosing colsum_batch exits
        if _COLSUM_STREAM is not None and torch.cuda.current_stream(t2d.device) != _COLSUM_STREAM:
            # requested under fork.on_side: `out` was allocated on the side stream but is written on the batch's stream at exit
            out.record_stream(_COLSUM_STREAM)
            t2d.record_stream(_COLSUM_STREAM)
        return out
    nbytes = lib.rp_colsum_workspace_bytes(rows, cols)
    ws = torch.empty(max(nbytes // 4, 1), device=t2d.device, dtype=torch.float32)
    _lib.check(lib.rp_colsum(_p(t2d), rows, cols, cols, _p(out), _p(ws), nbytes, _st()), "rp_colsum")
    return out


def layernorm_fwd(x2d, gamma, beta, eps=LN_EPS, want_stats=True):
    lib = _lib.load()
    _chk(x2d, gamma, beta)
    rows, C = x2d.shape
    y = torch.empty_like(x2d)
    mean = _empty(rows, like=x2d) if want_stats else None
    rstd = _empty(rows, like=x2d) if want_stats else None
    _lib.check(lib.rp_layernorm_fwd(_p(x2d), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, C, eps, _st()),
               "rp_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x2d, gamma, mean, rstd, add=None):
    """returns dx (+add), dgamma, dbeta [, column sums of add]"""
    lib = _lib.load()
    _chk(dy, x2d, gamma, mean, rstd, add)
    rows, C = x2d.shape
    nblk = lib.rp_layernorm_bwd_blocks(rows)
    np_ = 3 if add is not None else 2
    dx = torch.empty_like(x2d)
    part = _empty(nblk, np_ * C, like=x2d)
    _lib.check(lib.rp_layernorm_bwd(_p(dy), _p(x2d), _p(gamma), _p(mean), _p(rstd), _p(add), _p(dx), _p(part),
                                    None, rows, C, _st()), "rp_layernorm_bwd")
    sums = colsum(part)                                 # one launch: [dgamma | dbeta (| colsum(add))]
    if add is None:
        return dx, sums[:C], sums[C:]
    return dx, sums[:C], sums[C:2 * C], sums[2 * C:]


# Stored-P attention backward (exact-fp32 configuration, self attention): the training forward keeps exp2(s - running max) of every tile
# (Z * 4 MB per Block: 2.5 GB for the five Blocks at 64 pairs -- 288 GB of HBM) and the backward executes its four algorithmic
# products instead of five (no Q K^T recompute, no exponential).  RP_ATTN_STORE_P=0: the recompute form (rp_attn_bwd_dkdv_ds).
ATTN_STORE_P = os.environ.get("RP_ATTN_STORE_P", "1") == "1"


def attn_fwd(qkv, Z, stats_only=False, q_off=0, k_off=DIM, v_off=2 * DIM, q_xor=0, k_xor=0, save_p=False):
    """qkv [Z*576, 576] packed (q | k | v, head-major columns).  Returns (o [Z*576,192] or None, lse [Z,H,576]).
    k_xor: bit 0 takes K, bit 1 takes V from the partner image of the pair (3 = --noess cross attention).
    save_p (exact fp32, no xor): rp_attn_fwd_savep -- returns (o, lse, pst, mrun) for attn_bwd(..., saved_p=(pst, mrun))."""
    lib = _lib.load()
    _chk(qkv)
    ld = qkv.shape[1]
    o = None if stats_only else _empty(Z * N_TOK, DIM, like=qkv)
    lse = _empty(Z, HEADS, N_TOK, like=qkv)
    base = qkv.data_ptr()
    P = ctypes.c_void_p
    hd = DIM // HEADS
    if save_p:
        if stats_only or q_xor or k_xor or ATTN_BF16:
            raise RuntimeError("attn_fwd(save_p=True): exact-fp32 self attention only")
        pst = _empty(Z, HEADS, N_TOK // 32, N_TOK // 32, 1024, like=qkv)
        mrun = _empty(Z, HEADS, N_TOK // 32, N_TOK, like=qkv)
        with timed("attn_fwd_savep", 4.0 * Z * HEADS * N_TOK * N_TOK * hd, 4.0 * Z * N_TOK * (4 * DIM + HEADS * (N_TOK + 19))):
            _lib.check(lib.rp_attn_fwd_savep(P(base + 4 * q_off), P(base + 4 * k_off), P(base + 4 * v_off), _p(o), _p(lse), _p(pst),
                                             _p(mrun), Z, HEADS, ld, ld, ld, DIM, hd ** -0.5, _st()), "rp_attn_fwd_savep")
        return o, lse, pst, mrun
    with timed("attn_stats" if stats_only else "attn_fwd", (2.0 if stats_only else 4.0) * Z * HEADS * N_TOK * N_TOK * hd,
               4.0 * Z * N_TOK * ((2 if stats_only else 4) * DIM + HEADS)):
        _lib.check(lib.rp_attn_fwd(P(base + 4 * q_off), P(base + 4 * k_off), P(base + 4 * v_off), _p(o), _p(lse), Z, HEADS,
                                   ld, ld, ld, DIM, q_xor, k_xor, hd ** -0.5, 1 if stats_only else 0, ATTN_BF16, _st()),
                   "rp_attn_fwd")
    return o, lse


def attn_fwd_bf16(qkv, Z, stats_only=False, q_off=0, k_off=DIM, v_off=2 * DIM, q_xor=0, k_xor=0):
    """attn_fwd on the bf16 data path: qkv [Z*576, 576] BF16 packed (q | k | v) -> (o [Z*576,192] bf16 or None, lse [Z,H,576] fp32)
    (rp_attn_fwd_bf16, csrc/attention_bf16.hip)."""
    lib = _lib.load()
    _chk_act(qkv)
    if qkv.dtype != torch.bfloat16:
        raise RuntimeError("attn_fwd_bf16 needs bf16 q | k | v")
    ld = qkv.shape[1]
    o = None if stats_only else torch.empty(Z * N_TOK, DIM, device=qkv.device, dtype=torch.bfloat16)
    lse = torch.empty(Z, HEADS, N_TOK, device=qkv.device, dtype=torch.float32)
    base = qkv.data_ptr()
    P = ctypes.c_void_p
    hd = DIM // HEADS
    with timed("attn_stats_bf16" if stats_only else "attn_fwd_bf16", (2.0 if stats_only else 4.0) * Z * HEADS * N_TOK * N_TOK * hd,
               2.0 * Z * N_TOK * ((2 if stats_only else 4) * DIM) + 4.0 * Z * N_TOK * HEADS):
        _lib.check(lib.rp_attn_fwd_bf16(P(base + 2 * q_off), P(base + 2 * k_off), P(base + 2 * v_off), _p(o), _p(lse), Z, HEADS,
                                        ld, ld, ld, DIM, q_xor, k_xor, hd ** -0.5, 1 if stats_only else 0, _st()),
                   "rp_attn_fwd_bf16")
    return o, lse


def attn_bwd_bf16(qkv, o, lse2, do, Z, kv_xor=0, want_bias_partials=False):
    """dqkv (bf16 [Z*576, 576]) of attn_fwd_bf16: rp_attn_bwd_delta_bf16 + rp_attn_bwd_bf16 (recompute form, deterministic).
    want_bias_partials: also the [Z*18, 576] fp32 column sums of dq | dk | dv per 32-row block (the qkv bias gradient's partials)."""
    lib = _lib.load()
    for t in (qkv, o, do):
        _chk_act(t)
        if t.dtype != torch.bfloat16:
            raise RuntimeError("attn_bwd_bf16 needs bf16 operands")
    _chk(lse2)
    ld = qkv.shape[1]
    delta = torch.empty(Z, HEADS, N_TOK, device=qkv.device, dtype=torch.float32)
    _lib.check(lib.rp_attn_bwd_delta_bf16(_p(do), _p(o), _p(delta), Z, HEADS, DIM, _st()), "rp_attn_bwd_delta_bf16")
    dqkv = torch.empty_like(qkv)
    P = ctypes.c_void_p
    b, d = qkv.data_ptr(), dqkv.data_ptr()
    part = pb = None
    if want_bias_partials:
        part = torch.empty(Z * (N_TOK // 32), 3 * DIM, device=qkv.device, dtype=torch.float32)
        pb = part.data_ptr()
    hd = DIM // HEADS
    with timed("attn_bwd_bf16", 14.0 * Z * HEADS * N_TOK * N_TOK * hd, 2.0 * Z * N_TOK * 8 * DIM):
        _lib.check(lib.rp_attn_bwd_bf16(P(b), P(b + 2 * DIM), P(b + 4 * DIM), _p(do), _p(lse2), _p(delta), P(d), P(d + 2 * DIM),
                                        P(d + 4 * DIM), Z, HEADS, ld, ld, ld, DIM, ld, ld, ld, hd ** -0.5, kv_xor,
                                        P(pb) if pb else None, P(pb + 4 * DIM) if pb else None, P(pb + 8 * DIM) if pb else None,
                                        3 * DIM, _st()), "rp_attn_bwd_bf16")
    return (dqkv, part) if want_bias_partials else dqkv


ATTN_BWD_STORE_DS = os.environ.get("RP_ATTN_DS", "1") == "1"
EMM_BWD_STORE_DS = os.environ.get("RP_EMM_DS", "1") == "1"


def _ds_buffer(Z, like):
    """the stored-dS array of one attention / EMM backward: [Z,H,576,576] in 32x32 tiles, bf16 in the bf16 configuration"""
    return torch.empty(Z, HEADS, N_TOK, N_TOK, device=like.device, dtype=torch.bfloat16 if ATTN_BF16 else torch.float32)


def ds_matmul(ds, b_base, ldb, out_base, ldo, Z, b_xor=0, colpart_base=None, ldp=0):
    """out[z][i][h*64+d] = sum_j ds[z,h,i,j] b[z^b_xor][j][h*64+d]; b_base / out_base: device addresses of the first column.
    ds: the tiled array a stored-dS pass wrote (fp32, or bf16 from the bf16 configuration's producers).  colpart_base / ldp: device
    address + row stride of a [Z*18, ldp] array that receives the column sums of `out` per 32-row block (a bias-gradient partial)."""
    lib = _lib.load()
    _chk_act(ds)
    bf = int(ds.dtype == torch.bfloat16)
    with timed("ds_matmul", 2.0 * Z * HEADS * N_TOK * N_TOK * 64, Z * HEADS * N_TOK * ((2.0 if bf else 4.0) * N_TOK + 4.0 * 128)):
        _lib.check(lib.rp_ds_matmul(_p(ds), ctypes.c_void_p(b_base), ctypes.c_void_p(out_base), Z, HEADS, ldb, ldo, b_xor, bf,
                                    ctypes.c_void_p(colpart_base) if colpart_base else None, ldp, _st()), "rp_ds_matmul")


QKV_BIAS_FROM_PRODUCERS = True      # A/B aid


def attn_bwd(qkv, o, lse, do, Z, fork=None, kv_xor=0, want_bias_partials=False, saved_p=None):
    """dqkv of the fused attention.  With a _Fork the dQ pass runs on the side stream next to the dK/dV pass (they write
    disjoint column blocks of dqkv); the caller must fork.sync_main() before reading dqkv.
    kv_xor=1: backward of attn_fwd(..., k_xor=3) (keys/values from the partner image).
    saved_p = (pst, mrun) of attn_fwd(save_p=True): the stored-P form (rp_attn_bwd_dkdv_p + rp_ds_matmul)."""
    lib = _lib.load()
    _chk(qkv, o, lse, do)
    ld = qkv.shape[1]
    delta = _empty(Z, HEADS, N_TOK, like=qkv)
    _lib.check(lib.rp_attn_bwd_delta(_p(do), _p(o), _p(delta), Z, HEADS, DIM, _st()), "rp_attn_bwd_delta")
    dqkv = torch.empty_like(qkv)
    P = ctypes.c_void_p
    b, d = qkv.data_ptr(), dqkv.data_ptr()
    sc = (DIM // HEADS) ** -0.5
    if saved_p is not None:
        if kv_xor or ATTN_BF16:
            raise RuntimeError("stored-P attention backward: exact-fp32 self attention only")
        pst, mrun = saved_p
        _chk(pst, mrun)
        ds = _ds_buffer(Z, qkv)
        part = pb = None
        if want_bias_partials and QKV_BIAS_FROM_PRODUCERS:
            part = _empty(Z * (N_TOK // 32), 3 * DIM, like=qkv)
            pb = part.data_ptr()
        hd = DIM // HEADS
        with timed("attn_bwd_dkdv_p", 6.0 * Z * HEADS * N_TOK * N_TOK * hd, 4.0 * Z * (N_TOK * 5 * DIM + HEADS * N_TOK * (2 * N_TOK + 21))):
            _lib.check(lib.rp_attn_bwd_dkdv_p(P(b), P(b + 8 * DIM), _p(do), _p(lse), _p(delta), _p(pst), _p(mrun), P(d + 4 * DIM),
                                              P(d + 8 * DIM), _p(ds), Z, HEADS, ld, ld, DIM, ld, ld, sc,
                                              P(pb + 4 * DIM) if pb else None, P(pb + 8 * DIM) if pb else None, 3 * DIM, _st()),
                       "rp_attn_bwd_dkdv_p")
        with timed("ds_matmul_t", 2.0 * Z * HEADS * N_TOK * N_TOK * hd, Z * HEADS * N_TOK * (4.0 * N_TOK + 4.0 * 128)):
            _lib.check(lib.rp_ds_matmul_t(_p(ds), P(b + 4 * DIM), P(d), Z, HEADS, ld, ld, 0, P(pb) if pb else None, 3 * DIM, _st()),
                       "rp_ds_matmul_t")                                             # dQ = dS K
        return (dqkv, part) if want_bias_partials else dqkv
    if kv_xor:
        _lib.check(lib.rp_attn_bwd_cross(P(b), P(b + 4 * DIM), P(b + 8 * DIM), _p(do), _p(lse), _p(delta), P(d),
                                         P(d + 4 * DIM), P(d + 8 * DIM), Z, HEADS, ld, ld, ld, DIM, ld, ld, ld, sc, 1, ATTN_BF16, _st()),
                   "rp_attn_bwd_cross")
        return (dqkv, None) if want_bias_partials else dqkv
    if ATTN_BWD_STORE_DS and (fork is None or not fork.enabled):
        # the dK/dV pass stores scale*dS ([Z,H,576,576] in 32x32 tiles; fp32, bf16 in the bf16 configuration); dQ = dS K is then one
        # rp_ds_matmul: 5 executed GEMMs instead of 7 (the dQ pass would recompute S and dP) for 2 x 510 MB of extra HBM traffic
        ds = _ds_buffer(Z, qkv)
        part = pb = None
        if want_bias_partials and QKV_BIAS_FROM_PRODUCERS:
            # [Z*18, 576]: column sums of dq | dk | dv per 32-row block, written by the epilogues of the two kernels below: the qkv bias
            # gradient becomes a column sum over 2304 rows per 128 images instead of 73 728
            part = _empty(Z * (N_TOK // 32), 3 * DIM, like=qkv)
            pb = part.data_ptr()
        hd = DIM // HEADS
        # (algorithmic flops: dV, dP, dK -- the S recompute the kernel also executes is not counted)
        with timed("attn_bwd_dkdv_ds", 6.0 * Z * HEADS * N_TOK * N_TOK * hd,
                   (2.0 if ATTN_BF16 else 4.0) * Z * HEADS * N_TOK * N_TOK + 4.0 * Z * N_TOK * (6 * DIM + 2 * HEADS)):
            _lib.check(lib.rp_attn_bwd_dkdv_ds(P(b), P(b + 4 * DIM), P(b + 8 * DIM), _p(do), _p(lse), _p(delta), P(d + 4 * DIM),
                                               P(d + 8 * DIM), _p(ds), Z, HEADS, ld, ld, ld, DIM, ld, ld, sc, ATTN_BF16,
                                               P(pb + 4 * DIM) if pb else None, P(pb + 8 * DIM) if pb else None, 3 * DIM, _st()),
                       "rp_attn_bwd_dkdv_ds")
        ds_matmul(ds, b + 4 * DIM, ld, d, ld, Z, colpart_base=pb, ldp=3 * DIM)      # dQ = dS K: one streaming launch
        return (dqkv, part) if want_bias_partials else dqkv
    if fork is None or not fork.enabled:
        _lib.check(lib.rp_attn_bwd(P(b), P(b + 4 * DIM), P(b + 8 * DIM), _p(do), _p(lse), _p(delta), P(d), P(d + 4 * DIM),
                                   P(d + 8 * DIM), Z, HEADS, ld, ld, ld, DIM, ld, ld, ld, sc, ATTN_BF16, _st()), "rp_attn_bwd")
        return (dqkv, None) if want_bias_partials else dqkv
    fork.sync_side()                                   # delta (and do, dqkv allocation) visible to the side stream

    def dq_pass():
        _lib.check(lib.rp_attn_bwd_dq(P(b), P(b + 4 * DIM), P(b + 8 * DIM), _p(do), _p(lse), _p(delta), P(d), Z, HEADS,
                                      ld, ld, ld, DIM, ld, sc, ATTN_BF16, _st()), "rp_attn_bwd_dq")
    fork.on_side(dq_pass)
    _lib.check(lib.rp_attn_bwd_dkdv(P(b), P(b + 4 * DIM), P(b + 8 * DIM), _p(do), _p(lse), _p(delta), P(d + 4 * DIM),
                                    P(d + 8 * DIM), Z, HEADS, ld, ld, ld, DIM, ld, ld, sc, ATTN_BF16, _st()), "rp_attn_bwd_dkdv")
    return (dqkv, None) if want_bias_partials else dqkv


def preprocess(images, pad=0):
    """[B,2,3,H,W] BGR 0..255 -> [2B,3,224,224] RGB normalised, channels-last (src/model.py:115-118,124-125).
    pad > 0: the NHWC buffer [2B, 224+2pad, 224+2pad, 3] with the result inside a zero frame (input of conv_stem_fwd)."""
    lib = _lib.load()
    images = images.contiguous()
    _chk(images)
    B, two, C, H, W = images.shape
    Z = B * two
    side = 224 + 2 * pad
    out = torch.empty(Z, side, side, 3, device=images.device, dtype=torch.float32)
    _lib.check(lib.rp_preprocess_padded(_p(images), _p(out), Z, H, W, pad, _st()), "rp_preprocess_padded")
    return out if pad else out.permute(0, 3, 1, 2)          # pad = 0: [Z,3,224,224] view with channels-last strides


_LIN24 = {}


def lin24(device):
    """torch.linspace(-1, 1, 24) exactly as the reference builds it (vision_transformer.py:110-111)."""
    key = str(device)
    if key not in _LIN24:
        _LIN24[key] = torch.linspace(-1, 1, steps=24, dtype=torch.float32).to(device)
    return _LIN24[key]


def posenc(intrinsics, B, device, l1=False):
    lib = _lib.load()
    _chk(intrinsics)
    pos = torch.empty(B, N_TOK, 6, device=device, dtype=torch.float32)
    _lib.check(lib.rp_posenc(_p(intrinsics), _p(lin24(device)), _p(pos), B, 1 if l1 else 0, _st()), "rp_posenc")
    return pos


def emm_build_x(qkv, pos, Z):
    lib = _lib.load()
    _chk(qkv, pos)
    x = _empty(Z, HEADS, N_TOK, XW, like=qkv)
    _lib.check(lib.rp_emm_build_x(_p(qkv), _p(pos), _p(x), Z, HEADS, qkv.shape[1], _st()), "rp_emm_build_x")
    return x


EMM_STATS_ONE_PASS = True      # A/B aid
_stats_ws = {}


# Stored-S EMM (exact-fp32 configuration): the statistics pass also writes the score tiles (Z * 4 MB, kept for the backward), and the
# three later passes over S -- rp_emm_apply forward and swap, rp_emm_grad_ds -- read them instead of recomputing q k^T (96 of 260 MFMAs
# per tile).  RP_EMM_STORE_S=0: the recompute form.
EMM_STORE_S = os.environ.get("RP_EMM_STORE_S", "1") == "1"


def emm_stats(qkv, Z, single=False, want_s=False):
    """row / column log-sum-exp of S_z = scale q_{z^1} k_z^T (single softmax: rows only).  Both normalisers come from ONE pass over S
    (rp_emm_stats; two rp_attn_fwd(stats_only) passes in the bf16 configuration).
    want_s: -> (rlse, clse, s) with s the stored score tiles [Z,H,18,18,1024] (None where the one-pass fp32 kernel does not run)."""
    if single or not EMM_STATS_ONE_PASS:
        _, rlse = attn_fwd(qkv, Z, stats_only=True, q_off=0, k_off=DIM, q_xor=1, k_xor=0)
        if single:
            return (rlse, rlse, None) if want_s else (rlse, rlse)
        _, clse = attn_fwd(qkv, Z, stats_only=True, q_off=DIM, k_off=0, q_xor=0, k_xor=1)
        return (rlse, clse, None) if want_s else (rlse, clse)
    lib = _lib.load()
    _chk(qkv)
    ld = qkv.shape[1]
    rlse, clse = _empty(Z, HEADS, N_TOK, like=qkv), _empty(Z, HEADS, N_TOK, like=qkv)
    ws = None
    if not ATTN_BF16:
        key = (qkv.device, Z, torch.cuda.current_stream(qkv.device).cuda_stream)      # (one scratch per stream, like _mlp_ws)
        ws = _stats_ws.get(key)
        if ws is None:
            ws = _stats_ws[key] = torch.empty(lib.rp_emm_stats_workspace_bytes(Z, HEADS) // 4, device=qkv.device, dtype=torch.float32)
    b = qkv.data_ptr()
    s = _empty(Z, HEADS, N_TOK // 32, N_TOK // 32, 1024, like=qkv) if (want_s and EMM_STORE_S and not ATTN_BF16) else None
    with timed("emm_stats", 2.0 * Z * HEADS * N_TOK * N_TOK * 64,
               4.0 * Z * N_TOK * (2 * DIM + 2 * HEADS) + (4.0 * Z * HEADS * N_TOK * N_TOK if s is not None else 0.0)):
        _lib.check(lib.rp_emm_stats(ctypes.c_void_p(b), ctypes.c_void_p(b + 4 * DIM), _p(rlse), _p(clse), _p(ws), _p(s), Z, HEADS, ld, ld,
                                    (DIM // HEADS) ** -0.5, ATTN_BF16, _st()), "rp_emm_stats")
    return (rlse, clse, s) if want_s else (rlse, clse)


def pair_swap(x):
    """x[z] -> x[z ^ 1] for a tensor whose leading dim is the image index (images 2b, 2b+1 form pair b)."""
    return x.view(x.shape[0] // 2, 2, *x.shape[1:]).flip(1).reshape(x.shape).contiguous()


def emm_apply(qkv, x, rlse, clse, Z, swap=False, want_t=True, want_f=True, single=False, x_left=None, s=None):
    """s: the score tiles emm_stats(want_s=True) stored -> the kernel reads them instead of recomputing q k^T"""
    lib = _lib.load()
    _chk(qkv, x, rlse, clse, x_left, s)
    t = _empty(Z, HEADS, N_TOK, XW, like=qkv) if want_t else None
    f = _empty(Z, HEADS, NWG, XW, XW, like=qkv) if (want_f and not swap) else None
    # algorithmic: T = A X (96 per score element) -- plus S = q k^T (64) where it is recomputed; F = X^T T adds 2 * 576 * 96 * 96 per (image, head)
    with timed("emm_apply_s" if s is not None else "emm_apply",
               2.0 * Z * HEADS * (N_TOK * N_TOK * ((0 if s is not None else 64) + XW) + (N_TOK * XW * XW if f is not None else 0)),
               4.0 * Z * HEADS * N_TOK * ((N_TOK if s is not None else 2 * 64) + 2 * XW + 2)):
        _lib.check(lib.rp_emm_apply(_p(qkv), qkv.shape[1], _p(x), _p(x_left), _p(rlse), _p(clse), _p(s), _p(t), _p(f), Z, HEADS,
                                    (DIM // HEADS) ** -0.5, 1 if swap else 0, 1 if single else 0, ATTN_BF16, _st()), "rp_emm_apply")
    return t, f


def emm_finalize(fpart, Z):
    lib = _lib.load()
    g = _empty(Z * 70, GW, like=fpart)
    _lib.check(lib.rp_emm_finalize(_p(fpart), _p(g), Z, HEADS, GW, _st()), "rp_emm_finalize")
    return g


def emm_finalize_bwd(dg, Z):
    lib = _lib.load()
    _chk(dg)
    df = _empty(Z, HEADS, XW, XW, like=dg)
    _lib.check(lib.rp_emm_finalize_bwd(_p(dg), _p(df), Z, HEADS, GW, _st()), "rp_emm_finalize_bwd")
    return df


def rowdot96(a, b):
    lib = _lib.load()
    _chk(a, b)
    rows = a.numel() // XW
    out = _empty(*a.shape[:-1], like=a)
    _lib.check(lib.rp_rowdot96(_p(a), _p(b), _p(out), rows, _st()), "rp_rowdot96")
    return out


def _bmm96(X, D, transpose_d, residual=None):
    """[ZH,576,96] x [ZH,96,96](^T) -> [ZH,576,96]"""
    ZH = X.shape[0] * X.shape[1]
    out = torch.empty_like(X)
    gemm(X, D, N_TOK, XW, XW, b_layout=0 if transpose_d else 1, out=out, residual=residual, split_k=1, batch=ZH,
         strides=(N_TOK * XW, XW * XW, N_TOK * XW))
    return out


def emm_backward(qkv, x, t, rlse, clse, df, Z, single=False, cross=False, s=None):
    """Gradient of F = X_L^T A X wrt qkv (q, k through A; v through X_L and X).  df: [Z,H,96,96] zero-padded.
    cross (cross_features): X_L[z] = X[z^1]; otherwise X_L = X.  s: the forward's stored score tiles (emm_stats(want_s=True))."""
    lib = _lib.load()
    scale = (DIM // HEADS) ** -0.5
    sg = 1 if single else 0
    xl = pair_swap(x) if cross else x   # left operand, indexed by the problem z
    w = _bmm96(xl, df, False)       # W  = X_L dF    (rows i)
    wp = _bmm96(x, df, True)        # W' = X dF^T    (rows j)
    u, _ = emm_apply(qkv, xl, rlse, clse, Z, swap=True, want_f=False, single=single, s=s)     # U = A^T X_L
    rho = rowdot96(w, t)            # rho_i   = sum_j A_ij dA_ij
    gam = rho if single else rowdot96(wp, u)           # gamma_j = sum_i A_ij dA_ij (unused by the single softmax)
    dxl = _bmm96(t, df, True)                                   # d X_L = T dF^T  (belongs to image z^1 when cross)
    dx = _bmm96(u, df, False, residual=pair_swap(dxl) if cross else dxl)      # + d X = U dF
    dqkv = torch.empty_like(qkv)
    ld = qkv.shape[1]
    if EMM_BWD_STORE_DS:
        # the query-side pass stores scale*dS (tiled); dk_z = dS_z^T-major x q_{z^1} is one rp_ds_matmul instead of a second pass
        # that recomputes S and dA (68 of its 100 MFMAs per tile)
        ds = _ds_buffer(Z, qkv)
        # algorithmic: S recompute is not counted; dA = W X^T (96) and dq = dS k (64) per score element
        with timed("emm_grad_ds_s" if s is not None else "emm_grad_ds", 2.0 * Z * HEADS * N_TOK * N_TOK * (XW + 64),
                   4.0 * Z * HEADS * N_TOK * ((2 if s is not None else 1) * N_TOK + 3 * 64 + 2 * XW + 4)):
            _lib.check(lib.rp_emm_grad_ds(_p(qkv), ld, _p(x), _p(w), _p(rlse), _p(clse), _p(rho), _p(gam), _p(s), _p(dqkv), _p(ds), Z,
                                          HEADS, scale, sg, ATTN_BF16, _st()), "rp_emm_grad_ds")
        # dk_z = dS_z (key-major tiles) x q_{z^1}: one streaming launch (the stored-S pass writes the 16-byte-run tiles rp_ds_matmul_t takes)
        if s is not None:
            with timed("ds_matmul_t", 2.0 * Z * HEADS * N_TOK * N_TOK * 64, Z * HEADS * N_TOK * (4.0 * N_TOK + 4.0 * 128)):
                _lib.check(lib.rp_ds_matmul_t(_p(ds), ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(dqkv.data_ptr() + 4 * DIM), Z, HEADS,
                                              ld, ld, 1, None, 0, _st()), "rp_ds_matmul_t")
        else:
            ds_matmul(ds, qkv.data_ptr(), ld, dqkv.data_ptr() + 4 * DIM, ld, Z, b_xor=1)
    else:
        _lib.check(lib.rp_emm_grad(_p(qkv), ld, _p(x), _p(w), _p(rlse), _p(clse), _p(rho), _p(gam), _p(dqkv), Z, HEADS,
                                   scale, 0, sg, ATTN_BF16, _st()), "rp_emm_grad(q)")
        _lib.check(lib.rp_emm_grad(_p(qkv), ld, _p(xl), _p(wp), _p(rlse), _p(clse), _p(rho), _p(gam), _p(dqkv), Z, HEADS,
                                   scale, 1, sg, ATTN_BF16, _st()), "rp_emm_grad(k)")
    _lib.check(lib.rp_emm_build_x_bwd(_p(dx), _p(dqkv), Z, HEADS, ld, _st()), "rp_emm_build_x_bwd")
    return dqkv


# ---- the EMM on the bf16 data path (csrc/emm_bf16.hip): default flags only; the ablation variants keep the fp32-storage kernels ----
def _bf(*shape, like):
    return torch.empty(shape, device=like.device, dtype=torch.bfloat16)


def emm_forward_bf16(qkv, pos, Z, want_t=True):
    """(g [Z*70, 224] fp32, saved = (xa, t, rlse2, clse2)) of the EMM for bf16 qkv [Z*576, 576]: statistics (two statistics-only passes
    of the bf16 attention kernel, log2 units), X = [v | pos | 0], T = A X, F = X^T T, the reshape / transpose / flip of
    vision_transformer.py:229-230,238."""
    lib = _lib.load()
    _chk_act(qkv)
    _chk(pos)
    ld = qkv.shape[1]
    sc = (DIM // HEADS) ** -0.5
    _, rlse2 = attn_fwd_bf16(qkv, Z, stats_only=True, q_off=0, k_off=DIM, q_xor=1, k_xor=0)          # rows i: queries of the partner image
    _, clse2 = attn_fwd_bf16(qkv, Z, stats_only=True, q_off=DIM, k_off=0, q_xor=0, k_xor=1)          # columns j: keys as the owner rows
    xa = _bf(Z, HEADS, N_TOK, XW, like=qkv)
    _lib.check(lib.rp_emm_build_x_bf16(_p(qkv), _p(pos), _p(xa), Z, HEADS, ld, _st()), "rp_emm_build_x_bf16")
    t = _bf(Z, HEADS, N_TOK, XW, like=qkv)
    with timed("emm_apply_bf16", 2.0 * Z * HEADS * N_TOK * N_TOK * (64 + XW), 2.0 * Z * HEADS * N_TOK * (2 * 64 + 2 * XW)):
        _lib.check(lib.rp_emm_apply_bf16(_p(qkv), ld, _p(xa), _p(rlse2), _p(clse2), _p(t), Z, HEADS, sc, 0, _st()), "rp_emm_apply_bf16")
    f = torch.empty(Z, HEADS, XW, XW, device=qkv.device, dtype=torch.float32)
    _lib.check(lib.rp_emm_f_bf16(_p(xa), _p(t), _p(f), Z, HEADS, _st()), "rp_emm_f_bf16")
    g = _empty(Z * 70, GW, like=f)
    _lib.check(lib.rp_emm_finalize_parts(_p(f), _p(g), Z, HEADS, GW, 1, _st()), "rp_emm_finalize_parts")
    return g, (xa, t, rlse2, clse2)


def emm_backward_bf16(qkv, xa, t, rlse2, clse2, df, Z):
    """dqkv (bf16 [Z*576, 576]) of the EMM given df [Z,H,96,96] fp32 (zero-padded): W / W' / rho, U = A^T X, dX + gamma, then the two
    recompute passes for dq and dk."""
    lib = _lib.load()
    _chk(df)
    ld = qkv.shape[1]
    sc = (DIM // HEADS) ** -0.5
    w, wp = _bf(Z, HEADS, N_TOK, XW, like=qkv), _bf(Z, HEADS, N_TOK, XW, like=qkv)
    rho = torch.empty(Z, HEADS, N_TOK, device=qkv.device, dtype=torch.float32)
    gam = torch.empty_like(rho)
    _lib.check(lib.rp_emm_w_bf16(_p(xa), _p(t), _p(df), _p(w), _p(wp), _p(rho), Z, HEADS, _st()), "rp_emm_w_bf16")
    u = _bf(Z, HEADS, N_TOK, XW, like=qkv)
    _lib.check(lib.rp_emm_apply_bf16(_p(qkv), ld, _p(xa), _p(rlse2), _p(clse2), _p(u), Z, HEADS, sc, 1, _st()), "rp_emm_apply_bf16(swap)")
    dqkv = torch.empty_like(qkv)
    _lib.check(lib.rp_emm_dx_bf16(_p(t), _p(u), _p(wp), _p(df), _p(dqkv), ld, _p(gam), Z, HEADS, _st()), "rp_emm_dx_bf16")
    with timed("emm_grad_bf16", 2.0 * Z * HEADS * N_TOK * N_TOK * (64 + XW + 64), 2.0 * Z * HEADS * N_TOK * (3 * 64 + 2 * XW)):
        _lib.check(lib.rp_emm_grad_bf16(_p(qkv), ld, _p(xa), _p(w), _p(rlse2), _p(clse2), _p(rho), _p(gam), _p(dqkv), Z, HEADS, sc, 0, _st()),
                   "rp_emm_grad_bf16(q)")
    _lib.check(lib.rp_emm_grad_bf16(_p(qkv), ld, _p(xa), _p(wp), _p(rlse2), _p(clse2), _p(rho), _p(gam), _p(dqkv), Z, HEADS, sc, 1, _st()),
               "rp_emm_grad_bf16(k)")
    return dqkv


# ------------------------------------------------------------------------------------------------
# autograd Functions
# ------------------------------------------------------------------------------------------------
_GRAD_AT_APPLY = True


class _Fn(torch.autograd.Function):
    """Base of every Function here.  Inside Function.forward grad mode is always off and ctx.needs_input_grad ignores torch.no_grad()
    (it is just the inputs' requires_grad), so `any(ctx.needs_input_grad)` alone would send every no_grad / inference call down the
    training path (extra activation stores, no fused inference MLP).  apply() records the caller's grad mode; _train(ctx) combines
    the two."""

    @classmethod
    def apply(cls, *args, **kwargs):
        global _GRAD_AT_APPLY
        prev, _GRAD_AT_APPLY = _GRAD_AT_APPLY, torch.is_grad_enabled()
        try:
            return super().apply(*args, **kwargs)
        finally:
            _GRAD_AT_APPLY = prev


def _train(ctx):
    return _GRAD_AT_APPLY and any(ctx.needs_input_grad)


class TokensFn(_Fn):
    """x[z][n][c] = feat[z][c][n] + pos_embed[n][c]  (src/model.py:136-141,170-171).  A channels-last CNN map is
    already laid out [z][n][c] in memory, so the permutation is a view and only the add runs."""

    @staticmethod
    def forward(ctx, feat, pos_embed):
        lib = _lib.load()
        Z, C = feat.shape[0], feat.shape[1]
        N = feat.numel() // (Z * C)
        nhwc = feat.dim() == 4 and feat.is_contiguous(memory_format=torch.channels_last) and not feat.is_contiguous()
        ctx.nhwc, ctx.shape = nhwc, tuple(feat.shape)
        x = torch.empty(Z, N, C, device=feat.device, dtype=torch.float32)
        if nhwc:
            src = feat.permute(0, 2, 3, 1)              # [Z,H,W,C] view, contiguous
            _chk(src, pos_embed)
            _lib.check(lib.rp_tokens_fwd_nhwc(_p(src), _p(pos_embed), _p(x), Z, C, N, _st()), "rp_tokens_fwd_nhwc")
        else:
            feat = feat.contiguous()
            _chk(feat, pos_embed)
            _lib.check(lib.rp_tokens_fwd(_p(feat), _p(pos_embed), _p(x), Z, C, N, _st()), "rp_tokens_fwd")
        return x

    @staticmethod
    def backward(ctx, dx):
        lib = _lib.load()
        dx = dx.contiguous()
        Z, N, C = dx.shape
        dpe = colsum(dx.view(Z, N * C)).view(1, N, C)
        if ctx.nhwc:
            _, _, H, W = ctx.shape
            return dx.view(Z, H, W, C).permute(0, 3, 1, 2), dpe          # channels-last gradient, no copy
        dfeat = _empty(*ctx.shape, like=dx)
        _lib.check(lib.rp_tokens_bwd(_p(dx), _p(dfeat), Z, C, N, _st()), "rp_tokens_bwd")
        return dfeat, dpe


# The transformer MLP's forward: LayerNorm + fc1 + GELU + fc2 + residual as ONE kernel (csrc/mlp_fused.hip).  Inference
# keeps the hidden activation on chip; training runs the same kernel and stores xn, h and h_pre for the backward on the way
# (RP_FUSE_MLP_TRAIN=0: the LayerNorm+fc1 launch and the fc2 launch instead).
FUSE_MLP = True
FUSE_MLP_TRAIN = True
_mlp_ws = {}


def mlp_fused(x2d, gamma, beta, w1, b1, w2, b2, eps=LN_EPS, train=False, out_dtype=None, xn_dtype=None):
    """y = x + fc2(GELU(fc1(LayerNorm(x)) + b1)) + b2 for x [M,192], w1 [768,192], w2 [192,768] (rp_mlp_fused_fwd).
    train=True: returns (y, xn, mean, rstd, h, hpre) -- the same launch also stores what the backward needs.  At operand precision 1
    (the bf16 configuration) both products run on the bf16 MFMA from bf16 weight copies, and out_dtype=torch.bfloat16 stores h / hpre
    as bf16."""
    lib = _lib.load()
    _chk(x2d, gamma, beta, w1, b1, w2, b2)
    M = x2d.shape[0]
    bf = GEMM_PRECISION == 1
    obf = out_dtype == torch.bfloat16
    if obf and not (bf and train):
        raise RuntimeError("bf16-stored hidden tensors need operand precision 1 (the bf16 configuration) and the training form")
    y = torch.empty_like(x2d)
    key = (x2d.device, M, torch.cuda.current_stream(x2d.device).cuda_stream)
    ws = _mlp_ws.get(key)
    if ws is None:
        ws = _mlp_ws[key] = torch.empty(max(1, lib.rp_mlp_fused_workspace_bytes(M)) // 4 + 1, device=x2d.device, dtype=torch.float32)
    Hd = w1.shape[0]
    xn = mean = rstd = h = hpre = None
    xnbf = xn_dtype == torch.bfloat16 and bf and train
    if train:
        xn = torch.empty(x2d.shape, device=x2d.device, dtype=torch.bfloat16 if xnbf else torch.float32)
        mean, rstd = _empty(M, like=x2d), _empty(M, like=x2d)
        hdt = torch.bfloat16 if obf else torch.float32
        h, hpre = torch.empty(M, Hd, device=x2d.device, dtype=hdt), torch.empty(M, Hd, device=x2d.device, dtype=hdt)
    if bf:
        w1k, w2k = bf16_weight(w1), _chunk_permuted_bf16(w2)
    else:
        w1k, w2k = w1, w2
    with timed("mlp_fused_fwd" + ("" if train else "_eval") + ("_bf16" if bf else ""), 4.0 * M * DIM * Hd, 4.0 * (2 * M * DIM + 2 * DIM * Hd + (M * DIM + 2 * M * Hd if train else 0))):
        _lib.check(lib.rp_mlp_fused_fwd(_p(x2d), _p(gamma), _p(beta), _p(w1k), _p(b1), _p(w2k), _p(b2), _p(y), _p(ws), M, x2d.shape[1],
                                        Hd, eps, _p(xn), _p(mean), _p(rstd), _p(h), _p(hpre), 1 if bf else 0, (2 if obf else 0) | (8 if xnbf else 0) | (16 if bf and MLP_W2_CHUNK_MAJOR else 0), _st()),
                   "rp_mlp_fused_fwd")
    return (y, xn, mean, rstd, h, hpre) if train else y


FUSE_MLP_BWD = True
# the LayerNorm backward in front of fc1 on the epilogue of rp_mlp_fused_bwd (rp_mlp_fused_bwd_ln): dxn never reaches HBM.  1 = in the
# bf16 configuration (HBM-bound there: 19.67 -> 19.48 ms per 128 pairs), 2 = also with exact fp32 operands, where the kernel is
# matrix-bound, every workgroup's epilogue burst lands at the same moment and the fold LOSES 8 us per Block (profiles/r5_ab_mlp_bwd_ln.txt)
MLP_BWD_LN = int(os.environ.get("RP_MLP_BWD_LN", "1"))


_UNIT_PERM = {}


def _mlp_unit_perm(device):
    """column order of rp_mlp_fused_bwd's bf16 second weight: position 8q+e of every 32-unit chunk holds unit 4q+e (e<4) / 16+4q+e-4"""
    pm = _UNIT_PERM.get(device)
    if pm is None:
        pos = torch.arange(4 * DIM, device=device)
        c, r = pos // 32, pos % 32
        q, e = r // 8, r % 8
        pm = _UNIT_PERM[device] = c * 32 + torch.where(e < 4, 4 * q + e, 16 + 4 * q + e - 4)
    return pm


MLP_W2_CHUNK_MAJOR = True      # io_bf16 bit 4 of rp_mlp_fused_fwd / _bwd


def _chunk_permuted_bf16(w, transpose=False):
    """bf16 copy of a [192, 768] operand (w, or w^T when transpose) with the 768 hidden units of every 32-chunk in the order the fused
    MLP kernels' second product wants (_mlp_unit_perm); cached on w until it changes"""
    c = getattr(w, "_rp_bp", None)
    if c is not None and c[0] == w._version and c[1] == w.data_ptr() and c[2] == _PAD_GEN and c[4] == MLP_W2_CHUNK_MAJOR:
        return c[3]
    src = w.detach().t() if transpose else w.detach()
    o = src.index_select(1, _mlp_unit_perm(w.device)).to(torch.bfloat16).contiguous()
    if MLP_W2_CHUNK_MAJOR:      # [24 chunks][192][32]: every staged tile is 12 KB contiguous
        o = o.view(o.shape[0], -1, 32).permute(1, 0, 2).contiguous()
    if not (w.is_cuda and torch.cuda.is_current_stream_capturing()):
        try:
            w._rp_bp = (w._version, w.data_ptr(), _PAD_GEN, o, MLP_W2_CHUNK_MAJOR)      # (the layout is part of the key)
        except AttributeError:
            pass
    return o


def _mlp_bwd_bf16_weights(w1, w2):
    """(bf16 W2^T [768,192], bf16 W1^T [192,768] with the chunk-permuted unit order)"""
    return bf16_weight(transposed(w2)), _chunk_permuted_bf16(w1, transpose=True)


def mlp_fused_bwd(dy, hpre, w1, w2, out_dtype=None, ln=None):
    """(dhp, dxn, db1_partials) of the MLP backward-data chain (rp_mlp_fused_bwd): dhp = (dy W2) o GELU'(hpre), dxn = dhp W1.
    ln = (x, gamma, mean, rstd) of the LayerNorm in front of fc1: its backward rides on the epilogue (rp_mlp_fused_bwd_ln) and the
    result is (dhp, (dx, dgamma, dbeta, colsum(dy)), db1_partials) with dx = layernorm_bwd(dxn, ...) + dy -- layernorm_bwd's tuple.
    w1 [768,192], w2 [192,768] as stored by nn.Linear (transposed here: 2 x 590 KB); db1_partials [tiles,768] column-sums to db1.
    At operand precision 1 (the bf16 configuration) the products run on the bf16 MFMA from bf16 weight copies; hpre may then be a bf16
    tensor and out_dtype=torch.bfloat16 stores dhp as bf16."""
    lib = _lib.load()
    _chk(dy)
    _chk_act(hpre)
    M = dy.shape[0]
    bf = GEMM_PRECISION == 1
    io = (2 if out_dtype == torch.bfloat16 else 0) | (4 if hpre.dtype == torch.bfloat16 else 0)
    if io and not bf:
        raise RuntimeError("bf16-stored operands need operand precision 1 (the bf16 configuration)")
    if bf:
        w2t, w1t = _mlp_bwd_bf16_weights(w1, w2)
        io |= 16 if MLP_W2_CHUNK_MAJOR else 0
    else:
        w2t, w1t = transposed(w2), transposed(w1)
    dhp = torch.empty(hpre.shape, device=dy.device, dtype=out_dtype or torch.float32)
    dxn = torch.empty_like(dy)
    tiles = -(-M // lib.rp_mlp_fused_bwd_tile_rows())
    colpart = _empty(tiles, hpre.shape[1], like=dy)
    key = (dy.device, M, "bwd", torch.cuda.current_stream(dy.device).cuda_stream)
    ws = _mlp_ws.get(key)
    if ws is None:
        ws = _mlp_ws[key] = torch.empty(max(1, lib.rp_mlp_fused_bwd_workspace_bytes(M)) // 4 + 1, device=dy.device, dtype=torch.float32)
    if ln is not None:
        x, gamma, mean, rstd = ln
        _chk(x, gamma, mean, rstd)
        C = dy.shape[1]
        lnpart = _empty(lib.rp_mlp_fused_bwd_ln_part_rows(M), 3 * C, like=dy)
        with timed("mlp_fused_bwd_ln" + ("_bf16" if bf else ""), 4.0 * M * C * hpre.shape[1], M * (float(hpre.element_size()) * 2 * hpre.shape[1] + 4.0 * 4 * C)):
            _lib.check(lib.rp_mlp_fused_bwd_ln(_p(dy), _p(hpre), _p(w2t), _p(w1t), _p(dhp), _p(dxn), _p(colpart), _p(ws), M, C,
                                               hpre.shape[1], 1 if bf else 0, io, _p(x), _p(gamma), _p(mean), _p(rstd), _p(lnpart), _st()),
                       "rp_mlp_fused_bwd_ln")
        sums = colsum(lnpart)
        return dhp, (dxn, sums[:C], sums[C:2 * C], sums[2 * C:]), colpart
    with timed("mlp_fused_bwd" + ("_bf16" if bf else ""), 4.0 * M * dy.shape[1] * hpre.shape[1], M * (float(hpre.element_size()) * 2 * hpre.shape[1] + 4.0 * 2 * dy.shape[1])):
        _lib.check(lib.rp_mlp_fused_bwd(_p(dy), _p(hpre), _p(w2t), _p(w1t), _p(dhp), _p(dxn), _p(colpart), _p(ws), M, dy.shape[1],
                                        hpre.shape[1], 1 if bf else 0, io, _st()), "rp_mlp_fused_bwd")
    return dhp, dxn, colpart


def _mlp_block_fwd(x1, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b, train):
    """(y, xn2, m2, r2, h, hpre) of `x1 + Mlp(norm2(x1))`; the inference path returns y only (rest None)."""
    if (FUSE_MLP and GEMM_PRECISION in (0, 1) and x1.shape[1] == DIM and tuple(fc1_w.shape) == (4 * DIM, DIM)
            and tuple(fc2_w.shape) == (DIM, 4 * DIM) and (not train or FUSE_MLP_TRAIN) and x1.dtype == torch.float32):
        if train:
            return mlp_fused(x1, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b, train=True, out_dtype=torch.bfloat16 if _act_bf16() else None,
                             xn_dtype=torch.bfloat16 if _bf16_path() else None)
        return mlp_fused(x1, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b), None, None, None, None, None
    hd = torch.bfloat16 if _act_bf16() else None          # bf16 configuration: the [tokens, 768] hidden tensors live in bf16
    if train:
        h, hpre, xn2, m2, r2 = ln_linear(x1, n2w, n2b, fc1_w, fc1_b, act=1, want_pre=True, train=True, out_dtype=hd)
    else:
        h, xn2, m2, r2 = ln_linear(x1, n2w, n2b, fc1_w, fc1_b, act=1, train=False, out_dtype=hd)
        hpre = None
    y = linear(h, fc2_w, fc2_b, residual=x1)
    return y, xn2, m2, r2, h, hpre


def _mlp_fwd(xn, w1, b1, w2, b2, residual, train):
    if train:
        h, hpre = linear(xn, w1, b1, act=1, want_pre=True)
    else:
        h, hpre = linear(xn, w1, b1, act=1), None
    y = linear(h, w2, b2, residual=residual)
    return y, h, hpre


def linear_dw_db(dy, x):
    """(dW, db) of a Linear.  (Fusing the bias column sums into the weight-gradient GEMM was measured: the extra per-thread
    accumulators cost that kernel 20 %, more than the saved pass over dy -- kept as two kernels.)"""
    return linear_dw(dy, x), colsum(dy)


def _param_grads(fork, dy, x):
    """(dW, db) of a Linear on the side stream (caller has made dy visible with fork.sync_side())."""
    return fork.on_side(lambda: linear_dw_db(dy, x))


def _mlp_bwd(fork, dy, xn, h, hpre, w1, w2, want_db2=True, ln=None):
    fork.sync_side()
    if want_db2:
        dw2, db2 = _param_grads(fork, dy, h)
    else:                     # the caller gets colsum(dy) for free from the LayerNorm backward that adds dy
        dw2, db2 = fork.on_side(lambda: linear_dw(dy, h)), None
    if (FUSE_MLP_BWD and GEMM_PRECISION in (0, 1) and dy.shape[1] == DIM and tuple(w1.shape) == (4 * DIM, DIM)
            and tuple(w2.shape) == (DIM, 4 * DIM) and dy.dtype == torch.float32):
        # both input-gradient products as one kernel (dh stays on chip); fc1 bias gradient from its per-tile column sums
        odt = torch.bfloat16 if hpre.dtype == torch.bfloat16 else None
        fold = ln is not None and ln[4] is dy and MLP_BWD_LN >= (1 if GEMM_PRECISION == 1 else 2)      # (a Block's residual-branch gradient IS the MLP's dy)
        dh, dxn, part = mlp_fused_bwd(dy, hpre, w1, w2, out_dtype=odt, ln=ln[:4] if fold else None)
        db1 = colsum(part)
        fork.sync_side()
        dw1 = fork.on_side(lambda: linear_dw(dh, xn))
        if ln is not None and not fold:
            x_, gamma_, mean_, rstd_, add_ = ln
            return layernorm_bwd(dxn, x_, gamma_, mean_, rstd_, add=add_), dw1, db1, dw2, db2
        return dxn, dw1, db1, dw2, db2
    # grad wrt fc1 pre-activation (GELU' fused) and, from the same epilogue, its column sums = the fc1 bias gradient
    dh, db1 = linear_dx(dy, w2, dact=1, aux=hpre, want_colsum=True, out_dtype=torch.bfloat16 if hpre.dtype == torch.bfloat16 else None)
    fork.sync_side()
    dw1 = fork.on_side(lambda: linear_dw(dh, xn))
    if ln is not None:      # (x, gamma, mean, rstd, add): LayerNorm backward fused into the fc1 input-gradient GEMM
        return linear_dx_lnbwd(dh, w1, *ln), dw1, db1, dw2, db2
    dxn = linear_dx(dh, w1)
    return dxn, dw1, db1, dw2, db2


class BlockFn(_Fn):
    """Block.forward (vision_transformer.py:349-354) on x [Z,576,192]."""

    @staticmethod
    def forward(ctx, x, n1w, n1b, qkv_w, qkv_b, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b, cross=False):
        """cross=True: keys and values come from the partner image of each pair -- the --noess CrossBlock
        (vision_transformer.py:239-262,297-304), which is otherwise arithmetically a Block."""
        train = _train(ctx)
        x = x.contiguous()
        Z = x.shape[0]
        x2 = x.view(Z * N_TOK, DIM)
        bfp = _bf16_path()
        pst = mrun = None
        if bfp:      # bf16 q | k | v -> bf16 o (lse in log2 units); xn1 kept as the bf16 rows the product consumed
            qkv, xn1, m1, r1 = ln_linear(x2, n1w, n1b, qkv_w, qkv_b, train=train, out_dtype=torch.bfloat16, xn_dtype=torch.bfloat16)
            o, lse = attn_fwd_bf16(qkv, Z, k_xor=3 if cross else 0)
        else:
            qkv, xn1, m1, r1 = ln_linear(x2, n1w, n1b, qkv_w, qkv_b, train=train)
            if train and ATTN_STORE_P and ATTN_BWD_STORE_DS and not cross and not ATTN_BF16 and not USE_SIDE_STREAM:
                o, lse, pst, mrun = attn_fwd(qkv, Z, save_p=True)
            else:
                o, lse = attn_fwd(qkv, Z, k_xor=3 if cross else 0)
        ctx.cross = cross
        x1 = linear(o, proj_w, proj_b, residual=x2)
        y, xn2, m2, r2, h, hpre = _mlp_block_fwd(x1, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b, train)
        if train:
            register_transposed(proj_w, fc1_w, fc2_w)
            if bfp:
                register_transposed(qkv_w)          # W_qkv^T for the output-resident input-gradient kernel (rp_dx_lnbwd_bf16)
            ctx.save_for_backward(x2, m1, r1, xn1, qkv, o, lse, x1, m2, r2, xn2, h, hpre, n1w, qkv_w, proj_w, n2w,
                                  fc1_w, fc2_w, pst, mrun)
            ctx.Z = Z
        return y.view(Z, N_TOK, DIM)

    @staticmethod
    def backward(ctx, dy):
        (x2, m1, r1, xn1, qkv, o, lse, x1, m2, r2, xn2, h, hpre, n1w, qkv_w, proj_w, n2w, fc1_w,
         fc2_w, pst, mrun) = ctx.saved_tensors
        Z = ctx.Z
        dy = dy.contiguous().view(Z * N_TOK, DIM)
        fork = _Fork(dy.device)
        # the two LayerNorm backwards read dy / dx1 as their residual-branch operand anyway: they also return its column
        # sums, which ARE the fc2 / proj bias gradients (two 57 MB column-sum passes per block saved).  The block's four column
        # sums (LayerNorm partials x 2, fc1 and qkv bias gradients) run as one batched pair of launches at the end.
        with colsum_batch(), splitk_batch():
            (dx1, dn2w, dn2b, dfc2b), dfc1w, dfc1b, dfc2w, _ = _mlp_bwd(fork, dy, xn2, h, hpre, fc1_w, fc2_w, want_db2=False,
                                                                         ln=(x1, n2w, m2, r2, dy))
            fork.sync_side()
            dprojw = fork.on_side(lambda: linear_dw(dx1, o))
            if qkv.dtype == torch.bfloat16:          # the bf16 data path (forward ran attn_fwd_bf16)
                do = linear_dx(dx1, proj_w, out_dtype=torch.bfloat16)
                dqkv, bpart = attn_bwd_bf16(qkv, o, lse, do, Z, kv_xor=1 if ctx.cross else 0, want_bias_partials=True)
            else:
                do = linear_dx(dx1, proj_w)
                dqkv, bpart = attn_bwd(qkv, o, lse, do, Z, fork, kv_xor=1 if ctx.cross else 0, want_bias_partials=True,
                                       saved_p=None if pst is None else (pst, mrun))
            fork.sync_main()                                  # dQ pass (side) done before dqkv is consumed
            fork.sync_side()
            if bpart is not None:      # qkv bias gradient from the per-block column sums the attention backward's kernels left
                dqkvw, dqkvb = fork.on_side(lambda: linear_dw(dqkv, xn1)), colsum(bpart)
            else:
                dqkvw, dqkvb = _param_grads(fork, dqkv, xn1)
            dx, dn1w, dn1b, dprojb = linear_dx_lnbwd(dqkv, qkv_w, x2, n1w, m1, r1, add=dx1)
            fork.sync_main()
        return (dx.view(Z, N_TOK, DIM), dn1w, dn1b, dqkvw, dqkvb, dprojw, dprojb, dn2w, dn2b, dfc1w, dfc1b, dfc2w,
                dfc2b, None)


class CrossBlockFn(_Fn):
    """CrossBlock.forward, ess branch (vision_transformer.py:285-296): x [2B,576,192] -> [2B,70,192]."""

    @staticmethod
    def forward(ctx, x, pos, n1w, n1b, qkv_w, qkv_b, pf_w, pf_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b, single=False,
                cross=False):
        train = _train(ctx)
        x = x.contiguous()
        Z = x.shape[0]
        x2 = x.view(Z * N_TOK, DIM)
        ctx.single, ctx.cross = single, cross
        sc = None
        if _bf16_path() and not single and not cross:      # the bf16 data path (csrc/emm_bf16.hip): bf16 q | k | v, X, T; log2 normalisers
            qkv, xn, m1, r1 = ln_linear(x2, n1w, n1b, qkv_w, qkv_b, train=train, out_dtype=torch.bfloat16, xn_dtype=torch.bfloat16)
            g, (xa, t, rlse, clse) = emm_forward_bf16(qkv, pos, Z)
        else:
            qkv, xn, m1, r1 = ln_linear(x2, n1w, n1b, qkv_w, qkv_b, train=train)
            rlse, clse, sc = emm_stats(qkv, Z, single, want_s=True)
            xa = emm_build_x(qkv, pos, Z)
            t, fpart = emm_apply(qkv, xa, rlse, clse, Z, swap=False, want_t=train, single=single,
                                 x_left=xa if cross else None, s=sc)
            g = emm_finalize(fpart, Z)                                 # [Z*70, 224]
        pf_wp = _padded(pf_w, (0, GW - pf_w.shape[1]))
        f = linear(g, pf_wp, pf_b)                                     # [Z*70, 192]
        y, fn, m2, r2, h, hpre = _mlp_block_fwd(f, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b, train)
        if train:
            register_transposed(fc1_w, fc2_w)
            if qkv.dtype == torch.bfloat16:
                register_transposed(qkv_w)
            ctx.save_for_backward(x2, m1, r1, xn, qkv, rlse, clse, xa, t, g, f, m2, r2, fn, h, hpre, n1w, qkv_w, pf_wp,
                                  n2w, fc1_w, fc2_w, sc)
            ctx.Z = Z
            ctx.pf_cols = pf_w.shape[1]
        return y.view(Z, 70, DIM)

    @staticmethod
    def backward(ctx, dy):
        (x2, m1, r1, xn, qkv, rlse, clse, xa, t, g, f, m2, r2, fn, h, hpre, n1w, qkv_w, pf_wp, n2w, fc1_w,
         fc2_w, sc) = ctx.saved_tensors
        Z = ctx.Z
        dy = dy.contiguous().view(Z * 70, DIM)
        fork = _Fork(dy.device)
        with colsum_batch(), splitk_batch():
            (df_, dn2w, dn2b, dfc2b), dfc1w, dfc1b, dfc2w, _ = _mlp_bwd(fork, dy, fn, h, hpre, fc1_w, fc2_w, want_db2=False,
                                                                         ln=(f, n2w, m2, r2, dy))       # 4th: colsum(dy) = fc2 bias gradient
            fork.sync_side()
            dpfw_full, dpfb = _param_grads(fork, df_, g)
            dg = linear_dx(df_, pf_wp)                                      # [Z*70, 224]
            dF = emm_finalize_bwd(dg, Z)
            if qkv.dtype == torch.bfloat16:
                dqkv = emm_backward_bf16(qkv, xa, t, rlse, clse, dF, Z)
                fork.sync_side()
                dqkvw = fork.on_side(lambda: linear_dw(dqkv, xn))
                dqkvb = dqkv.sum(0, dtype=torch.float32)                # (one bf16 pass per step; the Blocks get theirs from kernel epilogues)
            else:
                dqkv = emm_backward(qkv, xa, t, rlse, clse, dF, Z, single=ctx.single, cross=ctx.cross, s=sc)
                fork.sync_side()
                dqkvw, dqkvb = _param_grads(fork, dqkv, xn)
            dx, dn1w, dn1b = linear_dx_lnbwd(dqkv, qkv_w, x2, n1w, m1, r1)
            fork.sync_main()
        dpfw = dpfw_full[:, :ctx.pf_cols].contiguous()
        return (dx.view(Z, N_TOK, DIM), None, dn1w, dn1b, dqkvw, dqkvb, dpfw, dpfb, dn2w, dn2b, dfc1w, dfc1b, dfc2w,
                dfc2b, None, None)


def _regress_fwd(feats, gs, w0, b0, w2, b2, w4, b4):
    """H-512-512-14 MLP + quaternion normalise (src/model.py:91-98,145-159) on feats [B,H]."""
    lib = _lib.load()
    B = feats.shape[0]
    h1 = linear(feats, w0, b0, act=2)
    h2 = linear(h1, w2, b2, act=2)
    w4p = _padded(w4, (0, 0, 0, 2))                                 # 14 -> 16 output rows (float4 alignment)
    b4p = _padded(b4, (0, 2))
    pred = linear(h2, w4p, b4p)[:, :14].contiguous()               # [B,14] == [B,2,7]
    gs = gs.contiguous()
    _chk(gs)
    out = _empty(B, 2, 7, like=feats)
    _lib.check(lib.rp_pose_normalize_fwd(_p(pred), _p(gs), _p(out), B, _st()), "rp_pose_normalize_fwd")
    return out, h1, h2, pred, w4p


def _regress_bwd(dout, feats, h1, h2, pred, w0, w2, w4p):
    lib = _lib.load()
    B = feats.shape[0]
    dout = dout.contiguous()
    dpred = _empty(B, 14, like=dout)
    _lib.check(lib.rp_pose_normalize_bwd(_p(pred), _p(dout), _p(dpred), B, _st()), "rp_pose_normalize_bwd")
    dp16 = torch.nn.functional.pad(dpred, (0, 2)).contiguous()
    dh2 = linear_dx(dp16, w4p, dact=2, aux=h2)
    dw4, db4 = linear_dw(dp16, h2)[:14].contiguous(), colsum(dpred)
    dh1 = linear_dx(dh2, w2, dact=2, aux=h1)
    dw2, db2 = linear_dw(dh2, h1), colsum(dh2)
    dfeats = linear_dx(dh1, w0)
    dw0, db0 = linear_dw(dh1, feats), colsum(dh1)
    return dfeats, dw0, db0, dw2, db2, dw4, db4


class HeadFn(_Fn):
    """final LayerNorm -> flatten [B,26880] -> 26880-512-512-14 MLP -> quaternion normalise
    (src/model.py:178,189,91-98,145-159).  y: [2B,70,192]; gs: [B,2,7] -> [B,2,7]."""

    @staticmethod
    def forward(ctx, y, gs, nw, nb, w0, b0, w2, b2, w4, b4):
        train = _train(ctx)
        y = y.contiguous()
        Z = y.shape[0]
        B = Z // 2
        y2 = y.view(Z * 70, DIM)
        fn, m, r = layernorm_fwd(y2, nw, nb)
        out, h1, h2, pred, w4p = _regress_fwd(fn.view(B, 2 * 70 * DIM), gs, w0, b0, w2, b2, w4, b4)
        if train:
            ctx.save_for_backward(y2, m, r, fn, h1, h2, pred, nw, w0, w2, w4p)
            ctx.B = B
        return out

    @staticmethod
    def backward(ctx, dout):
        y2, m, r, fn, h1, h2, pred, nw, w0, w2, w4p = ctx.saved_tensors
        B = ctx.B
        with colsum_batch():
            dfeats, dw0, db0, dw2, db2, dw4, db4 = _regress_bwd(dout, fn.view(B, -1), h1, h2, pred, w0, w2, w4p)
            dy, dnw, dnb = layernorm_bwd(dfeats.view(-1, DIM), y2, nw, m, r)
        return dy.view(2 * B, 70, DIM), None, dnw, dnb, dw0, db0, dw2, db2, dw4, db4


class RegressFn(_Fn):
    """pose regressor + quaternion normalise on already-pooled features [B,H] (the --noess head, src/model.py:79-86,188)."""

    @staticmethod
    def forward(ctx, feats, gs, w0, b0, w2, b2, w4, b4):
        train = _train(ctx)
        feats = feats.contiguous()
        _chk(feats)
        out, h1, h2, pred, w4p = _regress_fwd(feats, gs, w0, b0, w2, b2, w4, b4)
        if train:
            ctx.save_for_backward(feats, h1, h2, pred, w0, w2, w4p)
        return out

    @staticmethod
    def backward(ctx, dout):
        feats, h1, h2, pred, w0, w2, w4p = ctx.saved_tensors
        with colsum_batch():
            dfeats, dw0, db0, dw2, db2, dw4, db4 = _regress_bwd(dout, feats, h1, h2, pred, w0, w2, w4p)
        return dfeats, None, dw0, db0, dw2, db2, dw4, db4


class LayerNormFn(_Fn):
    """LayerNorm over the last dim (C = 192) on the rowwise HIP kernels: the final norm of the --noess model, whose output
    feeds a conv head instead of HeadFn (src/model.py:178,183-188)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        x2 = x.view(-1, x.shape[-1])
        y, m, r = layernorm_fwd(x2, w, b)
        if _train(ctx):
            ctx.save_for_backward(x2, m, r, w)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, m, r, w = ctx.saved_tensors
        dx, dw, db = layernorm_bwd(dy.contiguous().view(x2.shape), x2, w, m, r)
        return dx.view(dy.shape), dw, db


# ------------------------------------------------------------------------------------------------
# CNN front-end: channels-last BatchNorm2d + residual add + ReLU in two passes each way (csrc/batchnorm.hip)
# ------------------------------------------------------------------------------------------------
# BatchNorm backward, first pass fused into the producer of its incoming gradient (exact-fp32 configuration): BnActFn.forward hands out a
# token with (x, mean, rstd, gamma, beta) (`_BN_LAST`, attached to its output by bn_act); a hand-written convolution that consumes that output
# keeps it, masks its input gradient with the ReLU and posts the column-sum partials under the token (`_BN_PENDING`); BnActFn.backward finds
# them and runs only its second and third pass.  RP_CONV_F32_BN_BWD=0: off (A/B aid).
CONV_F32_BN_BWD = os.environ.get("RP_CONV_F32_BN_BWD", "1") != "0"
_BN_TOKENS = itertools.count(1)
_BN_LAST = None
_BN_PENDING = {}


class BnActFn(_Fn):
    """y = relu?(batch_norm(x) (+ residual)) for a channels-last NCHW x; same statistics / running-buffer semantics as
    torch.nn.BatchNorm2d (biased batch variance for the normalisation, unbiased for running_var, momentum update in
    training; running statistics in eval).  Returns a channels-last tensor."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, residual, training, momentum, eps, relu, stats=None):
        """stats: per-workgroup sums of x and x^2 [blocks,2,C] float64 from the producing convolution's epilogue (rp_conv3x3_c64_bf16):
        then the statistics pass over x is skipped (training only)"""
        lib = _lib.load()
        N, C, H, W = x.shape
        xr = x.permute(0, 2, 3, 1)                       # [N,H,W,C] view of the channels-last buffer
        if not xr.is_contiguous():
            xr = xr.contiguous()
        rr = None
        if residual is not None:
            rr = residual.permute(0, 2, 3, 1)
            if not rr.is_contiguous():
                rr = rr.contiguous()
        _chk(gamma, beta)
        bf = _chk_act(xr, rr)
        R = N * H * W
        y = torch.empty_like(xr)
        if training and stats is not None:
            mean, rstd = torch.empty(C, device=x.device), torch.empty(C, device=x.device)
            _lib.check(lib.rp_bn_stats_from_partials(_p(stats), stats.shape[0], R, C, _p(_zeros(C, xr.device)), _p(mean), _p(rstd),
                                                     _p(running_mean), _p(running_var), float(momentum), float(eps), _st()),
                       "rp_bn_stats_from_partials")
        elif training:
            mean, rstd = _empty(C, like=xr), _empty(C, like=xr)
            part = torch.empty(lib.rp_bn_partial_blocks(R) * 2 * C, device=x.device, dtype=torch.float64)
            _lib.check(lib.rp_bn_stats(_p(xr), R, C, _p(part), _p(mean), _p(rstd), _p(running_mean), _p(running_var),
                                       float(momentum), float(eps), bf, _st()), "rp_bn_stats")
        else:
            mean, rstd = running_mean, torch.rsqrt(running_var + eps)
        _lib.check(lib.rp_bn_apply_fwd(_p(xr), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(rr), _p(y), R, C, 1 if relu else 0,
                                       bf, _st()), "rp_bn_apply_fwd")
        global _BN_LAST
        _BN_LAST = None
        if _train(ctx):
            # without a residual the ReLU mask is re-evaluated from x in the backward: y is not kept alive for it
            keep_y = y if (relu and residual is not None) else None
            ctx.save_for_backward(xr, keep_y, mean, rstd, gamma, beta)
            ctx.cfg = (R, C, bool(relu), bool(training), residual is not None)
            ctx.token = None
            if training and relu and residual is None and not bf and CONV_F32_BN_BWD:
                # a consumer of this output that computes its own input gradient (the hand-written fp32 convolutions) may mask that gradient
                # with this ReLU and form this BatchNorm's backward column sums in its epilogue: see bn_act / conv2d / _BN_PENDING
                ctx.token = next(_BN_TOKENS)
                _BN_LAST = (ctx.token, xr, mean, rstd, gamma, beta)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        xr, y, mean, rstd, gamma, beta = ctx.saved_tensors
        R, C, relu, training, has_res = ctx.cfg
        dyr = dy.permute(0, 2, 3, 1)
        if not dyr.is_contiguous():
            dyr = dyr.contiguous()
        if dyr.dtype != xr.dtype:
            dyr = dyr.to(xr.dtype)
        bf = _chk_act(dyr, xr, y)
        dx = torch.empty_like(xr)
        dres = torch.empty_like(xr) if has_res and ctx.needs_input_grad[5] else None
        dgamma, dbeta, c12 = _empty(C, like=xr), _empty(C, like=xr), _empty(2 * C, like=xr)
        hit = _BN_PENDING.pop(ctx.token, None) if getattr(ctx, "token", None) is not None else None
        if hit is not None and hit[0] == dyr.data_ptr() and training and relu and not has_res and not bf:
            # dy IS the masked gradient g, written by the convolution kernel that produced it together with the column sums of g and g * xhat
            part = hit[1]
            _lib.check(lib.rp_bn_bwd_from_partials(_p(dyr), _p(xr), _p(mean), _p(rstd), _p(gamma), _p(part), part.shape[0], _p(dx), _p(dgamma),
                                                   _p(dbeta), _p(c12), R, C, _st()), "rp_bn_bwd_from_partials")
            return (dx.permute(0, 3, 1, 2), dgamma, dbeta, None, None, None, None, None, None, None, None)
        # (a gradient that was masked by a producer but arrives as another tensor takes the full path: masking twice is the identity)
        part = torch.empty(lib.rp_bn_partial_blocks(R) * 2 * C, device=xr.device, dtype=torch.float64)
        _lib.check(lib.rp_bn_bwd(_p(dyr), _p(y), _p(xr), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dx), _p(dres), _p(dgamma), _p(dbeta),
                                 _p(part), _p(c12), R, C, 1 if relu else 0, 1 if training else 0, bf, _st()), "rp_bn_bwd")
        return (dx.permute(0, 3, 1, 2), dgamma, dbeta, None, None,
                None if dres is None else dres.permute(0, 3, 1, 2), None, None, None, None, None)


# Convolution operand precision of the CNN front-end (MIOpen through PyTorch-ROCm): 0 = fp32 (the parity path),
# 1 = bf16 operands / fp32 accumulate -- part of the bf16 configuration (BASELINE.json configs[4], `bench.py --precision bf16`):
# the fp32 convolutions are 62 % of that configuration's step otherwise (profiles/r2_conv_probe*.txt: 14.3 -> 4.0 ms per 64 pairs).
# Round 3: the activations between the convolutions STAY bf16 (csrc/batchnorm.hip takes bf16 storage: statistics and arithmetic in
# fp32 / double): no cast kernels around the convolutions, half the BatchNorm / ReLU / pool traffic.
CNN_PRECISION = int(os.environ.get("RP_CNN_PRECISION", "0"))


def set_cnn_precision(p):
    global CNN_PRECISION
    if p not in (0, 1):
        raise ValueError("CNN precision must be 0 (fp32) or 1 (bf16 operands)")
    CNN_PRECISION = p


class _CastParamsFn(torch.autograd.Function):
    """bf16 copies of a list of fp32 parameters as ONE multi-tensor launch each way (forward: fp32 -> bf16, backward: the bf16
    gradients -> fp32).  Per convolution this was a cast kernel forward and another backward: 35 launches of 6-8 us per step."""

    @staticmethod
    def forward(ctx, *ps):
        ctx.set_materialize_grads(False)      # a copy nobody used gets no gradient (None), like the per-layer casts it replaces
        outs = [torch.empty_like(p, dtype=torch.bfloat16) for p in ps]
        torch._foreach_copy_(outs, [p.detach() for p in ps])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        live = [g for g in gs if g is not None]
        outs = [torch.empty_like(g, dtype=torch.float32) for g in live]
        if live:
            torch._foreach_copy_(outs, live)
        it = iter(outs)
        return tuple(None if g is None else next(it) for g in gs)


class conv_params_bf16:
    """`with conv_params_bf16(module):` -- at CNN precision 1 every nn.Conv2d below `module` finds its bf16 weight / bias copies ready
    (made by one _CastParamsFn call) instead of casting them itself; the copies live for the block only (the optimizer changes the
    masters between steps)."""

    def __init__(self, *roots):
        self.convs = [m for r in roots for m in r.modules() if isinstance(m, torch.nn.Conv2d)] if CNN_PRECISION == 1 else []

    def __enter__(self):
        if self.convs and self.convs[0].weight.is_cuda:
            ps = [m.weight for m in self.convs] + [m.bias for m in self.convs if m.bias is not None]
            cs = _CastParamsFn.apply(*ps)
            nb = iter(cs[len(self.convs):])
            for m, w in zip(self.convs, cs):
                m._rp_bf16 = (w, None if m.bias is None else next(nb))
        return self

    def __exit__(self, et, ev, tb):
        for m in self.convs:
            m._rp_bf16 = None
        return False


# bf16 configuration: resnet.layer1's four 3x3 / 64 -> 64 convolutions on the hand-written implicit GEMM (csrc/conv3x3_bf16.hip): forward
# (with the following BatchNorm's batch statistics from its epilogue) and input gradient (the same kernel on dY with the rotated,
# transposed filter); the weight gradient stays MIOpen's backward-weights.
CONV3X3_OWN = os.environ.get("RP_CONV3X3_OWN", "1") != "0"
CONV3X3_OWN_WGRAD = os.environ.get("RP_CONV3X3_OWN_WGRAD", "1") != "0"


class Conv3x3C64Fn(_Fn):
    @staticmethod
    def forward(ctx, x, w, want_stats):
        """x [N,64,56,56] bf16 channels-last, w [64,64,3,3] bf16 channels-last -> y (channels-last) [, stats partials (not differentiable)]"""
        xr, wr = x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1)
        if not xr.is_contiguous():
            xr = xr.contiguous()
        if not wr.is_contiguous():
            wr = wr.contiguous()
        ctx.save_for_backward(x, w)
        if want_stats:
            y, stats = conv3x3_c64_bf16(xr, wr, want_stats=True)
            ctx.mark_non_differentiable(stats)
            return y.permute(0, 3, 1, 2), stats
        return conv3x3_c64_bf16(xr, wr).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy, *_):
        x, w = ctx.saved_tensors
        dx = dw = None
        dy = dy.contiguous(memory_format=torch.channels_last)
        if ctx.needs_input_grad[0]:
            # dX = conv3x3(dY, w'), w'[ci][r][s][co] = w[co][2-r][2-s][ci]
            wt = w.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)
            dx = conv3x3_c64_bf16(dy.permute(0, 2, 3, 1), wt.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            if CONV3X3_OWN_WGRAD:
                xr = x.permute(0, 2, 3, 1)
                dw = conv3x3_c64_wgrad_bf16(xr if xr.is_contiguous() else xr.contiguous(), dy.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
            else:
                dw = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        return dx, dw, None


# bf16 configuration, the CNN tail's two 5x5 valid convolutions (src/modules/extractor.py:51-65): their input gradient IS a forward
# convolution of dY with the rotated, transposed filter (padding 4), and MIOpen's forward solvers run these shapes faster than its
# backward-data solvers once they are in the solver db (rel_pose_amd/miopen_db: profiles/r5_conv_probe_bf16_256.txt, 701 -> 515 us and
# 413 -> 335 us at 256 images).  Same sums in another order: equal to MIOpen's backward-data to bf16 rounding.
CONV_BWD_AS_FWD_MIN_K = int(os.environ.get("RP_CONV_BWD_AS_FWD_MIN_K", "5"))


class ConvBf16Fn(_Fn):
    """F.conv2d on bf16 channels-last operands (stride 1, dilation 1, groups 1) whose backward-data runs as a forward convolution;
    weight / bias gradients stay MIOpen's backward-weights."""

    @staticmethod
    def forward(ctx, x, w, b, padding):
        ctx.save_for_backward(x, w)
        ctx.conf = (tuple(padding), b is not None)
        return torch.nn.functional.conv2d(x, w, b, 1, padding)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        padding, has_b = ctx.conf
        dy = dy.contiguous(memory_format=torch.channels_last)
        k = w.shape[2]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wf = w.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)
            dx = torch.nn.functional.conv2d(dy, wf, None, 1, (k - 1 - padding[0], k - 1 - padding[1]))
        mask = [False, ctx.needs_input_grad[1], has_b and ctx.needs_input_grad[2]]
        if any(mask):
            g = torch.ops.aten.convolution_backward(dy, x, w, [w.shape[0]] if has_b else None, [1, 1], list(padding), [1, 1], False,
                                                    [0, 0], 1, mask)
            dw, db = (g[1] if mask[1] else None), (g[2] if mask[2] else None)
        return dx, dw, db, None


# exact-fp32 configuration: the weight gradient of resnet.layer1's 3x3 convolutions on the output-stationary fp32 kernel
# (csrc/conv3x3_wgrad_f32.hip); forward and input gradient stay MIOpen's
CONV3X3_WGRAD_F32 = os.environ.get("RP_CONV3X3_WGRAD_F32", "1") != "0"
# (256 workgroup partials are written and reduced whatever the batch: below ~56 images MIOpen's kernel is faster -- 12 images: 37 vs
# 55 us, 48: 110 vs 115, 64: 143 vs 137, 128: 263 vs 235, tools/lab/conv_wgrad_f32_time.py)
CONV3X3_WGRAD_F32_MIN_N = int(os.environ.get("RP_CONV3X3_WGRAD_F32_MIN_N", "56"))


# ... and, from CONV3X3_F32_MIN_N images up, their forward and input gradient on csrc/conv3x3_f32.hip (filter in registers, padded LDS
# ring): 128 images: forward MIOpen 267 us, input gradient 307 us (profiles/r2_conv_probe.txt).  A persistent workgroup per CU: small
# batches leave CUs idle, MIOpen's tiling wins there.
CONV3X3_F32 = os.environ.get("RP_CONV3X3_F32", "1") != "0"
CONV3X3_F32_MIN_N = int(os.environ.get("RP_CONV3X3_F32_MIN_N", "56"))


def _bn_mask(bn, y, want_stats):
    """RpBnMask of the convolution kernels' BatchNorm-backward epilogue (None = off): bn = (x_bn, mean, rstd, gamma, beta)"""
    if bn is None:
        return None
    xb, mean, rstd, gamma, beta = bn
    C = y.shape[-1]
    if not want_stats:
        raise RuntimeError("the BatchNorm-mask epilogue needs want_stats (its column sums are the point)")
    if not (xb.is_cuda and xb.is_contiguous() and xb.dtype == torch.float32 and xb.shape == y.shape):
        raise RuntimeError("the BatchNorm-mask epilogue needs the BatchNorm input as a contiguous fp32 tensor of the result's shape")
    for t in (mean, rstd, gamma, beta):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and tuple(t.shape) == (C,)):
            raise RuntimeError("the BatchNorm-mask epilogue needs contiguous fp32 [C] statistics and parameters")
    m = _lib.RpBnMask()
    m.x, m.mean, m.rstd, m.gamma, m.beta = xb.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    return ctypes.byref(m)


def conv3x3_c64_f32(x_nhwc, w_ohwi, input_gradient=False, want_stats=False, res=None, bn=None):
    """rp_conv3x3_c64_f32: y = conv3x3(x, w), stride 1, pad 1, exact fp32, for x [N,56,56,64] (NHWC memory) and w [64,3,3,64] (the memory
    of a channels-last [64,64,3,3] weight) -> y [N,56,56,64].  input_gradient: x is dY, the result dX of that convolution (the rotated,
    channel-swapped filter is read out of the forward weight by the kernel).  want_stats: also returns the per-workgroup sums of y and
    y^2 per channel [blocks,2,64] float64 from the kernel's epilogue (BnActFn's `stats`).  res: a tensor of y's shape added in the epilogue.
    bn = (x_bn [N,56,56,64], mean, rstd, gamma, beta) (with want_stats): the result is the gradient of relu(batch_norm(x_bn)); it is masked by
    that ReLU and the partials are the sums of g and g * xhat (rp_bn_bwd_from_partials finishes the BatchNorm's backward)."""
    lib = _lib.load()
    if not (x_nhwc.is_cuda and x_nhwc.is_contiguous() and x_nhwc.dtype == torch.float32 and tuple(x_nhwc.shape[1:]) == (56, 56, 64)):
        raise RuntimeError("conv3x3_c64_f32: contiguous fp32 [N,56,56,64] GPU tensor expected")
    if not (w_ohwi.is_cuda and w_ohwi.is_contiguous() and w_ohwi.dtype == torch.float32 and tuple(w_ohwi.shape) == (64, 3, 3, 64)):
        raise RuntimeError("conv3x3_c64_f32: contiguous fp32 [64,3,3,64] GPU filter expected")
    N = x_nhwc.shape[0]
    if res is not None and not (res.is_cuda and res.is_contiguous() and res.dtype == torch.float32 and res.shape == x_nhwc.shape):
        raise RuntimeError("conv3x3_c64_f32: res must be a contiguous fp32 tensor of the output's shape")
    y = torch.empty_like(x_nhwc)
    stats = torch.empty(lib.rp_conv3x3_c64_f32_blocks(N), 2, 64, device=x_nhwc.device, dtype=torch.float64) if want_stats else None
    with timed("conv3x3_c64_f32", 2.0 * N * 56 * 56 * 64 * 64 * 9, 4.0 * N * 56 * 56 * (128 if res is None else 192)):
        _lib.check(lib.rp_conv3x3_c64_f32(_p(x_nhwc), _p(w_ohwi), _p(y), _p(stats), _p(res), _bn_mask(bn, y, want_stats), N, 56, 56,
                                          1 if input_gradient else 0, _st()), "rp_conv3x3_c64_f32")
    return (y, stats) if want_stats else y


def _nhwc(t):
    r = t.permute(0, 2, 3, 1)
    return r if r.is_contiguous() else r.contiguous()


class Conv3x3C64F32Fn(_Fn):
    @staticmethod
    def forward(ctx, x, w, want_stats=False, share_input=False, bn_src=None):
        """want_stats: returns (y, stats) with the output's BatchNorm partial sums from the kernel's epilogue (None when MIOpen ran).
        bn_src = (token, x_bn, mean, rstd, gamma, beta) of the BatchNorm + ReLU that produced x (BnActFn): the input gradient is masked by
        that ReLU in the kernel and the BatchNorm's backward column sums are posted under the token.
        share_input: additionally returns x itself as an output -- the caller uses THAT tensor for the block's identity path, so the
        gradient of the identity path arrives here and is added in the input-gradient kernel's epilogue instead of by a pass of autograd's."""
        ctx.save_for_backward(x, w)
        own = CONV3X3_F32 and x.shape[0] >= CONV3X3_F32_MIN_N
        ctx.bn_src = bn_src if own else None
        stats = None
        if want_stats and own and CONV_F32_STATS:
            y, stats = conv3x3_c64_f32(_nhwc(x), _nhwc(w), want_stats=True)
            ctx.mark_non_differentiable(stats)
            y = y.permute(0, 3, 1, 2)
        else:
            y = conv3x3_c64_f32(_nhwc(x), _nhwc(w)).permute(0, 3, 1, 2) if own else torch.nn.functional.conv2d(x, w, None, 1, 1)
        if share_input:
            return y, stats, x.view_as(x)
        return (y, stats) if want_stats else y

    @staticmethod
    def backward(ctx, dy, *rest):
        x, w = ctx.saved_tensors
        dx = dw = None
        dy = dy.contiguous(memory_format=torch.channels_last)
        dshared = rest[1] if len(rest) > 1 else None              # gradient of the shared-input output (the identity path)
        if ctx.needs_input_grad[0]:
            if CONV3X3_F32 and x.shape[0] >= CONV3X3_F32_MIN_N:
                # dX = conv3x3(dY, w') with w'[ci][r][s][co] = w[co][2 - r][2 - s][ci], read out of w by the kernel (+ the identity path's gradient)
                bn = ctx.bn_src
                r = conv3x3_c64_f32(_nhwc(dy), _nhwc(w), input_gradient=True, res=None if dshared is None else _nhwc(dshared),
                                    want_stats=bn is not None, bn=None if bn is None else bn[1:])
                if bn is not None:
                    if len(_BN_PENDING) > 64:
                        _BN_PENDING.clear()
                    _BN_PENDING[bn[0]] = (r[0].data_ptr(), r[1])
                    r = r[0]
                dx = r.permute(0, 3, 1, 2)
            else:
                dx = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
                if dshared is not None:
                    dx = dx + dshared
        if ctx.needs_input_grad[1]:
            if CONV3X3_WGRAD_F32 and x.shape[0] >= CONV3X3_WGRAD_F32_MIN_N:
                dw = conv3x3_c64_wgrad_f32(_nhwc(x), dy.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
            else:
                dw = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        return dx, dw, None, None, None


# The 128-input-channel 3x3 convolutions on 28 x 28 maps (resnet.layer2's 128 -> 128 convolutions forward and input gradient, the forward of
# extractor_final_conv.conv1 128 -> 192) on csrc/conv3x3_c128_f32.hip from CONV3X3_C128_F32_MIN_N images up; their weight (and bias)
# gradients stay on MIOpen.  RP_CONV3X3_C128_F32=0: MIOpen for all of it (A/B aid).
CONV3X3_C128_F32 = os.environ.get("RP_CONV3X3_C128_F32", "1") != "0"
# the own fp32 convolutions hand the BatchNorm behind them its batch statistics from their epilogue (no statistics pass over y); A/B aid
CONV_F32_STATS = os.environ.get("RP_CONV_F32_STATS", "1") != "0"
CONV3X3_C128_F32_MIN_N = int(os.environ.get("RP_CONV3X3_C128_F32_MIN_N", "56"))


def conv3x3_c128_f32(x_nhwc, w_ohwi, bias=None, input_gradient=False, want_stats=False):
    """rp_conv3x3_c128_f32: y = bias + conv3x3(x, w), stride 1, pad 1, exact fp32, for x [N,28,28,128] (NHWC memory) and w [CO,3,3,128]
    (the memory of a channels-last [CO,128,3,3] weight), CO = 128 or 192 -> y [N,28,28,CO].  input_gradient (CO = 128, no bias): x is dY,
    the result dX of the convolution whose forward weight is w.  want_stats: also returns the per-chunk sums of y and y^2 per channel
    [chunks,2,CO] float64 from the kernel's epilogue (BnActFn's `stats`)."""
    lib = _lib.load()
    if not (x_nhwc.is_cuda and x_nhwc.is_contiguous() and x_nhwc.dtype == torch.float32 and tuple(x_nhwc.shape[1:]) == (28, 28, 128)):
        raise RuntimeError("conv3x3_c128_f32: contiguous fp32 [N,28,28,128] GPU tensor expected")
    CO = w_ohwi.shape[0]
    if not (w_ohwi.is_cuda and w_ohwi.is_contiguous() and w_ohwi.dtype == torch.float32 and tuple(w_ohwi.shape[1:]) == (3, 3, 128)
            and CO in (128, 192)):
        raise RuntimeError("conv3x3_c128_f32: contiguous fp32 [128 | 192,3,3,128] GPU filter expected")
    if input_gradient and (CO != 128 or bias is not None):
        raise RuntimeError("conv3x3_c128_f32: the input gradient needs the square 128 -> 128 filter and takes no bias")
    if bias is not None and not (bias.is_cuda and bias.is_contiguous() and bias.dtype == torch.float32 and tuple(bias.shape) == (CO,)):
        raise RuntimeError("conv3x3_c128_f32: contiguous fp32 [CO] GPU bias expected")
    N = x_nhwc.shape[0]
    y = torch.empty(N, 28, 28, CO, device=x_nhwc.device, dtype=torch.float32)
    stats = (torch.empty(lib.rp_conv3x3_c128_f32_blocks(N, CO) // (CO // 64), 2, CO, device=x_nhwc.device, dtype=torch.float64)
             if want_stats else None)
    with timed("conv3x3_c128_f32", 2.0 * N * 28 * 28 * 128 * CO * 9, 4.0 * N * 28 * 28 * (128 + CO)):
        _lib.check(lib.rp_conv3x3_c128_f32(_p(x_nhwc), _p(w_ohwi), _p(bias), _p(y), _p(stats), N, 28, 28, CO, 1 if input_gradient else 0,
                                           _st()), "rp_conv3x3_c128_f32")
    return (y, stats) if want_stats else y


class Conv3x3C128F32Fn(_Fn):
    """forward (and, for the square filter, input gradient) on rp_conv3x3_c128_f32; weight / bias gradients on MIOpen"""

    @staticmethod
    def forward(ctx, x, w, bias, want_stats=False):
        """want_stats: see Conv3x3C64F32Fn"""
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        if want_stats and CONV_F32_STATS:
            y, stats = conv3x3_c128_f32(_nhwc(x), _nhwc(w), bias, want_stats=True)
            ctx.mark_non_differentiable(stats)
            return y.permute(0, 3, 1, 2), stats
        y = conv3x3_c128_f32(_nhwc(x), _nhwc(w), bias).permute(0, 3, 1, 2)           # (channels-last NCHW view of the NHWC result)
        return (y, None) if want_stats else y

    @staticmethod
    def backward(ctx, dy, *_):
        x, w = ctx.saved_tensors
        dx = dw = db = None
        dy = dy.contiguous(memory_format=torch.channels_last)
        CO = w.shape[0]
        own_dx = ctx.needs_input_grad[0] and CO == 128
        if own_dx:
            dx = conv3x3_c128_f32(_nhwc(dy), _nhwc(w), input_gradient=True).permute(0, 3, 1, 2)
        mask = [bool(ctx.needs_input_grad[0]) and not own_dx, bool(ctx.needs_input_grad[1]), ctx.has_bias and bool(ctx.needs_input_grad[2])]
        if any(mask):
            g = torch.ops.aten.convolution_backward(dy, x, w, [CO] if ctx.has_bias else None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, mask)
            dx = g[0] if mask[0] else dx
            dw = g[1] if mask[1] else None
            db = g[2] if mask[2] else None
        return dx, dw, db, None


def conv3x3_c128_f32_ok(m, x):
    return (CONV3X3_C128_F32 and CNN_PRECISION == 0 and x.is_cuda and x.dtype == torch.float32 and tuple(m.weight.shape[1:]) == (128, 3, 3)
            and m.weight.shape[0] in (128, 192) and m.stride == (1, 1) and m.padding == (1, 1) and m.dilation == (1, 1) and m.groups == 1
            and getattr(m, "padding_mode", "zeros") == "zeros" and tuple(x.shape[1:]) == (128, 28, 28) and x.shape[0] >= CONV3X3_C128_F32_MIN_N)


def conv3x3_wgrad_f32_ok(m, x):
    return (CONV3X3_WGRAD_F32 and CNN_PRECISION == 0 and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled()
            and m.weight.requires_grad and tuple(m.weight.shape) == (64, 64, 3, 3) and m.bias is None and m.stride == (1, 1)
            and m.padding == (1, 1) and m.dilation == (1, 1) and m.groups == 1 and tuple(x.shape[1:]) == (64, 56, 56)
            and x.shape[0] >= CONV3X3_WGRAD_F32_MIN_N)


def conv3x3_f32_ok(m, x):
    """the hand-written exact-fp32 forward / input gradient applies (training or inference)"""
    return (CONV3X3_F32 and CNN_PRECISION == 0 and x.is_cuda and x.dtype == torch.float32 and tuple(m.weight.shape) == (64, 64, 3, 3)
            and m.bias is None and m.stride == (1, 1) and m.padding == (1, 1) and m.dilation == (1, 1) and m.groups == 1
            and tuple(x.shape[1:]) == (64, 56, 56) and x.shape[0] >= CONV3X3_F32_MIN_N)


def conv3x3_own_ok(m, x):
    return (CONV3X3_OWN and CNN_PRECISION == 1 and x.is_cuda and tuple(m.weight.shape) == (64, 64, 3, 3) and m.bias is None
            and m.stride == (1, 1) and m.padding == (1, 1) and m.dilation == (1, 1) and m.groups == 1 and tuple(x.shape[1:]) == (64, 56, 56))


def conv2d(m, x, want_stats=False):
    """nn.Conv2d module `m` applied to x at the configured operand precision.  want_stats: returns (y, stats) where stats are the
    output's BatchNorm partial sums when the hand-written convolution ran (None otherwise)."""
    if want_stats:
        if conv3x3_own_ok(m, x) and getattr(m, "_rp_bf16", None) is not None:
            xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
            return Conv3x3C64Fn.apply(xb, m._rp_bf16[0], True)
        if CNN_PRECISION == 0 and x.is_cuda and torch.is_grad_enabled():
            # exact-fp32 configuration: the own convolutions write the statistics partials in their epilogue
            bn_src = getattr(x, "_rp_bn", None) if CONV_F32_BN_BWD else None
            if conv3x3_f32_ok(m, x):
                return Conv3x3C64F32Fn.apply(x, m.weight, True, False, bn_src)
            if conv3x3_c128_f32_ok(m, x):
                return Conv3x3C128F32Fn.apply(x, m.weight, m.bias, True)
        return conv2d(m, x), None
    if CNN_PRECISION == 0 or not x.is_cuda:
        if conv3x3_wgrad_f32_ok(m, x) or conv3x3_f32_ok(m, x):
            return Conv3x3C64F32Fn.apply(x, m.weight, False, False, None)
        if conv3x3_c128_f32_ok(m, x):
            return Conv3x3C128F32Fn.apply(x, m.weight, m.bias, False)
        return m(x)
    bf = torch.bfloat16
    ready = getattr(m, "_rp_bf16", None)
    if ready is not None and conv3x3_own_ok(m, x):
        return Conv3x3C64Fn.apply(x if x.dtype == bf else x.to(bf), ready[0], False)
    if ready is not None:
        xb = x if x.dtype == bf else x.to(bf)
        if (m.stride == (1, 1) and m.dilation == (1, 1) and m.groups == 1 and m.kernel_size[0] == m.kernel_size[1]
                and m.kernel_size[0] >= CONV_BWD_AS_FWD_MIN_K):
            return ConvBf16Fn.apply(xb, ready[0], ready[1], m.padding)
        return torch.nn.functional.conv2d(xb, ready[0], ready[1], m.stride, m.padding, m.dilation, m.groups)
    # activations STAY bf16 between the convolutions (the BatchNorm / ReLU / pool kernels of csrc/batchnorm.hip take bf16 storage):
    # only the first convolution's input and the small weights are cast
    return torch.nn.functional.conv2d(x if x.dtype == bf else x.to(bf), m.weight.to(bf), None if m.bias is None else m.bias.to(bf),
                                      m.stride, m.padding, m.dilation, m.groups)


# A/B aid: the identity path's gradient of a BasicBlock is added in the epilogue of its first convolution's input-gradient kernel
CONV_F32_SHARE_INPUT = os.environ.get("RP_CONV_F32_SHARE_INPUT", "1") != "0"


def conv2d_shared(m, x):
    """(y, stats, x') for the FIRST convolution of a residual block whose identity path is x itself: y = m(x), stats as conv2d(...,
    want_stats=True), and x' = the tensor to use for the identity path.  With the own exact-fp32 64-channel convolution x' is an output of the
    convolution's autograd node, so the gradient of the identity path is handed to that node and added in its input-gradient kernel's
    epilogue (no separate add pass); otherwise x' is x."""
    if (CONV_F32_SHARE_INPUT and CNN_PRECISION == 0 and x.is_cuda and torch.is_grad_enabled() and x.requires_grad):
        if conv3x3_f32_ok(m, x):
            return Conv3x3C64F32Fn.apply(x, m.weight, True, True, None)
    y, st = conv2d(m, x, want_stats=True)
    return y, st, x


_NBT_PENDING = None      # inside `with batches_tracked_batch():` the num_batches_tracked buffers to bump at exit


class batches_tracked_batch:
    """The CNN front-end has 13 BatchNorm layers; `num_batches_tracked += 1` on each was a 5-us launch of its own.  Inside this block
    the increments are collected and applied as ONE torch._foreach_add_ at exit (same values, same buffers)."""

    def __enter__(self):
        global _NBT_PENDING
        self.prev, _NBT_PENDING = _NBT_PENDING, []
        return self

    def __exit__(self, et, ev, tb):
        global _NBT_PENDING
        cur, _NBT_PENDING = _NBT_PENDING, self.prev
        if et is None and cur:
            torch._foreach_add_(cur, 1)
        return False


def _bump_batches_tracked(bn):
    if _NBT_PENDING is not None and bn.num_batches_tracked is not None:
        _NBT_PENDING.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked += 1


def bn_act(bn, x, residual=None, relu=True, stats=None):
    """BatchNorm2d module `bn` applied to x, then (+ residual), then ReLU.  GPU tensors take the fused HIP path; CPU tensors
    (only the fixture generator uses the trunk on the CPU, as the reference's torchvision stand-in) take plain PyTorch."""
    if not x.is_cuda:
        y = bn(x)
        if residual is not None:
            y = y + residual
        return torch.relu(y) if relu else y
    if bn.training and bn.track_running_stats:
        _bump_batches_tracked(bn)
    y = BnActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, residual, bn.training, bn.momentum, bn.eps,
                      relu, stats if bn.training else None)
    if _BN_LAST is not None:
        y._rp_bn = _BN_LAST          # (picked up by conv2d when the consumer is a hand-written fp32 convolution)
    return y


class GeodesicLossFn(_Fn):
    """(mean |tau|, mean |phi|) of the geodesic pose loss as one kernel with exact derivatives (csrc/se3loss.hip): the
    PyTorch formulation costs ~240 tiny launches per step.  Ps, Gs: [B,2,7] fp32 on the GPU."""

    @staticmethod
    def forward(ctx, Ps, Gs):
        lib = _lib.load()
        Ps, Gs = Ps.contiguous(), Gs.contiguous()
        _chk(Ps, Gs)
        B = Gs.shape[0]
        losses = _empty(2, like=Gs)
        dmean = _empty(2, B, 2, 7, like=Gs)
        scratch = _empty(60 * B, like=Gs)
        _lib.check(lib.rp_geodesic_loss(_p(Ps), _p(Gs), _p(losses), _p(dmean), _p(scratch), B, _st()), "rp_geodesic_loss")
        ctx.save_for_backward(dmean)
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, gtr, grot):
        (dmean,) = ctx.saved_tensors
        return None, gtr * dmean[0] + grot * dmean[1]


class MaxPool3x3s2Fn(_Fn):
    """nn.MaxPool2d(3, 2, 1) on a channels-last tensor (the stem's pool, src/model.py:130): streaming forward that keeps a
    one-byte window position per output, gather backward (csrc/batchnorm.hip)."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        N, C, H, W = x.shape
        xr = x.permute(0, 2, 3, 1)
        if not xr.is_contiguous():
            xr = xr.contiguous()
        bf = _chk_act(xr)
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty(N, OH, OW, C, device=x.device, dtype=xr.dtype)
        idx = torch.empty(N, OH, OW, C, device=x.device, dtype=torch.uint8)
        _lib.check(lib.rp_maxpool3x3s2_fwd(_p(xr), _p(y), ctypes.c_void_p(idx.data_ptr()), N, H, W, C, bf, _st()), "rp_maxpool3x3s2_fwd")
        ctx.save_for_backward(idx)
        ctx.shape = (N, C, H, W)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        N, C, H, W = ctx.shape
        dyr = dy.permute(0, 2, 3, 1)
        if not dyr.is_contiguous():
            dyr = dyr.contiguous()
        bf = _chk_act(dyr)
        dx = torch.empty(N, H, W, C, device=dy.device, dtype=dyr.dtype)
        _lib.check(lib.rp_maxpool3x3s2_bwd(_p(dyr), ctypes.c_void_p(idx.data_ptr()), _p(dx), N, H, W, C, bf, _st()), "rp_maxpool3x3s2_bwd")
        return dx.permute(0, 3, 1, 2)


def conv_stem_fwd(x_padded_nhwc, w, want_stats=False):
    """rp_conv_stem_fwd: x_padded [N,H+6,W+6,3] (3-pixel zero frame), w = conv1.weight [64,3,7,7] in channels-last memory ->
    y [N,OH,OW,64] (NHWC memory) [, stats partials [blocks,2,64] float64: per-workgroup sums of y and y^2]."""
    lib = _lib.load()
    wr = w.permute(0, 2, 3, 1)
    if not wr.is_contiguous():
        wr = wr.contiguous()
    _chk(x_padded_nhwc, wr)
    N, Hp, Wp, _ = x_padded_nhwc.shape
    H, W = Hp - 6, Wp - 6
    y = torch.empty(N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 64, device=w.device, dtype=torch.float32)
    stats = torch.empty(lib.rp_conv_stem_blocks(N, H, W), 2, 64, device=w.device, dtype=torch.float64) if want_stats else None
    with timed("conv_stem_fwd", 2.0 * N * (H // 2) * (W // 2) * 64 * 147, 4.0 * N * (3.0 * H * W + 64.0 * (H // 2) * (W // 2))):
        _lib.check(lib.rp_conv_stem_fwd(_p(x_padded_nhwc), _p(wr), _p(y), _p(stats), N, H, W, _st()), "rp_conv_stem_fwd")
    return (y, stats) if want_stats else y


def conv3x3_c64_bf16(x_nhwc, w, scale=None, shift=None, want_stats=False):
    """rp_conv3x3_c64_bf16: y = conv3x3(act(x), w), stride 1, pad 1, for x [N,56,56,64] bf16 (NHWC memory) and w [64,3,3,64] bf16 (the
    memory of a channels-last [64,64,3,3] weight); act = max(0, x * scale + shift) when scale / shift [64] fp32 are given.  Returns y
    [N,56,56,64] bf16 [, stats partials [blocks,2,64] float64 of the stored y]."""
    lib = _lib.load()
    bfd = torch.bfloat16
    for t in (x_nhwc, w):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == bfd):
            raise RuntimeError("conv3x3_c64_bf16: contiguous bf16 GPU tensors expected")
    if tuple(x_nhwc.shape[1:]) != (56, 56, 64) or tuple(w.shape) != (64, 3, 3, 64):
        raise RuntimeError("conv3x3_c64_bf16: x [N,56,56,64], w [64,3,3,64] expected, got %s, %s" % (tuple(x_nhwc.shape), tuple(w.shape)))
    if scale is not None:
        _chk(scale, shift)
    N = x_nhwc.shape[0]
    y = torch.empty_like(x_nhwc)
    stats = torch.empty(lib.rp_conv3x3_c64_blocks(N), 2, 64, device=x_nhwc.device, dtype=torch.float64) if want_stats else None
    with timed("conv3x3_c64_bf16", 2.0 * N * 56 * 56 * 64 * 64 * 9, 2.0 * N * 56 * 56 * 128):
        _lib.check(lib.rp_conv3x3_c64_bf16(_p(x_nhwc), _p(w), _p(y), _p(scale), _p(shift), _p(stats), N, 56, 56, _st()), "rp_conv3x3_c64_bf16")
    return (y, stats) if want_stats else y


def conv3x3_c64_wgrad_bf16(x_nhwc, dy_nhwc):
    """rp_conv3x3_c64_wgrad_bf16: dW of y = conv3x3(x, w) (stride 1, pad 1, 64 -> 64, 56 x 56) from x and dY [N,56,56,64] bf16 (NHWC
    memory) -> [64,3,3,64] bf16 (the memory of a channels-last [64,64,3,3] weight)."""
    lib = _lib.load()
    for t in (x_nhwc, dy_nhwc):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.bfloat16 and tuple(t.shape[1:]) == (56, 56, 64)):
            raise RuntimeError("conv3x3_c64_wgrad_bf16: contiguous bf16 [N,56,56,64] GPU tensors expected")
    N = x_nhwc.shape[0]
    nb = lib.rp_conv3x3_c64_wgrad_workspace_bytes(N)
    ws = torch.empty(nb // 4, device=x_nhwc.device, dtype=torch.float32)
    dw = torch.empty(64, 3, 3, 64, device=x_nhwc.device, dtype=torch.bfloat16)
    with timed("conv3x3_c64_wgrad_bf16", 2.0 * N * 56 * 56 * 64 * 64 * 9, 2.0 * N * 56 * 56 * 128):
        _lib.check(lib.rp_conv3x3_c64_wgrad_bf16(_p(x_nhwc), _p(dy_nhwc), _p(dw), _p(ws), nb, N, 56, 56, _st()), "rp_conv3x3_c64_wgrad_bf16")
    return dw


def conv3x3_c64_wgrad_f32(x_nhwc, dy_nhwc):
    """rp_conv3x3_c64_wgrad_f32: dW of y = conv3x3(x, w) (stride 1, pad 1, 64 -> 64, 56 x 56) in exact fp32 from x and dY [N,56,56,64]
    fp32 (NHWC memory) -> [64,3,3,64] fp32 (the memory of a channels-last [64,64,3,3] weight gradient)."""
    lib = _lib.load()
    for t in (x_nhwc, dy_nhwc):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and tuple(t.shape[1:]) == (56, 56, 64)):
            raise RuntimeError("conv3x3_c64_wgrad_f32: contiguous fp32 [N,56,56,64] GPU tensors expected")
    N = x_nhwc.shape[0]
    nb = lib.rp_conv3x3_c64_wgrad_f32_workspace_bytes(N)
    ws = torch.empty(nb // 4, device=x_nhwc.device, dtype=torch.float32)
    dw = torch.empty(64, 3, 3, 64, device=x_nhwc.device, dtype=torch.float32)
    with timed("conv3x3_c64_wgrad_f32", 2.0 * N * 56 * 56 * 64 * 64 * 9, 4.0 * N * 56 * 56 * 128):
        _lib.check(lib.rp_conv3x3_c64_wgrad_f32(_p(x_nhwc), _p(dy_nhwc), _p(dw), _p(ws), nb, N, 56, 56, _st()), "rp_conv3x3_c64_wgrad_f32")
    return dw


def conv_stem_fwd_bf16(x_padded_nhwc, w, want_stats=False):
    """rp_conv_stem_fwd_bf16: as conv_stem_fwd (fp32 framed image, fp32 channels-last filter), bf16 operands on chip -> y
    [N,OH,OW,64] bf16 [, stats partials of the stored y]."""
    lib = _lib.load()
    wr = w.permute(0, 2, 3, 1)
    if not wr.is_contiguous():
        wr = wr.contiguous()
    _chk(x_padded_nhwc, wr)
    N, Hp, Wp, _ = x_padded_nhwc.shape
    H, W = Hp - 6, Wp - 6
    y = torch.empty(N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 64, device=w.device, dtype=torch.bfloat16)
    stats = torch.empty(lib.rp_conv_stem_bf16_blocks(N, H, W), 2, 64, device=w.device, dtype=torch.float64) if want_stats else None
    with timed("conv_stem_fwd_bf16", 2.0 * N * (H // 2) * (W // 2) * 64 * 147, N * (4.0 * 3 * H * W + 2.0 * 64 * (H // 2) * (W // 2))):
        _lib.check(lib.rp_conv_stem_fwd_bf16(_p(x_padded_nhwc), _p(wr), _p(y), _p(stats), N, H, W, _st()), "rp_conv_stem_fwd_bf16")
    return (y, stats) if want_stats else y


def conv_stem_wgrad_bf16(x_padded_nhwc, dy_nhwc):
    """rp_conv_stem_wgrad_bf16: dW of the stem convolution from the fp32 framed image [N,230,230,3] and dY [N,112,112,64] bf16 ->
    [64,7,7,3] fp32 (the memory of a channels-last [64,3,7,7] weight)."""
    lib = _lib.load()
    _chk(x_padded_nhwc)
    if not (dy_nhwc.is_cuda and dy_nhwc.is_contiguous() and dy_nhwc.dtype == torch.bfloat16):
        raise RuntimeError("conv_stem_wgrad_bf16: contiguous bf16 dY expected")
    N = x_padded_nhwc.shape[0]
    if tuple(x_padded_nhwc.shape[1:]) != (230, 230, 3) or tuple(dy_nhwc.shape) != (N, 112, 112, 64):
        raise RuntimeError("conv_stem_wgrad_bf16: x_padded [N,230,230,3], dY [N,112,112,64] expected")
    nb = lib.rp_conv_stem_wgrad_workspace_bytes(N)
    ws = torch.empty((nb + 3) // 4, device=dy_nhwc.device, dtype=torch.float32)
    dw = torch.empty(64, 7, 7, 3, device=dy_nhwc.device, dtype=torch.float32)
    with timed("conv_stem_wgrad_bf16", 2.0 * N * 112 * 112 * 64 * 147, N * (4.0 * 3 * 224 * 224 + 2.0 * 64 * 112 * 112)):
        _lib.check(lib.rp_conv_stem_wgrad_bf16(_p(x_padded_nhwc), _p(dy_nhwc), _p(dw), _p(ws), nb, N, 224, 224, _st()), "rp_conv_stem_wgrad_bf16")
    return dw


def conv_stem_wgrad_f32(x_padded_nhwc, dy_nhwc):
    """rp_conv_stem_wgrad_f32: dW of the stem convolution in exact fp32 from the framed image [N,230,230,3] and dY [N,112,112,64] fp32 ->
    [64,7,7,3] fp32 (the memory of a channels-last [64,3,7,7] weight)."""
    lib = _lib.load()
    _chk(x_padded_nhwc, dy_nhwc)
    N = x_padded_nhwc.shape[0]
    if tuple(x_padded_nhwc.shape[1:]) != (230, 230, 3) or tuple(dy_nhwc.shape) != (N, 112, 112, 64):
        raise RuntimeError("conv_stem_wgrad_f32: x_padded [N,230,230,3], dY [N,112,112,64] expected")
    nb = lib.rp_conv_stem_wgrad_f32_workspace_bytes(N)
    ws = torch.empty((nb + 3) // 4, device=dy_nhwc.device, dtype=torch.float32)
    dw = torch.empty(64, 7, 7, 3, device=dy_nhwc.device, dtype=torch.float32)
    with timed("conv_stem_wgrad_f32", 2.0 * N * 112 * 112 * 64 * 147, 4.0 * N * (3.0 * 224 * 224 + 64.0 * 112 * 112)):
        _lib.check(lib.rp_conv_stem_wgrad_f32(_p(x_padded_nhwc), _p(dy_nhwc), _p(dw), _p(ws), nb, N, 224, 224, _st()), "rp_conv_stem_wgrad_f32")
    return dw


STEM_CONV = True      # hand-written stem convolution forward
STEM_WGRAD = os.environ.get("RP_STEM_WGRAD", "1") != "0"    # ... and (224 x 224) its weight gradient: bf16 configuration and exact fp32
STEM_STATS = True    # ... with the BatchNorm batch statistics from its epilogue


class StemConvFn(_Fn):
    """resnet.conv1 (7x7 / 2, pad 3, 3 -> 64, no bias; src/model.py:127) on the zero-framed NHWC image: forward = csrc/conv_stem.hip;
    weight gradient = csrc/conv_stem_wgrad_f32.hip at 224 x 224 (otherwise MIOpen's backward-weights on the same framed buffer, padding
    0 there: identical arithmetic); the image itself needs no gradient."""

    @staticmethod
    def forward(ctx, xp, w, want_stats=False):
        """-> y (channels-last [N,64,OH,OW]) [, stats partials of y for the BatchNorm that follows (not differentiable)]"""
        ctx.save_for_backward(xp, w)
        if want_stats:
            y, stats = conv_stem_fwd(xp, w, want_stats=True)
            ctx.mark_non_differentiable(stats)
            return y.permute(0, 3, 1, 2), stats
        return conv_stem_fwd(xp, w).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy, *_):
        xp, w = ctx.saved_tensors
        dw = None
        if ctx.needs_input_grad[1]:
            dy = dy.contiguous(memory_format=torch.channels_last)
            if STEM_WGRAD and tuple(xp.shape[1:]) == (230, 230, 3) and dy.dtype == torch.float32 and tuple(dy.shape[1:]) == (64, 112, 112):
                dw = conv_stem_wgrad_f32(xp, dy.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)      # csrc/conv_stem_wgrad_f32.hip
            else:
                dw = torch.ops.aten.convolution_backward(dy, xp.permute(0, 3, 1, 2), w, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1,
                                                         [False, True, False])[1]
        return None, dw, None


class StemConvBf16Fn(_Fn):
    """resnet.conv1 in the bf16 configuration: forward = csrc/conv_stem_bf16.hip on the fp32 framed image and the fp32 master filter
    (rounded to bf16 on chip); weight gradient = MIOpen's bf16 backward-weights on a bf16 copy of the framed image (padding 0), returned
    in fp32 for the fp32 master; the image needs no gradient."""

    @staticmethod
    def forward(ctx, xp, w, want_stats=False):
        ctx.save_for_backward(xp, w)
        if want_stats:
            y, stats = conv_stem_fwd_bf16(xp, w, want_stats=True)
            ctx.mark_non_differentiable(stats)
            return y.permute(0, 3, 1, 2), stats
        return conv_stem_fwd_bf16(xp, w).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy, *_):
        xp, w = ctx.saved_tensors
        dw = None
        if ctx.needs_input_grad[1]:
            bf = torch.bfloat16
            dy = dy.contiguous(memory_format=torch.channels_last)
            if STEM_WGRAD and tuple(xp.shape[1:]) == (230, 230, 3) and dy.dtype == bf:
                dw = conv_stem_wgrad_bf16(xp, dy.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
            else:
                dw = torch.ops.aten.convolution_backward(dy.to(bf), xp.to(bf).permute(0, 3, 1, 2), w.to(bf), None, [2, 2], [0, 0], [1, 1], False,
                                                         [0, 0], 1, [False, True, False])[1].float()
        return None, dw, None


def stem_conv_ok(conv, images):
    return (STEM_CONV and images.is_cuda and tuple(conv.weight.shape) == (64, 3, 7, 7) and conv.bias is None
            and conv.stride == (2, 2) and conv.padding == (3, 3))


FUSE_STEM_POOL = True


class BnReluPoolFn(_Fn):
    """maxpool3x3s2(relu(batch_norm(x))) of the stem (reference src/model.py:127-130) without the two [N,112,112,64] intermediates:
    statistics pass, then ONE pass that normalises, clamps and pools; the backward gathers the pool gradient inside both
    BatchNorm-backward passes.  Bit-identical to BnActFn + MaxPool3x3s2Fn (csrc/batchnorm.hip)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, stats=None):
        """stats: sums of x and x^2 per producing workgroup [blocks,2,C] float64 (from the stem convolution's epilogue) -- then the
        statistics pass over x is skipped"""
        lib = _lib.load()
        N, C, H, W = x.shape
        xr = x.permute(0, 2, 3, 1)
        if not xr.is_contiguous():
            xr = xr.contiguous()
        _chk(gamma, beta)
        bf = _chk_act(xr)
        R = N * H * W
        if training and stats is not None:
            mean, rstd = _empty(C, like=xr), _empty(C, like=xr)
            _lib.check(lib.rp_bn_stats_from_partials(_p(stats), stats.shape[0], R, C, _p(_zeros(C, xr.device)), _p(mean), _p(rstd),
                                                     _p(running_mean), _p(running_var), float(momentum), float(eps), _st()),
                       "rp_bn_stats_from_partials")
        elif training:
            mean, rstd = _empty(C, like=xr), _empty(C, like=xr)
            part = torch.empty(lib.rp_bn_partial_blocks(R) * 2 * C, device=x.device, dtype=torch.float64)
            _lib.check(lib.rp_bn_stats(_p(xr), R, C, _p(part), _p(mean), _p(rstd), _p(running_mean), _p(running_var),
                                       float(momentum), float(eps), bf, _st()), "rp_bn_stats")
        else:
            mean, rstd = running_mean, torch.rsqrt(running_var + eps)
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty(N, OH, OW, C, device=x.device, dtype=xr.dtype)
        idx = torch.empty(N, OH, OW, C, device=x.device, dtype=torch.uint8)
        _lib.check(lib.rp_bn_relu_pool_fwd(_p(xr), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(y), ctypes.c_void_p(idx.data_ptr()),
                                           N, H, W, C, bf, _st()), "rp_bn_relu_pool_fwd")
        if _train(ctx):
            ctx.save_for_backward(xr, idx, mean, rstd, gamma, beta)
            ctx.cfg = (N, C, H, W, bool(training))
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        xr, idx, mean, rstd, gamma, beta = ctx.saved_tensors
        N, C, H, W, training = ctx.cfg
        dyr = dy.permute(0, 2, 3, 1)
        if not dyr.is_contiguous():
            dyr = dyr.contiguous()
        if dyr.dtype != xr.dtype:
            dyr = dyr.to(xr.dtype)
        bf = _chk_act(dyr, xr)
        dx = torch.empty_like(xr)
        dgamma, dbeta, c12 = _empty(C, like=xr), _empty(C, like=xr), _empty(2 * C, like=xr)
        part = torch.empty(lib.rp_bn_partial_blocks(N * H * W) * 2 * C, device=xr.device, dtype=torch.float64)
        _lib.check(lib.rp_bn_relu_pool_bwd(_p(dyr), ctypes.c_void_p(idx.data_ptr()), _p(xr), _p(mean), _p(rstd), _p(gamma), _p(beta),
                                           _p(dx), _p(dgamma), _p(dbeta), _p(part), _p(c12), N, H, W, C, 1 if training else 0, bf, _st()),
                   "rp_bn_relu_pool_bwd")
        return dx.permute(0, 3, 1, 2), dgamma, dbeta, None, None, None, None, None, None


_ZEROS = {}


def _zeros(n, device):
    key = (device, n)
    if key not in _ZEROS:
        _ZEROS[key] = torch.zeros(n, device=device, dtype=torch.float32)
    return _ZEROS[key]


def bn_relu_maxpool(bn, pool, x, stats=None):
    """pool(relu(bn(x))) for the stem (nn.BatchNorm2d `bn`, nn.MaxPool2d(3, 2, 1) `pool`); stats: see BnReluPoolFn."""
    if not x.is_cuda or not FUSE_STEM_POOL:
        return maxpool3x3s2(pool, bn_act(bn, x))
    if bn.training and bn.track_running_stats:
        _bump_batches_tracked(bn)
    return BnReluPoolFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, bn.momentum, bn.eps, stats)


def maxpool3x3s2(pool, x):
    """the stem's nn.MaxPool2d(3, 2, 1): HIP kernels on the GPU, the module itself on CPU tensors (fixture generation only)"""
    if not x.is_cuda:
        return pool(x)
    return MaxPool3x3s2Fn.apply(x)


def augment_pairs(images_u8, params, out_h, out_w):
    """rp_augment_pairs: uint8 [B,2,H,W,3] BGR + params [B,9] -> fp32 [B,2,3,out_h,out_w] BGR 0..255 (csrc/augment.hip)."""
    lib = _lib.load()
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 5 or images_u8.shape[1] != 2 or images_u8.shape[4] != 3:
        raise ValueError("augment_pairs wants uint8 [B,2,H,W,3], got %s %s" % (images_u8.dtype, tuple(images_u8.shape)))
    if not images_u8.is_cuda:
        raise RuntimeError("augment_pairs: HIP kernel, the batch must be resident on the GPU")
    images_u8 = images_u8.contiguous()
    params = params.to(torch.float32).contiguous()
    B, _, H, W, _ = images_u8.shape
    if tuple(params.shape) != (B, 9):
        raise ValueError("augment_pairs params must be [B,9]")
    out = torch.empty(B, 2, 3, out_h, out_w, device=images_u8.device, dtype=torch.float32)
    ws = torch.empty(B * lib.rp_augment_blocks(), device=images_u8.device, dtype=torch.float64)
    _lib.check(lib.rp_augment_pairs(_p(images_u8), _p(params), _p(out), _p(ws), B, H, W, out_h, out_w, _st()), "rp_augment_pairs")
    return out
