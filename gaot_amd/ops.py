"""Host-side op layer: thin ctypes calls into libgaot_hip.so + the torch.autograd.Functions the layers use.

PyTorch is plumbing here (device memory, streams, autograd tape).  Every arithmetic op on the hot path is a
hand-written gfx950 kernel reached through the C ABI in include/gaot_hip.h.  There is no CPU / eager
fallback: calling an op without the library or with host tensors raises.
"""
import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib as L

ACT = {"none": L.ACT_NONE, "gelu": L.ACT_GELU, "relu": L.ACT_RELU}
_ACT_BWD = {L.ACT_GELU: L.ACT_GELU_BWD, L.ACT_RELU: L.ACT_RELU_BWD}


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("gaot_amd ops take device tensors only (no CPU fallback); got a host tensor")


def _f32(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise TypeError(f"gaot_amd kernels are fp32; got {t.dtype}")


def _rowmajor(t: torch.Tensor) -> Tuple[torch.Tensor, int]:
    """2-D tensor with unit inner stride -> (tensor, leading dim)."""
    assert t.dim() == 2
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    ld = t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])
    return t, ld


# --------------------------------------------------------------------------------------------
# weights generation: kernels that update parameters through raw pointers (trainer.FlatAdamW, its hipGraph replay) do not
# move Parameter._version, so every cache of weight-derived values (AGNO kernel values, geoembed row bias, rollout graphs)
# keys on (Parameter._version..., weights_generation()).
# --------------------------------------------------------------------------------------------
_WEIGHTS_GEN = [0]


def weights_generation() -> int:
    return _WEIGHTS_GEN[0]


def bump_weights_generation() -> None:
    _WEIGHTS_GEN[0] += 1


# --------------------------------------------------------------------------------------------
# cut points of a staged backward (trainer.TrainStep with > 1 rank): the model calls cut(t) where the backward pass may be
# split; without a hook it is the identity.
# --------------------------------------------------------------------------------------------
_CUT_HOOK = [None]


def set_cut_hook(fn) -> None:
    _CUT_HOOK[0] = fn


def cut(t: torch.Tensor) -> torch.Tensor:
    fn = _CUT_HOOK[0]
    return t if fn is None else fn(t)


# --------------------------------------------------------------------------------------------
# Precision of the products on the bf16 matrix pipe (gaot_gemm_desc.pieces and the `pieces` argument of the attention / kernel-MLP
# entry points), per product kind:
#   "nt": activations x weight^T (every forward Linear), "nn": dY x weight (input gradients), "tn": dY^T x X (weight gradients),
#   "attn": Q / K / V / dO and the probabilities P / dS of the flash-attention kernels, "kmlp": the fused row-wise MLP kernels behind
#   a smooth activation (ReLU chains always keep exact products, _KernelMLP._pieces).
# 3 = THE DEFAULT ("f32"): every fp32 operand carried at fp32 width -- the arithmetic of the reference, which computes in fp32 end to
#     end (base_trainer.py:63-68).  Formed as two fp16 pieces of the power-of-two scaled operand wherever a product has its operands'
#     magnitude words (`pieces` = 4 on the C side: tile GEMMs, grouped weight gradients, attention for head_dim <= 64; the section
#     "Magnitude words" below), else as THREE bf16 pieces, six piece products (exact; the kernel MLP; everything with
#     GAOT_F32_PIECES=bf16x3).
# 2 = two pieces, both rounded to nearest, three piece products: 16 significant bits per operand, half the matrix-pipe work
#     ("bf16x2", opt-in: set_precision("bf16x2") / GAOT_PRECISION=bf16x2).  Measured at the bench configuration against the
#     reference algorithm evaluated in float64 (tools/grad_errors.py): exact products: output 1.25e-7, worst gradient tensor 8.8e-7;
#     two pieces: output 4.3e-7, worst gradient tensor 6.6e-6 (the fp32 reference itself: 7.1e-7 / <= 4.4e-7 outside the four
#     statistics-gated tensors): inside the 1e-5 output bar, but 4-15x further from float64 than the reference's fp32 on gradients.
# --------------------------------------------------------------------------------------------
_PIECES = {"nt": 3, "nn": 3, "tn": 3, "attn": 3, "kmlp": 3}
_PRECISIONS = {"f32": 3, "bf16x2": 2}


def set_gemm_pieces(all: Optional[int] = None, *, nt: Optional[int] = None, nn: Optional[int] = None, tn: Optional[int] = None,
                    attn: Optional[int] = None, kmlp: Optional[int] = None) -> dict:
    """pieces per operand (2 or 3) per product kind.  The positional argument sets the three GEMM kinds (nt, nn, tn); keywords set
    exactly the kind they name.  Returns the previous setting (a dict that can be passed back as keywords)."""
    old = dict(_PIECES)
    if all is not None:
        nt = all if nt is None else nt
        nn = all if nn is None else nn
        tn = all if tn is None else tn
    for k, v in (("nt", nt), ("nn", nn), ("tn", tn), ("attn", attn), ("kmlp", kmlp)):
        if v is not None:
            if v not in (2, 3):
                raise ValueError(f"pieces must be 2 or 3, got {v}")
            _PIECES[k] = v
    return old


def set_precision(name: str) -> dict:
    """"f32" (default): exact three-piece products everywhere; "bf16x2": two rounded bf16 pieces per operand everywhere (GEMM tiles,
    grouped weight gradients, attention, the GELU kernel MLP).  Storage and accumulation are fp32 either way.  Returns the previous
    per-kind setting (restore with set_gemm_pieces(**old))."""
    if name not in _PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}, got {name!r}")
    v = _PRECISIONS[name]
    return set_gemm_pieces(nt=v, nn=v, tn=v, attn=v, kmlp=v)


def precision() -> str:
    vals = set(_PIECES.values())
    return "f32" if vals == {3} else ("bf16x2" if vals == {2} else "mixed")


if os.environ.get("GAOT_PRECISION"):
    set_precision(os.environ["GAOT_PRECISION"])
if os.environ.get("GAOT_GEMM_PIECES"):          # tools: "3", "2" or "nt,nn,tn" (the GEMM kinds only)
    _parts = [int(t) for t in os.environ["GAOT_GEMM_PIECES"].split(",")]
    if len(_parts) not in (1, 3):
        raise ValueError("GAOT_GEMM_PIECES: one value or nt,nn,tn")
    set_gemm_pieces(*_parts) if len(_parts) == 1 else set_gemm_pieces(nt=_parts[0], nn=_parts[1], tn=_parts[2])


# --------------------------------------------------------------------------------------------
# Magnitude words: the fp16-piece products (gaot_gemm_desc.pieces = 4, the way the "f32" precision runs on the split tiles) scale each
# operand by the power of two that brings its largest magnitude to ~2^14 before splitting it into two fp16 pieces; the kernels read
# that magnitude from DEVICE memory (one float per tensor, so captured launches follow the data).  A word is
#   * published by the kernel that produced the tensor (GEMM epilogues: gaot_gemm_desc.c_absmax), or
#   * computed by gaot_absmax_grouped (weights: one launch per forward pass over every parameter, refresh_weight_amax; any other
#     tensor: on first use, amax_for),
# and travels as the attribute `_gaot_amax` = (word, tensor._version, pass id) of the tensor OBJECT (lost by reshapes / saved_tensors: the
# autograd Functions below carry the words of saved operands in their ctx).  Words live in a zeroed arena that is replaced, never
# reset, at the start of every forward pass (begin_pass), so a captured graph re-zeroes its own words at every replay.
# --------------------------------------------------------------------------------------------
_F16_PIECES = [os.environ.get("GAOT_F32_PIECES", "fp16x2") != "bf16x3"]      # False: the "f32" precision runs three bf16 pieces everywhere (A/B)


def set_f32_pieces(name: str) -> str:
    """how the "f32" precision forms its products on the split tiles: "fp16x2" (default: two fp16 pieces of the scaled operand, three
    piece products) or "bf16x3" (three bf16 pieces, six piece products).  Both carry every operand at fp32 width (bf16x3 exactly; fp16x2 up to the operand's last bit, common.h split2h_pair).  Returns the old name."""
    if name not in ("fp16x2", "bf16x3"):
        raise ValueError(f"f32 pieces must be 'fp16x2' or 'bf16x3', got {name!r}")
    old = "fp16x2" if _F16_PIECES[0] else "bf16x3"
    _F16_PIECES[0] = name == "fp16x2"
    return old


class _Arena:
    __slots__ = ("buf", "used", "n")

    def __init__(self, device, n=256):
        self.buf = torch.zeros(n * AMAX_SLOTS, device=device, dtype=torch.float32)       # n words of AMAX_SLOTS floats
        self.n = n
        self.used = 0


AMAX_SLOTS = 32 * 32           # floats per magnitude word: include/gaot_hip.h GAOT_AMAX_SLOTS * GAOT_AMAX_STRIDE (4 KB)
_ARENA = [None]
_WEIGHT_AMAX: list = []        # (lo, hi, word) of the current pass: every parameter and every fused weight group


_PASS_ID = [0]                 # serial number of the forward pass in progress: words cached on tensor objects are only trusted within it


def begin_pass(keep_weights: bool = False) -> None:
    """top of a forward pass: magnitude words of the previous pass are not reused (their arena lives on while a ctx still holds it),
    and words remembered on tensor OBJECTS by an earlier pass are ignored (a persistent input -- TrainStep / autograph / rollout static
    buffers -- keeps its object and, between in-place copies, even its `_version` across the eager warm-up and the capture that follows:
    a hit there would leave the captured graph without its absmax launch, scaling every replay by the warm-up batch's magnitude).
    keep_weights: the weight table (words + planes) of the previous pass stays (inference over unchanged weights, weights_current())."""
    _ARENA[0] = None
    _PASS_ID[0] += 1
    if not keep_weights:
        _WEIGHT_AMAX.clear()
        _WEIGHT_TABLE_FOR[0] = None


_WEIGHT_TABLE_FOR = [None]        # the parameter list OBJECT the table was built for (held: an id() alone could be reused by another list)
_PINNED_TABLES: list = []         # tables a captured inference graph reads without refreshing them itself: kept alive for good


def weights_current(params) -> bool:
    """is the weight table the one of exactly these parameters as they are now?  (every owner alive, same Parameter._version, same
    weights_generation(): nothing wrote a weight since refresh_weight_amax built it).  Writes that bypass both counters -- `p.data.copy_(ema)`,
    raw-pointer kernels -- must be followed by ops.bump_weights_generation() (trainer.FlatAdamW does; load_state_dict moves Parameter._version): the table
    holds the B operand itself (fp16 planes), not just a cache key."""
    if not _WEIGHT_AMAX or _WEIGHT_TABLE_FOR[0] is not params:
        return False
    gen = weights_generation()
    for e in _WEIGHT_AMAX:
        if e[5] != gen or not all((q := r()) is not None and q._version == v for r, v in zip(e[3], e[4])):
            return False
    return True


def pin_weight_table() -> None:
    """a graph being captured will read the current table's words and planes at every replay: never free them"""
    _PINNED_TABLES.append(list(_WEIGHT_AMAX))          # (never trimmed: a graph that reads a table may be replayed for as long as the process lives)
    _PLANES_IN_GRAPHS[0] = True


def _amax_words(n: int, device) -> List[torch.Tensor]:
    a = _ARENA[0]
    if a is None or a.buf.device != device or a.used + n > a.n:
        a = _Arena(device, max(256, n))
        _ARENA[0] = a
    out = [a.buf[(a.used + i) * AMAX_SLOTS:(a.used + i + 1) * AMAX_SLOTS] for i in range(n)]
    a.used += n
    return out


def _absmax_launch(pairs) -> None:
    """pairs: (2-D tensor with unit inner stride, word): ONE gaot_absmax_grouped launch"""
    if _AMAX_TRACE:
        import traceback
        fr = [f"{f.name}:{f.lineno}" for f in traceback.extract_stack()[-7:-1]]
        print(f"[absmax] {[tuple(t.shape) for t, _ in pairs][:4]}{'...' if len(pairs) > 4 else ''} <- {' < '.join(reversed(fr))}", flush=True)
    arr = (L.AbsmaxItem * len(pairs))()
    for i, (t, w) in enumerate(pairs):
        arr[i] = L.AbsmaxItem(t.data_ptr(), t.stride(0) if t.shape[0] > 1 else t.shape[1], t.shape[0], t.shape[1], w.data_ptr())
    L.check(L.load().gaot_absmax_grouped(arr, len(pairs), _stream()), "gaot_absmax_grouped")


def _version_of(o) -> int:
    """Tensor._version, or -1 for inference tensors (torch.inference_mode(): they have no version counter and cannot be written in
    place after the mode ends, so a constant stands in)"""
    try:
        return o._version
    except RuntimeError:
        return -1


def _amax_get(*objs):
    pid = _PASS_ID[0]
    for o in objs:
        a = getattr(o, "_gaot_amax", None)
        if a is not None and a[2] == pid and a[1] == _version_of(o) and a[0].device == o.device:
            return a[0]
    return None


def _amax_set(word, *objs) -> None:
    pid = _PASS_ID[0]
    for o in objs:
        if o is not None:
            o._gaot_amax = (word, _version_of(o), pid)


def amax_for(t2d: torch.Tensor, *aliases) -> torch.Tensor:
    """the magnitude word of a 2-D operand (unit inner stride): the one its producer published / an earlier use computed (carried by
    the tensor object or one of `aliases`, the objects it is a reshape of), else one absmax launch; remembered on all of them"""
    w = _amax_get(t2d, *aliases)
    if w is None:
        w = _amax_words(1, t2d.device)[0]
        _absmax_launch([(t2d, w)])
    _amax_set(w, t2d, *aliases)
    return w


def _want_word(device):
    """a fresh (zero) magnitude word for a producer kernel to publish into, or None when no fp16-piece product will ask for it"""
    return _amax_words(1, device)[0] if wants_amax() else None


def _publish(word, *objs) -> None:
    """hand the word a kernel published for its output to the tensor objects that carry that output"""
    if word is not None:
        _amax_set(word, *objs)


def wants_amax() -> bool:
    """does the current precision run fp16-piece products (i.e. are magnitude words of any use)?"""
    return _F16_PIECES[0] and 3 in _PIECES.values()


_PLANE_CACHE: dict = {}        # (address, rows, cols, device) -> (planes_k, planes_t): persistent fp16 plane buffers of a weight matrix
_PLANES_IN_GRAPHS = [False]    # a hipGraph was captured over the plane buffers (refresh_weight_amax / pin_weight_table): never evict again
_USE_PLANES = os.environ.get("GAOT_WEIGHT_PLANES", "1") != "0"        # A/B switch (bit-identical results either way)


def refresh_weight_amax(params, groups=()) -> None:
    """magnitude words of every parameter (and of every fused weight group read as one matrix) in ONE launch, and -- second launch --
    the two fp16 planes of every weight MATRIX scaled through its word, as stored and transposed (gaot_gemm_desc.b_planes: the tile
    kernels then stage their B operand without any split arithmetic).  The model calls it at the top of a forward pass; the products
    of that pass and of its backward look their weight operands up by address.  An entry is trusted while its parameter is alive and
    unchanged (weak reference, Parameter._version, weights_generation())."""
    import weakref
    need_t = torch.is_grad_enabled()
    if torch.cuda.is_current_stream_capturing():
        _PLANES_IN_GRAPHS[0] = True       # the graph being captured writes (and its products read) the plane buffers by address at every replay
    items, owners = [], []
    for g in groups:
        g = list(g)
        v = adjacent_rows(g)
        if v is not None and v.is_cuda and v.dtype == torch.float32:
            items.append(v)
            owners.append(g)
    grouped = {id(q) for own in owners for q in own}
    for q in params:
        if id(q) in grouped:
            continue          # a member of a fused group is found through the group's entry (address containment)
        if q.is_cuda and q.dtype == torch.float32 and q.dim() >= 1 and q.numel() > 0 and q.is_contiguous():
            items.append(q.detach().reshape(q.shape[0], -1))
            owners.append([q])
    if not items:
        return
    words = _amax_words(len(items), items[0].device)
    _absmax_launch(list(zip(items, words)))
    gen = weights_generation()
    pl_items = []
    for t, w, own in zip(items, words, owners):
        lo = t.data_ptr()
        rows, cols = t.shape
        pk = pt = None
        # planes for matrices the tile kernels can read as B: once per storage (a parameter inside a fused group is covered by the group)
        if (_USE_PLANES and rows % 16 == 0 and cols % 16 == 0 and min(rows, cols) >= 32 and not (len(own) == 1 and id(own[0]) in grouped)
                and not (lo & 15)):
            key = (lo, rows, cols, t.device.index)
            buf = _PLANE_CACHE.get(key)
            if buf is None:
                if len(_PLANE_CACHE) > 512 and not _PLANES_IN_GRAPHS[0]:
                    _PLANE_CACHE.clear()          # (only while no captured graph holds plane addresses: those buffers must outlive it)
                buf = (torch.empty(2 * rows * cols, device=t.device, dtype=torch.int16), torch.empty(2 * rows * cols, device=t.device, dtype=torch.int16))
                _PLANE_CACHE[key] = buf
            pk, pt = buf
            if not need_t:
                pt = None          # inference (rollouts): no input-gradient products, only the planes as stored are read
            pl_items.append(L.F16PlanesItem(lo, cols, rows, cols, w.data_ptr(), pk.data_ptr(), None if pt is None else pt.data_ptr()))
        _WEIGHT_AMAX.append((lo, lo + t.numel() * 4, w, [weakref.ref(q) for q in own], [q._version for q in own], gen, rows, cols, pk, pt))
    if pl_items:
        arr = (L.F16PlanesItem * len(pl_items))(*pl_items)
        L.check(L.load().gaot_split_f16_planes_grouped(arr, len(pl_items), _stream()), "gaot_split_f16_planes_grouped")
    _WEIGHT_TABLE_FOR[0] = params


def _weight_entry(w2d: torch.Tensor):
    lo = w2d.data_ptr()
    hi = lo + ((w2d.shape[0] - 1) * w2d.stride(0) + w2d.shape[1]) * 4 if w2d.shape[0] > 1 else lo + w2d.shape[1] * 4
    best = None
    gen = weights_generation()
    for e in _WEIGHT_AMAX:
        a, b, w, refs, vers, g_ = e[:6]
        if a <= lo and hi <= b and (best is None or b - a < best[1] - best[0]) and w.device == w2d.device and g_ == gen:
            if all((q := r()) is not None and q._version == v for r, v in zip(refs, vers)):
                best = e
    return best


def weight_amax(w2d: torch.Tensor) -> torch.Tensor:
    """magnitude word of a weight operand: the smallest registered parameter / fused group that contains the view (a bound over a
    superset is as good), else as amax_for"""
    e = _weight_entry(w2d)
    return e[2] if e is not None else amax_for(w2d)


def weight_operand(w2d: torch.Tensor, b_kmajor: bool):
    """(magnitude word, planes pointer or None, ld, plane stride) of a weight B operand: the pre-split fp16 planes of the registered
    matrix that contains the view -- as stored for x W^T (b_kmajor), transposed for dY W"""
    e = _weight_entry(w2d)
    if e is None:
        return amax_for(w2d), None, 0, 0
    lo, _, word, _, _, _, rows, cols, pk, pt = e
    if pk is None or (w2d.shape[0] > 1 and w2d.stride(0) != cols):
        return word, None, 0, 0
    off = (w2d.data_ptr() - lo) // 4
    r0, c0 = off // cols, off % cols
    # pieces [row][k / 16][piece][k % 16] (include/gaot_hip.h): a view must start on a 16-wide k group
    if b_kmajor:
        if c0 % 16:
            return word, None, 0, 0
        ptr, ld = pk.data_ptr() + 2 * (r0 * 2 * cols + 2 * c0), 2 * cols
    elif pt is None or r0 % 16:
        return word, None, 0, 0
    else:
        ptr, ld = pt.data_ptr() + 2 * (c0 * 2 * rows + 2 * r0), 2 * rows
    if ptr & 15:
        return word, None, 0, 0
    return word, ptr, ld, 16


_GEMM_TRACE = os.environ.get("GAOT_GEMM_TRACE", "0") == "1"      # tools: print every distinct product of gemm() with its tile family
_GEMM_TRACE_SEEN = set()
_AMAX_TRACE = os.environ.get("GAOT_AMAX_TRACE", "0") == "1"      # tools: print every fallback absmax launch with its call site
_PATH_CACHE: dict = {}
_PUBLISH_C = os.environ.get("GAOT_NO_CAMAX", "0") != "1"       # A/B switch (tools): GEMM epilogues publish the output's magnitude word


def _gemm_path(d, key) -> int:
    p_ = _PATH_CACHE.get(key)
    if p_ is None:
        p_ = int(L.load().gaot_gemm_path(C.byref(d)))
        if p_ < 0:
            L.check(-1, "gaot_gemm_path")
        _PATH_CACHE[key] = p_
    return p_


# --------------------------------------------------------------------------------------------
# raw calls
# --------------------------------------------------------------------------------------------
def gemm(M: int, N: int, K: int, A, lda, a_kmajor, B, ldb, b_kmajor, out, ldc, *, bias=None, rowbias=None,
         rowbias_period=0, ld_rowbias=0, rowscale=None, act=L.ACT_NONE, aux_in=None, aux_out=None, ld_aux=0,
         residual=None, ldr=0, A2=None, lda2=0, k_split=0, split_k=0, colsum=None, pieces: Optional[int] = None,
         a_amax=None, b_amax=None, b_is_weight: Optional[bool] = None, a2_amax=None, raw_slabs: bool = False,
         want_c_amax: bool = False):
    """`a_amax` / `b_amax`: magnitude words of the operands where the caller has them (fp16-piece products); missing ones are looked up /
    computed here.  `b_is_weight` (default: every product but the transposed-A one): B is a parameter (or a view of one).
    The output's own word, when the kernel published it, is left in `gemm.last_c_amax` (None otherwise).
    `want_c_amax`: publish the output's word from the fp32-MFMA tiles too (their vector epilogue does it for free; the fp16-piece tiles
    always do): for outputs whose consumer is an fp16-piece product reached through a reshape.
    `raw_slabs` (with split_k > 1, `out` may be None): no reduce launch; returns (workspace, n_slabs): the product is the sum of the
    [M, N] slabs workspace[z * M * N:], z < n_slabs, for a consumer that adds them itself (gaot_rmsnorm_bwd_slabs)."""
    _dev(A, B, out, bias, rowbias, rowscale, aux_in, aux_out, residual, A2, colsum)
    _f32(A, B) if out is None else _f32(A, B, out)
    split_k = max(1, split_k)
    if raw_slabs and split_k <= 1:
        raise ValueError("gemm: raw_slabs needs split_k > 1")
    ws = None
    if split_k > 1:
        ws = torch.empty(split_k * (M * N + M), device=A.device, dtype=torch.float32)
    kind = "tn" if not a_kmajor else ("nt" if b_kmajor else "nn")
    pc = pieces if pieces is not None else _PIECES[kind]
    d = L.GemmDesc(M, N, K, _p(A), lda, int(a_kmajor), _p(A2), lda2, k_split, _p(B), ldb, int(b_kmajor),
                   _p(out), ldc, _p(bias), _p(rowbias), rowbias_period, ld_rowbias, _p(rowscale), act,
                   _p(aux_in), _p(aux_out), ld_aux, _p(residual), ldr, split_k, _p(ws), _p(colsum),
                   pc, None, None, None, 0, 0, None, None, int(raw_slabs))
    gemm.last_c_amax = None
    if pc == 3 and _F16_PIECES[0]:
        # whether B comes with pre-split planes is part of the dispatcher's answer (the 64 x 64 all-DMA tiles of narrow outputs need them):
        # looked up BEFORE the dry run, part of the cache key
        is_w = b_is_weight if b_is_weight is not None else kind != "tn"
        wop = weight_operand(B, bool(b_kmajor)) if is_w else None
        has_planes = wop is not None and wop[1] is not None and (b_amax is None or b_amax is wop[0])
        key = (M, N, K, int(a_kmajor), int(b_kmajor), act, split_k, colsum is None, bias is None, rowbias is None, rowscale is None,
               aux_in is None, aux_out is None, residual is None, lda % 4, ldb % 4, ldc % 4, ld_aux % 4, ldr % 4,
               (A.data_ptr() | B.data_ptr() | (0 if out is None else out.data_ptr())) & 15, _GEMM_MODE,
               None if A2 is None else (k_split, lda2 % 4, A2.data_ptr() & 15), raw_slabs, has_planes)
        d.pieces, d.a_absmax, d.b_absmax = 4, A.data_ptr(), B.data_ptr()      # (placeholders: the dry run reads no memory)
        d.a2_absmax = None if A2 is None else A2.data_ptr()
        if has_planes:
            d.b_planes, d.ld_bplanes, d.b_plane_stride = wop[1], wop[2], wop[3]
        path = _gemm_path(d, key)
        if _GEMM_TRACE and key not in _GEMM_TRACE_SEEN:          # tools: every distinct product once, with the tile family the dispatcher answers
            _GEMM_TRACE_SEEN.add(key)
            print(f"[gemm] {kind} M={M} N={N} K={K} split_k={split_k} raw={int(raw_slabs)} act={act} planes={int(has_planes)} A2={int(A2 is not None)} -> "
                  f"{ {1: 'fp32-MFMA tiles', 2: 'skinny', 3: 'fp16-piece tiles'}.get(path, path) }", flush=True)
        d.pieces, d.a_absmax, d.b_absmax, d.a2_absmax = 3, None, None, None
        d.b_planes, d.ld_bplanes, d.b_plane_stride = None, 0, 0
        if path == 3:
            if a_amax is None:
                a_amax = amax_for(A)
            if A2 is not None:
                d.a2_absmax = (a2_amax if a2_amax is not None else amax_for(A2)).data_ptr()
            if is_w:
                wword, pl, pl_ld, pl_stride = wop
                if b_amax is None:
                    b_amax = wword
                if pl is not None and b_amax is wword:
                    d.b_planes, d.ld_bplanes, d.b_plane_stride = pl, pl_ld, pl_stride
            elif b_amax is None:
                b_amax = amax_for(B)
            d.pieces, d.a_absmax, d.b_absmax = 4, a_amax.data_ptr(), b_amax.data_ptr()
            if colsum is None and _PUBLISH_C and not raw_slabs:          # (split-K products: the reduce launch publishes it)
                cw = _amax_words(1, out.device)[0]
                d.c_absmax = cw.data_ptr()
                gemm.last_c_amax = cw
        elif want_c_amax and path == 1 and colsum is None and _PUBLISH_C and not raw_slabs and split_k <= 1 and 3 in _PIECES.values():
            cw = _amax_words(1, out.device)[0]
            d.c_absmax = cw.data_ptr()
            gemm.last_c_amax = cw
    L.check(L.load().gaot_gemm_f32(C.byref(d), _stream()), "gaot_gemm_f32")
    if raw_slabs:
        return ws, int(L.load().gaot_gemm_slab_count(K, split_k))
    return out


gemm.last_c_amax = None


# tuning switch (tools / A-B runs only): GAOT_GEMM_MODE = argument of gaot_debug_set_gemm_glds (default 4: fp32 MFMA tiles
# + the split-bf16 tiles where the heuristic picks them; 1 = fp32 MFMA tiles only); the split-K choice below follows it
_FUSED_KERNEL_MLP = os.environ.get("GAOT_FUSED_KERNEL_MLP", "1") != "0"     # A/B switch for tools; the chain path is also HIP
_GEMM_MODE = int(os.environ.get("GAOT_GEMM_MODE", "4"))
# integral-transform kernel choice: 2 (default) = edge-partitioned kernels on degree-skewed plans + batch-inside decoder forward,
# 1 = edge-partitioned everywhere, 0 = the row-parallel kernels of round 1 (A/B switch for tools and tests)
_GNO_EP = int(os.environ.get("GAOT_GNO_EP", "2"))
_NARROW_SPLIT = int(os.environ.get("GAOT_NARROW_SPLIT", "1"))        # A/B switch: split-K for narrow-output activation products
_NARROW_TILES_1K = int(os.environ.get("GAOT_NARROW_TILES_1K", "1"))  # A/B switch: no K slabs where the 64 x 64 all-DMA tiles take a narrow product whole
_NARROW_SPLIT_1K = int(os.environ.get("GAOT_NARROW_SPLIT_1K", "1"))  # A/B switch: K slabs of a narrow-output product with K = 1 024 (FFN down-projection)


def set_gno_ep(mode: int) -> int:
    global _GNO_EP
    old, _GNO_EP = _GNO_EP, int(mode)
    return old
if _GEMM_MODE != 4:
    L.load().gaot_debug_set_gemm_glds(_GEMM_MODE)
if os.environ.get("GAOT_ATTN_SPLIT", "1") != "1":          # A/B switch: argument of gaot_debug_set_attention_split (0 = fp32 MFMA)
    L.load().gaot_debug_set_attention_split(int(os.environ["GAOT_ATTN_SPLIT"]))


def set_gemm_mode(mode: int) -> int:
    """runtime form of GAOT_GEMM_MODE (tests / A-B runs): returns the previous mode"""
    global _GEMM_MODE
    old, _GEMM_MODE = _GEMM_MODE, int(mode)
    L.load().gaot_debug_set_gemm_glds(int(mode))
    return old


def set_attention_split(mode: int) -> int:
    """runtime form of GAOT_ATTN_SPLIT: 1 = split-bf16 attention kernels for head_dim 32 (default), 0 = fp32-MFMA kernels"""
    return int(L.load().gaot_debug_set_attention_split(int(mode)))


def _split_for_reduction(Mo: int, No: int, K: int) -> int:
    """split-K factor for weight-gradient products (tiny output, long reduction)."""
    t128 = ((Mo + 127) // 128) * ((No + 127) // 128)
    if _GEMM_MODE >= 4 and K % 32 == 0 and Mo % 4 == 0 and No % 4 == 0 and min(Mo, No) >= 128 and t128 >= 8:
        # split-bf16 128x128 tiles (gemm_split.hip): ~512 workgroups, at least 256 reduction rows per slab
        # ... and at most 1 024 per slab (the split tiles' accumulation cap, gemm.hip)
        return int(max(1, min(512 // t128, K // 256), -(-K // 1024)))
    tiles = ((Mo + 63) // 64) * ((No + 63) // 64)
    want = max(1, 1024 // max(1, tiles))          # ~1024 workgroups of 64x64 (tools/gemm_bench.py sweep)
    if Mo * No <= 4096 and min(Mo, No) <= 16:     # skinny path: HBM-latency bound, wants many short row chunks
        return int(max(1, min(1024, (K + 127) // 128)))
    return int(max(1, min(want, 256, (K + 255) // 256)))


def linear_nt(x2: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None, **epi) -> torch.Tensor:
    """out[M,N] = x2[M,K] @ w[N,K]^T (+ epilogue)."""
    x2, lda = _rowmajor(x2)
    w, ldb = _rowmajor(w)
    M, K = x2.shape
    N = w.shape[0]
    assert w.shape[1] == K, (x2.shape, w.shape)
    if out is None:
        out = torch.empty(M, N, device=x2.device, dtype=torch.float32)
    if "split_k" not in epi and epi.get("act", L.ACT_NONE) == L.ACT_NONE and epi.get("aux_out") is None and epi.get("aux_in") is None:
        epi["split_k"] = _split_for_narrow_output(M, N, K, lambda: weight_operand(w, True)[1] is not None)
    return gemm(M, N, K, x2, lda, 1, w, ldb, 1, out, out.stride(0) if M > 1 else N, **epi)


def _split_for_narrow_output(Mo: int, No: int, K: int, planes=None) -> int:
    """activation-side products whose output is only two 128-wide tiles across (N = 256) with a long reduction (K >= 2048,
    du @ [w1;w3]): 128 output tiles leave half the CUs idle; two K halves on the split-bf16 tiles + one reduce measured
    78 -> 61 us at 8192 x 256 x 2048 (tools/gemm_n256_sweep.py)"""
    if _GEMM_MODE < 4 or _NARROW_SPLIT == 0 or K < 1024 or K % 64 or Mo % 4 or No % 4:
        return 1
    t128 = ((Mo + 127) // 128) * ((No + 127) // 128)
    if 100 <= t128 < 200:
        # fp16 / two-piece tiles (tools/gemm_splitk_sweep.py, product + reduce): K = 2048: 76 / 51 / 47 us at 1 / 2 / 4 slabs on the
        # 128-row tiles (64-row tiles without slabs: 66); K = 1024: 34 (64-row, no slabs) / 31.5 (two slabs)
        return 4 if K >= 2048 else _NARROW_SPLIT_1K
    if _NARROW_TILES_1K and 48 <= t128 < 100 and 128 < No <= 256 and K <= 1024 and Mo % 64 == 0 and (planes is None or planes()):
        # two tiles across, half a round of 64-row tiles (4 096 tokens x 256, the FFN down-projection at K = 1 024): the 64 x 64 all-DMA
        # tiles (gemm.hip, 256 workgroups) take the whole reduction in one launch; two slabs on the fp32-MFMA tiles + the reduce were
        # 24.8 + 5.3 us per launch (profiles/r5z_c4_kernel_stats_before.txt, r5z_c4_step_sequence.txt).  `planes`: whether the B operand has
        # pre-split weight planes -- those tiles need them (gemm.hip `ad_narrow`); without (GAOT_WEIGHT_PLANES=0, an unregistered or misaligned
        # weight view, B an activation) the product keeps its K slabs below instead of running unsplit on ~128 fp32-MFMA workgroups
        return 1
    if 48 <= t128 < 100 and Mo >= 128 and No >= 128:
        # fewer than 100 output tiles (4 096 tokens x 384 at the 3-D configuration): on the fp32-MFMA tiles 130 us at K = 3 072;
        # enough K slabs for ~256 workgroups on the split-bf16 tiles, each at most 1 024 deep (the accumulation cap of gemm.hip)
        s = min(-(-256 // t128), K // 512)
        while s > 0 and K / s > 1024 and s < K // 256:
            s += 1
        s = max(1, s)
        # [r6] slabs that would not fill the split tiles either (s x tiles < 250: they ran on the fp32-MFMA tiles -- 4 096 x 384 x 1 152 in two
        # slabs of 39 us + a reduce) with a reduction of at most 2 048: unsplit on the 64 x 64 all-DMA tiles, whose kernel flushes its
        # accumulators every 1 024 values of k (gemm.hip `long_k`; needs the weight planes like every product on those tiles)
        if s > 1 and s * t128 < 250 and 1024 < K <= 2048 and K % 32 == 0 and Mo % 64 == 0 and 128 < No <= 384 and planes is not None and planes() \
                and int(L.load().gaot_debug_set_gemm_ad_flush(-1)) == 1:          # (-1: query)
            return 1
        return s
    if t128 >= 200 and K > 1024 and Mo >= 128 and No >= 128:
        # plenty of tiles, long reduction (C3: 16 384 x 256 x 2 048): the split-bf16 tiles accumulate at most 1 024 values of k per
        # workgroup (gemm.hip), so the reduction arrives in slabs of that depth rather than falling back to the fp32-MFMA tiles
        # (165 vs 123 us per launch at that shape)
        return -(-K // 1024)
    return 1


def matmul_nn(g: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None, **epi) -> torch.Tensor:
    """out[M,K] = g[M,N] @ w[N,K]   (input gradient of a Linear with weight w)."""
    g, lda = _rowmajor(g)
    w, ldb = _rowmajor(w)
    M, N = g.shape
    K = w.shape[1]
    assert w.shape[0] == N
    if out is None:
        out = torch.empty(M, K, device=g.device, dtype=torch.float32)
    if "split_k" not in epi:
        epi["split_k"] = _split_for_narrow_output(M, K, N, lambda: weight_operand(w, False)[1] is not None)
    return gemm(M, K, N, g, lda, 1, w, ldb, 0, out, out.stride(0) if M > 1 else K, **epi)


def matmul_tn(g: torch.Tensor, x2: torch.Tensor, out: Optional[torch.Tensor] = None,
              colsum_out: Optional[torch.Tensor] = None, final: bool = False, g_amax=None, x_amax=None) -> torch.Tensor:
    """out[N,K] = g[M,N]^T @ x2[M,K]   (weight gradient); split-K over the long M reduction.
    colsum_out[N] (optional) receives sum_m g[m,:] -- the bias gradient -- from the same pass over g.
    `final`: `out` (and `colsum_out`) are a PARAMETER's registered gradient slices (ops._claim): nothing reads them before the
    backward pass has ended, so inside a deferred_wgrad() scope the product may join the grouped launch at its end."""
    g, lda = _rowmajor(g)
    x2, ldb = _rowmajor(x2)
    M, N = g.shape
    K = x2.shape[1]
    assert x2.shape[0] == M
    if (final and out is not None and _WGRAD_DEPTH[0] > 0 and not _slot_shared(out) and not _slot_shared(colsum_out)
            and _wgrad_groupable(g, lda, x2, ldb, out, colsum_out, N, K, M)):
        # a caller-provided destination (the parameter's slice of the flat gradient buffer) inside a deferral scope: the product
        # joins the grouped launch at the end of the backward pass (flush_wgrad); operands stay alive in the queue until then
        # the queue keeps ALIASES, not the tensor objects handed back to autograd (an extra reference to those makes AccumulateGrad clone)
        if _PIECES["tn"] == 3 and _F16_PIECES[0]:      # the operands' magnitude words, now (the tensors' producers may have published them)
            g_amax = g_amax if g_amax is not None else amax_for(g)
            x_amax = x_amax if x_amax is not None else amax_for(x2)
        _WGRAD_QUEUE.append((g, lda, x2, ldb, out.detach(), out.stride(0), None if colsum_out is None else colsum_out.detach(), N, K, M, g_amax, x_amax))
        _note_deferred(out)
        if colsum_out is not None:
            _note_deferred(colsum_out)
        return out
    if out is None:
        out = torch.empty(N, K, device=g.device, dtype=torch.float32)
    ldc = out.stride(0) if N > 1 else K
    return gemm(N, K, M, g, lda, 0, x2, ldb, 0, out, ldc, split_k=_split_for_reduction(N, K, M), colsum=colsum_out, a_amax=g_amax, b_amax=x_amax)


# --------------------------------------------------------------------------------------------
# Deferred, grouped weight gradients.  Autograd reaches the Linear layers of a backward pass one by one; each weight gradient is
# a long reduction over all tokens into a small matrix (2048 x 256 ... 256 x 256 at the example model, K = 8 192), which on its
# own fills the chip only through 16-32 split-K slabs and a separate reduce launch.  Inside a `deferred_wgrad()` scope
# (trainer.TrainStep, autograph: callers that own the gradient buffers, so nobody reads a gradient before the pass has ended)
# matmul_tn only queues the product; flush_wgrad() runs ALL queued products as one launch (gaot_gemm_tn_grouped).
# --------------------------------------------------------------------------------------------
# Zero-initialised scratch that single-launch reductions leave zero again (last-workgroup tickets, the MSE partials): one set per
# OWNER.  Launches of one owner are ordered on one stream; two owners in flight at once (an autograph replay overlapping a TrainStep, two
# steps on two streams) must not share a ticket, so TrainStep / autograph entries run their launches inside
# `with ops.scratch_owner(their dict)`; everything else shares the per-device default set (one stream at a time per device).
_SCRATCH_DEFAULT: dict = {}
_SCRATCH_STACK: list = []


class scratch_owner:
    def __init__(self, store: dict):
        self.store = store

    def __enter__(self):
        _SCRATCH_STACK.append(self.store)
        return self

    def __exit__(self, et, ev, tb):
        _SCRATCH_STACK.pop()
        return False


def _scratch(kind: str, device, make, what: str):
    """the current owner's scratch of `kind` on `device`, created (and zero-filled) on first use -- which must not happen inside a graph
    capture: the fill would be captured, not executed"""
    store = _SCRATCH_STACK[-1] if _SCRATCH_STACK else _SCRATCH_DEFAULT
    key = (kind, device.index if device.type == "cuda" else -1)
    v = store.get(key)
    if v is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError(f"{what}: run once outside graph capture first (its scratch is zeroed when it is allocated)")
        v = store[key] = make()
    return v


_WGRAD_DEPTH = [0]
_WGRAD_QUEUE: list = []
_COLSUM_QUEUE: list = []
_WGRAD_COUNTERS_RETIRED: list = []
_WGRAD_GROUPED = os.environ.get("GAOT_WGRAD_GROUPED", "1") != "0"        # A/B switch


def _wgrad_groupable(g, lda, x2, ldb, out, colsum_out, Mo, No, K) -> bool:
    if not _WGRAD_GROUPED or _GEMM_MODE < 4:
        return False
    if Mo % 4 or No % 4 or K % 32 or K < 1024 or min(Mo, No) < 32 or lda % 4 or ldb % 4 or out.dim() != 2 or out.stride(1) != 1 or out.stride(0) % 4:
        return False
    if (g.data_ptr() | x2.data_ptr() | out.data_ptr()) & 15:
        return False
    return g.dtype == x2.dtype == out.dtype == torch.float32 and g.is_cuda


class deferred_wgrad:
    """`with ops.deferred_wgrad(): loss.backward()`: weight gradients with a registered destination are computed by ONE grouped
    launch when the outermost scope ends.  The gradient buffers must not be read inside the scope."""

    def __enter__(self):
        _WGRAD_DEPTH[0] += 1
        return self

    def __exit__(self, et, ev, tb):
        _WGRAD_DEPTH[0] -= 1
        if _WGRAD_DEPTH[0] == 0:
            if et is None:
                flush_wgrad()
            else:
                _WGRAD_QUEUE.clear()
                _COLSUM_QUEUE.clear()
        return False


def wgrad_launch(items) -> None:
    """items: (g [K,M], ldg, x [K,N], ldx, out [M,N], ldo, colsum or None, M, N, K[, g_amax, x_amax]).  One gaot_gemm_tn_grouped call."""
    lib = L.load()
    n = len(items)
    arr = (L.WgradItem * n)()
    f16 = _PIECES["tn"] == 3 and _F16_PIECES[0]
    for i, it in enumerate(items):
        g, ldg, x2, ldx, out, ldo, cs, Mo, No, K = it[:10]
        ga, xa = (it[10], it[11]) if len(it) > 10 else (None, None)
        if f16:
            ga = ga if ga is not None else amax_for(g)
            xa = xa if xa is not None else amax_for(x2)
        arr[i] = L.WgradItem(g.data_ptr(), ldg, x2.data_ptr(), ldx, out.data_ptr(), ldo, None if cs is None else cs.data_ptr(), Mo, No, K,
                             None if ga is None else ga.data_ptr(), None if xa is None else xa.data_ptr())
    cnt = C.c_int32(0)
    need = int(lib.gaot_gemm_tn_grouped_workspace(arr, n, C.byref(cnt)))
    if need < 0:
        L.check(-1, "gaot_gemm_tn_grouped_workspace")
    dev = items[0][0].device
    ws = torch.empty(max(need, 4), device=dev, dtype=torch.float32)
    # ticket counters: one zero-initialised buffer per scratch owner and device (every launch leaves it zero again).  Never created
    # inside a graph capture: its zero fill would be captured, not executed.  A buffer that has become too small is retired, not freed
    # (a captured graph may hold its address).
    store = _SCRATCH_STACK[-1] if _SCRATCH_STACK else _SCRATCH_DEFAULT
    key = ("wgrad_counters", dev.index)
    ctr = store.get(key)
    if ctr is None or ctr.numel() < cnt.value:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("grouped weight gradients: the ticket counters must exist before graph capture (run one eager step first)")
        if ctr is not None:
            _WGRAD_COUNTERS_RETIRED.append(ctr)
        ctr = store[key] = torch.zeros(max(65536, 4 * cnt.value), device=dev, dtype=torch.int32)
    L.check(lib.gaot_gemm_tn_grouped(arr, n, 4 if f16 else _PIECES["tn"], _p(ws), _p(ctr), _stream()), "gaot_gemm_tn_grouped")


def flush_wgrad() -> None:
    if _COLSUM_QUEUE:
        cs = list(_COLSUM_QUEUE)
        _COLSUM_QUEUE.clear()
        arr = (L.ColsumItem * len(cs))()
        for i, (x2, ld, out, M, N, oc, old_) in enumerate(cs):
            arr[i] = L.ColsumItem(x2.data_ptr(), ld, out.data_ptr(), M, N, oc, old_)
        L.check(L.load().gaot_colsum_grouped(arr, len(cs), _stream()), "gaot_colsum_grouped")
    if not _WGRAD_QUEUE:
        return
    items = list(_WGRAD_QUEUE)
    _WGRAD_QUEUE.clear()
    # a grouped launch lasts as long as one workgroup's K loop: a queue of one or two small products (the encoder's stage group of a
    # staged backward: ONE 64 x 64 product over 32 768 rows, 113 us as a grouped launch) runs faster as plain split-K products
    if sum(((it[7] + 127) // 128) * ((it[8] + 127) // 128) for it in items) <= 4:
        for g, ldg, x2, ldx, out, ldo, cs, Mo, No, K, ga, xa in items:
            gemm(Mo, No, K, g, ldg, 0, x2, ldx, 0, out, ldo, split_k=_split_for_reduction(Mo, No, K), colsum=cs, a_amax=ga, b_amax=xa)
        return
    wgrad_launch(items)


def colsum(x2: torch.Tensor, out: Optional[torch.Tensor] = None, final: bool = False) -> torch.Tensor:
    """column sums of [M, N].  `final` (with `out` = a parameter's registered gradient slice, see matmul_tn): inside a
    deferred_wgrad() scope the sum joins ONE grouped launch at the end of the backward pass (gaot_colsum_grouped)."""
    x2, ld = _rowmajor(x2)
    M, N = x2.shape
    lib = L.load()
    if (final and out is not None and _WGRAD_DEPTH[0] > 0 and _WGRAD_GROUPED and N % 4 == 0 and ld % 4 == 0 and M <= 8192
            and not ((x2.data_ptr() | out.data_ptr()) & 15) and not _slot_shared(out)):
        if out.is_contiguous():
            _COLSUM_QUEUE.append((x2, ld, out.detach(), M, N, 0, 0))
            _note_deferred(out.view(-1))
            return out
        if out.dim() == 2 and out.stride(1) == 1 and out.numel() == N and out.shape[1] % 4 == 0 and out.stride(0) % 4 == 0:
            # a column block of a wider gradient matrix (a slice handed out by split_cols)
            _COLSUM_QUEUE.append((x2, ld, out.detach(), M, N, out.shape[1], out.stride(0)))
            _note_deferred(out)
            return out
    strided = None
    if out is None:
        out = torch.empty(N, device=x2.device, dtype=torch.float32)
    elif not out.is_contiguous():
        # a column block of a wider gradient matrix outside the grouped launch (a parameter used twice in the pass -- the scales of a
        # multiscale MAGNO share their weights -- never defers): gaot_colsum writes N contiguous floats, so sum into a temporary first
        if out.numel() != N:
            raise ValueError("colsum: a strided destination must hold exactly N elements")
        strided, out = out, torch.empty(N, device=x2.device, dtype=torch.float32)
    scratch = torch.empty(int(lib.gaot_colsum_scratch(M, N)), device=x2.device, dtype=torch.float32)
    L.check(lib.gaot_colsum(_p(x2), ld, M, N, _p(out), _p(scratch), _stream()), "gaot_colsum")
    if strided is not None:
        strided.copy_(out.view(strided.shape))
        return strided
    return out


def batchsum(x: torch.Tensor, B: int) -> torch.Tensor:
    """x [B, ...] contiguous -> sum over the leading dim."""
    x = x.contiguous()
    rn = x.numel() // B
    out = torch.empty(x.shape[1:] if x.shape[0] == B else (rn,), device=x.device, dtype=torch.float32)
    L.check(L.load().gaot_batchsum(_p(x), B, rn, _p(out), _stream()), "gaot_batchsum")
    return out


# --------------------------------------------------------------------------------------------
# Linear (nn.Linear / Conv1d k=1) with fused bias, periodic row bias, residual and split input
# --------------------------------------------------------------------------------------------
# --------------------------------------------------------------------------------------------
# Gradient slots: trainer.FlatGradBucket registers, per parameter, the view of the flat gradient buffer where that
# parameter's gradient has to end up.  The FIRST use of a parameter in a forward pass claims its slot and the matching
# backward writes the weight gradient straight into it (no per-parameter temporary + pack copy); any further use of the
# same parameter before the next FlatGradBucket.clear() takes the ordinary path and autograd accumulates as usual.
# --------------------------------------------------------------------------------------------
# The slot lives ON its owner -- an attribute of the parameter (or of the forward-pass temporary that inherited a column block of it:
# split_cols) -- not in a table keyed by id(): rounds 3-5 kept `{id(tensor): slot}` and paid for it with a silent path change when CPython
# handed a freed temporary's id to a new model's Parameter.  An attribute cannot outlive or be confused with another object;
# temporaries take theirs with them when the autograd graph dies.  `_SLOT_OWNERS` only lists the registered parameters (to reset and to
# clear them).
_SLOT_ATTR = "_gaot_grad_slot"          # [view of the flat gradient buffer, claimed in this forward pass]
_SLOT_OWNERS: list = []


def register_grad_slots(params, views):
    for q in _SLOT_OWNERS:
        if hasattr(q, _SLOT_ATTR):
            delattr(q, _SLOT_ATTR)
    _SLOT_OWNERS[:] = list(params)
    for p, v in zip(params, views):
        setattr(p, _SLOT_ATTR, [v, False])


def release_grad_slots():
    for q in _SLOT_OWNERS:
        s = getattr(q, _SLOT_ATTR, None)
        if s is not None:
            s[1] = False
    _DEFERRED_DESTS.clear()
    _SHARED_SLOTS.clear()


def grad_slot_of(t) -> Optional[torch.Tensor]:
    """the registered gradient slice of a tensor (None: not registered)"""
    s = getattr(t, _SLOT_ATTR, None)
    return None if s is None else s[0]


def save_grad_slots():
    """the current registration, for a scope that registers its own (autograph's capture): restore_grad_slots puts it back"""
    return [(q, getattr(q, _SLOT_ATTR, None)) for q in _SLOT_OWNERS]


def restore_grad_slots(saved) -> None:
    register_grad_slots([], [])
    _SLOT_OWNERS[:] = [q for q, _ in saved]
    for q, s in saved:
        if s is not None:
            setattr(q, _SLOT_ATTR, s)


# Slots of parameters used MORE THAN ONCE in a forward pass (the reference shares agno / geoembed / lifting / recovery / projection
# weights across all `scales`, magno.py:277-300): the first use holds the slot and writes its contribution in place, every further
# use returns an ordinary gradient tensor and autograd sums them -- so the slot is READ (by that sum) before the backward pass has
# ended and must never be the destination of a deferred launch.  _claim() records the slice of a parameter claimed twice; matmul_tn /
# colsum drop `final` for destinations inside such a slice (decided at backward time: every forward use precedes the backward pass,
# and the decision is the same at capture and at replay).
_SHARED_SLOTS: list = []


def _mark_shared(slot: torch.Tensor) -> None:
    lo = slot.data_ptr()
    _SHARED_SLOTS.append((lo, lo + slot.numel() * 4))


def _slot_shared(t: Optional[torch.Tensor]) -> bool:
    if t is None or not _SHARED_SLOTS:
        return False
    p_ = t.data_ptr()
    return any(lo <= p_ < hi for lo, hi in _SHARED_SLOTS)


# destinations (address ranges inside the flat gradient buffer) that a deferred, grouped launch writes at the END of the backward
# pass.  AccumulateGrad may have taken a COPY of the tensor a backward node returned for such a slice (it clones instead of adopting
# whenever it sees another reference) -- a copy of memory the product had not reached yet.  Whoever gathers gradients into the flat
# buffer afterwards (FlatGradBucket.pack, autograph's settle) must therefore not copy such a tensor over its slice.
_DEFERRED_DESTS: list = []


def deferred_dest(ptr: int, nbytes: int = 4) -> bool:
    """does [ptr, ptr + nbytes) -- a parameter's whole gradient slice -- overlap a destination of a deferred launch?  (A slice whose
    column blocks were handed out by split_cols may be deferred in part only: the slice, not its base address, is what counts.)"""
    return any(lo < ptr + nbytes and ptr < hi for lo, hi in _DEFERRED_DESTS)


def _note_deferred(t: torch.Tensor) -> None:
    lo = t.data_ptr()
    _DEFERRED_DESTS.append((lo, lo + ((t.shape[0] - 1) * t.stride(0) + t.shape[-1]) * 4 if t.dim() == 2 else lo + t.numel() * 4))


def _claim(w) -> Optional[torch.Tensor]:
    if w is None or not w.requires_grad:
        return None
    s = getattr(w, _SLOT_ATTR, None)
    if s is None or s[0].device != w.device:
        return None
    if s[1]:
        _mark_shared(s[0])       # second use in this forward pass: the holder of the slot must not defer (see _SHARED_SLOTS)
        return None
    s[1] = True
    return s[0]


def _claim_view(t) -> Optional[torch.Tensor]:
    """the gradient slice of the PARAMETER that `t` is a reshaped view of (a Conv1d weight with its trailing singleton dimension
    squeezed): same storage address and element count"""
    if t is None or not t.requires_grad:
        return None
    if getattr(t, _SLOT_ATTR, None) is not None:
        return _claim(t)
    ptr, n = t.data_ptr(), t.numel()
    for p in _SLOT_OWNERS:
        s = getattr(p, _SLOT_ATTR, None)
        if s is not None and p.data_ptr() == ptr and p.numel() == n and s[0].device == t.device and t.is_contiguous():
            if s[1]:
                _mark_shared(s[0])
                return None
            s[1] = True
            return s[0]
    return None


class _Linear(torch.autograd.Function):
    """y = x @ w[:, :K]^T (+ x2 @ w[:, K:]^T) + b + rowbias[m % P] + residual"""

    @staticmethod
    def forward(ctx, x, w, b, residual, rowbias, x2, publish=False):
        _dev(x, w)
        shp = x.shape
        K = shp[-1]
        xm = x.reshape(-1, K)
        M = xm.shape[0]
        N = w.shape[0]
        w2d = w.reshape(N, -1)            # Conv1d weights carry a trailing singleton dim
        res2 = residual.reshape(M, N) if residual is not None else None
        epi = dict(bias=b)
        if publish:
            epi["want_c_amax"] = True
        if res2 is not None:
            res2, ldr = _rowmajor(res2)
            epi.update(residual=res2, ldr=ldr)
        if rowbias is not None:
            rb, ldrb = _rowmajor(rowbias.reshape(-1, N))
            epi.update(rowbias=rb, rowbias_period=rb.shape[0], ld_rowbias=ldrb)
        cw = None
        if x2 is None:
            assert w2d.shape[1] == K
            y = linear_nt(xm, w2d, a_amax=_amax_get(x, xm), **epi)
            cw = gemm.last_c_amax
            x2m = None
        else:
            K2 = x2.shape[-1]
            x2m = x2.reshape(-1, K2)
            assert w2d.shape[1] == K + K2
            xm_, lda = _rowmajor(xm)
            x2m_, lda2 = _rowmajor(x2m)
            wc, ldb = _rowmajor(w2d)
            y = torch.empty(M, N, device=x.device, dtype=torch.float32)
            if K % 32 == 0:
                gemm(M, N, K + K2, xm_, lda, 1, wc, ldb, 1, y, N, A2=x2m_, lda2=lda2, k_split=K, a_amax=_amax_get(xm_, xm, x),
                     a2_amax=_amax_get(x2m_, x2m, x2), **epi)
                cw = gemm.last_c_amax
            else:
                linear_nt(xm_, wc[:, :K], out=y, **epi)
                linear_nt(x2m_, wc[:, K:], out=y, residual=y, ldr=N)
        ctx.save_for_backward(xm, x2m, w2d)
        ctx.amax_x = _amax_get(xm, x)          # the input's magnitude word, for the weight-gradient product (saved_tensors are new objects)
        ctx.amax_x2 = _amax_get(x2m, x2) if x2 is not None else None
        _publish(ctx.amax_x, x)
        _publish(ctx.amax_x2, x2)
        ctx.slots = (_claim(w), _claim(b))
        ctx.meta = (shp, x2.shape if x2 is not None else None, w.shape, b is not None,
                    residual.shape if residual is not None else None,
                    rowbias.shape if rowbias is not None else None)
        yr = y.reshape(*shp[:-1], N)
        _publish(cw, y, yr)
        return yr

    @staticmethod
    def backward(ctx, dy):
        xm, x2m, w2d = ctx.saved_tensors
        shp, shp2, wshape, has_b, res_shape, rb_shape = ctx.meta
        N = w2d.shape[0]
        g = dy.reshape(-1, N)
        g, _ = _rowmajor(g)
        K = xm.shape[1]
        need = ctx.needs_input_grad
        dx = dw = db = dres = drb = dx2 = None
        # dY's magnitude word: its producer's, or the one the first product that needs it computes (it stays on `g` for the others)
        if need[0]:
            dx2d = matmul_nn(g, w2d[:, :K], a_amax=_amax_get(g, dy))
            dx = dx2d.reshape(shp)
            _publish(gemm.last_c_amax, dx2d, dx)
        if need[5] and x2m is not None:
            dx2d_ = matmul_nn(g, w2d[:, K:], a_amax=_amax_get(g, dy))
            dx2 = dx2d_.reshape(shp2)
            _publish(gemm.last_c_amax, dx2d_, dx2)
        want_db = has_b and need[2]
        wslot, bslot = ctx.slots
        if need[1]:
            dw = wslot.detach().view(N, w2d.shape[1]) if wslot is not None else torch.empty(N, w2d.shape[1], device=g.device, dtype=torch.float32)
            if want_db:      # bias gradient rides the weight-gradient product (same pass over dY)
                db = bslot.detach() if bslot is not None else torch.empty(N, device=g.device, dtype=torch.float32)
            # final: both destinations are the parameters' own gradient slices (a bias gradient without its slice is read by
            # autograd right away, so it keeps the product immediate)
            fin = wslot is not None and (db is None or bslot is not None)
            matmul_tn(g, xm, out=dw[:, :K], colsum_out=db, final=fin, g_amax=_amax_get(g, dy), x_amax=ctx.amax_x)
            if x2m is not None:
                matmul_tn(g, x2m, out=dw[:, K:], final=wslot is not None, g_amax=_amax_get(g, dy), x_amax=ctx.amax_x2)
            dw = dw.reshape(wshape)
        elif want_db:
            db = colsum(g)
        if res_shape is not None and need[3]:
            dres = dy.reshape(res_shape)
        if rb_shape is not None and need[4]:
            P = 1
            for s in rb_shape[:-1]:
                P *= s
            drb = batchsum(g.reshape(-1, P * N), g.shape[0] // P).reshape(rb_shape)
        return dx, dw, db, dres, drb, dx2, None


def linear(x, w, b=None, residual=None, rowbias=None, x2=None, publish: bool = False):
    """`publish`: the output's magnitude word is wanted whichever tile family runs the product (gemm(want_c_amax=))"""
    return _Linear.apply(x, w, b, residual, rowbias, x2, publish)


class _MatMul(torch.autograd.Function):
    """y[M,N] = a[M,K] @ b[K,N] with both operands as they lie in memory (no transposed copy of either): the folded
    projection o recovery weight of the decoder, W_proj @ W_rec[:, :C] (magno.py:640-641 after 345-350)."""

    @staticmethod
    def forward(ctx, a, b):
        _dev(a, b)
        a2, lda = _rowmajor(a)
        b2, ldb = _rowmajor(b)
        M, K = a2.shape
        N = b2.shape[1]
        assert b2.shape[0] == K, (a.shape, b.shape)
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
        gemm(M, N, K, a2, lda, 1, b2, ldb, 0, out, N)
        ctx.save_for_backward(a2, b2)
        return out

    @staticmethod
    def backward(ctx, g):
        a2, b2 = ctx.saved_tensors
        g, _ = _rowmajor(g)
        da = linear_nt(g, b2) if ctx.needs_input_grad[0] else None          # g [M,N] @ b[K,N]^T
        db = matmul_tn(a2, g) if ctx.needs_input_grad[1] else None          # a[M,K]^T @ g [M,N]
        return da, db


def matmul(a, b):
    return _MatMul.apply(a, b)


class _ProjFold(torch.autograd.Function):
    """The decoder's output projection folded into the recovery block (magno.py:345-350 then 640-641; both linear, nothing in
    between):   weff = W @ Wr_a  [OC, C]   and   rproj = rowb @ W^T + b  [Q, OC],   so that  y = agno @ weff^T + rproj.
    ONE node for both uses of the projection weight W: its two gradient contributions are summed inside the second product
    (residual operand) and land in W's own gradient slice; Wr_a's gradient goes straight into the column block of the recovery
    weight's slice that split_cols handed out."""

    @staticmethod
    def fusable(hw2, ldh, wa2, rb2, ldr) -> bool:
        OC, Cc = hw2.shape
        return (_FUSED_PROJ_FOLD and 1 <= OC <= 4 and Cc % 4 == 0 and Cc <= 256 and wa2.shape[1] <= 1024 and ldh % 4 == 0 and ldr % 4 == 0
                and not ((hw2.data_ptr() | rb2.data_ptr()) & 15))

    @staticmethod
    def forward(ctx, hw, hb, w_a, rowb):
        _dev(hw, w_a, rowb)
        hw2, ldh = _rowmajor(hw)
        wa2, lda = _rowmajor(w_a)
        rb2, ldr = _rowmajor(rowb)
        OC, Cc = hw2.shape
        Q, Cout = rb2.shape[0], wa2.shape[1]
        weff = torch.empty(OC, Cout, device=hw.device, dtype=torch.float32)
        ctx.fused = _ProjFold.fusable(hw2, ldh, wa2, rb2, ldr)
        if ctx.fused:          # one launch: the row-bias product and the tiny weight product
            rproj = torch.empty(Q, OC, device=hw.device, dtype=torch.float32)
            L.check(L.load().gaot_proj_fold_fwd(_p(hw2), ldh, _p(hb), _p(wa2), lda, _p(rb2), ldr, Q, Cc, Cout, OC, _p(weff), _p(rproj), _stream()),
                    "gaot_proj_fold_fwd")
        else:
            gemm(OC, Cout, Cc, hw2, ldh, 1, wa2, lda, 0, weff, Cout)
            rproj = linear_nt(rb2, hw2, bias=hb)
        ctx.save_for_backward(hw2, wa2, rb2)
        ctx.slots = (_claim_view(hw), _claim(hb), _claim(w_a))
        ctx.has_b = hb is not None
        return weff, rproj

    @staticmethod
    def backward(ctx, g_weff, g_rproj):
        hw2, wa2, rb2 = ctx.saved_tensors
        s_w, s_b, s_a = ctx.slots
        need = ctx.needs_input_grad
        dhw = dhb = dwa = drowb = None
        if ctx.fused and g_weff is not None and g_rproj is not None:
            # every gradient of the node from ONE pass over rowb / g_rproj (the last workgroup finishes the small matrices)
            lib = L.load()
            OC, Cc = hw2.shape
            Q, Cout = rb2.shape[0], wa2.shape[1]
            g1, g2 = g_weff.contiguous(), g_rproj.contiguous()
            dev_ = hw2.device
            if need[0]:
                dhw = s_w.detach().view(hw2.shape) if s_w is not None else torch.empty(OC, Cc, device=dev_, dtype=torch.float32)
            if ctx.has_b and need[1]:
                dhb = s_b.detach() if s_b is not None else torch.empty(OC, device=dev_, dtype=torch.float32)
            if need[2]:
                dwa = s_a.detach() if s_a is not None else torch.empty(Cc, Cout, device=dev_, dtype=torch.float32)
                if dwa.dim() != 2 or dwa.stride(1) != 1:
                    dwa = torch.empty(Cc, Cout, device=dev_, dtype=torch.float32)
            if need[3]:
                drowb = torch.empty(Q, Cc, device=dev_, dtype=torch.float32)
            tk = _scratch("ticket", dev_, lambda: torch.zeros(1, device=dev_, dtype=torch.int32), "_ProjFold.backward")
            ws = torch.empty(int(lib.gaot_proj_fold_workspace(Q, Cc, OC)), device=dev_, dtype=torch.float32)
            L.check(lib.gaot_proj_fold_bwd(_p(g1), _p(g2), _p(hw2), hw2.stride(0) if OC > 1 else Cc, _p(wa2), wa2.stride(0), _p(rb2), rb2.stride(0),
                                           Q, Cc, Cout, OC, _p(drowb), _p(dhw), (dhw.stride(0) if OC > 1 else Cc) if dhw is not None else 0, _p(dhb),
                                           _p(dwa), dwa.stride(0) if dwa is not None else 0, _p(ws), _p(tk), _stream()), "gaot_proj_fold_bwd")
            return dhw, dhb, dwa, drowb
        g1, _ = _rowmajor(g_weff)
        g2, _ = _rowmajor(g_rproj)
        if need[0]:
            part = matmul_tn(g2, rb2)                                              # g_rproj^T @ rowb   [OC, C]
            out = s_w.detach().view(hw2.shape) if s_w is not None else None
            dhw = linear_nt(g1, wa2, out=out, residual=part, ldr=part.stride(0) if part.shape[0] > 1 else part.shape[1])   # + g_weff @ Wr_a^T
        if ctx.has_b and need[1]:
            dhb = colsum(g2, out=s_b.detach() if s_b is not None else None)
        if need[2]:
            dwa = matmul_tn(hw2, g1, out=s_a.detach() if s_a is not None else None)   # W^T @ g_weff   [C, C]
        if need[3]:
            drowb = matmul_nn(g2, hw2)                                             # g_rproj @ W       [Q, C]
        return dhw, dhb, dwa, drowb


_FUSED_PROJ_FOLD = os.environ.get("GAOT_FUSED_PROJ_FOLD", "1") != "0"        # A/B switch (tools): 0 = the node as library products


def proj_fold(hw, hb, w_a, rowb):
    return _ProjFold.apply(hw, hb, w_a, rowb)


class _SplitCols(torch.autograd.Function):
    """w [N, K] -> (w[:, :c], w[:, c:]) as views; the backward writes both column blocks into ONE gradient (autograd's
    own slice nodes would zero-fill two full-size gradients, copy a block into each and add them)."""

    @staticmethod
    def forward(ctx, w, c, slot):
        ctx.c, ctx.shape, ctx.slot = c, w.shape, slot
        return w[:, :c], w[:, c:]

    @staticmethod
    def backward(ctx, g1, g2):
        N, K = ctx.shape
        s_ = ctx.slot
        if s_ is not None:
            # the views carried the parameter's own slice: a block whose consumer wrote its gradient in place (possibly by a
            # deferred launch that has not run yet -- never READ such a block here) is final; any other block is copied in
            for g_, blk in ((g1, s_[:, :ctx.c]), (g2, s_[:, ctx.c:])):
                if g_ is None:
                    blk.zero_()
                elif not (g_.data_ptr() == blk.data_ptr() and g_.stride(0) == K):
                    blk.copy_(g_)
            return s_.detach(), None, None
        ref = g1 if g1 is not None else g2
        if g1 is None:
            g1 = ref.new_zeros(N, ctx.c)
        if g2 is None:
            g2 = ref.new_zeros(N, K - ctx.c)
        return torch.cat([g1, g2], dim=1), None, None


def split_cols(w, c: int, param=None):
    """(w[:, :c], w[:, c:]).  `param`: the nn.Parameter w is a reshaped view of (a Conv1d weight with its trailing singleton
    dimension squeezed).  When that parameter's gradient slice is registered, the two views inherit its column blocks as THEIR
    gradient slices: the layers that consume the views write their weight gradients straight into the parameter's slice and the
    backward of the split hands that slice on (no concatenation, no copy into the flat buffer)."""
    owner = w if param is None else param
    slot = _claim(owner) if torch.is_grad_enabled() else None
    s2 = slot.detach().view(w.shape[0], -1) if (slot is not None and w.dim() == 2 and slot.numel() == w.numel()
                                                 and c % 4 == 0 and (w.shape[1] - c) % 4 == 0) else None
    if slot is not None and s2 is None:
        getattr(owner, _SLOT_ATTR)[1] = False         # not forwarded: give the claim back
    a, b = _SplitCols.apply(w, c, s2)
    if s2 is not None:
        for v, blk in ((a, s2[:, :c]), (b, s2[:, c:])):
            setattr(v, _SLOT_ATTR, [blk, False])          # a temporary of this forward pass: its slot dies with it
    return a, b


def adjacent_rows(ws: Sequence[torch.Tensor]) -> Optional[torch.Tensor]:
    """[sum N_i, K] view over weights that already sit back to back in one storage (the flat parameter buffer of
    trainer.FlatAdamW lays fused groups out that way), or None.  Replaces a torch.cat per step by pointer checks."""
    base = ws[0].detach()
    K = base.shape[-1]
    sp, off, rows = base.untyped_storage().data_ptr(), base.storage_offset(), 0
    for w in ws:
        if (w.dim() != 2 or w.shape[1] != K or not w.is_contiguous() or w.untyped_storage().data_ptr() != sp
                or w.storage_offset() != off + rows * K):
            return None
        rows += w.shape[0]
    return base.as_strided((rows, K), (K, 1), off)


def adopt_adjacent_storage(params: Sequence[torch.nn.Parameter]) -> bool:
    """Re-home the weights a fused GEMM reads as ONE matrix (q|k|v, w1|w3) into one storage, back to back, once: `param.data`
    becomes a view of it (names, shapes, values, state_dict and the optimizer's references are untouched; in-place updates keep the
    views).  trainer.FlatAdamW does this for its whole flat buffer; this is for plain loops (the reference trainer's own, with
    torch.optim.AdamW): without it every forward concatenates the group again -- six torch.cat launches per step at the example model.
    Returns True if the group is (now) adjacent.  Called by the owning modules at the top of their forward; a no-op pointer check
    once adopted, and again after anything that re-creates the storages (`.to()`, `.cuda()`)."""
    if adjacent_rows(params) is not None:
        return True
    K = params[0].shape[-1]
    if torch.cuda.is_current_stream_capturing() or not all(
            isinstance(p_, torch.nn.Parameter) and p_.is_cuda and p_.dtype == torch.float32 and p_.dim() == 2 and p_.shape[1] == K for p_ in params):
        return False
    with torch.no_grad():
        buf = torch.cat([p_.detach() for p_ in params], dim=0)
        r = 0
        for p_ in params:
            p_.data = buf[r:r + p_.shape[0]]
            r += p_.shape[0]
    bump_weights_generation()
    return True


def stacked_rows(ws: Sequence[torch.Tensor]) -> torch.Tensor:
    v = adjacent_rows(ws)
    return v if v is not None else torch.cat([w.detach() for w in ws], dim=0)


class _LinearCat(torch.autograd.Function):
    """y = x @ cat(ws, 0)^T for several bias-free Linear layers sharing an input (q|k|v): one GEMM forward, one
    input-gradient GEMM and ONE weight-gradient GEMM backward (the per-layer gradients are row blocks of it)."""

    @staticmethod
    def forward(ctx, x, *ws):
        _dev(x, *ws)
        shp = x.shape
        xm = x.reshape(-1, shp[-1])
        W = stacked_rows(ws)
        y = linear_nt(xm, W, a_amax=_amax_get(x, xm))
        cw = gemm.last_c_amax
        ctx.save_for_backward(xm, W)
        ctx.amax_x = _amax_get(xm, x)
        _publish(ctx.amax_x, x)
        slots = [_claim(w) for w in ws]
        ctx.slot = adjacent_rows(slots) if all(s_ is not None for s_ in slots) else None
        ctx.meta = (shp, [w.shape[0] for w in ws])
        yr = y.reshape(*shp[:-1], W.shape[0])
        _publish(cw, y, yr)
        return yr

    @staticmethod
    def backward(ctx, dy):
        xm, W = ctx.saved_tensors
        shp, rows = ctx.meta
        g, _ = _rowmajor(dy.reshape(-1, W.shape[0]))
        dx = None
        if ctx.needs_input_grad[0]:
            dx2d = matmul_nn(g, W, a_amax=_amax_get(g, dy))
            dx = dx2d.reshape(shp)
            _publish(gemm.last_c_amax, dx2d, dx)
        dws = [None] * len(rows)
        if any(ctx.needs_input_grad[1:]):
            dws = list(matmul_tn(g, xm, out=ctx.slot, final=ctx.slot is not None, g_amax=_amax_get(g, dy), x_amax=ctx.amax_x).split(rows, dim=0))
        return (dx, *dws)


def linear_cat(x, ws):
    return _LinearCat.apply(x, *ws)


# --------------------------------------------------------------------------------------------
# MLP chain: z_i = h_{i-1} W_i^T + b_i, h_i = act_i(z_i); activation derivative of layer i is fused into the
# epilogue of layer i+1's input-gradient GEMM.
# --------------------------------------------------------------------------------------------
class _MLPChain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, acts, *wb):
        _dev(x)
        n = len(wb) // 2
        shp = x.shape
        h = x.reshape(-1, shp[-1])
        saved_in, saved_aux, ws = [], [], []
        for i in range(n):
            w, b = wb[2 * i], wb[2 * i + 1]
            w2d = w.reshape(w.shape[0], -1)
            act = ACT[acts[i]]
            saved_in.append(h)
            M, N = h.shape[0], w2d.shape[0]
            y = torch.empty(M, N, device=x.device, dtype=torch.float32)
            if act == L.ACT_GELU:
                z = torch.empty_like(y)
                linear_nt(h, w2d, out=y, bias=b, act=act, aux_out=z, ld_aux=N)
                saved_aux.append(z)
            else:
                linear_nt(h, w2d, out=y, bias=b, act=act)
                saved_aux.append(y if act == L.ACT_RELU else None)
            ws.append(w2d)
            h = y
        ctx.acts = acts
        ctx.n = n
        ctx.slots = [(_claim(wb[2 * i]), _claim(wb[2 * i + 1])) for i in range(n)]
        ctx.shapes = (shp, [w.shape for w in wb[0::2]], [b is not None for b in wb[1::2]])
        ctx.save_for_backward(*saved_in, *[a if a is not None else saved_in[0].new_empty(0) for a in saved_aux], *ws)
        return h.reshape(*shp[:-1], h.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        n = ctx.n
        sv = ctx.saved_tensors
        ins, auxs, ws = sv[:n], sv[n:2 * n], sv[2 * n:]
        shp, wshapes, has_b = ctx.shapes
        g = dy.reshape(-1, ws[-1].shape[0])
        g, _ = _rowmajor(g)
        last_act = ACT[ctx.acts[-1]]
        if last_act in (L.ACT_RELU, L.ACT_GELU):      # derivative of the final activation (not fusable: no following GEMM)
            g2 = torch.empty_like(g)
            L.check(L.load().gaot_act_bwd(_p(g), _p(auxs[-1].contiguous()), g.numel(), last_act, _p(g2), _stream()), "gaot_act_bwd")
            g = g2
        grads: List[Optional[torch.Tensor]] = [None] * (2 * n)
        for i in range(n - 1, -1, -1):
            want_db = has_b[i] and ctx.needs_input_grad[3 + 2 * i]
            wslot, bslot = ctx.slots[i]
            if ctx.needs_input_grad[2 + 2 * i]:
                db = None
                if want_db:
                    db = bslot.detach() if bslot is not None else torch.empty(g.shape[1], device=g.device, dtype=torch.float32)
                out = wslot.detach().view(g.shape[1], ins[i].shape[1]) if wslot is not None else None
                grads[2 * i] = matmul_tn(g, ins[i], out=out, colsum_out=db,
                                         final=out is not None and (db is None or bslot is not None)).reshape(wshapes[i])
                grads[2 * i + 1] = db
            elif want_db:
                grads[2 * i + 1] = colsum(g)
            if i > 0:
                pa = ACT[ctx.acts[i - 1]]
                if pa == L.ACT_NONE:
                    g = matmul_nn(g, ws[i])
                else:
                    g = matmul_nn(g, ws[i], act=_ACT_BWD[pa], aux_in=auxs[i - 1], ld_aux=auxs[i - 1].shape[1])
            elif ctx.needs_input_grad[0]:
                g = matmul_nn(g, ws[0])
        dx = g.reshape(shp) if ctx.needs_input_grad[0] else None
        return (dx, None, *grads)


class _MSELoss(torch.autograd.Function):
    """nn.MSELoss() (mean) of the reference trainers (base_trainer.py:71): two small launches forward, one backward."""

    @staticmethod
    def forward(ctx, pred, target):
        _dev(pred, target)
        p, t = pred.contiguous(), target.contiguous()
        assert p.shape == t.shape, (p.shape, t.shape)
        part = torch.empty(256, device=p.device, dtype=torch.float32)
        loss = torch.empty((), device=p.device, dtype=torch.float32)
        L.check(L.load().gaot_mse_loss_fwd(_p(p), _p(t), p.numel(), _p(part), _p(loss), _stream()), "gaot_mse_loss_fwd")
        ctx.save_for_backward(p, t)
        return loss

    @staticmethod
    def backward(ctx, gl):
        p, t = ctx.saved_tensors
        dp = torch.empty_like(p)
        L.check(L.load().gaot_mse_loss_bwd(_p(p), _p(t), p.numel(), _p(gl.contiguous()), _p(dp), _stream()), "gaot_mse_loss_bwd")
        return dp, None


def mse_loss(pred, target):
    return _MSELoss.apply(pred, target)




def mse_loss_and_grad(pred, target, tick: Optional[torch.Tensor] = None):
    """(loss, d loss / d pred) of nn.MSELoss() for a unit seed in ONE launch (trainer.TrainStep: `pred.backward(dpred)` continues
    the backward pass; the loss carries no graph).  Same bits as mse_loss().  `tick`: an optimizer's device-resident step counter,
    advanced by the same launch (FlatAdamW.step(ticked=True) then skips its own 1-thread tick launch)."""
    _dev(pred, target)
    p, t = pred.detach().contiguous(), target.contiguous()
    assert p.shape == t.shape, (p.shape, t.shape)
    ws = _scratch("mse", p.device, lambda: (torch.empty(256, device=p.device, dtype=torch.float32), torch.zeros(1, device=p.device, dtype=torch.int32)),
                  "mse_loss_and_grad")
    loss = torch.empty((), device=p.device, dtype=torch.float32)
    dp = torch.empty_like(p)
    L.check(L.load().gaot_mse_loss_fwd_bwd(_p(p), _p(t), p.numel(), _p(ws[0]), _p(ws[1]), _p(loss), _p(dp), _p(tick), _stream()),
            "gaot_mse_loss_fwd_bwd")
    return loss, dp.view_as(pred)


class _KernelMLP(torch.autograd.Function):
    """Fused kernel MLP over edge rows (csrc/kernel_mlp.hip): x [E, c_in <= 16] -> 64 -> ... -> 64, GELU between layers.
    One launch forward; backward recomputes the chain and returns every parameter gradient from one launch (+ reduce)."""

    @staticmethod
    def eligible(x, weights, biases, acts) -> bool:
        n = len(weights)
        if not (2 <= n <= 4) or x.dim() != 2 or x.requires_grad or not x.is_cuda or x.shape[0] == 0:
            return False
        if list(acts) not in (["gelu"] * (n - 1) + ["none"], ["relu"] * (n - 1) + ["none"]) or any(b is None for b in biases):
            return False
        cin = x.shape[1]
        if cin > 16:
            return False
        # every layer at most 64 wide, widths multiples of 4 (narrower layers -- lifting_channels 48 at the 3-D configuration -- run
        # zero-padded at 64 inside the kernels)
        prev = cin
        for w, b in zip(weights, biases):
            if w.dim() != 2 or w.shape[1] != prev or not (4 <= w.shape[0] <= 64 and w.shape[0] % 4 == 0) or tuple(b.shape) != (w.shape[0],):
                return False
            prev = w.shape[0]
        return True

    @staticmethod
    def _ptrs(ts):
        arr = (C.c_void_p * len(ts))()
        for i, t in enumerate(ts):
            arr[i] = t.data_ptr()
        return arr

    @staticmethod
    def _pieces(act) -> int:
        """the "kmlp" setting behind a smooth activation; EXACT products behind ReLU whatever the setting: a ReLU chain's
        gradient is discontinuous in its pre-activations -- at 5e-6 of relative error a few of 10^6 gates flip and the first layers'
        gradients move by 2e-3 (measured, tools/kmlp_ab.py), where the exact products move them by 3e-7"""
        return _PIECES["kmlp"] if act == L.ACT_GELU else 3

    @staticmethod
    def forward(ctx, x, n, act, *wb):
        x = x.contiguous()
        # a weight that is a column block of a wider matrix (the geoembed half of the recovery weight: split_cols) is read in
        # place through its row stride; anything else non-contiguous is copied
        ws = [w if (w.dim() == 2 and w.stride(1) == 1 and w.stride(0) >= w.shape[1] and w.stride(0) % 4 == 0 and not (w.data_ptr() & 15))
              else w.contiguous() for w in wb[:n]]
        bs = [b.contiguous() for b in wb[n:]]
        _dev(x, *ws, *bs)
        E, cin = x.shape
        widths = (C.c_int32 * n)(*[int(w.shape[0]) for w in ws])
        ldw = (C.c_int32 * n)(*[int(w.stride(0)) if w.shape[0] > 1 else 0 for w in ws])
        out = torch.empty(E, ws[-1].shape[0], device=x.device, dtype=torch.float32)
        L.check(L.load().gaot_kernel_mlp_fwd_w(_p(x), E, cin, n, _KernelMLP._ptrs(ws), _KernelMLP._ptrs(bs), act, widths, ldw, _KernelMLP._pieces(act), _p(out), _stream()),
                "gaot_kernel_mlp_fwd")
        ctx.save_for_backward(x, *ws, *bs)
        ctx.n, ctx.act = n, act
        ctx.slots = [_claim(t) for t in wb]        # the parameters' slices of the flat gradient buffer (weights, then biases)
        return out

    @staticmethod
    def backward(ctx, dk):
        n = ctx.n
        sv = ctx.saved_tensors
        x, ws, bs = sv[0], list(sv[1:1 + n]), list(sv[1 + n:])
        E, cin = x.shape
        lib = L.load()
        dk = dk.contiguous()
        psize = (n - 1) * 4096 + 64 * cin + 64 * n
        grads = torch.empty(psize, device=x.device, dtype=torch.float32)
        wsp = torch.empty(int(lib.gaot_kernel_mlp_bwd_workspace(E, cin, n)), device=x.device, dtype=torch.float32)
        wo = [int(w.shape[0]) for w in ws]
        widths = (C.c_int32 * n)(*wo)
        ldw = (C.c_int32 * n)(*[int(w.stride(0)) if w.shape[0] > 1 else 0 for w in ws])
        o = (n - 1) * 4096
        slots = ctx.slots
        if (_WGRAD_DEPTH[0] > 0 and _WGRAD_GROUPED and all(s_ is not None for s_ in slots) and all(v == 64 for v in wo)
                and (64 * cin) % 4 == 0 and all(ctx.needs_input_grad[3:])):
            # every parameter has its slice and every layer is 64 wide (the blocks of the partial rows ARE the parameters' layouts):
            # the per-workgroup partial rows are summed per parameter, straight into the slices, by the grouped column-sum launch
            # at the end of the backward pass -- no row sum here, no packed gradient buffer, no copy into the flat buffer later
            L.check(lib.gaot_kernel_mlp_bwd_w(_p(x), E, cin, n, _KernelMLP._ptrs(ws), _KernelMLP._ptrs(bs), ctx.act, widths, ldw, _KernelMLP._pieces(ctx.act), _p(dk), _p(wsp),
                                              _p(wsp), _stream()), "gaot_kernel_mlp_bwd")
            rows = int(lib.gaot_kernel_mlp_bwd_rows(E))
            part = wsp[:rows * psize].view(rows, psize)
            blocks = [(o, 64 * cin)] + [(m * 4096, 4096) for m in range(n - 1)] + [(o + 64 * cin + 64 * i, 64) for i in range(n)]
            outs = []
            for (off, width), slot in zip(blocks, slots):
                dst = slot.detach()
                outs.append(colsum(part[:, off:off + width], out=dst if (dst.dim() == 2 and not dst.is_contiguous()) else dst.view(-1),
                                   final=True).view(slot.shape))
            return (None, None, None, *outs)
        L.check(lib.gaot_kernel_mlp_bwd_w(_p(x), E, cin, n, _KernelMLP._ptrs(ws), _KernelMLP._ptrs(bs), ctx.act, widths, ldw, _KernelMLP._pieces(ctx.act), _p(dk), _p(grads),
                                          _p(wsp), _stream()), "gaot_kernel_mlp_bwd")
        # 64 x 64 blocks (64 x cin for the first layer) whose leading [out, in] corner is the gradient; the padding carries zeros
        dws = [grads[o:o + 64 * cin].view(64, cin)[:wo[0]]]
        for m in range(n - 1):
            blk = grads[m * 4096:(m + 1) * 4096].view(64, 64)[:wo[m + 1], :wo[m]]
            dws.append(blk if wo[m] == 64 else blk.contiguous())
        ob = o + 64 * cin
        dbs = [grads[ob + 64 * i:ob + 64 * i + wo[i]] for i in range(n)]
        return (None, None, None, *dws, *dbs)


class _KernelMLPPair(torch.autograd.Function):
    """Two fused row-wise MLP chains in ONE launch each way (gaot_kernel_mlp_fwd_pair / _bwd_pair): chain A = the kernel MLP of an integral
    transform over the E edge rows, chain B = the geometry-embedding chain of the same transform over the Q query rows.  Neither reads the
    other; alone chain B is a launch of a few dozen workgroups.  Arithmetic, gradient slots and deferred column sums per chain exactly as
    _KernelMLP (whose pieces this reuses)."""

    @staticmethod
    def _desc(x, ws, bs, act, keep, out=None, dk=None, grads=None, wsp=None):
        n = len(ws)
        w_arr, b_arr = _KernelMLP._ptrs(ws), _KernelMLP._ptrs(bs)
        widths = (C.c_int32 * n)(*[int(w.shape[0]) for w in ws])
        ldw = (C.c_int32 * n)(*[int(w.stride(0)) if w.shape[0] > 1 else 0 for w in ws])
        keep += [w_arr, b_arr, widths, ldw]
        return L.KmlpDesc(x.data_ptr(), x.shape[0], x.shape[1], n, w_arr, b_arr, act, widths, ldw, _KernelMLP._pieces(act),
                          None if out is None else out.data_ptr(), None if dk is None else dk.data_ptr(),
                          None if grads is None else grads.data_ptr(), None if wsp is None else wsp.data_ptr())

    @staticmethod
    def _norm(x, n, wb):
        x = x.contiguous()
        ws = [w if (w.dim() == 2 and w.stride(1) == 1 and w.stride(0) >= w.shape[1] and w.stride(0) % 4 == 0 and not (w.data_ptr() & 15))
              else w.contiguous() for w in wb[:n]]
        bs = [b.contiguous() for b in wb[n:]]
        _dev(x, *ws, *bs)
        return x, ws, bs

    @staticmethod
    def forward(ctx, xa, na, acta, xb, nb, actb, *wb):
        wba, wbb = wb[:2 * na], wb[2 * na:]
        xa, wsa, bsa = _KernelMLPPair._norm(xa, na, wba)
        xb, wsb, bsb = _KernelMLPPair._norm(xb, nb, wbb)
        outa = torch.empty(xa.shape[0], wsa[-1].shape[0], device=xa.device, dtype=torch.float32)
        outb = torch.empty(xb.shape[0], wsb[-1].shape[0], device=xb.device, dtype=torch.float32)
        keep = []
        da = _KernelMLPPair._desc(xa, wsa, bsa, acta, keep, out=outa)
        db = _KernelMLPPair._desc(xb, wsb, bsb, actb, keep, out=outb)
        L.check(L.load().gaot_kernel_mlp_fwd_pair(C.byref(da), C.byref(db), _stream()), "gaot_kernel_mlp_fwd_pair")
        ctx.save_for_backward(xa, xb, *wsa, *bsa, *wsb, *bsb)
        ctx.meta = (na, acta, nb, actb)
        ctx.slots_a = [_claim(t) for t in wba]
        ctx.slots_b = [_claim(t) for t in wbb]
        return outa, outb

    @staticmethod
    def _finish(lib, x, ws, n, slots, wsp, grads, need_all):
        """the parameter gradients of one chain from the kernel's output: per-workgroup partial rows summed into the parameters' slices by
        the grouped column sum at the end of the pass (wsp is grads), or the packed gradient vector cut into its blocks"""
        E, cin = x.shape
        wo = [int(w.shape[0]) for w in ws]
        o = (n - 1) * 4096
        psize = o + 64 * cin + 64 * n
        if grads is None:
            rows = int(lib.gaot_kernel_mlp_bwd_rows(E))
            part = wsp[:rows * psize].view(rows, psize)
            blocks = [(o, 64 * cin)] + [(m * 4096, 4096) for m in range(n - 1)] + [(o + 64 * cin + 64 * i, 64) for i in range(n)]
            outs = []
            for (off, width), slot in zip(blocks, slots):
                dst = slot.detach()
                outs.append(colsum(part[:, off:off + width], out=dst if (dst.dim() == 2 and not dst.is_contiguous()) else dst.view(-1),
                                   final=True).view(slot.shape))
            return outs
        dws = [grads[o:o + 64 * cin].view(64, cin)[:wo[0]]]
        for m in range(n - 1):
            blk = grads[m * 4096:(m + 1) * 4096].view(64, 64)[:wo[m + 1], :wo[m]]
            dws.append(blk if wo[m] == 64 else blk.contiguous())
        ob = o + 64 * cin
        return dws + [grads[ob + 64 * i:ob + 64 * i + wo[i]] for i in range(n)]

    @staticmethod
    def _direct(slots, ws, x, needs):
        return (_WGRAD_DEPTH[0] > 0 and _WGRAD_GROUPED and all(s_ is not None for s_ in slots) and all(int(w.shape[0]) == 64 for w in ws)
                and (64 * x.shape[1]) % 4 == 0 and all(needs))

    @staticmethod
    def backward(ctx, dka, dkb):
        na, acta, nb, actb = ctx.meta
        sv = ctx.saved_tensors
        xa, xb = sv[0], sv[1]
        wsa, bsa = list(sv[2:2 + na]), list(sv[2 + na:2 + 2 * na])
        wsb, bsb = list(sv[2 + 2 * na:2 + 2 * na + nb]), list(sv[2 + 2 * na + nb:])
        lib = L.load()
        dka = (dka if dka is not None else torch.zeros(xa.shape[0], wsa[-1].shape[0], device=xa.device)).contiguous()
        dkb = (dkb if dkb is not None else torch.zeros(xb.shape[0], wsb[-1].shape[0], device=xb.device)).contiguous()
        keep, res = [], []
        for x, ws, n in ((xa, wsa, na), (xb, wsb, nb)):
            psize = (n - 1) * 4096 + 64 * x.shape[1] + 64 * n
            wsp = torch.empty(int(lib.gaot_kernel_mlp_bwd_workspace(x.shape[0], x.shape[1], n)), device=x.device, dtype=torch.float32)
            res.append([wsp, psize])
        need = ctx.needs_input_grad
        direct_a = _KernelMLPPair._direct(ctx.slots_a, wsa, xa, need[6:6 + 2 * na])
        direct_b = _KernelMLPPair._direct(ctx.slots_b, wsb, xb, need[6 + 2 * na:])
        ga = None if direct_a else torch.empty(res[0][1], device=xa.device, dtype=torch.float32)
        gb = None if direct_b else torch.empty(res[1][1], device=xb.device, dtype=torch.float32)
        da = _KernelMLPPair._desc(xa, wsa, bsa, acta, keep, dk=dka, grads=res[0][0] if direct_a else ga, wsp=res[0][0])
        db = _KernelMLPPair._desc(xb, wsb, bsb, actb, keep, dk=dkb, grads=res[1][0] if direct_b else gb, wsp=res[1][0])
        L.check(lib.gaot_kernel_mlp_bwd_pair(C.byref(da), C.byref(db), _stream()), "gaot_kernel_mlp_bwd_pair")
        outs_a = _KernelMLPPair._finish(lib, xa, wsa, na, ctx.slots_a, res[0][0], ga, True)
        outs_b = _KernelMLPPair._finish(lib, xb, wsb, nb, ctx.slots_b, res[1][0], gb, True)
        return (None, None, None, None, None, None, *outs_a, *outs_b)


_MLP_PAIR = os.environ.get("GAOT_MLP_PAIR", "1") != "0"        # A/B switch (tools): 0 = the two chains as two launches


def mlp_chain_pair(xa, weights_a, biases_a, acts_a, xb, weights_b, biases_b, acts_b):
    """(chain_a(xa), chain_b(xb)): both through ONE launch each way when both are fused-kernel chains (see _KernelMLPPair), else one by one"""
    if (_FUSED_KERNEL_MLP and _MLP_PAIR and _KernelMLP.eligible(xa, weights_a, biases_a, acts_a) and _KernelMLP.eligible(xb, weights_b, biases_b, acts_b)
            and xa.device == xb.device):
        return _KernelMLPPair.apply(xa, len(weights_a), ACT[acts_a[0]], xb, len(weights_b), ACT[acts_b[0]],
                                    *weights_a, *biases_a, *weights_b, *biases_b)
    return mlp_chain(xa, weights_a, biases_a, acts_a), mlp_chain(xb, weights_b, biases_b, acts_b)


def mlp_chain(x, weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]], acts: Sequence[str]):
    if _FUSED_KERNEL_MLP and _KernelMLP.eligible(x, weights, biases, acts):
        return _KernelMLP.apply(x, len(weights), ACT[acts[0]], *weights, *biases)
    wb = []
    for w, b in zip(weights, biases):
        wb += [w, b]
    return _MLPChain.apply(x, tuple(acts), *wb)


# --------------------------------------------------------------------------------------------
# GNO integral transform
# --------------------------------------------------------------------------------------------
def gno_gather_reduce(w, src, splits, cols, edge_map, n_out, escale=None):
    _dev(src, splits)
    B, n_src_rows, Cc = src.shape
    src = src.contiguous()
    out = torch.empty(B, n_out, Cc, device=src.device, dtype=torch.float32)
    L.check(L.load().gaot_gno_gather_reduce(_p(w), _p(src), B, n_src_rows, Cc, _p(splits), _p(cols), _p(edge_map),
                                            n_out, _p(escale), _p(out), _stream()), "gaot_gno_gather_reduce")
    return out


class _GNOTransform(torch.autograd.Function):
    """out[b,q,:] = sum_{e in seg(q)} a_e * k[e,:] * f[b, j(e), :]     (agno.py:198,245-271)"""

    @staticmethod
    def forward(ctx, k, f, plan, escale):
        k = k.contiguous()
        f = f.contiguous()
        out = gno_gather_reduce(k, f, plan.splits, plan.index, None, plan.Q, escale)
        ctx.plan = plan
        ctx.save_for_backward(k, f, escale if escale is not None else k.new_empty(0))
        ctx.has_scale = escale is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        k, f, esc = ctx.saved_tensors
        esc = esc if ctx.has_scale else None
        plan = ctx.plan
        dout = dout.contiguous()
        dk = df = da = None
        B, n_src, Cc = f.shape
        lib = L.load()
        need_a = ctx.has_scale and ctx.needs_input_grad[3]      # learned (dot-product) attention only
        if ctx.needs_input_grad[0] or need_a:
            dk = torch.empty_like(k)
            L.check(lib.gaot_gno_edge_grad(_p(dout), _p(f), B, plan.Q, n_src, Cc, _p(plan.index), _p(plan.edge_query),
                                           plan.E, None if need_a else _p(esc), _p(dk), _stream()), "gaot_gno_edge_grad")
            if need_a:      # d/da_e = <sum_b dOut*f , k_e> ; then dk_e = a_e * (sum_b dOut*f): one pass over the [E,C] rows
                da = torch.zeros_like(esc)
                L.check(lib.gaot_edge_rowdot_scale(_p(dk), _p(k), _p(esc), plan.E, Cc, _p(da), _stream()), "gaot_edge_rowdot_scale")
        if ctx.needs_input_grad[1]:
            df = gno_gather_reduce(k, dout, plan.t_splits, plan.edge_query, plan.t_edge, n_src, esc)
        return dk, df, None, da


def gno_transform(k, f, plan, escale=None):
    return _GNOTransform.apply(k, f, plan, escale)


class _GNOLiftTransform(torch.autograd.Function):
    """Encoder transform with the linear lifting folded in (csrc/gno.hip lift_* kernels):
        out[b,q,:] = sum_e a_e k_e (*) (Wl pn[b,j(e),:] + bl)
    without ever forming the lifted [B,n,C] tensor; backward gives dk, dWl, dbl from one launch + one column sum."""

    @staticmethod
    def eligible(pn, wl, C: int, escale) -> bool:
        return (pn.dim() == 3 and 1 <= pn.shape[-1] <= 4 and C % 4 == 0 and not pn.requires_grad
                and (escale is None or not escale.requires_grad) and wl.reshape(wl.shape[0], -1).shape[1] == pn.shape[-1])

    @staticmethod
    def forward(ctx, k, pn, wl, bl, plan, escale):
        _dev(k, pn, wl)
        k, pn = k.contiguous(), pn.contiguous()
        B, n_src, ci = pn.shape
        Cc = k.shape[1]
        w2 = wl.reshape(Cc, ci).contiguous()
        out = torch.empty(B, plan.Q, Cc, device=k.device, dtype=torch.float32)
        lib = L.load()
        if _GNO_EP == 1 or (_GNO_EP == 2 and plan.rows_skewed and Cc <= 256):
            # degree-skewed rows: edge-partitioned kernel with segmented reductions (csrc/gno_ep.hip)
            ws = torch.empty(int(lib.gaot_gno_ep_workspace(plan.E, Cc, B)), device=k.device, dtype=torch.float32)
            L.check(lib.gaot_gno_lift_gather_reduce_ep(_p(k), _p(pn), _p(w2), _p(bl), B, n_src, ci, Cc, _p(plan.splits), _p(plan.index),
                                                       _p(plan.edge_query), plan.Q, plan.E, _p(escale), _p(out), _p(ws), _p(plan.e_dev), _stream()),
                    "gaot_gno_lift_gather_reduce_ep")
        else:
            ow = _want_word(k.device)
            L.check(lib.gaot_gno_lift_gather_reduce(_p(k), _p(pn), _p(w2), _p(bl), B, n_src, ci, Cc, _p(plan.splits),
                                                    _p(plan.index), plan.Q, _p(escale), _p(out), _p(ow), _stream()),
                    "gaot_gno_lift_gather_reduce")
            _publish(ow, out)
        ctx.plan = plan
        ctx.save_for_backward(k, pn, w2, bl if bl is not None else k.new_empty(0), escale if escale is not None else k.new_empty(0))
        ctx.meta = (wl.shape, bl is not None, escale is not None, _claim(wl), _claim(bl))
        return out

    @staticmethod
    def backward(ctx, dout):
        k, pn, w2, bl, esc = ctx.saved_tensors
        wshape, has_b, has_e, wslot, bslot = ctx.meta
        plan = ctx.plan
        B, n_src, ci = pn.shape
        Cc = k.shape[1]
        lib = L.load()
        dout = dout.contiguous()
        dk = torch.empty_like(k)
        if plan.E == 0:
            return dk, None, torch.zeros(wshape, device=k.device), (torch.zeros(Cc, device=k.device) if has_b else None), None, None
        nparts = int(lib.gaot_gno_lift_edge_grad_parts(plan.E, Cc))
        part = torch.empty(nparts, (ci + 1) * Cc, device=k.device, dtype=torch.float32)
        L.check(lib.gaot_gno_lift_edge_grad(_p(dout), _p(k), _p(pn), _p(w2), _p(bl) if has_b else None, B, plan.Q, n_src, ci, Cc,
                                            _p(plan.index), _p(plan.edge_query), plan.E, _p(esc) if has_e else None, _p(dk),
                                            _p(part), _stream()), "gaot_gno_lift_edge_grad")
        if ci == 1 and wslot is not None and (not has_b or bslot is not None):
            # one input channel: dWl and dbl are the two column blocks of the partial rows -> straight into their gradient slices
            dwl = colsum(part[:, :Cc], out=wslot.detach().view(-1), final=True).view(wshape)
            dbl = colsum(part[:, Cc:], out=bslot.detach(), final=True) if has_b else None
            return dk, None, dwl, dbl, None, None
        sums = colsum(part)                                  # [(ci + 1) * C] = [dWl^T | dbl]
        dwl = sums[:ci * Cc].reshape(ci, Cc).t().reshape(wshape) if ci > 1 else sums[:Cc].reshape(wshape)
        dbl = sums[ci * Cc:] if has_b else None
        return dk, None, dwl, dbl, None, None


class _GNOProjTransform(torch.autograd.Function):
    """Decoder transform with the following point-wise linear maps folded in (csrc/gno.hip proj_* kernels):
        y[b,q,o] = sum_ch weff[o,ch] (sum_e a_e k[e,ch] f[b,j(e),ch]) + rowb[q,o] + bias[o]        (o < 4)
    The [B,Q,C] transform output and its gradient never exist."""

    @staticmethod
    def eligible(f, weff, escale) -> bool:
        return (f.dim() == 3 and weff.dim() == 2 and 1 <= weff.shape[0] <= 4 and f.shape[-1] % 4 == 0 and f.shape[-1] <= 256
                and weff.shape[1] == f.shape[-1] and (escale is None or not escale.requires_grad))

    @staticmethod
    def forward(ctx, k, f, weff, rowb, bias, plan, escale):
        _dev(k, f, weff)
        k, f, weff = k.contiguous(), f.contiguous(), weff.contiguous()
        B, n_src, Cc = f.shape
        OC = weff.shape[0]
        rb = rowb.contiguous() if rowb is not None else None
        y = torch.empty(B, plan.Q, OC, device=k.device, dtype=torch.float32)
        if _GNO_EP != 0 and B > 1:      # batch inside the lane group: every kernel-value row is read once per 4 samples
            L.check(L.load().gaot_gno_proj_gather_reduce_bin(_p(k), _p(f), _p(weff), _p(rb), _p(bias), B, n_src, Cc, OC, _p(plan.splits),
                                                             _p(plan.index), plan.Q, _p(escale), _p(y), _p(plan.row_order), _stream()),
                    "gaot_gno_proj_gather_reduce_bin")
        else:
            L.check(L.load().gaot_gno_proj_gather_reduce(_p(k), _p(f), _p(weff), _p(rb), _p(bias), B, n_src, Cc, OC, _p(plan.splits),
                                                         _p(plan.index), plan.Q, _p(escale), _p(y), _stream()),
                    "gaot_gno_proj_gather_reduce")
        ctx.plan = plan
        ctx.save_for_backward(k, f, weff, escale if escale is not None else k.new_empty(0))
        ctx.meta = (escale is not None, rowb.shape if rowb is not None else None, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        k, f, weff, esc = ctx.saved_tensors
        has_e, rb_shape, has_bias = ctx.meta
        plan = ctx.plan
        B, n_src, Cc = f.shape
        OC = weff.shape[0]
        lib = L.load()
        dy = dy.contiguous()
        need = ctx.needs_input_grad
        dk = torch.empty_like(k)
        df = torch.empty_like(f) if need[1] else None
        drowb = dbias = None
        if plan.E == 0:
            dk.zero_()
            dweff = torch.zeros_like(weff)
            if df is not None:
                df.zero_()
        else:
            nparts = int(lib.gaot_gno_lift_edge_grad_parts(plan.E, Cc))
            part = torch.empty(nparts, OC * Cc, device=k.device, dtype=torch.float32)
            # the edge-partitioned dF kernel keeps four samples' sums per lane group (every k row read once per four samples): also
            # ahead of the row-parallel form on regular plans once the batch is that large (C2: 24.6 -> 18.9 us, tools/gno_c2_kernels.py)
            ep = df is not None and (_GNO_EP == 1 or (_GNO_EP == 2 and (plan.t_rows_skewed or B >= 4)))
            L.check(lib.gaot_gno_proj_backward(_p(dy), _p(k), _p(f), _p(weff), B, plan.Q, n_src, Cc, OC, _p(plan.index),
                                               _p(plan.edge_query), plan.E, _p(plan.t_splits), _p(plan.t_edge),
                                               _p(esc) if has_e else None, _p(dk), _p(part), None if ep else _p(df), _stream()),
                    "gaot_gno_proj_backward")
            if ep:      # dF over the (skewed) transposed CSR: edge-partitioned, segmented
                ws = torch.empty(int(lib.gaot_gno_ep_workspace(plan.E, Cc, B)), device=k.device, dtype=torch.float32)
                ow = _want_word(k.device)          # dF's magnitude word: the processor's last input-gradient product reads dF as its A operand
                L.check(lib.gaot_gno_proj_gather_t_ep_w(_p(k), _p(dy), _p(weff), B, plan.Q, n_src, Cc, OC, _p(plan.index), _p(plan.edge_query),
                                                        plan.E, _p(plan.t_splits), _p(plan.t_edge), _p(esc) if has_e else None, _p(df), _p(ws),
                                                        _p(plan.e_dev), _p(ow), _stream()), "gaot_gno_proj_gather_t_ep_w")
                _publish(ow, df)
            dweff = colsum(part).reshape(OC, Cc)
        if rb_shape is not None and need[3]:
            drowb = batchsum(dy.reshape(B, -1), B).reshape(rb_shape)
        if has_bias and need[4]:
            dbias = colsum(dy.reshape(-1, OC))
        return dk, df, dweff, drowb, dbias, None, None


def gno_proj_transform(k, f, weff, rowb, bias, plan, escale=None):
    return _GNOProjTransform.apply(k, f, weff, rowb, bias, plan, escale)


def gno_lift_transform(k, pn, wl, bl, plan, escale=None):
    return _GNOLiftTransform.apply(k, pn, wl, bl, plan, escale)


class _SegmentSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, score, plan):
        score = score.contiguous()
        attn = torch.zeros_like(score)
        L.check(L.load().gaot_segment_softmax_fwd(_p(score), _p(plan.splits), plan.Q, _p(attn), _stream()), "gaot_segment_softmax_fwd")
        ctx.plan = plan
        ctx.save_for_backward(attn)
        return attn

    @staticmethod
    def backward(ctx, dattn):
        (attn,) = ctx.saved_tensors
        ds = torch.zeros_like(attn)
        L.check(L.load().gaot_segment_softmax_bwd(_p(attn), _p(dattn.contiguous()), _p(ctx.plan.splits), ctx.plan.Q, _p(ds),
                                                  _stream()), "gaot_segment_softmax_bwd")
        return ds, None


def segment_softmax(score, plan):
    return _SegmentSoftmax.apply(score, plan)


class _SegmentSum(torch.autograd.Function):
    """out[b,q,:] = rowscale[q] * sum_{e in seg(q)} x[b,e,:]  (batched per-edge values: 'nonlinear' transforms)"""

    @staticmethod
    def forward(ctx, x, plan, rowscale):
        x = x.contiguous()
        B, E, Cc = x.shape
        out = torch.empty(B, plan.Q, Cc, device=x.device, dtype=torch.float32)
        L.check(L.load().gaot_gno_segment_sum(_p(x), B, E, Cc, _p(plan.splits), plan.Q, _p(rowscale), _p(out), _stream()),
                "gaot_gno_segment_sum")
        ctx.plan = plan
        ctx.rowscale = rowscale
        ctx.E = E
        return out

    @staticmethod
    def backward(ctx, dout):
        plan = ctx.plan
        dout = dout.contiguous()
        B, _, Cc = dout.shape
        dx = torch.empty(B, ctx.E, Cc, device=dout.device, dtype=torch.float32)
        L.check(L.load().gaot_segment_broadcast(_p(dout), B, ctx.E, Cc, plan.Q, _p(plan.edge_query), _p(ctx.rowscale), _p(dx), _stream()),
                "gaot_segment_broadcast")
        zero_pad_rows(dx, plan)
        return dx, None, None


def zero_pad_rows(x, plan):
    """x [.., E, C] per-edge rows of a padded union (plan.StaticUnion): the rows past the union's real edge count are set to zero, so a
    reduction over all E rows (the weight gradients of an MLP over the edge rows) sees exact zeros there.  No-op for ordinary plans."""
    if plan.e_dev is None or x.numel() == 0:
        return x
    E, Cc = x.shape[-2], x.shape[-1]
    L.check(L.load().gaot_edge_zero_pads(_p(x), x.numel() // (E * Cc), E, Cc, _p(plan.e_dev), E, _stream()), "gaot_edge_zero_pads")
    return x


def segment_sum(x, plan, rowscale=None):
    return _SegmentSum.apply(x, plan, rowscale)


class _EdgeDotScore(torch.autograd.Function):
    """score[e] = scale * <qn[query(e)], kn[j(e)]>  (dot-product attention, agno.py:215-217): one gather-dot kernel; the two
    node gradients are segment reductions over the CSR / transposed CSR (no atomics)."""

    @staticmethod
    def forward(ctx, qn, kn, plan, scale):
        _dev(qn, kn)
        qn, kn = qn.contiguous(), kn.contiguous()
        score = torch.zeros(max(plan.E, 1), device=qn.device, dtype=torch.float32)
        L.check(L.load().gaot_edge_dot_score(_p(qn), _p(kn), qn.shape[1], _p(plan.index), _p(plan.edge_query), plan.E, float(scale),
                                             _p(score), _stream()), "gaot_edge_dot_score")
        ctx.plan, ctx.scale = plan, float(scale)
        ctx.save_for_backward(qn, kn)
        return score

    @staticmethod
    def backward(ctx, ds):
        qn, kn = ctx.saved_tensors
        plan = ctx.plan
        ds = (ds * ctx.scale).contiguous()
        dqn = gno_gather_reduce(None, kn[None], plan.splits, plan.index, None, plan.Q, ds)[0]
        dkn = gno_gather_reduce(None, qn[None], plan.t_splits, plan.edge_query, plan.t_edge, plan.n_src, ds)[0]
        return dqn, dkn, None, None


def edge_dot_score(qn, kn, plan, scale):
    return _EdgeDotScore.apply(qn, kn, plan, scale)


class _SegmentMax(torch.autograd.Function):
    """PointNet pooling (gemb.py:217): per-query maximum over the edge rows, 0 for queries without neighbours."""

    @staticmethod
    def forward(ctx, h, plan):
        _dev(h)
        h = h.contiguous()
        out = torch.empty(plan.Q, h.shape[1], device=h.device, dtype=torch.float32)
        L.check(L.load().gaot_segment_max_fwd(_p(h), h.shape[1], _p(plan.splits), plan.Q, _p(out), _stream()), "gaot_segment_max_fwd")
        ctx.plan = plan
        ctx.save_for_backward(h, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        h, out = ctx.saved_tensors
        dh = torch.empty_like(h)
        zero_pad_rows(dh, ctx.plan)           # (the kernel below writes the rows' edges only)
        L.check(L.load().gaot_segment_max_bwd(_p(h), _p(out), _p(dout.contiguous()), h.shape[1], _p(ctx.plan.splits), ctx.plan.Q, _p(dh),
                                              _stream()), "gaot_segment_max_bwd")
        return dh, None


def segment_max(h, plan):
    return _SegmentMax.apply(h, plan)


def _ptr_array(ts):
    arr = (C.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = None if t is None else t.data_ptr()
    return arr


class _ScaleMix(torch.autograd.Function):
    """multiscale combination (magno.py:291-303): sum_i w[q,i] * t_i, or the plain mean over scales when w is None"""

    @staticmethod
    def forward(ctx, w, *ts):
        _dev(*ts)
        ts = [t.contiguous() for t in ts]
        B, Q, Cc = ts[0].shape
        wc = w.contiguous() if w is not None else None
        out = torch.empty_like(ts[0])
        L.check(L.load().gaot_scale_mix_fwd(_ptr_array(ts), len(ts), _p(wc), B, Q, Cc, _p(out), _stream()), "gaot_scale_mix_fwd")
        ctx.save_for_backward(*ts, *( [wc] if wc is not None else []))
        ctx.n, ctx.has_w = len(ts), wc is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        sv = ctx.saved_tensors
        ts, wc = list(sv[:ctx.n]), (sv[ctx.n] if ctx.has_w else None)
        B, Q, Cc = ts[0].shape
        dout = dout.contiguous()
        dts = [torch.empty_like(t) if ctx.needs_input_grad[1 + i] else None for i, t in enumerate(ts)]
        dw = torch.empty_like(wc) if (ctx.has_w and ctx.needs_input_grad[0]) else None
        L.check(L.load().gaot_scale_mix_bwd(_ptr_array(ts), _ptr_array(dts), ctx.n, _p(wc), B, Q, Cc, _p(dout), _p(dw), _stream()),
                "gaot_scale_mix_bwd")
        return (dw, *dts)


def scale_mix(ts, w=None):
    return _ScaleMix.apply(w, *ts)


class _NonlinearTransform(torch.autograd.Function):
    """'nonlinear' / 'nonlinear_kernelonly' integral transform (agno.py:230-271) around a caller-supplied kernel MLP:
    out[b,q,:] = sum_e a_e k[b,e,:] (*) f[b,j(e),:]   with k = MLP([y_j, x_i, f(y_j)]) evaluated per sample.
    This Function covers the two ends (the gather+concat that builds the MLP rows is `edge_cat`, the product/reduction is
    here); the MLP in between is the ordinary HIP GEMM chain with its own autograd."""

    @staticmethod
    def forward(ctx, k, f, plan, escale, mul_f):
        _dev(k, f)
        k, f = k.contiguous(), f.contiguous()
        B, n_src, Cc = f.shape
        out = torch.empty(B, plan.Q, Cc, device=k.device, dtype=torch.float32)
        L.check(L.load().gaot_gno_bk_reduce(_p(k), _p(f), B, n_src, Cc, plan.E, _p(plan.splits), _p(plan.index), plan.Q, _p(escale),
                                            int(mul_f), _p(out), _stream()), "gaot_gno_bk_reduce")
        ctx.plan, ctx.mul_f = plan, bool(mul_f)
        ctx.save_for_backward(k, f, escale if escale is not None else k.new_empty(0))
        ctx.has_e = escale is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        k, f, esc = ctx.saved_tensors
        esc = esc if ctx.has_e else None
        plan = ctx.plan
        B, n_src, Cc = f.shape
        dout = dout.contiguous()
        need = ctx.needs_input_grad
        dk = torch.empty_like(k) if need[0] else None
        df = torch.empty_like(f) if (need[1] and ctx.mul_f) else None
        da = torch.zeros_like(esc) if (ctx.has_e and need[3]) else None
        if plan.E == 0:
            return (dk.zero_() if dk is not None else None), (df.zero_() if df is not None else None), None, da, None
        L.check(L.load().gaot_gno_bk_backward(_p(dout), _p(k), _p(f), None, 0, B, n_src, Cc, plan.E, plan.Q, _p(plan.index), _p(plan.edge_query),
                                              _p(plan.t_splits), _p(plan.t_edge), _p(esc), int(ctx.mul_f), _p(dk), _p(df), _p(da), _stream()),
                "gaot_gno_bk_backward")
        return dk, df, None, da, None


def nonlinear_transform(k, f, plan, escale, mul_f: bool):
    return _NonlinearTransform.apply(k, f, plan, escale, mul_f)


class _EdgeCat(torch.autograd.Function):
    """rows of the 'nonlinear' kernel MLP: x[b,e,:] = [feat[e,:], f[b,j(e),:]]; backward sums the f-slice of the row gradients
    per source node over the transposed CSR (feat carries no gradient: geometry only)."""

    @staticmethod
    def forward(ctx, feat, f, plan):
        _dev(feat, f)
        feat, f = feat.contiguous(), f.contiguous()
        B, n_src, Cc = f.shape
        W0 = feat.shape[1]
        x = torch.empty(B, plan.E, W0 + Cc, device=f.device, dtype=torch.float32)
        L.check(L.load().gaot_edge_cat(_p(feat), W0, _p(f), B, n_src, Cc, _p(plan.index), plan.E, _p(x), _stream()), "gaot_edge_cat")
        ctx.plan, ctx.dims = plan, (B, n_src, Cc, W0)
        return x

    @staticmethod
    def backward(ctx, dx):
        plan = ctx.plan
        B, n_src, Cc, W0 = ctx.dims
        dx = dx.contiguous()
        df = torch.empty(B, n_src, Cc, device=dx.device, dtype=torch.float32)
        if plan.E == 0:
            return None, df.zero_(), None
        L.check(L.load().gaot_gno_bk_backward(_p(dx), None, None, _p(dx), W0, B, n_src, Cc, plan.E, plan.Q, _p(plan.index), _p(plan.edge_query),
                                              _p(plan.t_splits), _p(plan.t_edge), None, 0, None, _p(df), None, _stream()), "gaot_gno_bk_backward")
        return None, df, None


def edge_cat(feat, f, plan):
    return _EdgeCat.apply(feat, f, plan)


class _CondAffine(torch.autograd.Function):
    """ConditionedNorm's modulation (mlp.py:118-124): y[b,s,:] = x[b,s,:] * scale[b,:] + shift[b,:]"""

    @staticmethod
    def forward(ctx, x, scale, shift):
        _dev(x, scale, shift)
        x, scale, shift = x.contiguous(), scale.contiguous(), shift.contiguous()
        B, D = scale.shape
        S = x.numel() // (B * D)
        y = torch.empty_like(x)
        L.check(L.load().gaot_cond_affine_fwd(_p(x), _p(scale), _p(shift), B, S, D, _p(y), _stream()), "gaot_cond_affine_fwd")
        ctx.save_for_backward(x, scale)
        ctx.dims = (B, S, D)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scale = ctx.saved_tensors
        B, S, D = ctx.dims
        lib = L.load()
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        chunks = int(lib.gaot_cond_affine_bwd_chunks(S))
        part = torch.empty(chunks, B, 2 * D, device=x.device, dtype=torch.float32)
        L.check(lib.gaot_cond_affine_bwd(_p(x), _p(dy), _p(scale), B, S, D, _p(dx), _p(part), _stream()), "gaot_cond_affine_bwd")
        sums = batchsum(part, chunks) if chunks > 1 else part[0]
        return dx, sums[:, :D], sums[:, D:]


def cond_affine(x, scale, shift):
    return _CondAffine.apply(x, scale, shift)


class _Rope(torch.autograd.Function):
    """rotary embedding of the q and k heads of the fused projection output (attn.py:106-108); an orthogonal map, so the
    backward is the transposed rotation of the incoming gradient.  (The kernel works in place; the copy keeps the projection's
    own output untouched -- autograd forbids in-place edits of a custom Function's view outputs.)"""

    @staticmethod
    def forward(ctx, qkv, n_heads, D, cos_sin):
        _dev(qkv, cos_sin)
        assert qkv.dim() == 3
        out = qkv.contiguous().clone()
        B, S, W = out.shape
        L.check(L.load().gaot_rope_inplace(_p(out), B, S, W, n_heads, D, _p(cos_sin), 0, _stream()), "gaot_rope_inplace")
        ctx.args = (n_heads, D)
        ctx.save_for_backward(cos_sin)
        return out

    @staticmethod
    def backward(ctx, g):
        (cos_sin,) = ctx.saved_tensors
        n_heads, D = ctx.args
        g = g.contiguous().clone()
        B, S, W = g.shape
        L.check(L.load().gaot_rope_inplace(_p(g), B, S, W, n_heads, D, _p(cos_sin), 1, _stream()), "gaot_rope_inplace")
        return g, None, None, None


def rope(qkv, n_heads, D, cos_sin):
    return _Rope.apply(qkv, n_heads, D, cos_sin)


# --------------------------------------------------------------------------------------------
# processor ops
# --------------------------------------------------------------------------------------------
class _RMSNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, eps):
        _dev(x, w)
        shp = x.shape
        D = shp[-1]
        xm = x.reshape(-1, D).contiguous()
        M = xm.shape[0]
        y = torch.empty_like(xm)
        rstd = torch.empty(M, device=x.device, dtype=torch.float32)
        yw = _want_word(x.device)
        L.check(L.load().gaot_rmsnorm_fwd(_p(xm), _p(w), M, D, float(eps), _p(y), _p(rstd), _p(yw), _stream()), "gaot_rmsnorm_fwd")
        ctx.save_for_backward(xm, w, rstd)
        ctx.shp = shp
        ctx.slot = _claim(w)
        yr = y.reshape(shp)
        _publish(yw, y, yr)
        return yr

    @staticmethod
    def backward(ctx, dy):
        xm, w, rstd = ctx.saved_tensors
        M, D = xm.shape
        lib = L.load()
        g = dy.reshape(M, D).contiguous()
        dx = torch.empty_like(xm)
        P = int(lib.gaot_rmsnorm_bwd_partials(M))
        part = torch.empty(P, D, device=xm.device, dtype=torch.float32)
        dxw = _want_word(xm.device)
        L.check(lib.gaot_rmsnorm_bwd(_p(xm), _p(w), _p(rstd), _p(g), None, None, M, D, _p(dx), _p(part), _p(dxw), _stream()), "gaot_rmsnorm_bwd")
        dw = None
        if ctx.needs_input_grad[1]:
            dw = colsum(part, out=ctx.slot.detach() if ctx.slot is not None else None, final=ctx.slot is not None)
        dxr = dx.reshape(ctx.shp)
        _publish(dxw, dx, dxr)
        return dxr, dw, None


def rms_norm(x, w, eps):
    return _RMSNorm.apply(x, w, eps)


class _RMSNormFork(torch.autograd.Function):
    """(x, rmsnorm(x)[, x]): the pre-norm residual fork.  Handing the untouched stream back through the SAME node lets the
    backward add the residual branch's gradient inside the norm-gradient kernel (dx_add) instead of autograd launching
    a separate elementwise add per fork.  `n_alias` = 2 hands out a second alias of the stream (the long-range skip of the
    U-shaped processor, attn.py:281-299): its gradient is added in the same kernel (dx_add2)."""

    @staticmethod
    def forward(ctx, x, w, eps, n_alias=1):
        ctx.set_materialize_grads(False)      # an output nobody differentiated stays None (no zero tensors, no kernel reading them)
        _dev(x, w)
        shp = x.shape
        D = shp[-1]
        xm = x.reshape(-1, D).contiguous()
        M = xm.shape[0]
        y = torch.empty_like(xm)
        rstd = torch.empty(M, device=x.device, dtype=torch.float32)
        yw = _want_word(x.device)
        L.check(L.load().gaot_rmsnorm_fwd(_p(xm), _p(w), M, D, float(eps), _p(y), _p(rstd), _p(yw), _stream()), "gaot_rmsnorm_fwd")
        ctx.save_for_backward(xm, w, rstd)
        ctx.shp = shp
        ctx.slot = _claim(w)
        yr = y.reshape(shp)
        _publish(yw, y, yr)
        xw = _amax_get(x, xm)            # the aliases of the stream carry the stream's word on
        a1 = x.view_as(x)
        _publish(xw, a1)
        if n_alias == 2:
            a2 = x.view_as(x)
            _publish(xw, a2)
            return a1, yr, a2
        return a1, yr

    @staticmethod
    def backward(ctx, dres, dy, dskip=None):
        xm, w, rstd = ctx.saved_tensors
        M, D = xm.shape
        if dy is None:
            tot = dres if dskip is None else (dskip if dres is None else dres + dskip)
            return tot, None, None, None
        lib = L.load()
        g = dy.reshape(M, D).contiguous()
        add = dres.reshape(M, D).contiguous() if dres is not None else None
        add2 = dskip.reshape(M, D).contiguous() if dskip is not None else None
        dx = torch.empty_like(xm)
        P = int(lib.gaot_rmsnorm_bwd_partials(M))
        part = torch.empty(P, D, device=xm.device, dtype=torch.float32)
        dxw = _want_word(xm.device)
        L.check(lib.gaot_rmsnorm_bwd(_p(xm), _p(w), _p(rstd), _p(g), _p(add), _p(add2), M, D, _p(dx), _p(part), _p(dxw), _stream()), "gaot_rmsnorm_bwd")
        dw = None
        if ctx.needs_input_grad[1]:
            dw = colsum(part, out=ctx.slot.detach() if ctx.slot is not None else None, final=ctx.slot is not None)
        dxr = dx.reshape(ctx.shp)
        _publish(dxw, dx, dxr)
        return dxr, dw, None, None


def rms_norm_fork(x, w, eps, with_skip_alias: bool = False):
    """returns (x, rmsnorm(x)); use the returned x for the residual branch.  with_skip_alias: (x, rmsnorm(x), x_skip) -- a second
    alias of the stream for the long-range skip connection."""
    return _RMSNormFork.apply(x, w, eps, 2 if with_skip_alias else 1)


class _SwiGLU(torch.autograd.Function):
    """u = [u1 | u3] -> silu(u1) * u3   (attn.py:151)"""

    @staticmethod
    def forward(ctx, u):
        _dev(u)
        shp = u.shape
        F2 = shp[-1]
        um = u.reshape(-1, F2).contiguous()
        M, F = um.shape[0], F2 // 2
        g = torch.empty(M, F, device=u.device, dtype=torch.float32)
        L.check(L.load().gaot_swiglu_fwd(_p(um), M, F, _p(g), _stream()), "gaot_swiglu_fwd")
        ctx.save_for_backward(um)
        ctx.shp = shp
        return g.reshape(*shp[:-1], F)

    @staticmethod
    def backward(ctx, dg):
        (um,) = ctx.saved_tensors
        M, F2 = um.shape
        d = dg.reshape(M, F2 // 2).contiguous()
        du = torch.empty_like(um)
        L.check(L.load().gaot_swiglu_bwd(_p(um), _p(d), M, F2 // 2, _p(du), _stream()), "gaot_swiglu_bwd")
        return du.reshape(ctx.shp)


def swiglu(u):
    return _SwiGLU.apply(u)


class _SwiGLUFFN(torch.autograd.Function):
    """y = (silu(x w1^T) * (x w3^T)) w2^T (+ residual)   (attn.py:150-156) as three GEMMs forward / four backward:
    the gate is the epilogue of the [w1;w3] product (u is written once for the backward pass and never re-read) and the
    gate's gradient is the epilogue of dg = dy w2, so neither g's gradient nor a separate gate pass ever touches HBM."""

    @staticmethod
    def fusable(K: int, F: int) -> bool:
        return K % 32 == 0 and F % 4 == 0

    @staticmethod
    def forward(ctx, x, w1, w3, w2, residual, res_is_x):
        _dev(x, w1, w3, w2)
        shp = x.shape
        K = shp[-1]
        xm, lda = _rowmajor(x.reshape(-1, K))
        M, F, No = xm.shape[0], w1.shape[0], w2.shape[0]
        w13 = stacked_rows([w1, w3])
        if res_is_x:          # y = x + ffn(x): the stream's gradient is added in the epilogue of du @ w13 (backward)
            residual = x
        u = torch.empty(M, 2 * F, device=x.device, dtype=torch.float32)
        g = torch.empty(M, F, device=x.device, dtype=torch.float32)
        gemm(M, 2 * F, K, xm, lda, 1, w13, K, 1, g, F, act=L.ACT_SWIGLU, aux_out=u, ld_aux=2 * F, a_amax=_amax_get(x, xm))
        _publish(gemm.last_c_amax, g)
        epi = {}
        if residual is not None:
            res2, ldr = _rowmajor(residual.reshape(M, No))
            epi.update(residual=res2, ldr=ldr)
        y = linear_nt(g, w2, **epi)
        cw = gemm.last_c_amax
        ctx.amax = (_amax_get(xm, x), _amax_get(g))
        _publish(ctx.amax[0], x)
        ctx.save_for_backward(xm, u, g, w13, w2)
        s1, s3, s2 = _claim(w1), _claim(w3), _claim(w2)
        ctx.slots = (adjacent_rows([s1, s3]) if (s1 is not None and s3 is not None) else None, s2)
        ctx.meta = (shp, residual.shape if (residual is not None and not res_is_x) else None, bool(res_is_x))
        yr = y.reshape(*shp[:-1], No)
        _publish(cw, y, yr)
        return yr

    @staticmethod
    def backward(ctx, dy):
        xm, u, g, w13, w2 = ctx.saved_tensors
        shp, res_shape, res_is_x = ctx.meta
        M, F = g.shape
        No, K = w2.shape[0], xm.shape[1]
        d, ldd = _rowmajor(dy.reshape(M, No))
        need = ctx.needs_input_grad
        w2c, ldw2 = _rowmajor(w2)
        du = torch.empty(M, 2 * F, device=d.device, dtype=torch.float32)
        ax, ag = ctx.amax
        gemm(M, F, No, d, ldd, 1, w2c, ldw2, 0, du, 2 * F, act=L.ACT_SWIGLU_BWD, aux_in=u, ld_aux=2 * F, a_amax=_amax_get(d, dy))
        _publish(gemm.last_c_amax, du)
        slot13, slot2 = ctx.slots
        dw2 = matmul_tn(d, g, out=slot2.detach() if slot2 is not None else None, final=slot2 is not None, g_amax=_amax_get(d, dy), x_amax=ag) if need[3] else None
        dx = None
        if need[0]:
            dx2d = matmul_nn(du, w13, residual=d, ldr=ldd) if res_is_x else matmul_nn(du, w13)
            dx = dx2d.reshape(shp)
            _publish(gemm.last_c_amax, dx2d, dx)
        dw1 = dw3 = None
        if need[1] or need[2]:
            dw13 = matmul_tn(du, xm, out=slot13, final=slot13 is not None, g_amax=_amax_get(du), x_amax=ax)
            dw1, dw3 = dw13[:F], dw13[F:]
        dres = dy.reshape(res_shape) if (res_shape is not None and need[4]) else None
        return dx, dw1, dw3, dw2, dres, None


class _NormedSwiGLUFFN(torch.autograd.Function):
    """h = rmsnorm(x) * wn;  y = h + w2 (silu(w1 h) * w3 h): the FFN half of a transformer block (attn.py:229-233: the FFN's residual
    is its own NORMALISED input) as one node.  Forward: the three launches of rms_norm + _SwiGLUFFN.  Backward: dY w2^T with the gate's
    derivative in its epilogue, then du [w1; w3] as a split-K product whose K slabs are never reduced on their own: the norm-gradient
    kernel sums them (+ dY, the residual route) while it forms dx -- one launch instead of reduce + norm gradient."""

    @staticmethod
    def forward(ctx, x, wn, eps, w1, w3, w2):
        _dev(x, wn, w1, w3, w2)
        shp = x.shape
        K = shp[-1]
        xm = x.reshape(-1, K).contiguous()
        M, F = xm.shape[0], w1.shape[0]
        lib = L.load()
        h = torch.empty_like(xm)
        rstd = torch.empty(M, device=x.device, dtype=torch.float32)
        hw = _want_word(x.device)
        L.check(lib.gaot_rmsnorm_fwd(_p(xm), _p(wn), M, K, float(eps), _p(h), _p(rstd), _p(hw), _stream()), "gaot_rmsnorm_fwd")
        w13 = stacked_rows([w1, w3])
        u = torch.empty(M, 2 * F, device=x.device, dtype=torch.float32)
        g = torch.empty(M, F, device=x.device, dtype=torch.float32)
        gemm(M, 2 * F, K, h, K, 1, w13, K, 1, g, F, act=L.ACT_SWIGLU, aux_out=u, ld_aux=2 * F, a_amax=hw)
        gw = gemm.last_c_amax
        y = linear_nt(g, w2, residual=h, ldr=K, a_amax=gw)
        cw = gemm.last_c_amax
        ctx.amax = (hw, gw)
        ctx.save_for_backward(xm, wn, rstd, h, u, g, w13, w2)
        s1, s3, s2 = _claim(w1), _claim(w3), _claim(w2)
        ctx.slots = (adjacent_rows([s1, s3]) if (s1 is not None and s3 is not None) else None, s2, _claim(wn))
        ctx.shp = shp
        yr = y.reshape(shp)
        _publish(cw, y, yr)
        return yr

    @staticmethod
    def backward(ctx, dy):
        xm, wn, rstd, h, u, g, w13, w2 = ctx.saved_tensors
        M, F = g.shape
        K = xm.shape[1]
        lib = L.load()
        d, ldd = _rowmajor(dy.reshape(M, K))
        need = ctx.needs_input_grad
        w2c, ldw2 = _rowmajor(w2)
        du = torch.empty(M, 2 * F, device=d.device, dtype=torch.float32)
        hw, gw = ctx.amax
        dword = _amax_get(d, dy)
        gemm(M, F, K, d, ldd, 1, w2c, ldw2, 0, du, 2 * F, act=L.ACT_SWIGLU_BWD, aux_in=u, ld_aux=2 * F, a_amax=dword)
        duw = gemm.last_c_amax
        _publish(duw, du)
        slot13, slot2, slotn = ctx.slots
        dw2 = matmul_tn(d, g, out=slot2.detach() if slot2 is not None else None, final=slot2 is not None, g_amax=_amax_get(d, dy), x_amax=gw) if need[5] else None
        dx = dwn = None
        if need[0] or need[1]:
            dxm = torch.empty_like(xm)
            P = int(lib.gaot_rmsnorm_bwd_partials(M))
            part = torch.empty(P, K, device=xm.device, dtype=torch.float32)
            dxw = _want_word(xm.device)
            split = _split_for_narrow_output(M, K, 2 * F, lambda: weight_operand(w13, False)[1] is not None)
            dd = d if ldd == K else d.contiguous()
            if split > 1 and K in (256, 384, 512):
                w13c, ldw13 = _rowmajor(w13)
                ws, nz = gemm(M, K, 2 * F, du, 2 * F, 1, w13c, ldw13, 0, None, K, split_k=split, raw_slabs=True, a_amax=duw)
                L.check(lib.gaot_rmsnorm_bwd_slabs(_p(xm), _p(wn), _p(rstd), _p(ws), nz, M * K, _p(dd), None, None, M, K, _p(dxm), _p(part), _p(dxw),
                                                   _stream()), "gaot_rmsnorm_bwd_slabs")
            else:
                dh = matmul_nn(du, w13, residual=d, ldr=ldd)
                L.check(lib.gaot_rmsnorm_bwd(_p(xm), _p(wn), _p(rstd), _p(dh), None, None, M, K, _p(dxm), _p(part), _p(dxw), _stream()), "gaot_rmsnorm_bwd")
            if need[1]:
                dwn = colsum(part, out=slotn.detach() if slotn is not None else None, final=slotn is not None)
            dx = dxm.reshape(ctx.shp)
            _publish(dxw, dxm, dx)
        dw1 = dw3 = None
        if need[3] or need[4]:
            dw13 = matmul_tn(du, h, out=slot13, final=slot13 is not None, g_amax=_amax_get(du), x_amax=hw)
            dw1, dw3 = dw13[:F], dw13[F:]
        return dx, dwn, None, dw1, dw3, dw2


_NORMED_FFN = os.environ.get("GAOT_NORMED_FFN", "1") != "0"          # A/B switch (tools): 0 = rms_norm and swiglu_ffn as separate nodes


def normed_swiglu_ffn(x, wn, eps, w1, w3, w2):
    """rmsnorm(x) + SwiGLU feed-forward of it, the residual on the normalised stream (one autograd node; see _NormedSwiGLUFFN);
    None when the fused epilogues' shape rules do not hold (the caller composes rms_norm and swiglu_ffn instead)."""
    if _NORMED_FFN and x.is_cuda and _SwiGLUFFN.fusable(x.shape[-1], w1.shape[0]) and w2.shape[0] == x.shape[-1]:
        return _NormedSwiGLUFFN.apply(x, wn, eps, w1, w3, w2)
    return None


def swiglu_ffn(x, w1, w3, w2, residual=None):
    """SwiGLU feed-forward; falls back to the unfused HIP kernels when the fused epilogues' shape rules do not hold.
    `residual is x` (the reference block adds the FFN to its own normalised input, attn.py:231-232) is fused both ways."""
    if _SwiGLUFFN.fusable(x.shape[-1], w1.shape[0]):
        if residual is x and w2.shape[0] == x.shape[-1]:
            return _SwiGLUFFN.apply(x, w1, w3, w2, None, True)
        return _SwiGLUFFN.apply(x, w1, w3, w2, residual, False)
    return linear(swiglu(linear(x, torch.cat([w1, w3], dim=0))), w2, residual=residual)


class _Attention(torch.autograd.Function):
    """qkv [B,S,(H + 2 Hkv) * D] fused projection output -> softmax(q k^T / sqrt(D)) v  as [B,S,H*D].
    p_drop > 0: attention dropout (attn.py:110-114) -- the softmax output is multiplied by a keep mask / (1 - p) drawn from a
    counter-based hash of a device-resident seed word (`dropout_state`), regenerated by the backward."""

    @staticmethod
    def forward(ctx, qkv, H, Hkv, D, p_drop=0.0):
        _dev(qkv)
        qkv_in = qkv
        qkv = qkv.contiguous()
        B, S, W = qkv.shape
        assert W == (H + 2 * Hkv) * D
        o = torch.empty(B, S, H * D, device=qkv.device, dtype=torch.float32)
        lse = torch.empty(B, H, S, device=qkv.device, dtype=torch.float32)
        q = qkv.view(-1)
        kq = q[H * D:]
        vq = q[(H + Hkv) * D:]
        lib = L.load()
        seed = None
        if p_drop > 0.0:
            if not p_drop < 1.0:
                raise ValueError(f"attention dropout probability must be in [0, 1), got {p_drop}")
            seed = torch.empty(1, device=qkv.device, dtype=torch.int64)
            L.check(lib.gaot_attention_seed_next(_p(dropout_state(qkv.device)), 0, _p(seed), _stream()), "gaot_attention_seed_next")
            L.check(lib.gaot_attention_fwd_dropout(_p(q), _p(kq), _p(vq), W, W, W, B, S, H, Hkv, D, _p(o), H * D, _p(lse), float(p_drop),
                                                   _p(seed), _stream()), "gaot_attention_fwd_dropout")
            _LAST_DROPOUT_SEED[0] = seed
        else:
            pc, qw = _PIECES["attn"], None
            if pc == 3 and _F16_PIECES[0] and 32 <= D <= 64 and D % 4 == 0:       # fp16 pieces: the operands' magnitude word (published by the projection's epilogue)
                qw = amax_for(qkv.view(B * S, W), qkv, qkv_in)
                pc = 4
            # (a workspace where the shape takes the key-split forward: two halves of the keys per query block, joined by a second launch)
            nws = int(lib.gaot_attention_fwd_workspace(B, S, H, D)) if pc == 4 else 0
            ws = torch.empty(nws, device=qkv.device, dtype=torch.float32) if nws > 0 else None
            L.check(lib.gaot_attention_fwd_ws(_p(q), _p(kq), _p(vq), W, W, W, B, S, H, Hkv, D, _p(o), H * D, _p(lse), pc, _p(qw), _p(ws), _stream()),
                    "gaot_attention_fwd_ws")
            ctx.qkv_amax = qw
        ctx.save_for_backward(qkv, o, lse, seed if seed is not None else qkv.new_empty(0))
        ctx.dims = (B, S, H, Hkv, D, float(p_drop))
        if p_drop > 0.0:
            ctx.qkv_amax = None
        if p_drop == 0.0:      # every output row is a convex combination of V rows: max |o| <= max |v| <= max |qkv| (a bound is as good as the maximum)
            _publish(_amax_get(qkv, qkv_in), o)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse, seed = ctx.saved_tensors
        B, S, H, Hkv, D, p_drop = ctx.dims
        W = (H + 2 * Hkv) * D
        lib = L.load()
        do_in = do
        do = do.contiguous()
        dqkv = torch.empty_like(qkv)
        ws = torch.empty(int(lib.gaot_attention_bwd_workspace(B, S, H, D)), device=qkv.device, dtype=torch.float32)
        flat = qkv.view(-1)
        dflat = dqkv.view(-1)
        if Hkv == H:
            dk_t, dv_t, ldk, ldv = dflat[H * D:], dflat[2 * H * D:], W, W
        else:   # kernel emits per-query-head dK/dV; reduce over the group afterwards
            dk_full = torch.empty(B, S, H * D, device=qkv.device, dtype=torch.float32)
            dv_full = torch.empty_like(dk_full)
            dk_t, dv_t, ldk, ldv = dk_full.view(-1), dv_full.view(-1), H * D, H * D
        if p_drop > 0.0:
            L.check(lib.gaot_attention_bwd_dropout(_p(flat), _p(flat[H * D:]), _p(flat[(H + Hkv) * D:]), W, W, W, _p(o), _p(do), H * D,
                                                   _p(lse), B, S, H, Hkv, D, _p(dflat), _p(dk_t), _p(dv_t), W, ldk, ldv, _p(ws),
                                                   p_drop, _p(seed), _stream()), "gaot_attention_bwd_dropout")
        else:
            pc, qw, gw = _PIECES["attn"], ctx.qkv_amax, None
            if pc == 3 and _F16_PIECES[0] and 32 <= D <= 64 and D % 4 == 0:
                if qw is None:
                    qw = amax_for(qkv.view(B * S, W), qkv)
                gw = amax_for(do.view(B * S, H * D), do, do_in)
                pc = 4
            dw_ = _want_word(qkv.device) if Hkv == H else None          # dq | dk | dv feed the q|k|v input-gradient and weight-gradient products
            L.check(lib.gaot_attention_bwd(_p(flat), _p(flat[H * D:]), _p(flat[(H + Hkv) * D:]), W, W, W, _p(o), _p(do), H * D,
                                           _p(lse), B, S, H, Hkv, D, _p(dflat), _p(dk_t), _p(dv_t), W, ldk, ldv, _p(ws), pc, _p(qw), _p(gw), _p(dw_), _stream()),
                    "gaot_attention_bwd")
            _publish(dw_, dqkv)
        if Hkv != H:
            r = H // Hkv
            dqkv[..., H * D:(H + Hkv) * D] = dk_full.view(B, S, Hkv, r, D).sum(3).reshape(B, S, Hkv * D)
            dqkv[..., (H + Hkv) * D:] = dv_full.view(B, S, Hkv, r, D).sum(3).reshape(B, S, Hkv * D)
        return dqkv, None, None, None, None


_DROPOUT_STATE = {}                 # per device: int64 [2] = (seed, counter) -- advanced on the device by every dropout forward
_LAST_DROPOUT_SEED = [None]         # the seed word of the latest dropout forward (tests rebuild the mask from it)


def dropout_state(device) -> torch.Tensor:
    """the device-resident (seed, counter) pair of the attention-dropout generator; seeded from torch's default generator the first
    time it is used on a device (so `torch.manual_seed` before the first step makes runs repeatable), `seed_dropout` re-seeds."""
    key = torch.device(device).index or 0
    st = _DROPOUT_STATE.get(key)
    if st is None:
        st = torch.tensor([int(torch.randint(0, 2 ** 62, (1,)).item()), 0], dtype=torch.int64).to(device)
        _DROPOUT_STATE[key] = st
    return st


def seed_dropout(seed: int, device=None) -> None:
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    dropout_state(device).copy_(torch.tensor([int(seed), 0], dtype=torch.int64))


def attention(qkv, H, Hkv, D, dropout_p: float = 0.0):
    return _Attention.apply(qkv, H, Hkv, D, float(dropout_p))


class _Patchify(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sizes, P, inverse):
        _dev(x)
        x = x.contiguous()
        B = x.shape[0]
        dim = len(sizes)
        H, W = sizes[0], sizes[1]
        Dz = sizes[2] if dim == 3 else 0
        nodes = H * W * (Dz if dim == 3 else 1)
        pvol = P ** dim
        if inverse:
            Cc = x.shape[2] // pvol
            out = torch.empty(B, nodes, Cc, device=x.device, dtype=torch.float32)
        else:
            Cc = x.shape[2]
            out = torch.empty(B, nodes // pvol, pvol * Cc, device=x.device, dtype=torch.float32)
        ow = _want_word(x.device)
        L.check(L.load().gaot_patchify(_p(x), B, H, W, Dz, P, Cc, _p(out), int(inverse), _p(ow), _stream()), "gaot_patchify")
        ctx.args = (tuple(sizes), P, inverse)
        _publish(ow, out)
        return out

    @staticmethod
    def backward(ctx, g):
        sizes, P, inverse = ctx.args
        return _Patchify.apply(g, sizes, P, not inverse), None, None, None


class _Reshaped(torch.autograd.Function):
    """x.view(shape) -- and its gradient's view back -- that keep the magnitude word of what they alias (words travel on tensor OBJECTS, a
    plain reshape drops them and the next fp16-piece product would spend a launch on the maximum)"""

    @staticmethod
    def forward(ctx, x, shape):
        ctx.shape = x.shape
        out = x.view(shape)
        _publish(_amax_get(x), out)
        return out

    @staticmethod
    def backward(ctx, g):
        gi = g.contiguous().view(ctx.shape)
        _publish(_amax_get(g), gi)
        return gi, None


def reshaped(x, shape):
    return _Reshaped.apply(x, tuple(shape))


def patchify(x, sizes, P):
    return _Patchify.apply(x, tuple(sizes), P, False)


def unpatchify(x, sizes, P):
    return _Patchify.apply(x, tuple(sizes), P, True)
