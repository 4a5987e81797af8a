"""torch.autograd.Function wrappers over the C ABI (include/applecider_hip.h).

Every function here launches hand-written HIP kernels from libapplecider_hip.so on
the current HIP stream with raw device pointers.  There is no CPU or ATen fallback:
tensors must be fp32 GPU tensors.  PyTorch is used for memory, streams and autograd
bookkeeping only.
"""

from __future__ import annotations

import ctypes as C
import itertools
import weakref
from typing import Optional, Sequence

import torch
from torch.autograd import Function

from . import _lib
from ._lib import (ACT_CODES, ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, AC_GEMM_NN,
                   AC_GEMM_NT, AC_GEMM_TN, AdamSeg, GemmDesc, Mat, RowMap)

_MATH = _lib.MATH_F32


_MATH_NAMES = {"f32": _lib.MATH_F32, "bf16": _lib.MATH_BF16, "bf16x3": _lib.MATH_BF16X3, "f16": _lib.MATH_BF16}
_MODE = "f32"
_H16 = torch.bfloat16    # torch dtype that labels the 16-bit buffers of the loaded library's operand format
_MATH_EPOCH = 0          # bumped by set_math: 16-bit parameter mirrors made under another mode are stale
# modes whose logits are held to the path's parity bar (<= 1e-3 relative to the CPU oracle, identical
# argmax) by tests/test_gpu_parity_modes.py; plain bf16 is the fast, unqualified mode
QUALIFIED_MODES = ("f32", "bf16x3")


def set_math(mode: str):
    """Matrix-core arithmetic of every product on the path:
      'f32'     exact fp32 matrix cores (v_mfma_f32_32x32x2_f32), fp32 data flow
      'bf16x3'  split bf16: fp32 data flow, every product = 3 bf16 MFMAs on (hi, lo) operand halves,
                ~2^-16 relative per product (the qualified fast mode)
      'bf16'    bf16 MFMA inputs (one rounding to 8 mantissa bits per operand), bf16 operand copies and
                bf16-only hand-overs in HBM, fp32 accumulate / parameters / optimizer state
      'f16'     INFERENCE ONLY (BASELINE configs[4]): the data flow of 'bf16' on the fp16 build of the
                library (libapplecider_hip_f16.so: IEEE fp16 operands, v_mfma_f32_32x32x16_f16); backward
                through this mode raises"""
    global _MATH, _MODE, _H16, _MATH_EPOCH
    _MATH = _MATH_NAMES[mode]
    _lib.select("f16" if mode == "f16" else "bf16")
    _H16 = torch.float16 if mode == "f16" else torch.bfloat16
    if mode != _MODE:
        _MATH_EPOCH += 1
        _step_cache.clear()
    _MODE = mode


def get_math() -> str:
    return _MODE


ACT_GELU_FAST = 5   # AC_ACT_GELU_FAST (include/applecider_hip.h)


def _kact(code: int) -> int:
    """Activation code handed to the row / elementwise kernels: GELU means the rational-erf form in every math mode but
    the exact-fp32 one (the GEMM epilogues make the same choice from ac_gemm_desc.math), so that a step's forward and
    backward kernels differentiate one function."""
    return ACT_GELU_FAST if (code == ACT_GELU and _MATH != _lib.MATH_F32) else code


def _no_f16_backward():
    if _MODE == "f16":
        raise RuntimeError("math mode 'f16' is inference only (fp16 gradients underflow): train in "
                           "'bf16x3' / 'bf16' / 'f32'")


# --------------------------------------------------------------------------- helpers
def _lib_():
    return _lib.load()


_DEV_INDEX = None


def _stream():
    """Raw HIP stream handle of torch's current stream.  torch.cuda.current_stream() builds a Stream
    object through several Python layers (9 us, ~340 calls per training step); the raw getter is a
    single C call.  One process drives one GPU, so the device index is looked up once."""
    global _DEV_INDEX
    if _DEV_INDEX is None:
        _DEV_INDEX = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(_DEV_INDEX)


# ---------------------------------------------------------------------------------------------------------------
# Zeroed temporaries of the backward pass (split-K / atomic accumulation targets, strided-gather gradients): slices of
# ONE zero-filled buffer per stream and step instead of one fill launch each (32 ATen fills per step at B = 512).
# A slice is a view: it keeps its buffer alive, nothing is ever handed out twice, and a buffer that runs out is replaced
# by a fresh zero-filled one - there is no lifetime rule for callers to respect.  zero_pools_new_step() (called by
# FlatParameters.zero_grad) sizes the next buffer to what the last step took.  Not used while a hipGraph is being
# captured (a replay would find the slices dirty): plain torch.zeros there.
_ZERO_POOL = True
_ZERO_POOL_MIN, _ZERO_POOL_MAX = 4 << 20, 1 << 30


class _ZeroPool:
    __slots__ = ("buf", "off", "taken", "target")

    def __init__(self):
        self.buf, self.off, self.taken, self.target = None, 0, 0, _ZERO_POOL_MIN


_zero_pools: dict = {}
_ITEMSIZE = {torch.float32: 4, torch.int32: 4, torch.bfloat16: 2, torch.float16: 2, torch.uint8: 1, torch.int64: 8,
             torch.float64: 8, torch.int16: 2}


def _zeros(shape, device, dtype=torch.float32) -> torch.Tensor:
    shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)))
    n = 1
    for v in shape:
        n *= v
    nbytes = n * _ITEMSIZE.get(dtype, 4)
    if (not _ZERO_POOL or nbytes == 0 or nbytes > _ZERO_POOL_MAX or device.type != "cuda"
            or torch.cuda.is_current_stream_capturing()):
        return torch.zeros(shape, device=device, dtype=dtype)
    key = (device.index, _stream())
    pool = _zero_pools.get(key)
    if pool is None:
        pool = _zero_pools[key] = _ZeroPool()
    nb = (nbytes + 255) & ~255
    if pool.buf is None or pool.off + nb > pool.buf.numel():
        size = max(nb, min(pool.target, _ZERO_POOL_MAX))
        pool.buf, pool.off = torch.zeros(size, device=device, dtype=torch.uint8), 0
    out = pool.buf[pool.off:pool.off + nbytes].view(dtype).view(shape)
    pool.off += nb
    pool.taken += nb
    return out


def zero_pools_new_step():
    """Start a step with fresh, right-sized zero buffers (one fill per stream and step)."""
    for pool in _zero_pools.values():
        if pool.taken:
            pool.target = max(_ZERO_POOL_MIN, int(pool.taken * 1.05) + 4096)
        pool.buf, pool.off, pool.taken = None, 0, 0


def _is16only(t) -> bool:
    return getattr(t, "_ac16_only", False)


def _mark16only(t: torch.Tensor, t16: torch.Tensor) -> torch.Tensor:
    """`t` is an fp32 placeholder that was never written: its value exists as the bf16 tensor t16
    only (bf16 math mode, producer and consumer are kernels of this package that agreed on it).
    Any other use of the placeholder is caught by _chk."""
    t._ac16, t._ac16_only = t16, True
    return t


def _chk(t: torch.Tensor, name="tensor", allow16: bool = False):
    if not allow16 and _is16only(t):
        raise RuntimeError(f"{name}: this tensor's fp32 payload was not materialised (bf16-only hand-over "
                           "between two kernels); its consumer must read the bf16 side tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: applecider_amd kernels need a GPU tensor (no CPU fallback)")
    if _DEV_INDEX is not None and t.device.index != _DEV_INDEX:
        raise RuntimeError(f"{name}: tensor lives on cuda:{t.device.index} but this process launches on "
                           f"cuda:{_DEV_INDEX} (one process drives one GPU: call torch.cuda.set_device "
                           "before the first kernel)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _p(t: Optional[torch.Tensor], elem_off: int = 0):
    if t is None:
        return None
    return t.data_ptr() + elem_off * t.element_size()


def rowmap(s3=0, r1=0, r2=0, s1=0, s2=0) -> RowMap:
    return RowMap(int(r1), int(r2), int(s1), int(s2), int(s3))


def mat(ptr, s3=0, r1=0, r2=0, s1=0, s2=0, goff: Optional[torch.Tensor] = None) -> Mat:
    return Mat(ptr, rowmap(s3, r1, r2, s1, s2), _p(goff))


def gemm(mode, M, N, K, a: Mat, b: Mat, c: Mat, *, bias=None, act=ACT_NONE, pre_out=None,
         ld_pre=0, dact=ACT_NONE, aux=None, ld_aux=0, colscale=None, residual=None, ld_res=0,
         accumulate=0, split_k=1, alpha=1.0, force_simple=0, math=None, tile=0, c16=None, ld_c16=0,
         mask16=None, ld_mask16=0, drop_p=0.0, drop_seed=0, group=None, b_planes=None, ld_bpl=0, colsum=None):
    """group = [(a_ptr, b_ptr, c_ptr), ...] (device addresses): that many independent products of this shape in ONE
    launch (ac_gemm_grouped; a / b / c then only carry strides)."""
    d = GemmDesc()
    d.mode, d.math = mode, (_MATH if math is None else math)
    d.M, d.N, d.K = int(M), int(N), int(K)
    d.act, d.dact, d.accumulate = act, dact, accumulate
    d.split_k, d.force_simple, d.alpha, d.tile = int(split_k), force_simple, alpha, tile
    d.a, d.b, d.c = a, b, c
    d.bias = _p(bias)
    d.pre_out, d.ld_pre = _p(pre_out), ld_pre
    d.aux, d.ld_aux = _p(aux), ld_aux
    d.colscale = _p(colscale)
    d.residual, d.ld_res = _p(residual), ld_res
    d.c16, d.ld_c16 = _p(c16), ld_c16
    d.mask16, d.ld_mask16 = _p(mask16), ld_mask16
    d.drop_p, d.drop_seed = float(drop_p), int(drop_seed)
    d.drop_step = _p(_STEP_DEV) if drop_p > 0.0 else None
    d.colsum = _p(colsum)           # column sums of the stored values += (atomics): see ac_gemm_desc.colsum
    if b_planes is not None:      # split-bf16: B from cached (hi, lo) planes (a weight: split once per optimizer step)
        d.b_hi, d.b_lo, d.ld_bpl = _p(b_planes[0]), _p(b_planes[1]), int(ld_bpl)
    if group is not None:
        flat = [ptr for trip in group for ptr in trip]
        arr = (C.c_void_p * len(flat))(*flat)
        _lib.check(_lib_().ac_gemm_grouped(C.byref(d), len(group), arr, _stream()), "ac_gemm_grouped")
        return
    _lib.check(_lib_().ac_gemm(C.byref(d), _stream()), "ac_gemm")


def bf16_operands() -> bool:
    """bf16 mode stages GEMM operands as bf16 copies in HBM (cast once per call) when shapes allow."""
    return _MATH == _lib.MATH_BF16


def _ln_sub_shape(C: int) -> bool:
    """Row widths the sub-wave LayerNorm kernels cover (ac_rows.hip ln_sub_shape)."""
    if C % 4:
        return False
    V = C // 4
    for g in (64, 32, 16, 8):
        if V % g == 0:
            return V // g in (1, 2, 3, 6)
    return False


def _side16_alloc(shape, C: int, device):
    """bf16 side output of a producer kernel (bf16 math mode only)."""
    if not bf16_operands() or not _ln_sub_shape(C):
        return None
    return torch.empty(shape, device=device, dtype=_H16)


def cast16_act(x: torch.Tensor, K: int) -> torch.Tensor:
    """[rows, K] bf16 copy of an activation: the producer's side output when it left one."""
    side = getattr(x, "_ac16", None)
    if side is not None and side.shape == x.shape:
        return side.reshape(-1, K)
    return cast16(x.reshape(-1, K))


def cast16(t: torch.Tensor) -> torch.Tensor:
    y = torch.empty(t.shape, device=t.device, dtype=_H16)
    _lib.check(_lib_().ac_cast_bf16(_p(t), _p(y), t.numel(), _stream()), "ac_cast_bf16")
    return y


def cast16_into(src: torch.Tensor, dst: torch.Tensor):
    _lib.check(_lib_().ac_cast_bf16(_p(src), _p(dst), src.numel(), _stream()), "ac_cast_bf16")


def transpose_cast_segments(src, dst, segs, nseg, tiles):
    _lib.check(_lib_().ac_transpose_cast_segments(_p(src), _p(dst), _p(segs), nseg, tiles, _stream()),
               "ac_transpose_cast_segments")


def cast16_T(t2d: torch.Tensor) -> torch.Tensor:
    """[R, C] fp32 -> [C, R] bf16."""
    R, Cc = t2d.shape
    y = torch.empty(Cc, R, device=t2d.device, dtype=_H16)
    _lib.check(_lib_().ac_transpose_cast_bf16(_p(t2d), Cc, _p(y), R, R, Cc, _stream()),
               "ac_transpose_cast_bf16")
    return y


def split16(t: torch.Tensor):
    """(hi, lo) bf16 planes of an fp32 tensor: hi = bf16(t), lo = bf16(t - hi)  (math mode bf16x3)."""
    hi = torch.empty(t.shape, device=t.device, dtype=_H16)
    lo = torch.empty_like(hi)
    _lib.check(_lib_().ac_split_bf16(_p(t), _p(hi), _p(lo), t.numel(), _stream()), "ac_split_bf16")
    return hi, lo


def split16_T(t2d: torch.Tensor):
    """[R, C] fp32 -> (hi, lo) planes of the transpose [C, R]."""
    R, Cc = t2d.shape
    hi = torch.empty(Cc, R, device=t2d.device, dtype=_H16)
    lo = torch.empty_like(hi)
    _lib.check(_lib_().ac_transpose_split_bf16(_p(t2d), Cc, _p(hi), _p(lo), R, R, Cc, _stream()),
               "ac_transpose_split_bf16")
    return hi, lo


def split16_into(src: torch.Tensor, hi: torch.Tensor, lo: torch.Tensor):
    _lib.check(_lib_().ac_split_bf16(_p(src), _p(hi), _p(lo), src.numel(), _stream()), "ac_split_bf16")


def _plane_mirror(w):
    """(hi, lo) views of `w` in the planes of its flat parameter buffer (optim.FlatParameters.refresh_mirrors: one
    split launch per optimizer step for ALL parameters), or None when `w` does not live in such a buffer."""
    ent = _mirror_owner.get(id(w))
    if ent is None or ent[0]() is not w:
        return None
    fp = ent[1]()
    if fp is None or fp.flat is None or not fp.flat.is_cuda:
        return None
    off, n = fp.offsets[ent[2]], w.numel()
    if w.data_ptr() != fp.flat.data_ptr() + 4 * off or off % 8:
        return None
    if (fp.mirror_dirty or getattr(fp, "flat_hi", None) is None or fp.mirror_version != fp.flat._version
            or fp.mirror_pver[ent[2]] != w._version or getattr(fp, "mirror_epoch", -1) != _MATH_EPOCH):
        fp.refresh_mirrors()
    return fp.flat_hi[off:off + n].view(w.shape), fp.flat_lo[off:off + n].view(w.shape)


def split16_w(w):
    if x3_mode():
        m = _plane_mirror(w)
        if m is not None:
            return m
    return _cached(w, "s", split16)


def split16_wT(w2d):
    return _cached(w2d, "st", split16_T)


def _pad_rows_split(x, B, L, Cn, pad_lo, Lp):
    hi = torch.empty(B, Lp, Cn, device=x.device, dtype=_H16)
    lo = torch.empty_like(hi)
    _lib.check(_lib_().ac_pad_rows_split(_p(x), _p(hi), _p(lo), B, L, Cn, pad_lo, Lp, _stream()),
               "ac_pad_rows_split")
    return hi, lo


def x3_mode() -> bool:
    return _MATH == _lib.MATH_BF16X3


_X3_VARIANT = 0   # 0: v_mfma_f32_16x16x32 form (default) ; 2: the 32x32x16 form (A/B measurements, tests)


def conv_window_x3(a_planes, a_batch_stride, a_row_stride, a_col_off, row_base, B, L, Cw, k, w_planes,
                   w_row_stride, w_tap_stride, flip, N, c_ptr, ldc, bias, accumulate, tap_row_step=0, c_block=0,
                   c_block_stride=0) -> bool:
    """Split-bf16 conv through the LDS-resident-window kernel: three passes over (hi, lo) operand
    planes — hi*hi (+bias), lo*hi, hi*lo — accumulating in the fp32 output.  False when the shape is
    not covered (nothing has been written then)."""
    (ah, al), (wh, wl) = a_planes, w_planes
    short_seq = L < 128 and L >= 8 and (L & (L - 1)) == 0 and (B * L) % 256 == 0   # whole samples per 256-row tile
    if _CONVWIN and _CONVWIN_X3_FUSED and Cw % 64 == 0 and (L % 128 == 0 or short_seq) and N % 4 == 0:
        # one launch: both planes of the window in LDS (chunked over channels / taps), 3 MFMAs per pair
        d = _lib.ConvWinDesc()
        d.a, d.a_batch_stride, d.a_row_stride = _p(ah), a_batch_stride, a_row_stride
        d.a_col_off, d.row_base = a_col_off, row_base
        d.B, d.L, d.C, d.k = B, L, Cw, k
        d.w, d.w_row_stride, d.w_tap_stride = _p(wh), w_row_stride, w_tap_stride
        d.flip, d.N, d.c, d.ldc = int(flip), N, c_ptr, ldc
        d.bias, d.accumulate = _p(bias), int(accumulate)
        d.a_lo_off = (al.data_ptr() - ah.data_ptr()) // 2
        d.w_lo_off = (wl.data_ptr() - wh.data_ptr()) // 2
        d.variant = _X3_VARIANT
        d.tap_row_step, d.c_block, d.c_block_stride = int(tap_row_step), int(c_block), int(c_block_stride)
        rc = _lib_().ac_conv1d_window_x3(C.byref(d), _stream())
        if rc == 0:
            return True
        if rc != _lib.AC_EINVAL:
            _lib.check(rc, "ac_conv1d_window_x3")
    if tap_row_step or c_block:
        return False    # the Toeplitz / blocked-column form exists on the ring kernel only
    if not conv_window(ah, a_batch_stride, a_row_stride, a_col_off, row_base, B, L, Cw, k, wh, w_row_stride,
                       w_tap_stride, flip, N, c_ptr, ldc, bias, accumulate):
        return False
    for a_, w_ in ((al, wh), (ah, wl)):
        if not conv_window(a_, a_batch_stride, a_row_stride, a_col_off, row_base, B, L, Cw, k, w_,
                           w_row_stride, w_tap_stride, flip, N, c_ptr, ldc, None, True):
            raise RuntimeError("conv_window_x3: the kernel accepted the first pass only")
    return True


_step_cache: dict = {}


def clear_step_cache():
    """Drop cached bf16 copies of the weights (call whenever parameters change: optimizer step)."""
    _step_cache.clear()


def _cached(t: torch.Tensor, kind: str, fn):
    # Only long-lived Parameters are cached, and an entry is valid only for the very same tensor
    # object at the same version: a freed temporary's address can be handed to another tensor.
    if not isinstance(t, torch.nn.Parameter):
        return fn(t)
    key = (id(t), kind)
    hit = _step_cache.get(key)
    if hit is not None and hit[0]() is t and hit[1] == t._version and hit[2] == t.data_ptr():
        return hit[3]
    v = fn(t)
    _step_cache[key] = (weakref.ref(t), t._version, t.data_ptr(), v)
    return v


# bf16 mirrors of a flat parameter buffer (applecider_amd.optim.FlatParameters): the whole buffer is
# cast, and every 2-D weight transposed, by two launches per optimizer step; the bf16 operand of a
# product is then a view.  Keyed by parameter identity; validated against the flat buffer's version
# counter (any torch-level write to a parameter bumps it) and the optimizer's dirty flag.
_mirror_owner: dict = {}


def register_mirror(fp):
    for i, p in enumerate(fp.params):
        _mirror_owner[id(p)] = (weakref.ref(p), weakref.ref(fp), i)


def _mirror(w, transposed: bool):
    ent = _mirror_owner.get(id(w))
    if ent is None or ent[0]() is not w:
        return None
    fp = ent[1]()
    if fp is None or fp.flat is None or not fp.flat.is_cuda:
        return None
    off, n = fp.offsets[ent[2]], w.numel()
    if w.data_ptr() != fp.flat.data_ptr() + 4 * off:
        return None
    # stale when the optimizer stepped (raw-pointer update: dirty flag), when the flat buffer was
    # written through torch (broadcast), or when this parameter was (load_state_dict, init)
    if (fp.mirror_dirty or fp.flat16 is None or fp.mirror_version != fp.flat._version
            or fp.mirror_pver[ent[2]] != w._version or getattr(fp, "mirror_epoch", -1) != _MATH_EPOCH):
        fp.refresh_mirrors()
    if transposed:
        if w.dim() != 2:
            return None
        return fp.flatT16[off:off + n].view(w.shape[1], w.shape[0])
    return fp.flat16[off:off + n].view(w.shape)


# Side streams on which model branches run (models/applecider.py).  Whoever consumes gradients
# outside autograd's own stream bookkeeping (the DDP bucket launcher) must order itself after them.
_side_streams: list = []


def register_side_streams(streams):
    for st in streams:
        if all(st is not r for r in _side_streams):
            _side_streams.append(st)


def wait_side_streams():
    """Make the current stream wait for everything queued so far on the registered side streams."""
    if not _side_streams:
        return
    cur = torch.cuda.current_stream()
    for st in _side_streams:
        if st != cur and st.device == cur.device:
            cur.wait_stream(st)


def ensure_mirrors(fp):
    """Refresh the bf16 parameter mirrors now (on the current stream) if they are stale: callers that
    fan work out over several streams do this before the fork, so that no branch triggers the lazy
    refresh while another is already reading the mirrors."""
    if fp is None or fp.flat is None or not fp.flat.is_cuda or not (bf16_operands() or x3_mode()):
        return
    have = fp.flat_hi if x3_mode() else fp.flat16
    stale = (fp.mirror_dirty or have is None or fp.mirror_version != fp.flat._version
             or getattr(fp, "mirror_epoch", -1) != _MATH_EPOCH)
    if not stale:
        stale = any(v != p._version for v, p in zip(fp.mirror_pver, fp.params))
    if stale:
        fp.refresh_mirrors()


def cast16_w(w):
    """bf16 copy of a weight, computed once per optimizer step."""
    m = _mirror(w, False)
    return m if m is not None else _cached(w, "c", cast16)


def cast16_wT(w2d):
    m = _mirror(w2d, True)
    return m if m is not None else _cached(w2d, "t", cast16_T)


# Gradient sinks: when a parameter already owns a contiguous fp32 .grad (the flat gradient buffer of
# applecider_amd.optim), backward kernels accumulate straight into it (atomics / C += v) instead of
# materialising a temporary that autograd then adds — one pass less over every gradient.  Whoever
# needs to know that a parameter's gradient is complete (ddp.GradBuckets) registers a callback.
_grad_callbacks: list = []
_SINKS = True


def enable_grad_sinks(on: bool):
    global _SINKS
    _SINKS = bool(on)


def _sink(p):
    if not _SINKS or p is None or not isinstance(p, torch.nn.Parameter) or not p.requires_grad:
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or not g.is_cuda:
        return None
    return g


def _grad_written(p):
    for cb in _grad_callbacks:
        cb(p)


def _big(M, N, K) -> bool:
    return float(M) * N * K >= 262144.0


_CONVWIN = True
_TOEPLITZ_RING = True   # split-bf16: the Cin = 1 conv bank's forward on the ring window kernel (tests switch it)
_LN_PLANES = True   # split-bf16: LayerNorm backward emits the (hi, lo) planes of d(conv outputs) (tests switch it)


def _x3_bank_covered(B, L, Cin, Cout) -> bool:
    """Every gradient product of a SpectraNet conv bank runs on the plane-fed kernels (ac_conv1d_window_x3 for the
    input gradient, ac_conv1d_wgrad_bf16 for every k) — the shape rules of those entry points."""
    short = L < 128 and L >= 8 and (L & (L - 1)) == 0 and (B * L) % 256 == 0
    return (_CONVWIN and _CONVWIN_X3_FUSED and _WGRAD_WIN and Cout % 128 == 0 and Cin % 64 == 0 and L % 64 == 0
            and (L % 128 == 0 or short))


_CAT16 = True   # bf16 conv-bank output in front of the fused LayerNorm (bf16 math mode); tests switch it


class _View16:
    """Pointer + element offset of a bf16 tensor, for `_p()` (column block of the cat buffer)."""

    def __init__(self, t, elem_off):
        self.t, self.off = t, elem_off

    def data_ptr(self):
        return self.t.data_ptr() + 2 * self.off

    def element_size(self):
        return 2
_CONVWIN_X3_FUSED = True   # tests: False = three passes of the bf16 window kernel instead of the fused one
_CONVWIN_VARIANT = 0   # 1: keep N <= 64 products on the 4-wave window kernel (A/B tests)


def enable_conv_window(on: bool):
    global _CONVWIN
    _CONVWIN = bool(on)


def conv_window(a16, a_batch_stride, a_row_stride, a_col_off, row_base, B, L, Cw, k, w16, w_row_stride,
                w_tap_stride, flip, N, c_ptr, ldc, bias, accumulate, c16_ptr=None, ldc16=0) -> bool:
    """LDS-resident-window conv1d (ac_conv1d_window_bf16).  Returns False when the shape is not
    covered (the caller then uses the generic gather-GEMM)."""
    if not _CONVWIN or Cw not in (64, 128, 256) or L % 128:
        return False
    d = _lib.ConvWinDesc()
    d.a, d.a_batch_stride, d.a_row_stride = _p(a16), a_batch_stride, a_row_stride
    d.a_col_off, d.row_base = a_col_off, row_base
    d.B, d.L, d.C, d.k = B, L, Cw, k
    d.w, d.w_row_stride, d.w_tap_stride = _p(w16), w_row_stride, w_tap_stride
    d.flip, d.N, d.c, d.ldc = int(flip), N, c_ptr, ldc
    d.bias, d.accumulate = _p(bias), int(accumulate)
    d.variant = _CONVWIN_VARIANT
    d.c16, d.ldc16 = c16_ptr, ldc16
    rc = _lib_().ac_conv1d_window_bf16(C.byref(d), _stream())
    if rc == _lib.AC_EINVAL:
        return False
    _lib.check(rc, "ac_conv1d_window_bf16")
    return True


_WGRAD_WIN = True   # tests switch the LDS-window weight-gradient kernel off to compare with the TN product


def conv_wgrad(dy, dy_lo, dy_batch_stride, dy_row_stride, dy_row_base, dy_col_off, x, x_lo, x_batch_stride,
               x_row_stride, x_row_base, x_rows, B, L, Cout, Cin, k, dw, tap_row_step=0, dy_block=0, dy_block_stride=0,
               x_elem_off=0, ldw=None) -> bool:
    """dw[Cout, k*Cin] += conv weight gradient through the LDS-window kernel (ac_conv1d_wgrad_bf16).
    dy / x are bf16 tensors (dy_lo / x_lo: the lo planes in split-bf16 mode, else None).  False when
    the shape is not covered (the caller then runs the generic TN product)."""
    # bf16: short kernels (k < 7) keep the generic TN product (a chunk of 8 taps would be mostly empty);
    # split-bf16: every k, so that a covered conv bank never needs the fp32 padded input
    # 64 / L whole samples per K step; k < 7 would leave most of the 8-tap chunk empty at these small products
    short_seq = L in (16, 32) and (B * L) % 64 == 0 and _X3_VARIANT != 2 and k >= 7
    if not _WGRAD_WIN or (k < 7 and dy_lo is None) or (L % 64 and not short_seq) or Cout % 128 or Cin % 64:
        return False
    d = _lib.WgradDesc()
    d.dy, d.dy_batch_stride, d.dy_row_stride = _p(dy), dy_batch_stride, dy_row_stride
    d.dy_row_base, d.dy_col_off = dy_row_base, dy_col_off
    d.x, d.x_batch_stride, d.x_row_stride = _p(x, x_elem_off), x_batch_stride, x_row_stride
    d.x_row_base, d.x_rows = x_row_base, x_rows
    d.B, d.L, d.Cout, d.Cin, d.k = B, L, Cout, Cin, k
    d.tap_row_step, d.dy_block, d.dy_block_stride = int(tap_row_step), int(dy_block), int(dy_block_stride)
    tiles = (Cout // 128) * (Cin // 64) * (-(-k // 8))
    steps = (B * L) // 64
    d.split_k = max(1, min(steps // 16, -(-512 // tiles)))
    d.dw, d.ldw = _p(dw), (k * Cin if ldw is None else ldw)
    d.variant = _X3_VARIANT
    d.dy_lo_off = (dy_lo.data_ptr() - dy.data_ptr()) // 2 if dy_lo is not None else 0
    d.x_lo_off = (x_lo.data_ptr() - x.data_ptr()) // 2 if x_lo is not None else 0
    rc = _lib_().ac_conv1d_wgrad_bf16(C.byref(d), _stream())
    if rc == _lib.AC_EINVAL:
        return False
    _lib.check(rc, "ac_conv1d_wgrad_bf16")
    return True


_XCD_PIECES = True    # A/B: False keeps the split factors as they come


def _xcd_pieces(split: int) -> int:
    """Split-K factors from 6 up become multiples of 8: the kernels then deal the K pieces to the 8 XCDs and walk the
    tiles of one piece after another on one L2 (ac_gemm.hip map_workgroup) instead of spreading every piece's operand
    rows over all eight."""
    if not _XCD_PIECES or split < 6:
        return split
    return max(8, (split + 4) // 8 * 8)


def _split_for(m_out: int, n_out: int, k_red: int) -> int:
    """Split-K factor of a weight-gradient product (measured, tools/bench_split.py): long reductions
    get >= 64 K tiles per workgroup and up to 4 workgroups per CU; skinny outputs that cannot fill
    the chip that way are cut down to ~8 K tiles per workgroup, aiming at one workgroup per CU (each
    extra workgroup pays a fixed prologue + atomic epilogue of a few microseconds)."""
    tiles = max(1, -(-m_out // 128) * -(-n_out // 128))
    nkt = -(-k_red // 64)
    if x3_mode():
        # split-bf16 kernel: 32-deep K tiles and ~3x the matrix-core time per tile — workgroups amortise their
        # prologue over half as many K rows (measured, tools/archive/exp_tn_split.py: 1024x1536x8192 178 -> 139 us,
        # 512x128x66048 80 -> 72 us, 512x1028x262144 1417 -> 1333 us)
        split = max(1, min(2048 // tiles, nkt // 32))
        want = 512 if nkt >= 256 else 256      # short reductions: one workgroup per CU is enough
        if tiles * split < want:
            split = max(1, min(want // tiles, nkt // 8))
        return _xcd_pieces(split)
    split = max(1, min(1024 // tiles, nkt // 64))
    if tiles * split < 256:
        split = max(1, min(256 // tiles, nkt // 8))
    return _xcd_pieces(split)


def _tn_plan(m_out: int, n_out: int, k_red: int):
    """(tile, split_k) of a bf16 weight-gradient product.  Long reductions over a wide output use the
    128 x 256 tile (8 waves, one workgroup per CU): per FLOP it pulls 25 % fewer operand bytes through
    L2 than two 128 x 128 workgroups, which is what bounds these products (709 -> 860 TF on the stage-2
    conv gradient).  Its split is chosen to land on a whole number of 256-workgroup rounds."""
    nkt = -(-k_red // 64)
    if n_out >= 1024 and m_out >= 128 and nkt >= 512:
        tiles = -(-m_out // 128) * -(-n_out // 256)
        best = None
        for rounds in (1, 2, 4):
            split = max(1, (256 * rounds) // tiles)
            if nkt // split < 64:
                continue
            waste = (-(tiles * split) % 256) / 256.0 / max(1, -(-(tiles * split) // 256))
            if best is None or waste < best[0] - 1e-9:
                best = (waste, split)
        if best is not None:
            return 4, best[1]
    if m_out >= 1024 and n_out >= 1024 and nkt >= 64:
        # short reductions over a large output (last SpectraNet stage): 256 x 128 tiles, about one
        # workgroup per CU (207 vs 279 us at 1024 x 6656 x 8192, tools/bench_tn_late.py)
        tiles = -(-m_out // 256) * -(-n_out // 128)
        return 3, max(1, min(256 // tiles, nkt // 16))
    return 0, _split_for(m_out, n_out, k_red)


_seed_counter = itertools.count(1)
_seed_offset = 0


def set_seed_offset(rank: int):
    """Decorrelates the dropout streams of data-parallel replicas that share torch's global seed
    (ddp.init_from_env calls this with the rank)."""
    global _seed_offset
    _seed_offset = int(rank)


_STEP_DEV = None


def enable_device_step(device=None) -> torch.Tensor:
    """Creates the device-resident step counter of this process: from now on every dropout / mask launch
    is handed its address (the `step` argument of ac_dropout / ac_mha_* / ac_mpt_mask, ac_gemm_desc.drop_step)
    and mixes counter[0] into the seed it carries, so a captured hipGraph of a training step draws new
    masks at every replay.  Advance it once per step with step_advance() (GraphedTrainStep does).
    Returns the counter (int64[1]; the kernels read it as uint64)."""
    global _STEP_DEV
    if _STEP_DEV is None:
        _STEP_DEV = torch.zeros(1, dtype=torch.int64, device=device or torch.device("cuda", torch.cuda.current_device()))
    return _STEP_DEV


def disable_device_step():
    """Back to plain host seeds."""
    global _STEP_DEV
    _STEP_DEV = None


def step_advance(counter: Optional[torch.Tensor] = None):
    """counter[0] += 1 on the current stream (default: the registered dropout step counter)."""
    c = _STEP_DEV if counter is None else counter
    if c is None:
        raise RuntimeError("no device step counter: call hipops.enable_device_step() first")
    _lib.check(_lib_().ac_step_advance(_p(c), _stream()), "ac_step_advance")


def next_seed() -> int:
    """Fresh dropout seed derived from torch's global seed, the replica's rank and a per-process
    counter (deterministic per process)."""
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + next(_seed_counter) * 0xD1B54A32D192ED03
            + _seed_offset * 0xA24BAED4963EE407) & 0xFFFFFFFFFFFFFFFF


_table_cache: dict = {}


def _table(key, builder, device):
    k = (key, str(device))
    t = _table_cache.get(k)
    if t is None:
        t = torch.tensor(builder(), dtype=torch.int32, device=device)
        _table_cache[k] = t
    return t


def colsum(x2d_ptr, ld, rows, cols, device) -> torch.Tensor:
    out = torch.empty(cols, device=device, dtype=torch.float32)
    _lib.check(_lib_().ac_colsum(x2d_ptr, ld, _p(out), rows, cols, 0, _stream()), "ac_colsum")
    return out


_SMALL_GRID_SPLIT = True   # tests / A-B: False = one workgroup per output tile whatever the grid


def _small_grid_split(M: int, N: int, K: int) -> int:
    """Split-K factor of a product whose OUTPUT grid cannot fill the chip (fp32 data flow: f32 / bf16x3 modes).
    A 128 x 128 tile per workgroup gives the ConvNeXt stage-2 / stage-3 products 108 / 24 workgroups on 256 CUs, one
    wave per SIMD, and their 48 / 96 K tiles then run at the load -> split -> LDS -> barrier latency of a tile
    (1.2 us) instead of its matrix-core time (0.35 us): 22-90 TFLOP/s.  Cut over K into >= 8-tile pieces until the
    grid reaches ~2 workgroups per CU; the pieces meet in the output with fp32 atomics."""
    if not _SMALL_GRID_SPLIT or (M % 4) or (N % 4) or (K % 4) or not _big(M, N, K):
        return 1   # (below _big the library runs its scalar kernel, which has no split form)
    tiles = -(-M // 128) * -(-N // 128)
    nkt = K // 32
    if tiles > 128 or nkt < 16:
        return 1
    return _exact_split(nkt, max(1, min(512 // tiles, nkt // 8)))


def _exact_split(nkt: int, split: int) -> int:
    """The largest split <= `split` whose LAST piece still holds a K tile.  The library gives every piece
    ceil(nkt / split) tiles; a piece that starts beyond the last tile returns without storing, and in slab form
    (accumulate = 3) its slab would reach ac_splitk_reduce unwritten (M, N <= 128, K = 4160 -> nkt = 130,
    split 16 -> 9 tiles per piece -> piece 15 starts at tile 135).  ac_gemm refuses such a launch with AC_EINVAL."""
    per = -(-nkt // max(1, split))
    return -(-nkt // per)


_PLANE_B = True   # tests / A-B: False = nn.Linear products split their weight operand on the fly
_FUSE_DACT = True  # tests / A-B: False = activation backward + bias gradient as their own pass (ac_act_bwd_colsum)


def _wplanes(wparam, N: int, K: int) -> dict:
    """gemm() keywords that feed a weight [N, K] to a split-bf16 product as cached (hi, lo) planes (B operand of the
    NT forward product and of the NN input-gradient product)."""
    if not (_PLANE_B and x3_mode() and K % 8 == 0 and tuple(wparam.shape) == (N, K) and wparam.is_contiguous()
            and wparam.dtype == torch.float32):
        return {}
    return {"b_planes": split16_w(wparam), "ld_bpl": K}


# --------------------------------------------------------------------------- Linear
class _Linear(Function):
    """y = [drop](act(x @ w.T + b) [* colscale]) [+ residual]   (nn.Linear + fused epilogue).
    drop_p > 0 (fp32 data flow only): nn.Dropout on the product's output inside the GEMM epilogue; the
    backward applies the same mask in the pass that also forms act' and the bias gradient."""

    @staticmethod
    def forward(ctx, x, w, b, act, residual, colscale, drop_p=0.0):
        ctx.x16only = _is16only(x)   # producer handed the activation over in bf16 only
        x = _chk(x, "x", allow16=True)
        wparam = w
        w = _chk(w, "w")
        N, K = w.shape
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        need_grad = any(ctx.needs_input_grad)
        if residual is not None:
            residual = _chk(residual, "residual").reshape(M, N)
            if act not in (ACT_NONE, ACT_GELU, ACT_RELU):
                raise ValueError("residual epilogue supports act none/gelu/relu only")
        save_pre = need_grad and (act == ACT_GELU or (act == ACT_RELU and residual is not None)
                                  or colscale is not None)
        pre = torch.empty_like(y) if save_pre else None
        ctx.b16 = bf16_operands() and K % 8 == 0 and N % 8 == 0 and _big(M, N, K)
        if ctx.x16only and not ctx.b16:
            raise RuntimeError("bf16-only activation reached a product that is not on the bf16 path")
        x16 = None
        ctx.drop_p, ctx.drop_seed = float(drop_p), 0
        if drop_p > 0.0:
            if ctx.b16 or colscale is not None or N % 2 or act not in (ACT_NONE, ACT_RELU, ACT_GELU):
                raise ValueError("fused dropout: fp32 data flow, even width, act none / relu / gelu, no layer scale")
            ctx.drop_seed = next_seed()
            if act == ACT_RELU and pre is None and need_grad:
                pre = torch.empty_like(y)      # y is post-dropout: keep the pre-activation for act'
                save_pre = True
        if ctx.b16:
            x16, w16 = cast16_act(x, K), cast16_w(w)
            gemm(AC_GEMM_NT, M, N, K, mat(_p(x16), K), mat(_p(w16), K), mat(_p(y), N), bias=b,
                 act=act, pre_out=pre, ld_pre=N, colscale=colscale, residual=residual, ld_res=N,
                 math=_lib.MATH_BF16_IN)
        else:
            split = _small_grid_split(M, N, K) if (act == ACT_NONE and drop_p == 0.0 and N % 4 == 0) else 1
            if split > 1:
                # forward stays bit-reproducible: every K piece stores its partial tile into its own slab, one pass
                # sums the slabs in order and applies bias / layer scale / skip (which need the complete sum)
                part = torch.empty(split, M, N, device=x.device, dtype=torch.float32)
                gemm(AC_GEMM_NT, M, N, K, mat(_p(x2), K), mat(_p(w), K), mat(_p(part), N), accumulate=3,
                     split_k=split, **_wplanes(wparam, N, K))
                _lib.check(_lib_().ac_splitk_reduce(_p(part), split, _p(b), _p(pre), _p(colscale), _p(residual),
                                                    _p(y), M, N, _stream()), "ac_splitk_reduce")
            else:
                gemm(AC_GEMM_NT, M, N, K, mat(_p(x2), K), mat(_p(w), K), mat(_p(y), N), bias=b, act=act,
                     pre_out=pre, ld_pre=N, colscale=colscale, residual=residual, ld_res=N,
                     drop_p=ctx.drop_p, drop_seed=ctx.drop_seed, **_wplanes(wparam, N, K))
        if act == ACT_RELU and ROUTING_TAP is not None:
            _tap("relu", pre if pre is not None else y)     # (sign pattern = the gates; tests: ROUTING_TAP)
        ctx.act, ctx.has_res = act, residual is not None
        ctx.shape_x = x.shape
        ctx.has_b = b is not None
        aux = pre if save_pre else (y if act != ACT_NONE else None)
        ctx.save_for_backward(x16 if ctx.b16 else x2, w, aux, colscale)  # bf16 mode keeps the bf16 copy
        ctx.wp, ctx.bp, ctx.csp = w, b, colscale
        out = y.reshape(*x.shape[:-1], N)
        # Activation backward in the CONSUMER's product (fp32 data flow): when this output (GELU / ReLU, optional
        # dropout) feeds another Linear, that layer's input-gradient product applies act' (+ the dropout mask) in its
        # epilogue and leaves this layer's bias gradient in its sink (ac_gemm_desc.dact / colsum) - the separate
        # act' / mask / column-sum pass over the hidden tensor (ac_act_bwd_colsum: 3 reads + 1 write of it) is gone.
        # `link` is shared by the two autograd nodes; the consumer fills it in during backward.
        ctx.link = None
        if (_FUSE_DACT and need_grad and not ctx.b16 and act in (ACT_GELU, ACT_RELU) and colscale is None and residual is None
                and N % 4 == 0 and aux is not None):
            ctx.link = {"act": act, "aux": aux, "bias": b, "drop_p": ctx.drop_p, "drop_seed": ctx.drop_seed, "N": N, "M": M,
                        "done": False, "dx": None}
            out._ac_link = ctx.link
        src = getattr(x, "_ac_link", None)
        ctx.src = src if (src is not None and not ctx.b16 and src["N"] == K and src["M"] == M) else None
        return out

    @staticmethod
    def backward(ctx, dy):
        _no_f16_backward()
        x2, w, aux, colscale = ctx.saved_tensors
        N, K = w.shape
        M = x2.shape[0]
        dy2 = _chk(dy, "dy").reshape(M, N)
        dcs = None
        g = dy2
        if colscale is not None:
            # y = pre*gamma (+res): dpre = dy*gamma ; dgamma = sum_m dy*pre
            g = torch.empty_like(dy2)
            cssink = _sink(ctx.csp)       # the layer scale's gradient accumulates straight into its sink
            dcs = cssink if cssink is not None else torch.zeros_like(colscale)
            # the bias gradient of the Linear under the scale comes out of the same pass (into its sink)
            lsink = _sink(ctx.bp) if (not ctx.b16 and ctx.has_b and ctx.needs_input_grad[2]) else None
            _lib.check(_lib_().ac_layerscale_bwd(_p(dy2), _p(aux), _p(colscale), _p(g), None, _p(dcs),
                                                 _p(lsink), M, N, _stream()), "ac_layerscale_bwd")
            if lsink is not None:
                _grad_written(ctx.bp)
            if cssink is not None:
                dcs = None
                _grad_written(ctx.csp)
        bias_done = colscale is not None and lsink is not None
        db_tmp = None
        link = getattr(ctx, "link", None)
        fused_dact = link is not None and link["done"]
        if fused_dact:
            # the consumer's product already applied act' (+ mask) and summed this layer's bias gradient.  The gradient
            # must be THE tensor that product wrote: anything else (autograd summed several consumers) would carry the
            # activation backward only in part.
            # (same storage AND same version: autograd sums several consumers' gradients in place when it can)
            if link["dx"] != (dy.data_ptr(), dy._version, tuple(dy.shape)):
                raise RuntimeError("fused activation backward: the hidden gradient is not the tensor the consumer's product "
                                   "wrote (several consumers of an activated Linear output?) - set hipops._FUSE_DACT = False")
            link["done"], link["dx"] = False, None
            bias_done = True
        if colscale is None and (ctx.act != ACT_NONE or ctx.drop_p > 0.0) and not fused_dact:
            g = torch.empty_like(dy2)
            want_b = ctx.has_b and ctx.needs_input_grad[2]
            bsink = _sink(ctx.bp) if (not ctx.b16 and want_b and N % 2 == 0) else None
            if bsink is not None or ctx.drop_p > 0.0:
                # fp32 data flow: dropout mask, activation backward and the bias gradient in one pass
                tgt = bsink
                if tgt is None:
                    tgt = db_tmp = _zeros((N,), dy.device)
                _lib.check(_lib_().ac_act_bwd_colsum(_p(dy2), _p(aux) if ctx.act != ACT_NONE else None, _p(g), N,
                                                     _p(tgt), M, N, _kact(ctx.act), 1, ctx.drop_p, ctx.drop_seed,
                                                     _p(_STEP_DEV), _stream()), "ac_act_bwd_colsum")
                if bsink is not None:
                    _grad_written(ctx.bp)
                bias_done = True
            else:
                _lib.check(_lib_().ac_act_bwd(_p(dy2), _p(aux), _p(g), M * N, _kact(ctx.act), _stream()),
                           "ac_act_bwd")
        dx = dw = None
        db = db_tmp if (db_tmp is not None and ctx.has_b and ctx.needs_input_grad[2]) else None
        g16 = None
        if ctx.b16:
            bsink = _sink(ctx.bp) if (ctx.has_b and ctx.needs_input_grad[2]) else None
            if bsink is not None and N % 2 == 0:
                # one pass over g: the bf16 operand copy and the bias gradient (into its sink)
                g16 = torch.empty(M, N, device=dy.device, dtype=_H16)
                _lib.check(_lib_().ac_cast_bf16_colsum(_p(g), N, _p(g16), N, _p(bsink), M, N, 1,
                                                       _stream()), "ac_cast_bf16_colsum")
                _grad_written(ctx.bp)
                bias_done = True
            else:
                g16 = cast16(g)
        if ctx.needs_input_grad[0]:
            split_dx = 1 if ctx.b16 else _small_grid_split(M, K, N)
            dx = (_zeros((M, K), dy.device) if split_dx > 1 else torch.empty(M, K, device=dy.device, dtype=torch.float32))
            if ctx.b16 and ctx.x16only:
                # the producer of x reads its output gradient in bf16 (LayerNorm backward of the conv
                # bank): write only that; the fp32 tensor autograd carries is a placeholder
                wT16 = cast16_wT(ctx.wp if ctx.wp.shape == w.shape else w)
                dx16 = torch.empty(M, K, device=dy.device, dtype=_H16)
                gemm(AC_GEMM_NT, M, K, N, mat(_p(g16), N), mat(_p(wT16), N), mat(None, K), c16=dx16,
                     ld_c16=K, math=_lib.MATH_BF16_IN)
                dx = _mark16only(dx, dx16.reshape(ctx.shape_x))
            elif ctx.b16:  # dX = g @ W as NT against the k-contiguous copy W^T [K, N]
                wT16 = cast16_wT(ctx.wp if ctx.wp.shape == w.shape else w)
                gemm(AC_GEMM_NT, M, K, N, mat(_p(g16), N), mat(_p(wT16), N), mat(_p(dx), K),
                     math=_lib.MATH_BF16_IN)
            else:
                split = split_dx
                src = getattr(ctx, "src", None)
                bsrc = _sink(src["bias"]) if (src is not None and src["bias"] is not None) else None
                if split > 1:
                    gemm(AC_GEMM_NN, M, K, N, mat(_p(g), N), mat(_p(w), K), mat(_p(dx), K), accumulate=2,
                         split_k=split, **_wplanes(ctx.wp, N, K))
                elif (src is not None and K % 4 == 0 and (src["bias"] is None or bsrc is not None)
                      and src["aux"].is_contiguous()):
                    # d(hidden) with the producer's activation backward, dropout mask and bias gradient in the epilogue
                    gemm(AC_GEMM_NN, M, K, N, mat(_p(g), N), mat(_p(w), K), mat(_p(dx), K), dact=_kact(src["act"]),
                         aux=src["aux"], ld_aux=K, drop_p=src["drop_p"], drop_seed=src["drop_seed"], colsum=bsrc,
                         **_wplanes(ctx.wp, N, K))
                    if bsrc is not None:
                        _grad_written(src["bias"])
                    src["done"] = True
                else:
                    gemm(AC_GEMM_NN, M, K, N, mat(_p(g), N), mat(_p(w), K), mat(_p(dx), K), **_wplanes(ctx.wp, N, K))
            if not _is16only(dx):
                dx = dx.reshape(ctx.shape_x)
            elif dx.shape != ctx.shape_x:
                dx = _mark16only(dx.reshape(ctx.shape_x), dx._ac16)
            if getattr(ctx, "src", None) is not None and ctx.src["done"]:
                ctx.src["dx"] = (dx.data_ptr(), dx._version, tuple(dx.shape))   # what the producer's backward must receive
        if ctx.needs_input_grad[1]:
            wsink = _sink(ctx.wp)
            dw = wsink if wsink is not None else torch.zeros(N, K, device=dy.device, dtype=torch.float32)
            if ctx.b16:  # x2 is the bf16 copy saved by forward
                gemm(AC_GEMM_TN, N, K, M, mat(_p(g16), N), mat(_p(x2), K), mat(_p(dw), K),
                     accumulate=2, split_k=_split_for(N, K, M), math=_lib.MATH_BF16_IN)
            else:
                gemm(AC_GEMM_TN, N, K, M, mat(_p(g), N), mat(_p(x2), K), mat(_p(dw), K),
                     accumulate=2, split_k=_split_for(N, K, M))
            if wsink is not None:
                dw = None
                _grad_written(ctx.wp)
        if ctx.has_b and ctx.needs_input_grad[2] and not bias_done:
            bsink = _sink(ctx.bp)
            if bsink is not None:
                _lib.check(_lib_().ac_colsum(_p(g), N, _p(bsink), M, N, 1, _stream()), "ac_colsum")
                _grad_written(ctx.bp)
            else:
                db = colsum(_p(g), N, M, N, dy.device)
        dres = dy if ctx.has_res else None
        return dx, dw, db, None, dres, dcs, None


class _MLP(Function):
    """y = [drop2](fc2(drop1(act(fc1(x))))) [* colscale] [+ residual] with the hidden activation kept
    in bf16 only (bf16 math mode).  ConvNeXt block MLP (timm convnext_tiny, astrominn.py:12-17:
    act = GELU, colscale = layer-scale gamma, residual = block input) and the encoder feed-forward
    (Time2Vec.py:96-101: act = ReLU, dropout on the hidden and on the output).

    HBM traffic of the 4C-wide hidden per row, forward + backward: 88 B per hidden element pair
    instead of 192 (the fp32 activation, its cast pass, the separate act-backward pass and the fp32
    hidden gradient are gone).  GELU keeps the fp32 pre-activation for act'; ReLU (+dropout) needs
    nothing but the bf16 output itself, which is its own backward mask."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act, p1, p2, residual, colscale):
        x = _chk(x, "x")
        Hd, K = w1.shape
        N = w2.shape[0]
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
        dev = x.device
        if act not in (ACT_GELU, ACT_RELU):
            raise ValueError("fused MLP supports GELU and ReLU")
        seed1 = next_seed() if p1 > 0 else 0
        seed2 = next_seed() if p2 > 0 else 0
        x16, w1_16, w2_16 = cast16_act(x, K), cast16_w(w1), cast16_w(w2)
        need_grad = any(ctx.needs_input_grad)
        pre1 = torch.empty(M, Hd, device=dev, dtype=torch.float32) if (act == ACT_GELU and need_grad) else None
        g16 = torch.empty(M, Hd, device=dev, dtype=_H16)
        gemm(AC_GEMM_NT, M, Hd, K, mat(_p(x16), K), mat(_p(w1_16), K), mat(None, Hd), bias=b1, act=act,
             pre_out=pre1, ld_pre=Hd, c16=g16, ld_c16=Hd, drop_p=p1, drop_seed=seed1,
             math=_lib.MATH_BF16_IN)
        y = torch.empty(M, N, device=dev, dtype=torch.float32)
        pre2 = torch.empty_like(y) if (colscale is not None and need_grad) else None
        res2 = _chk(residual, "residual").reshape(M, N) if residual is not None else None
        gemm(AC_GEMM_NT, M, N, Hd, mat(_p(g16), Hd), mat(_p(w2_16), Hd), mat(_p(y), N), bias=b2,
             pre_out=pre2, ld_pre=N, colscale=colscale, drop_p=p2, drop_seed=seed2, residual=res2,
             ld_res=N, math=_lib.MATH_BF16_IN)
        ctx.save_for_backward(x16, w1, w2, pre1, g16, pre2, colscale)
        ctx.params = (w1, b1, w2, b2)
        ctx.act, ctx.p1, ctx.p2, ctx.seed1, ctx.seed2 = act, p1, p2, seed1, seed2
        ctx.shape_x, ctx.has_res = x.shape, residual is not None
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        _no_f16_backward()
        x16, w1, w2, pre1, g16, pre2, colscale = ctx.saved_tensors
        w1p, b1p, w2p, b2p = ctx.params
        Hd, K = w1.shape
        N = w2.shape[0]
        M = x16.shape[0]
        dev = dy.device
        dy2 = _chk(dy, "dy").reshape(M, N)
        lib, st = _lib_(), _stream()
        # ---- through dropout2 / layer scale to the gradient of fc2's linear output
        g2 = dy2
        if ctx.p2 > 0:
            g2 = torch.empty_like(dy2)
            _lib.check(lib.ac_dropout(_p(dy2), _p(g2), M * N, ctx.p2, ctx.seed2, 0, _p(_STEP_DEV), st), "ac_dropout")
        dcs = db2 = None
        b2sink = _sink(b2p) if (b2p is not None and ctx.needs_input_grad[4]) else None
        g2_16 = torch.empty(M, N, device=dev, dtype=_H16)
        if colscale is not None:
            # layer scale: bf16 gradient of the linear output, dgamma and the bias gradient in one pass
            dcs = torch.zeros_like(colscale)
            if b2p is not None and ctx.needs_input_grad[4] and b2sink is None:
                db2 = torch.zeros(N, device=dev, dtype=torch.float32)
            dbp = b2sink if b2sink is not None else db2
            _lib.check(lib.ac_layerscale_bwd(_p(g2), _p(pre2), _p(colscale), None, _p(g2_16), _p(dcs),
                                             _p(dbp), M, N, st), "ac_layerscale_bwd")
            if b2sink is not None:
                _grad_written(b2p)
        elif b2sink is not None and N % 2 == 0:
            _lib.check(lib.ac_cast_bf16_colsum(_p(g2), N, _p(g2_16), N, _p(b2sink), M, N, 1, st),
                       "ac_cast_bf16_colsum")
            _grad_written(b2p)
        else:
            g2_16 = cast16(g2)
            db2 = _bias_grad(b2p, g2, M, N, ctx.needs_input_grad[4])
        # ---- dW2 = g2^T @ hidden
        dw2 = _weight_grad(w2p, g2_16, N, g16, Hd, M, ctx.needs_input_grad[3])
        # ---- hidden gradient, bf16 only: (g2 @ W2) * act'(.) [* dropout1 mask]
        dh16 = torch.empty(M, Hd, device=dev, dtype=_H16)
        w2T16 = cast16_wT(w2p if w2p.shape == w2.shape else w2)
        if ctx.act == ACT_GELU:
            gemm(AC_GEMM_NT, M, Hd, N, mat(_p(g2_16), N), mat(_p(w2T16), N), mat(None, Hd),
                 dact=ACT_GELU, aux=pre1, ld_aux=Hd, c16=dh16, ld_c16=Hd, math=_lib.MATH_BF16_IN)
        else:
            gemm(AC_GEMM_NT, M, Hd, N, mat(_p(g2_16), N), mat(_p(w2T16), N), mat(None, Hd),
                 mask16=g16, ld_mask16=Hd, alpha=1.0 / (1.0 - ctx.p1) if ctx.p1 > 0 else 1.0,
                 c16=dh16, ld_c16=Hd, math=_lib.MATH_BF16_IN)
        db1 = None
        if b1p is not None and ctx.needs_input_grad[2]:
            sink = _sink(b1p)
            out = sink if sink is not None else torch.zeros(Hd, device=dev, dtype=torch.float32)
            _lib.check(lib.ac_colsum_bf16(_p(dh16), Hd, _p(out), M, Hd, 1, st), "ac_colsum_bf16")
            if sink is not None:
                _grad_written(b1p)
            else:
                db1 = out
        dw1 = _weight_grad(w1p, dh16, Hd, x16, K, M, ctx.needs_input_grad[1])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, device=dev, dtype=torch.float32)
            w1T16 = cast16_wT(w1p if w1p.shape == w1.shape else w1)
            gemm(AC_GEMM_NT, M, K, Hd, mat(_p(dh16), Hd), mat(_p(w1T16), Hd), mat(_p(dx), K),
                 math=_lib.MATH_BF16_IN)
            dx = dx.reshape(ctx.shape_x)
        dres = dy if ctx.has_res else None
        return dx, dw1, db1, dw2, db2, None, None, None, dres, dcs


def _bias_grad(bp, g, M, N, needed):
    """colsum of an fp32 gradient into the parameter's sink (or a fresh tensor)."""
    if bp is None or not needed:
        return None
    sink = _sink(bp)
    if sink is not None:
        _lib.check(_lib_().ac_colsum(_p(g), N, _p(sink), M, N, 1, _stream()), "ac_colsum")
        _grad_written(bp)
        return None
    return colsum(_p(g), N, M, N, g.device)


def _weight_grad(wp, g16, N, a16, K, M, needed):
    """dW[N, K] = g16[M, N]^T @ a16[M, K] (bf16 operands) into the parameter's sink."""
    if not needed:
        return None
    sink = _sink(wp)
    dw = sink if sink is not None else torch.zeros(N, K, device=g16.device, dtype=torch.float32)
    gemm(AC_GEMM_TN, N, K, M, mat(_p(g16), N), mat(_p(a16), K), mat(_p(dw), K), accumulate=2,
         split_k=_split_for(N, K, M), math=_lib.MATH_BF16_IN)
    if sink is not None:
        _grad_written(wp)
        return None
    return dw


def mlp(x, w1, b1, w2, b2, act, p1=0.0, p2=0.0, training=True, residual=None, colscale=None):
    """Two-layer perceptron.  bf16 math mode with 8-aligned widths runs the fused bf16-hidden path;
    otherwise the two Linear products with their separate dropout / residual kernels."""
    act_code = ACT_CODES[act] if not isinstance(act, int) else act
    if not training:
        p1 = p2 = 0.0
    Hd, K = w1.shape
    N = w2.shape[0]
    M = x.numel() // K
    if (bf16_operands() and K % 8 == 0 and Hd % 8 == 0 and N % 8 == 0 and _big(M, Hd, K)
            and _big(M, N, Hd) and act_code in (ACT_GELU, ACT_RELU) and x.is_cuda):
        return _MLP.apply(x, w1, b1, w2, b2, act_code, float(p1), float(p2), residual, colscale)
    h = linear(x, w1, b1, act=act_code, drop_p=p1)
    return linear(h, w2, b2, residual=residual, colscale=colscale, drop_p=p2)


def linear(x, w, b=None, act=None, residual=None, colscale=None, drop_p: float = 0.0):
    """drop_p > 0: dropout on the product's output (before the residual), fused into the product in the fp32
    data flow; the caller passes 0 in eval mode."""
    act_code = ACT_CODES[act] if not isinstance(act, int) else act
    if drop_p > 0.0 and (bf16_operands() or colscale is not None or w.shape[0] % 2 or not x.is_cuda):
        # not covered by the fused epilogue here: product, then the dropout kernel, then the residual
        y = dropout(_Linear.apply(x, w, b, act_code, None, colscale, 0.0), drop_p, True)
        return add(residual, y) if residual is not None else y
    return _Linear.apply(x, w, b, act_code, residual, colscale, float(drop_p))


# --------------------------------------------------------------------------- LayerNorm
class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, act):
        x = _chk(x, "x")
        Cn = x.shape[-1]
        rows = x.numel() // Cn
        y = torch.empty_like(x)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
        y16 = _side16_alloc(x.shape, Cn, x.device)
        _lib.check(_lib_().ac_layernorm_fwd(_p(x), Cn, _p(gamma), _p(beta), _p(y), Cn, _p(mean),
                                            _p(rstd), rows, Cn, eps, _kact(act), _p(y16), Cn, 0, _stream()),
                   "ac_layernorm_fwd")
        if y16 is not None:
            y._ac16 = y16   # the matrix product that consumes y reads this instead of casting
        ctx.act = act
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        ctx.gp, ctx.bp = gamma, beta
        return y

    @staticmethod
    def backward(ctx, dy):
        _no_f16_backward()
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        dy = _chk(dy, "dy")
        Cn = x.shape[-1]
        rows = x.numel() // Cn
        dx = torch.empty_like(x)
        gs, bs = _sink(ctx.gp), _sink(ctx.bp)
        both = gs is not None and bs is not None
        dg = gs if both else torch.zeros_like(gamma)
        db = bs if both else torch.zeros_like(beta)
        _lib.check(_lib_().ac_layernorm_bwd(_p(dy), Cn, _p(x), Cn, _p(mean), _p(rstd), _p(gamma),
                                            _p(beta), _p(dx), Cn, _p(dg), _p(db), None, rows, Cn,
                                            _kact(ctx.act), None, 0, 0, 0, 0, 0, 0, _stream()), "ac_layernorm_bwd")
        if both:
            _grad_written(ctx.gp)
            _grad_written(ctx.bp)
            return dx, None, None, None, None
        return dx, dg, db, None, None


def layer_norm(x, gamma, beta, eps=1e-5, act=None):
    return _LayerNorm.apply(x, gamma, beta, float(eps), ACT_CODES[act])


class _BatchNormAct(Function):
    """act(BatchNorm1d(x)) over the columns of x [rows, C] (rows = B*L positions of a channels-last
    sequence): SpectraNetBlock with use_ln=False (spectranet.py:21,33-37).  running_mean / running_var are
    updated in place in training mode, as nn.BatchNorm1d does."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, eps, momentum, act):
        x = _chk(x, "x")
        Cn = x.shape[-1]
        rows = x.numel() // Cn
        y = torch.empty_like(x)
        stats = torch.empty(4 * Cn, device=x.device, dtype=torch.float32)
        sums = torch.zeros(2 * Cn, device=x.device, dtype=torch.float32) if training else None
        _lib.check(_lib_().ac_batchnorm_fwd(_p(x), Cn, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                                            _p(y), Cn, _p(stats), _p(sums), rows, Cn, eps, momentum,
                                            int(training), act, _stream()), "ac_batchnorm_fwd")
        ctx.save_for_backward(x, gamma, stats)
        ctx.cfg = (rows, Cn, int(training), act)
        ctx.gp, ctx.bp = gamma, beta
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, stats = ctx.saved_tensors
        rows, Cn, training, act = ctx.cfg
        dy = _chk(dy, "dy")
        dx = torch.empty_like(x)
        gs, bs = _sink(ctx.gp), _sink(ctx.bp)
        both = gs is not None and bs is not None
        dg = gs if both else torch.zeros_like(gamma)
        db = bs if both else torch.zeros(Cn, device=x.device, dtype=torch.float32)
        work = torch.zeros(5 * Cn, device=x.device, dtype=torch.float32)
        _lib.check(_lib_().ac_batchnorm_bwd(_p(dy), Cn, _p(x), Cn, _p(gamma), _p(stats), _p(dx), Cn, _p(dg), _p(db),
                                            _p(work), rows, Cn, training, act, _stream()), "ac_batchnorm_bwd")
        if both:
            _grad_written(ctx.gp)
            _grad_written(ctx.bp)
            return dx, None, None, None, None, None, None, None, None
        return dx, dg, db, None, None, None, None, None, None


def batchnorm_act(x, gamma, beta, running_mean, running_var, training, eps=1e-5, momentum=0.1, act=None):
    return _BatchNormAct.apply(x, gamma, beta, running_mean, running_var, bool(training), float(eps),
                               float(momentum), ACT_CODES[act])


# --------------------------------------------------------------------------- elementwise
class _Dropout(Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        x = _chk(x, "x")
        y = torch.empty_like(x)
        _lib.check(_lib_().ac_dropout(_p(x), _p(y), x.numel(), p, seed, 0, _p(_STEP_DEV), _stream()), "ac_dropout")
        ctx.p, ctx.seed = p, seed
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _chk(dy, "dy")
        dx = torch.empty_like(dy)
        _lib.check(_lib_().ac_dropout(_p(dy), _p(dx), dy.numel(), ctx.p, ctx.seed, 0, _p(_STEP_DEV), _stream()),
                   "ac_dropout")
        return dx, None, None


def dropout(x, p: float, training: bool):
    if not training or p <= 0.0:
        return x
    return _Dropout.apply(x, float(p), next_seed())


class _Gate(Function):
    """out = a*g (+ s)."""

    @staticmethod
    def forward(ctx, a, g, s):
        a, g = _chk(a, "a"), _chk(g, "g")
        s = _chk(s, "s") if s is not None else None
        out = torch.empty_like(a)
        _lib.check(_lib_().ac_gate_fwd(_p(a), _p(g), _p(s), _p(out), a.numel(), _stream()),
                   "ac_gate_fwd")
        ctx.has_s = s is not None
        ctx.save_for_backward(a, g)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, g = ctx.saved_tensors
        dout = _chk(dout, "dout")
        da, dg = torch.empty_like(a), torch.empty_like(g)
        _lib.check(_lib_().ac_gate_bwd(_p(dout), _p(a), _p(g), _p(da), _p(dg), a.numel(),
                                       _stream()), "ac_gate_bwd")
        return da, dg, (dout if ctx.has_s else None)


def gate(a, g, s=None):
    return _Gate.apply(a, g, s)


def gather_cols(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """dst[:, j] = src[:, idx[j]] (no gradient: the source is input metadata)."""
    src = _chk(src, "src")
    rows, ld = src.shape
    n = idx.numel()
    dst = torch.empty(rows, n, device=src.device, dtype=torch.float32)
    _lib.check(_lib_().ac_gather_cols(_p(src), ld, _p(idx), _p(dst), n, rows, n, _stream()),
               "ac_gather_cols")
    return dst


class _Cat(Function):
    """torch.cat(tensors, dim=1) for 2-D fp32 tensors."""

    @staticmethod
    def forward(ctx, *ts):
        ts = [_chk(t, "cat input") for t in ts]
        rows = ts[0].shape[0]
        widths = [t.shape[1] for t in ts]
        total = sum(widths)
        out = torch.empty(rows, total, device=ts[0].device, dtype=torch.float32)
        off = 0
        for t, wd in zip(ts, widths):
            _lib.check(_lib_().ac_copy2d(_p(t), wd, _p(out, off), total, rows, wd, _stream()),
                       "ac_copy2d")
            off += wd
        ctx.widths = widths
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _chk(dout, "dout")
        rows, total = dout.shape
        grads, off = [], 0
        for wd in ctx.widths:
            g = torch.empty(rows, wd, device=dout.device, dtype=torch.float32)
            _lib.check(_lib_().ac_copy2d(_p(dout, off), total, _p(g), wd, rows, wd, _stream()),
                       "ac_copy2d")
            grads.append(g)
            off += wd
        return tuple(grads)


def cat_cols(tensors: Sequence[torch.Tensor]):
    return _Cat.apply(*tensors)


class _Add(Function):
    @staticmethod
    def forward(ctx, a, b, alpha):
        a, b = _chk(a, "a"), _chk(b, "b")
        y = torch.empty_like(a)
        _lib.check(_lib_().ac_add(_p(a), _p(b), _p(y), a.numel(), alpha, _stream()), "ac_add")
        ctx.alpha = alpha
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.alpha == 1.0:
            return dy, dy, None
        dy = _chk(dy, "dy")
        g = torch.empty_like(dy)
        # g = (dy + 0*dy) * alpha  ==  dy * alpha   (reuse the add kernel: a + a scaled by alpha/2)
        _lib.check(_lib_().ac_add(_p(dy), _p(dy), _p(g), dy.numel(), ctx.alpha * 0.5, _stream()),
                   "ac_add")
        return g, g, None


def add(a, b, alpha=1.0):
    return _Add.apply(a, b, float(alpha))


class _MoETop2(Function):
    @staticmethod
    def forward(ctx, scores, expert_out):
        scores, expert_out = _chk(scores, "scores"), _chk(expert_out, "expert_out")
        B, E = scores.shape
        Cn = expert_out.shape[-1]
        out = torch.empty(B, Cn, device=scores.device, dtype=torch.float32)
        sel = torch.empty(B, 2, device=scores.device, dtype=torch.int32)
        _lib.check(_lib_().ac_moe_top2_fwd(_p(scores), _p(expert_out), _p(out), _p(sel), B, E, Cn,
                                           _stream()), "ac_moe_top2_fwd")
        ctx.save_for_backward(scores, expert_out, sel)
        ctx.mark_non_differentiable(sel)
        return out, sel

    @staticmethod
    def backward(ctx, dout, _dsel):
        scores, eo, sel = ctx.saved_tensors
        dout = _chk(dout, "dout")
        B, E = scores.shape
        Cn = eo.shape[-1]
        ds, deo = torch.empty_like(scores), torch.empty_like(eo)
        _lib.check(_lib_().ac_moe_top2_bwd(_p(dout), _p(scores), _p(eo), _p(sel), _p(ds), _p(deo),
                                           B, E, Cn, _stream()), "ac_moe_top2_bwd")
        return ds, deo


# --------------------------------------------------------------------------- fused tower blocks
_TOWER_PARAMS = ("w1", "b1", "lnm_g", "lnm_b", "lng_g", "lng_b", "wm", "bm", "wg", "bg", "ws", "bs")
FUSED_TOWERS = True   # AstroMiNN's towers / experts through ac_tower_blocks_* (tests switch it off to compare)


class TowerPlan:
    """Static description of one grouped launch of ResidualTowerBlocks (ac_tower_blocks_fwd/bwd).

    blocks     list of dicts: n_in, hid, n_out, eps, cols (list of metadata columns or None), y_off (element
               offset of the block's output inside the output buffer: an int or a function of B), ldy
    out_shape  B -> shape of the output buffer ([B, width] concatenation, or [E, B, C] stacked experts)
    extra_off  column offset at which an `extra` [B, w] tensor is copied into a 2-D output (the image
               tower's features inside the concatenation), or None
    need_dx    whether dL/dx is wanted (experts: yes; towers on raw metadata: no)
    group_base first dropout group id (blocks of different launches must not share ids)"""

    def __init__(self, blocks, out_shape, extra_off=None, need_dx=False, group_base=0, p_drop=0.25):
        if not 1 <= len(blocks) <= 8:
            raise ValueError("1..8 blocks per grouped launch")
        self.blocks, self.out_shape, self.extra_off = blocks, out_shape, extra_off
        self.need_dx, self.group_base, self.p_drop = need_dx, group_base, float(p_drop)


def _tower_descs(plan, B, x, ldx, params, out, save, sv_off):
    arr = (_lib.TowerDesc * len(plan.blocks))()
    for i, (blk, d) in enumerate(zip(plan.blocks, arr)):
        pr = params[12 * i:12 * i + 12]
        d.x, d.ldx = _p(x), ldx
        for name, t in zip(_TOWER_PARAMS, pr):
            setattr(d, name, _p(t))
        d.n_in, d.hid, d.n_out, d.eps = blk["n_in"], blk["hid"], blk["n_out"], blk["eps"]
        d.group_id = plan.group_base + i
        cols = blk.get("cols")
        d.gather = 1 if cols is not None else 0
        if cols is not None:
            for j, c in enumerate(cols):
                d.cols[j] = c
        d.y, d.ldy = _p(out, _yoff(blk, B)), blk["ldy"]
        d.save = _p(save, sv_off[i])
    return arr


def _yoff(blk, B):
    off = blk["y_off"]
    return off(B) if callable(off) else off


class _TowerBlocks(Function):
    @staticmethod
    def forward(ctx, x, extra, plan, training, *params):
        x = _chk(x, "x")
        B, ldx = x.shape
        dev = x.device
        out = torch.empty(*plan.out_shape(B), device=dev, dtype=torch.float32)
        svw = [b["hid"] + 2 * b["n_out"] + 2 for b in plan.blocks]
        sv_off = [0]
        for w in svw:
            sv_off.append(sv_off[-1] + B * w)
        save = torch.empty(sv_off[-1], device=dev, dtype=torch.float32)
        drop = bool(training) and plan.p_drop > 0.0
        seed = next_seed() if drop else 0
        arr = _tower_descs(plan, B, x, ldx, params, out, save, sv_off)
        _lib.check(_lib_().ac_tower_blocks_fwd(arr, len(plan.blocks), B, plan.p_drop, int(drop), seed,
                                               _p(_STEP_DEV), _stream()), "ac_tower_blocks_fwd")
        if extra is not None:
            extra = _chk(extra, "extra")
            w = extra.shape[1]
            _lib.check(_lib_().ac_copy2d(_p(extra), w, _p(out, plan.extra_off), out.shape[1], B, w, _stream()),
                       "ac_copy2d")
            ctx.extra_w = w
        ctx.has_extra = extra is not None
        ctx.plan, ctx.drop, ctx.seed, ctx.sv_off = plan, drop, seed, sv_off
        ctx.params = params
        ctx.save_for_backward(x, save)
        return out

    @staticmethod
    def backward(ctx, dout):
        _no_f16_backward()
        x, save = ctx.saved_tensors
        plan, params = ctx.plan, ctx.params
        B, ldx = x.shape
        dev = x.device
        dout = _chk(dout, "dout")
        arr = _tower_descs(plan, B, x, ldx, params, dout, save, ctx.sv_off)   # y slot unused by backward
        dx = _zeros(x.shape, x.device) if (plan.need_dx and ctx.needs_input_grad[0]) else None
        grads, written = [], []
        for i, (blk, d) in enumerate(zip(plan.blocks, arr)):
            d.dy, d.lddy = _p(dout, _yoff(blk, B)), blk["ldy"]
            d.dx, d.lddx = _p(dx), ldx
            for name, t in zip(_TOWER_PARAMS, params[12 * i:12 * i + 12]):
                if t is None:
                    grads.append(None)
                    continue
                sink = _sink(t)
                g = sink if sink is not None else torch.zeros_like(t)
                setattr(d, "d" + name, _p(g))
                grads.append(None if sink is not None else g)
                if sink is not None:
                    written.append(t)
        _lib.check(_lib_().ac_tower_blocks_bwd(arr, len(plan.blocks), B, plan.p_drop, int(ctx.drop), ctx.seed,
                                               _p(_STEP_DEV), _stream()), "ac_tower_blocks_bwd")
        for t in written:
            _grad_written(t)
        dextra = None
        if ctx.has_extra and ctx.needs_input_grad[1]:
            dextra = dout[:, plan.extra_off:plan.extra_off + ctx.extra_w].contiguous()
        return (dx, dextra, None, None, *grads)


def tower_blocks(x, extra, plan: TowerPlan, training: bool, params):
    """Grouped fused ResidualTowerBlocks: x [B, ldx] -> the plan's output buffer.  `params` = 12 tensors per
    block in _TOWER_PARAMS order (ws, bs None for an identity skip)."""
    return _TowerBlocks.apply(x, extra, plan, training, *params)


def moe_top2(scores, expert_out):
    """scores [B,E], expert_out [E,B,C] -> (out [B,C], sel [B,2])."""
    return _MoETop2.apply(scores, expert_out)


class _Stack(Function):
    """torch.stack(tensors, 0) for equally shaped 2-D tensors ([B,C] x E -> [E,B,C])."""

    @staticmethod
    def forward(ctx, *ts):
        ts = [_chk(t, "stack input") for t in ts]
        B, Cn = ts[0].shape
        out = torch.empty(len(ts), B, Cn, device=ts[0].device, dtype=torch.float32)
        for e, t in enumerate(ts):
            _lib.check(_lib_().ac_copy2d(_p(t), Cn, _p(out, e * B * Cn), Cn, B, Cn, _stream()),
                       "ac_copy2d")
        return out

    @staticmethod
    def backward(ctx, dout):
        return tuple(dout.unbind(0))


def stack0(tensors):
    return _Stack.apply(*tensors)


class _L2Norm(Function):
    @staticmethod
    def forward(ctx, x):
        x = _chk(x, "x")
        rows, Cn = x.shape
        y = torch.empty_like(x)
        nrm = torch.empty(rows, device=x.device, dtype=torch.float32)
        _lib.check(_lib_().ac_l2norm_fwd(_p(x), _p(y), _p(nrm), rows, Cn, _stream()),
                   "ac_l2norm_fwd")
        ctx.save_for_backward(y, nrm)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, nrm = ctx.saved_tensors
        dy = _chk(dy, "dy")
        dx = torch.empty_like(dy)
        _lib.check(_lib_().ac_l2norm_bwd(_p(dy), _p(y), _p(nrm), _p(dx), y.shape[0], y.shape[1],
                                         _stream()), "ac_l2norm_bwd")
        return dx


def l2_normalize(x):
    return _L2Norm.apply(x)


def softmax_rows(x: torch.Tensor) -> torch.Tensor:
    """Inference-only softmax over the last dim (use_probabilities)."""
    x = _chk(x.detach(), "x")
    y = torch.empty_like(x)
    Cn = x.shape[-1]
    _lib.check(_lib_().ac_softmax_fwd(_p(x), _p(y), x.numel() // Cn, Cn, _stream()),
               "ac_softmax_fwd")
    return y


# --------------------------------------------------------------------------- image branch
def stem_patchify(img: torch.Tensor):
    """NCHW image -> ([B*OH*OW, 64] patches, OH, OW).  No gradient (network input)."""
    img = _chk(img, "image")
    B, Cin, H, W = img.shape
    OH, OW = (H - 4) // 4 + 1, (W - 4) // 4 + 1
    patches = torch.empty(B * OH * OW, 64, device=img.device, dtype=torch.float32)
    _lib.check(_lib_().ac_stem_patchify(_p(img), _p(patches), B, Cin, H, W, _stream()),
               "ac_stem_patchify")
    return patches, OH, OW


_DWCONV_VARIANT = 0   # 0: automatic (pipelined LDS-DMA kernels where they apply); 1: the round-2 kernels (A/B, tests)


class _DWConv7(Function):
    """x [B,H,W,C] NHWC, w [49,C], b [C]."""

    @staticmethod
    def forward(ctx, x, w, b, with_shortcut=False):
        x, w = _chk(x, "x"), _chk(w, "w")
        B, H, W_, Cn = x.shape
        y = torch.empty_like(x)
        if _DWCONV_VARIANT:
            _lib.check(_lib_().ac_dwconv7x7_fwd_v(_p(x), _p(w), _p(b), _p(y), B, H, W_, Cn, _DWCONV_VARIANT, _stream()),
                       "ac_dwconv7x7_fwd_v")
        else:
            _lib.check(_lib_().ac_dwconv7x7_fwd(_p(x), _p(w), _p(b), _p(y), B, H, W_, Cn, _stream()),
                       "ac_dwconv7x7_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        ctx.wp, ctx.bp = w, b
        if with_shortcut:
            # second output: the input itself, for the block's shortcut.  Both gradients then arrive in ONE backward
            # call and the kernel adds the shortcut's while it stores d x (autograd's own sum of the two was an
            # elementwise launch per block: 17 per step, 3 x the activation in traffic)
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, w = ctx.saved_tensors
        dy = _chk(dy, "dy")
        dres = _chk(dres, "dres") if dres is not None else None
        B, H, W_, Cn = x.shape
        dx = torch.empty_like(x)
        ws, bs = _sink(ctx.wp), _sink(ctx.bp)
        both = ws is not None and (bs is not None or not ctx.has_b)
        dw = ws if both else torch.zeros_like(w)
        db = bs if (both and ctx.has_b) else torch.zeros(Cn, device=x.device, dtype=torch.float32)
        _lib.check(_lib_().ac_dwconv7x7_bwd_res(_p(dy), _p(x), _p(w), _p(dres), _p(dx), _p(dw), _p(db), B, H, W_,
                                                Cn, _DWCONV_VARIANT, _stream()), "ac_dwconv7x7_bwd_res")
        if both:
            _grad_written(ctx.wp)
            if ctx.has_b:
                _grad_written(ctx.bp)
            return dx, None, None, None
        return dx, dw, (db if ctx.has_b else None), None


def dwconv7x7(x, w, b):
    return _DWConv7.apply(x, w, b)


def dwconv7x7_shortcut(x, w, b):
    """(dwconv(x), x): the second output feeds the ConvNeXt block's shortcut, so that the block input's two gradient
    contributions meet inside the depthwise backward kernel (ac_dwconv7x7_bwd_res)."""
    return _DWConv7.apply(x, w, b, True)


class _PatchConv2x2(Function):
    """2x2 stride-2 convolution on NHWC as a gather-GEMM.  w is [Cout, (ky,kx,ci)]."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w = _chk(x, "x"), _chk(w, "w")
        B, H, W_, Cn = x.shape
        Cout = w.shape[0]
        OH, OW = H // 2, W_ // 2
        M = B * OH * OW
        if Cn % 32:
            raise ValueError("PatchConv2x2 needs C % 32 == 0")
        goff = _table(("pc2", W_, Cn),
                      lambda: [(ky * W_ + kx) * Cn + cb * 32 for ky in range(2) for kx in range(2)
                               for cb in range(Cn // 32)], x.device)
        amap = dict(r1=OH * OW, s1=H * W_ * Cn, r2=OW, s2=2 * W_ * Cn, s3=2 * Cn)
        y = torch.empty(M, Cout, device=x.device, dtype=torch.float32)
        ctx.b16 = bf16_operands() and Cout % 8 == 0
        if ctx.b16:
            x16, w16 = cast16(x), cast16_w(w)  # keep both alive until the launch is enqueued
            gemm(AC_GEMM_NT, M, Cout, 4 * Cn, mat(_p(x16), goff=goff, **amap),
                 mat(_p(w16), 4 * Cn), mat(_p(y), Cout), bias=b, math=_lib.MATH_BF16_IN)
        else:
            gemm(AC_GEMM_NT, M, Cout, 4 * Cn, mat(_p(x), goff=goff, **amap), mat(_p(w), 4 * Cn),
                 mat(_p(y), Cout), bias=b)
        ctx.save_for_backward(x, w, goff)
        ctx.amap, ctx.has_b = amap, b is not None
        ctx.wp, ctx.bp = w, b
        return y.reshape(B, OH, OW, Cout)

    @staticmethod
    def backward(ctx, dy):
        x, w, goff = ctx.saved_tensors
        B, H, W_, Cn = x.shape
        Cout = w.shape[0]
        OH, OW = H // 2, W_ // 2
        M = B * OH * OW
        dy2 = _chk(dy, "dy").reshape(M, Cout)
        dx = dw = db = None
        dy16 = cast16(dy2) if ctx.b16 else None
        if ctx.needs_input_grad[0]:
            dx = _zeros(x.shape, x.device)  # border pixels dropped by the stride get zero gradient
            if ctx.b16:
                wT16 = cast16_wT(ctx.wp)
                gemm(AC_GEMM_NT, M, 4 * Cn, Cout, mat(_p(dy16), Cout), mat(_p(wT16), Cout),
                     mat(_p(dx), goff=goff, **ctx.amap), math=_lib.MATH_BF16_IN)
            else:
                gemm(AC_GEMM_NN, M, 4 * Cn, Cout, mat(_p(dy2), Cout), mat(_p(w), 4 * Cn),
                     mat(_p(dx), goff=goff, **ctx.amap))
        if ctx.needs_input_grad[1]:
            wsink = _sink(ctx.wp)
            dw = wsink if wsink is not None else torch.zeros_like(w)
            if ctx.b16:
                x16 = cast16(x)
                gemm(AC_GEMM_TN, Cout, 4 * Cn, M, mat(_p(dy16), Cout),
                     mat(_p(x16), goff=goff, **ctx.amap), mat(_p(dw), 4 * Cn), accumulate=2,
                     split_k=_split_for(Cout, 4 * Cn, M), math=_lib.MATH_BF16_IN)
            else:
                gemm(AC_GEMM_TN, Cout, 4 * Cn, M, mat(_p(dy2), Cout), mat(_p(x), goff=goff, **ctx.amap),
                     mat(_p(dw), 4 * Cn), accumulate=2, split_k=_split_for(Cout, 4 * Cn, M))
            if wsink is not None:
                dw = None
                _grad_written(ctx.wp)
        if ctx.has_b:
            bsink = _sink(ctx.bp)
            if bsink is not None:
                _lib.check(_lib_().ac_colsum(_p(dy2), Cout, _p(bsink), M, Cout, 1, _stream()), "ac_colsum")
                _grad_written(ctx.bp)
            else:
                db = colsum(_p(dy2), Cout, M, Cout, dy.device)
        return dx, dw, db


def patch_conv2x2(x, w, b):
    return _PatchConv2x2.apply(x, w, b)


class _AvgPool(Function):
    @staticmethod
    def forward(ctx, x):
        x = _chk(x, "x")
        B, HW, Cn = x.shape
        y = torch.empty(B, Cn, device=x.device, dtype=torch.float32)
        _lib.check(_lib_().ac_avgpool_fwd(_p(x), _p(y), B, HW, Cn, _stream()), "ac_avgpool_fwd")
        ctx.shape = (B, HW, Cn)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, HW, Cn = ctx.shape
        dy = _chk(dy, "dy")
        dx = torch.empty(B, HW, Cn, device=dy.device, dtype=torch.float32)
        _lib.check(_lib_().ac_avgpool_bwd(_p(dy), _p(dx), B, HW, Cn, _stream()), "ac_avgpool_bwd")
        return dx


def avgpool_tokens(x):
    """[B, HW, C] -> [B, C] mean over HW."""
    return _AvgPool.apply(x)


# --------------------------------------------------------------------------- spectra branch
# Tests: a list here receives ("pool4" | "global", index tensor) from every max-pool forward, in call order — the
# routing a gradient comparison needs (max-pooling makes the gradient discontinuous at near-tie windows; the oracle can
# be evaluated under THIS routing: oracle.functional.spectranet_forward(routing=)).  None = off.
ROUTING_TAP = None


def _tap(kind, idx):
    if ROUTING_TAP is not None:
        ROUTING_TAP.append((kind, idx))


class _MaxPool4(Function):
    @staticmethod
    def forward(ctx, x):
        x = _chk(x, "x")
        B, L, Cn = x.shape
        Lo = L // 4
        y = torch.empty(B, Lo, Cn, device=x.device, dtype=torch.float32)
        idx = torch.empty(B, Lo, Cn, device=x.device, dtype=torch.uint8)
        _lib.check(_lib_().ac_maxpool4_fwd(_p(x), _p(y), Lo * Cn, _p(idx), B, L, Cn, _stream()),
                   "ac_maxpool4_fwd")
        _tap("pool4", idx)
        ctx.save_for_backward(idx)
        ctx.shape = (B, L, Cn)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        B, L, Cn = ctx.shape
        dy = _chk(dy, "dy")
        dx = torch.empty(B, L, Cn, device=dy.device, dtype=torch.float32)
        _lib.check(_lib_().ac_maxpool4_bwd(_p(dy), (L // 4) * Cn, _p(idx), _p(dx), B, L, Cn,
                                           _stream()), "ac_maxpool4_bwd")
        return dx


def maxpool4(x):
    return _MaxPool4.apply(x)


class _GlobalMax(Function):
    @staticmethod
    def forward(ctx, x):
        x = _chk(x, "x")
        B, L, Cn = x.shape
        y = torch.empty(B, Cn, device=x.device, dtype=torch.float32)
        idx = torch.empty(B, Cn, device=x.device, dtype=torch.int32)
        _lib.check(_lib_().ac_globalmax_fwd(_p(x), _p(y), _p(idx), B, L, Cn, _stream()),
                   "ac_globalmax_fwd")
        _tap("global", idx)
        ctx.save_for_backward(idx)
        ctx.shape = (B, L, Cn)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        B, L, Cn = ctx.shape
        dy = _chk(dy, "dy")
        dx = torch.empty(B, L, Cn, device=dy.device, dtype=torch.float32)
        _lib.check(_lib_().ac_globalmax_bwd(_p(dy), _p(idx), _p(dx), B, L, Cn, _stream()),
                   "ac_globalmax_bwd")
        return dx


def global_max(x):
    return _GlobalMax.apply(x)


# --------------------------------------------------------------------------- frequency-domain conv products
# (module-level switches below are set by tests / A-B tools through the attribute; the product reads no environment variable)
_FFTCONV = True       # long-tap Conv1d products of the SpectraNet bank in the frequency domain (f32 / bf16x3 modes)
_FFT_MATH = None      # arithmetic of the per-frequency products: None = the math mode's (fp32 or split-bf16 matrix cores)
_FFT_FORCE = False    # tests: the transform form wherever the kernels cover the shape, whatever the cost rule says
_FFT_MARGIN = 1.0     # the direct form must cost this many times the transform form's estimate before it is replaced
_fft_tables: dict = {}


_FFT_OVERLAP_SAVE = True   # A/B: False = one sequence per sample only
_FFT_RADIX3 = True         # A/B: False = power-of-two transform lengths only
_FFT_SHARE = True          # A/B: False = no sharing of spectra inside a bank
# Transform lengths 9 * 2^m (1152 points for stage 2's k = 251, 288 for stage 3, 72 for stage 4's k = 11).  They were the
# first plans on which the co-residency problem of the transform kernels showed (profiles/r03_fft_coresidency_
# investigation.txt); with every transform workgroup alone on its CU they are as exact as the others.
_FFT_RADIX9 = True


def _fft_size(size):
    """(logm, radix3, N) of a transform size given as log2 N (power of two) or as (logm, radix3): N = (3 if radix3 else 1) << logm."""
    logm, r3 = (size, 0) if isinstance(size, int) else (int(size[0]), int(size[1]))
    return logm, r3, (1, 3, 9)[r3] << logm


def _fft_tw(size, device) -> torch.Tensor:
    """Twiddle table of ac_fft_*: M = 2^logm entries — level e (transform size M >> e) = exp(-2 pi i j / (M >> e)),
    j < M >> (e + 1), levels back to back, one pad — then, for N = 3 M, exp(-2 pi i t / N), t < 2 M.  Built once per
    size in fp64 on the host."""
    logm, r3, N = _fft_size(size)
    key = (logm, r3, str(device))
    t = _fft_tables.get(key)
    if t is None:
        M = 1 << logm
        parts = []
        for e in range(logm):
            n = M >> e
            ang = -2.0 * torch.pi * torch.arange(n // 2, dtype=torch.float64) / n
            parts.append(torch.stack([torch.cos(ang), torch.sin(ang)], dim=1))
        parts.append(torch.zeros(1, 2, dtype=torch.float64))
        if r3:
            ang = -2.0 * torch.pi * torch.arange(2 * N // 3, dtype=torch.float64) / N
            parts.append(torch.stack([torch.cos(ang), torch.sin(ang)], dim=1))
        t = _fft_tables[key] = torch.cat(parts).to(torch.float32).to(device)
    return t


def _fft_sizes():
    """Transform sizes the kernels cover, ascending: 2^m (32 ... 2048), 3 * 2^m (24 ... 1536), 9 * 2^m (72 ... 1152)."""
    return sorted([((m, 0), 1 << m) for m in range(5, 12)] + [((m, 1), 3 << m) for m in range(3, 10)] +
                  [((m, 2), 9 << m) for m in range(3, 8)], key=lambda t: (t[1], t[0][1]))


# relative cost per point of the >= 1024-point transform kernels in fft_plan (one workgroup per CU, 8 sequences)
_FFT_LONG_WEIGHT = 1.4


def fft_plan(L: int, k: int):
    """(logm, radix3, blocks, step) of the transform form of a 'same' Conv1d with k taps on L positions; the transform
    length is N = 2^logm or 3 * 2^logm (a convolution needs L + k // 2 points, rarely a power of two).
    blocks = 1: one sequence of N >= L + k // 2 points per sample (the circular wrap of the k // 2 positions past
    either end of the linear convolution falls outside the cropped output).  blocks > 1: overlap-save — windows of
    N <= 512 points advancing by step = N - k + 1 rows, at the price of a second transform of x in the backward pass
    (the weight gradient pairs the dy windows with x masked to each block's own rows).  None: not covered."""
    best = None
    need = L + k // 2
    for (logm, r3), N in _fft_sizes():
        if N >= need and N >= k and (_FFT_RADIX3 or not r3) and (_FFT_RADIX9 or r3 < 2):
            # (measured, tools/bench_fftconv.py: stage 2's k = 251 costs the same as one 2048-point sequence, 1.91 ms,
            # and as four 512-point windows, 1.96 ms — the window form needs a fifth transform)
            # (the >= 1024-point kernels hold one workgroup per CU and move 2.1 TB/s against 3.6 for the shorter ones:
            # stage 2's k = 31 costs 1.78 ms as one 1152-point sequence, 1.29 ms as three 384-point windows; with the
            # long transforms on 16 waves and the x / dx transforms shared with k = 251 the step is still 0.07 ms
            # slower with weight 1.0: tools/archive/gpu_r3_ze.sh)
            best = (N * (_FFT_LONG_WEIGHT if N >= 1024 else 1.0), logm, r3, 1, L)
            break
    if _FFT_OVERLAP_SAVE:
        for (logm, r3), N in _fft_sizes():
            if N > 512 or (r3 and not _FFT_RADIX3) or (r3 == 2 and not _FFT_RADIX9):
                continue
            V = N - k + 1
            if V < 0.6 * N:       # windows that overlap by more than 40 % transform most points twice
                continue
            blocks = -(-L // V)
            if blocks < 2:
                continue
            cost = blocks * N * 1.08
            if best is None or cost <= best[0]:      # (ties: the longer windows, fewer of them)
                best = (cost, logm, r3, blocks, V)
    return None if best is None else best[1:]


def fft_logn(L: int, k: int):
    """log2 of the transform length when the plan is a power of two (else None)."""
    plan = fft_plan(L, k)
    return None if plan is None or plan[1] else plan[0]


def fftconv_covered(B: int, L: int, Cin: int, Cout: int, k: int) -> bool:
    """Whether the three products of a 'same' Conv1d run in the frequency domain.  Shape limits of the kernels, and
    a cost rule from tools/bench_fftconv.py (profiles/r03_fftconv_kernels_*.txt): the four (five with overlap-save)
    transforms move rows + spectrum once each at ~3.6 TB/s (2.15 TB/s on the >= 1024-point kernels), the three
    per-frequency products their operands at ~3 TB/s or their FLOP at ~220 TFLOP/s, whatever k is; the direct window
    kernels sustain ~480 TFLOP/s of the 6 B L Cin Cout k FLOP (~300 on the 16-position stage)."""
    if not (_FFTCONV and _MATH in (_lib.MATH_F32, _lib.MATH_BF16X3) and _MODE != "f16" and Cin % 16 == 0
            and Cout % 16 == 0 and k % 2 == 1):
        return False
    plan = fft_plan(L, k)
    if plan is None:
        return False
    if _FFT_FORCE:
        return True
    logm, r3, blocks, _ = plan
    N = _fft_size((logm, r3))[2]
    pts, F = blocks * N, N // 2 + 1
    t_bytes = (2.0 * (Cin + Cout) + (Cin if blocks > 1 else 0)) * B * 4 * (pts + L)
    p_bytes = 3.0 * (B * pts * (Cin + Cout) * 4 + F * 4 * Cin * Cout * 4)
    p_flops = 3.0 * F * 2 * B * blocks * 4 * Cin * Cout       # the split-bf16 product kernel sustains ~220 TFLOP/s
    fft_ms = t_bytes / (2.15e9 if N >= 1024 else 3.6e9) + max(p_bytes / 3.0e9, p_flops / 220e9) + 0.05
    direct_ms = 6.0 * B * L * Cin * Cout * k / (300e9 if L <= 16 else 480e9)
    return direct_ms >= _FFT_MARGIN * fft_ms


def _fft_rows_desc(rows, rows_lo, elem_off, spec, bias, batch_stride, row_stride, col_off, B, L, Cn, size, blocks, step,
                   shift, n_lo, n_hi, accumulate):
    logm, r3, _ = _fft_size(size)
    d = _lib.FftRowsDesc()
    d.rows, d.rows_lo, d.spec = _p(rows, elem_off), _p(rows_lo, elem_off), _p(spec)
    d.tw, d.bias = _p(_fft_tw(size, spec.device)), _p(bias)
    d.batch_stride, d.row_stride, d.col_off = batch_stride, row_stride, col_off
    d.B, d.L, d.C, d.logn, d.radix3 = B, L, Cn, logm, r3
    d.blocks, d.block_step, d.shift = blocks, step, shift
    d.n_lo, d.n_hi, d.accumulate = n_lo, n_hi, int(accumulate)
    return d


def fft_rows_fwd(src, src_lo, elem_off, batch_stride, row_stride, col_off, B, L, Cn, shift, size, blocks=1, step=0,
                 n_lo=0, n_hi=0) -> torch.Tensor:
    """Spectrum [N/2 + 1, B * blocks, 2 Cn] of the channels-last rows src[b, l, col_off + c] (fp32, or (hi, lo) bf16
    planes); row l of block r sits at sequence index l - r * step + shift (inside [n_lo, n_hi) when given).
    size = log2 N, or (logm, radix3) for N = 3 * 2^logm."""
    N = _fft_size(size)[2]
    spec = torch.empty(N // 2 + 1, B * blocks, 2 * Cn, device=src.device, dtype=torch.float32)
    d = _fft_rows_desc(src, src_lo, elem_off, spec, None, batch_stride, row_stride, col_off, B, L, Cn, size, blocks, step,
                       shift, n_lo, n_hi, 0)
    _lib.check(_lib_().ac_fft_rows_fwd(C.byref(d), _stream()), "ac_fft_rows_fwd")
    return spec


def fft_rows_inv(spec, B, Cn, size, dst, batch_stride, row_stride, col_off, L, shift, bias, accumulate, blocks=1, step=0):
    d = _fft_rows_desc(dst, None, 0, spec, bias, batch_stride, row_stride, col_off, B, L, Cn, size, blocks, step, shift,
                       0, 0, accumulate)
    _lib.check(_lib_().ac_fft_rows_inv(C.byref(d), _stream()), "ac_fft_rows_inv")


def fft_taps_fwd(w, Cout, Cin, k, size) -> torch.Tensor:
    """Block spectrum [N/2 + 1, 2 Cout, 2 Cin] of the (flipped) taps w [Cout, k * Cin]."""
    logm, r3, N = _fft_size(size)
    hb = torch.empty(N // 2 + 1, 2 * Cout, 2 * Cin, device=w.device, dtype=torch.float32)
    _lib.check(_lib_().ac_fft_taps_fwd(_p(w), Cout, Cin, k, logm, r3, _p(_fft_tw(size, w.device)), _p(hb), _stream()),
               "ac_fft_taps_fwd")
    return hb


def fft_taps_inv(mp, Cout, Cin, k, size, dw):
    logm, r3, _ = _fft_size(size)
    _lib.check(_lib_().ac_fft_taps_inv(_p(mp), Cout, Cin, k, logm, r3, _p(_fft_tw(size, mp.device)), _p(dw), _stream()),
               "ac_fft_taps_inv")


def gemm_batched(mode, M, N, K, a: Mat, b: Mat, c: Mat, batch, bs_a, bs_b, bs_c, math=None, accumulate=0):
    d = GemmDesc()
    if math is None:
        math = _FFT_MATH if _FFT_MATH is not None else (_lib.MATH_BF16X3 if _MATH == _lib.MATH_BF16X3 else _lib.MATH_F32)
    d.mode, d.math = mode, math
    d.M, d.N, d.K = int(M), int(N), int(K)
    d.split_k, d.alpha, d.accumulate = 1, 1.0, int(accumulate)
    d.a, d.b, d.c = a, b, c
    _lib.check(_lib_().ac_gemm_batched(C.byref(d), int(batch), int(bs_a), int(bs_b), int(bs_c), _stream()),
               "ac_gemm_batched")


def fftconv_forward(x, w, B, L, Cin, Cout, k, out, ld_out, col_off, bias, xf=None):
    """out[b, l, col_off + co] = bias[co] + sum_{t, ci} x[b, l + t - k//2, ci] w[co, t, ci] through the frequency
    domain (spectranet.py:18-20,25).  x [B, L, Cin] fp32 contiguous, w [Cout, k * Cin] tap-major.  Returns what the
    gradient products reuse."""
    logm, r3, blocks, step = fft_plan(L, k)
    logn = (logm, r3)
    F, p = _fft_size(logn)[2] // 2 + 1, k // 2
    dev = x.device
    # one sequence per sample: x at shift 0, y read at k - 1 - p.  Overlap-save: windows from row r * step - p, y at k - 1
    if xf is None or blocks > 1:      # (xf: the spectrum another convolution of the bank made with the same plan)
        xf = fft_rows_fwd(x, None, 0, L * Cin, Cin, 0, B, L, Cin, 0 if blocks == 1 else p, logn, blocks, step)
    hb = fft_taps_fwd(w, Cout, Cin, k, logn)
    Bb = B * blocks
    yf = torch.empty(F, Bb, 2 * Cout, device=dev, dtype=torch.float32)
    gemm_batched(AC_GEMM_NT, Bb, 2 * Cout, 2 * Cin, mat(_p(xf), 2 * Cin), mat(_p(hb), 2 * Cin), mat(_p(yf), 2 * Cout),
                 F, Bb * 2 * Cin, 4 * Cout * Cin, Bb * 2 * Cout)
    fft_rows_inv(yf, B, Cout, logn, out, L * ld_out, ld_out, col_off, L, k - 1 - p if blocks == 1 else k - 1, bias, False,
                 blocks, step)
    # the weight gradient pairs the spectrum of dy with that of x: the same one when a sample is one sequence; with
    # overlap-save x masked to each block's own rows, transformed in the backward pass from x itself
    return (xf if blocks == 1 else x), hb, (logn, blocks, step)


def fftconv_backward(saved, dy, dy_lo, dy_elem_off, dy_batch_stride, dy_row_stride, dy_col_off, B, L, Cin, Cout, k,
                     dx, dx_accumulate, dw, dx_group=None):
    """Gradient products of fftconv_forward: dx [B, L, Cin] (nullable; stored or accumulated) and dw [Cout, k * Cin]
    (accumulated into).  dy: fp32 rows, or (dy, dy_lo) bf16 planes."""
    xs, hb, (logn, blocks, step) = saved
    F, p = _fft_size(logn)[2] // 2 + 1, k // 2
    dev, Bb = hb.device, B * blocks
    gf = fft_rows_fwd(dy, dy_lo, dy_elem_off, dy_batch_stride, dy_row_stride, dy_col_off, B, L, Cout,
                      k - 1 - p if blocks == 1 else p, logn, blocks, step)
    if dx is not None and dx_group is not None and blocks == 1:
        # convolutions of one bank with the same plan: their input-gradient spectra add up in ONE buffer and are
        # transformed back once (the caller's fftconv_finish_dx)
        first = dx_group.get("dxf") is None
        if first:
            dx_group["dxf"], dx_group["size"] = torch.empty(F, Bb, 2 * Cin, device=dev, dtype=torch.float32), logn
        gemm_batched(AC_GEMM_NN, Bb, 2 * Cin, 2 * Cout, mat(_p(gf), 2 * Cout), mat(_p(hb), 2 * Cin),
                     mat(_p(dx_group["dxf"]), 2 * Cin), F, Bb * 2 * Cout, 4 * Cout * Cin, Bb * 2 * Cin,
                     accumulate=0 if first else 1)
    elif dx is not None:
        dxf = torch.empty(F, Bb, 2 * Cin, device=dev, dtype=torch.float32)
        gemm_batched(AC_GEMM_NN, Bb, 2 * Cin, 2 * Cout, mat(_p(gf), 2 * Cout), mat(_p(hb), 2 * Cin), mat(_p(dxf), 2 * Cin),
                     F, Bb * 2 * Cout, 4 * Cout * Cin, Bb * 2 * Cin)
        fft_rows_inv(dxf, B, Cin, logn, dx, L * Cin, Cin, 0, L, 0, None, dx_accumulate, blocks, step)
    if dw is not None:
        xf = xs if blocks == 1 else fft_rows_fwd(xs, None, 0, L * Cin, Cin, 0, B, L, Cin, 0, logn, blocks, step, 0, step)
        mp = torch.empty(F, 2 * Cout, 2 * Cin, device=dev, dtype=torch.float32)
        gemm_batched(AC_GEMM_TN, 2 * Cout, 2 * Cin, Bb, mat(_p(gf), 2 * Cout), mat(_p(xf), 2 * Cin), mat(_p(mp), 2 * Cin),
                     F, Bb * 2 * Cout, Bb * 2 * Cin, 4 * Cout * Cin)
        fft_taps_inv(mp, Cout, Cin, k, logn, dw)


def fftconv_finish_dx(dx_group, dx, dx_accumulate, B, L, Cin):
    """The one inverse transform of a bank's summed input-gradient spectra (see fftconv_backward)."""
    fft_rows_inv(dx_group["dxf"], B, Cin, dx_group["size"], dx, L * Cin, Cin, 0, L, 0, None, dx_accumulate)


def _pad_rows(x, B, L, Cn, pad_lo, Lp):
    y = torch.empty(B, Lp, Cn, device=x.device, dtype=torch.float32)
    _lib.check(_lib_().ac_pad_rows(_p(x), _p(y), B, L, Cn, pad_lo, Lp, _stream()), "ac_pad_rows")
    return y


def _pad_rows16(x, B, L, Cn, pad_lo, Lp):
    y = torch.empty(B, Lp, Cn, device=x.device, dtype=_H16)
    _lib.check(_lib_().ac_pad_rows_bf16(_p(x), _p(y), B, L, Cn, pad_lo, Lp, _stream()),
               "ac_pad_rows_bf16")
    return y


class _ConvGroup1d(Function):
    """The parallel 'same' Conv1d bank of a SpectraNetBlock (spectranet.py:18-20,25):
    ycat[b, l, j*Cout + co] = bias_j[co] + sum_{t,ci} x[b, l + t - k_j//2, ci] * w_j[co, t, ci].

    x is [B, L, Cin]; each w_j is stored tap-major as [Cout, k_j*Cin].  Implicit GEMM over
    a zero-padded copy of x: the im2col row of (b, l) is a contiguous k_j*Cin slice, so A is
    just an overlapping-row view (row stride Cin).  For Cin == 1 the rows would be 4-byte
    shifted; there the 8 residues l%8 become 8 shifted copies of the filter instead
    (Toeplitz trick) so that A rows are 32-byte strided.
    With ln_gamma/ln_beta the LayerNorm over the 3*Cout channels + GELU of spectranet.py:31-35 is
    applied in the same autograd node, and its backward kernel also emits the column sums of
    d(ycat) — the three bias gradients — so no separate pass over the [B*L, 3*Cout] gradient.
    args: x, ksizes(tuple), ln_gamma, ln_beta, ln_eps, out16_only, w0, b0, w1, b1, ...
    """

    @staticmethod
    def forward(ctx, x, ksizes, ln_gamma, ln_beta, ln_eps, out16_only, pw_w, pw_b, *wb):
        x = _chk(x, "x")
        B, L, Cin = x.shape
        ws = [_chk(w, "w") for w in wb[0::2]]
        bs = list(wb[1::2])
        nconv = len(ksizes)
        Cout = ws[0].shape[0]
        Ncat = nconv * Cout
        if Cout % 32:
            raise ValueError("ConvGroup1d needs Cout % 32 == 0")
        for k in ksizes:
            if k % 2 == 0:
                raise ValueError("ConvGroup1d supports odd kernel sizes ('same' padding k//2)")
        Pmax = max(ksizes) // 2
        dev = x.device
        ctx.ksizes, ctx.dims = tuple(ksizes), (B, L, Cin, Cout, Pmax)
        ctx.fft = {}       # conv index -> (spectrum of x, block spectrum of the taps, logn) of the frequency-domain form
        ctx.has_b = [b is not None for b in bs]
        b16 = ctx.b16 = bf16_operands()
        mth = _lib.MATH_BF16_IN if b16 else None
        # bf16 mode with the fused LayerNorm: the concatenated conv outputs exist in bf16 only — they
        # are written once by the conv epilogues and read by LayerNorm forward and backward (4.7 GB of
        # HBM traffic per step at the benchmark shape when kept in fp32)
        cat16 = ctx.cat16 = bool(b16 and ln_gamma is not None and _ln_sub_shape(Ncat) and _CAT16)
        ycat = torch.empty(B, L, Ncat, device=dev, dtype=_H16 if cat16 else torch.float32)
        if Cin == 1:
            if L % 8:
                raise ValueError("Cin == 1 path needs L % 8 == 0")
            al = 8 if b16 else 4  # 16-byte granules in operand elements
            Lp = (L + 2 * Pmax + 48 + 7) // 8 * 8
            Lq = L // 8
            # split-bf16 mode: the Toeplitz products run on the ring window kernel (weights by LDS-DMA, fragments
            # prefetched across the barrier) over (hi, lo) planes of the padded flux: "taps" of 64 window elements,
            # window rows 8 elements apart (ac_convwin_desc.tap_row_step = 8) — 296 -> ~530 TFLOP/s on the k = 1021
            # product.  K is padded to 64, so the zero tail of the padded flux grows by up to 63 + 7 elements.
            toep = bool(not b16 and x3_mode() and _CONVWIN and _CONVWIN_X3_FUSED and _TOEPLITZ_RING and Cout % 8 == 0
                        and (Lq % 256 == 0 or (Lq < 128 and Lq >= 8 and (Lq & (Lq - 1)) == 0 and (B * Lq) % 256 == 0)))
            if toep:
                kp_max = max((k + 7 + (Pmax - k // 2) % 8 + 63) // 64 * 64 for k in ksizes)
                Lp = max(Lp, (L + Pmax + kp_max + 8 + 7) // 8 * 8)
            xpad = (_pad_rows16 if b16 else _pad_rows)(x, B, L, 1, Pmax, Lp)
            xpl = split16(xpad) if toep else None
            saved_meta = []
            for j, k in enumerate(ksizes):
                off = Pmax - k // 2
                shift = off % al
                base = off - shift
                Kp = (k + 7 + shift + al - 1) // al * al
                goff_c = _table(("tz", Ncat, Cout, j),
                                lambda j=j: [r * Ncat + j * Cout + cb * 32 for r in range(8)
                                             for cb in range(Cout // 32)], dev)
                if toep:
                    shift8 = off % 8
                    Kp64 = (k + 7 + shift8 + 63) // 64 * 64
                    wexp8 = torch.empty(8 * Cout, Kp64, device=dev, dtype=torch.float32)
                    _lib.check(_lib_().ac_toeplitz_expand(_p(ws[j]), _p(wexp8), Cout, k, Kp64, shift8, _stream()),
                               "ac_toeplitz_expand")
                    bexp = bs[j].detach().repeat(8) if bs[j] is not None else None
                    done = conv_window_x3(xpl, Lp, 8, off - shift8, 0, B, Lq, 64, Kp64 // 64, split16(wexp8), Kp64, 64,
                                          False, 8 * Cout, _p(ycat, j * Cout), 8 * Ncat, bexp, False,
                                          tap_row_step=8, c_block=Cout, c_block_stride=Ncat)
                    if done:
                        saved_meta.append((base, shift, Kp, goff_c))
                        continue
                wexp = torch.empty(8 * Cout, Kp, device=dev, dtype=torch.float32)
                _lib.check(_lib_().ac_toeplitz_expand(_p(ws[j]), _p(wexp), Cout, k, Kp, shift,
                                                      _stream()), "ac_toeplitz_expand")
                # bias[(r,co)] = b[co]; 8*Cout floats of plumbing
                bexp = bs[j].detach().repeat(8) if bs[j] is not None else None
                wop = cast16(wexp) if b16 else wexp
                if cat16:
                    gemm(AC_GEMM_NT, B * Lq, 8 * Cout, Kp,
                         mat(_p(xpad, base), r1=Lq, r2=Lq, s1=Lp, s3=8), mat(_p(wop), Kp),
                         mat(None, 8 * Ncat, goff=goff_c), bias=bexp, c16=ycat, ld_c16=8 * Ncat, math=mth)
                else:
                    gemm(AC_GEMM_NT, B * Lq, 8 * Cout, Kp,
                         mat(_p(xpad, base), r1=Lq, r2=Lq, s1=Lp, s3=8),
                         mat(_p(wop), Kp),
                         mat(_p(ycat), 8 * Ncat, goff=goff_c), bias=bexp, math=mth)
                saved_meta.append((base, shift, Kp, goff_c))
            ctx.meta = saved_meta
            ctx.Lp = Lp
            ctx.xpl = xpl      # (hi, lo) planes of the padded flux: the Toeplitz weight-gradient products read them too
        else:
            if Cin % 32:
                raise ValueError("ConvGroup1d needs Cin == 1 or Cin % 32 == 0")
            Lp = L + 2 * Pmax
            # split-bf16 mode: the conv products run on the LDS-window kernels over (hi, lo) planes of the
            # padded input and of the taps.  When every product of this bank (forward, input gradient,
            # weight gradient) is covered by them, the fp32 padded copy is not built at all.
            xplanes = _pad_rows_split(x, B, L, Cin, Pmax, Lp) if (x3_mode() and _CONVWIN and Cin % 8 == 0) else None
            planes_only = bool(xplanes is not None and _CONVWIN_X3_FUSED and _WGRAD_WIN and L % 128 == 0
                               and Cin % 64 == 0 and Cout % 128 == 0)
            xpad = None if planes_only else (_pad_rows16 if b16 else _pad_rows)(x, B, L, Cin, Pmax, Lp)
            xf_shared = {}
            for j, k in enumerate(ksizes):
                off = Pmax - k // 2
                if not b16 and fftconv_covered(B, L, Cin, Cout, k):
                    # long taps: the three products of this convolution run in the frequency domain (ac_fft.hip);
                    # convolutions of the bank with the same transform plan share the spectrum of x
                    plan = fft_plan(L, k)
                    shared = xf_shared.get(plan) if (_FFT_SHARE and plan[2] == 1) else None
                    ctx.fft[j] = fftconv_forward(x, ws[j], B, L, Cin, Cout, k, ycat, Ncat, j * Cout, bs[j], xf=shared)
                    if plan[2] == 1:
                        xf_shared[plan] = ctx.fft[j][0]
                    continue
                if xplanes is not None and conv_window_x3(
                        xplanes, Lp * Cin, Cin, 0, off, B, L, Cin, k, split16_w(ws[j]), k * Cin, Cin, False, Cout,
                        _p(ycat, j * Cout), Ncat, bs[j], False):
                    continue
                if xpad is None:
                    raise RuntimeError("split-bf16 conv bank: the window kernel refused a shape it covers")
                wop = cast16_w(ws[j]) if b16 else ws[j]
                if cat16:
                    if conv_window(xpad, Lp * Cin, Cin, 0, off, B, L, Cin, k, wop, k * Cin, Cin, False,
                                   Cout, None, Ncat, bs[j], False, c16_ptr=_p(ycat, j * Cout), ldc16=Ncat):
                        continue
                    gemm(AC_GEMM_NT, B * L, Cout, k * Cin,
                         mat(_p(xpad, off * Cin), r1=L, r2=L, s1=Lp * Cin, s3=Cin), mat(_p(wop), k * Cin),
                         mat(None, Ncat), bias=bs[j], c16=_View16(ycat, j * Cout), ld_c16=Ncat, math=mth)
                    continue
                if b16 and conv_window(xpad, Lp * Cin, Cin, 0, off, B, L, Cin, k, wop, k * Cin, Cin,
                                       False, Cout, _p(ycat, j * Cout), Ncat, bs[j], False):
                    continue
                gemm(AC_GEMM_NT, B * L, Cout, k * Cin,
                     mat(_p(xpad, off * Cin), r1=L, r2=L, s1=Lp * Cin, s3=Cin),
                     mat(_p(wop), k * Cin),
                     mat(_p(ycat, j * Cout), Ncat), bias=bs[j], math=mth)
            ctx.Lp = Lp
            ctx.xplanes = xplanes
        ctx.fused_ln = ln_gamma is not None
        ctx.params = (list(wb[0::2]), ln_gamma, ln_beta)
        ctx.bparams = list(wb[1::2])
        ctx.tail = False
        if pw_w is not None:
            # LayerNorm + GELU + 1x1 conv + MaxPool(4) behind the conv bank in one kernel (ac_tail.hip): nothing of row
            # length Ncat is written — z = gelu(LN(ycat)) exists only as (hi, lo) planes of 32 rows in LDS, the backward
            # pass rebuilds it from ycat.  The caller checked tail_covered().
            rows, Cpw = B * L, pw_w.shape[0]
            mean = torch.empty(rows, device=dev, dtype=torch.float32)
            rstd = torch.empty(rows, device=dev, dtype=torch.float32)
            pooled = torch.empty(B, L // 4, Cpw, device=dev, dtype=torch.float32)
            pidx = torch.empty(B, L // 4, Cpw, device=dev, dtype=torch.uint8)
            wh, wl = split16_w(pw_w)
            spectail_fwd(ycat, ln_gamma, ln_beta, ln_eps, wh, wl, pw_b, mean, rstd, pooled, pidx, rows, Ncat, Cpw)
            _tap("pool4", pidx)
            ctx.tail, ctx.pw = True, (pw_w, pw_b)
            ctx.save_for_backward(xpad, *ws, ycat, mean, rstd, ln_gamma, ln_beta, pidx, pw_w)
            return pooled
        if ctx.fused_ln:
            rows = B * L
            y = torch.empty(B, L, Ncat, device=dev, dtype=torch.float32)
            mean = torch.empty(rows, device=dev, dtype=torch.float32)
            rstd = torch.empty(rows, device=dev, dtype=torch.float32)
            y16 = _side16_alloc(ycat.shape, Ncat, dev)
            only16 = out16_only and y16 is not None   # the consumer reads the bf16 copy: skip fp32
            _lib.check(_lib_().ac_layernorm_fwd(_p(ycat), Ncat, _p(ln_gamma), _p(ln_beta),
                                                None if only16 else _p(y), Ncat,
                                                _p(mean), _p(rstd), rows, Ncat, ln_eps, _kact(ACT_GELU),
                                                _p(y16), Ncat, 1 if cat16 else 0, _stream()),
                       "ac_layernorm_fwd")
            if only16:
                _mark16only(y, y16)
            elif y16 is not None:
                y._ac16 = y16
            ctx.save_for_backward(xpad, *ws, ycat, mean, rstd, ln_gamma, ln_beta)
            return y
        ctx.save_for_backward(xpad, *ws)
        return ycat

    @staticmethod
    def backward(ctx, dycat):
        _no_f16_backward()
        B, L, Cin, Cout, Pmax = ctx.dims
        ksizes = ctx.ksizes
        nconv = len(ksizes)
        Ncat = nconv * Cout
        dev = dycat.device
        dy16in = dycat._ac16 if _is16only(dycat) else None   # gradient handed over in bf16 only
        dycat = _chk(dycat, "dycat", allow16=True)
        if dy16in is not None and not ctx.fused_ln:
            raise RuntimeError("bf16-only gradient reached a conv bank without the fused LayerNorm")
        dgam = dbet = bias_sums = None
        Lp = ctx.Lp
        b16 = ctx.b16
        mth = _lib.MATH_BF16_IN if b16 else None
        need_dx = Cin != 1 and ctx.needs_input_grad[0]
        Lpd = L + 2 * Pmax
        dypad = dyop = None
        planes_direct, ctx_dyplanes, dy1planes = False, None, None
        dpw_w = dpw_b = None
        if ctx.fused_ln:
            saved = ctx.saved_tensors
            xpad, ws = saved[0], list(saved[1:1 + nconv])
            ycat, mean, rstd, ln_gamma, ln_beta = saved[1 + nconv:6 + nconv]
            dpool = pidx = pw_w = None
            if ctx.tail:
                # fused tail (ac_tail.hip): the 1x1 conv's weight gradient = scatter(d pooled)^T . z with z rebuilt from
                # ycat in the kernel, its bias gradient = column sums of d pooled (every pooled gradient reaches exactly
                # one row); the (hi, lo) planes of d ycat come from ac_spectail_bwd_dx below, in LayerNorm's place
                pidx, pw_w = saved[6 + nconv:]
                Cpw, rows = pw_w.shape[0], B * L
                dpool = _chk(dycat, "dpooled")
                if ctx.needs_input_grad[6]:
                    wsink = _sink(ctx.pw[0])
                    dpw_w = wsink if wsink is not None else torch.zeros(Cpw, Ncat, device=dev, dtype=torch.float32)
                    spectail_bwd_dw(ycat, mean, rstd, ln_gamma, ln_beta, dpool, pidx, dpw_w, rows, Ncat, Cpw)
                    if wsink is not None:
                        dpw_w = None
                        _grad_written(ctx.pw[0])
                if ctx.pw[1] is not None and ctx.needs_input_grad[7]:
                    bsink = _sink(ctx.pw[1])
                    if bsink is not None:
                        _lib.check(_lib_().ac_colsum(_p(dpool), Cpw, _p(bsink), rows // 4, Cpw, 1, _stream()), "ac_colsum")
                        _grad_written(ctx.pw[1])
                    else:
                        dpw_b = colsum(_p(dpool), Cpw, rows // 4, Cpw, dev)
                dycat, dy16in = None, None
            gsink, bsink = _sink(ctx.params[1]), _sink(ctx.params[2])
            ln_direct = gsink is not None and bsink is not None
            dgam = gsink if ln_direct else torch.zeros_like(ln_gamma)
            dbet = bsink if ln_direct else torch.zeros_like(ln_beta)
            fuse_bias = Ncat % 4 == 0 and Ncat <= 3072
            bias_sums = _zeros((Ncat,), dev) if fuse_bias else None
            # bf16 mode: the LayerNorm backward writes the bf16 operand of the gradient products
            # directly (into the zero-padded buffer when the input gradient is needed); the fp32
            # gradient of the conv outputs is never materialised
            direct16 = b16 and fuse_bias and _ln_sub_shape(Ncat)
            dpre, seg = None, (0, 0, 0)
            # split-bf16 mode: LayerNorm's backward writes the (hi, lo) operand planes of the gradient products
            # itself, zero-padded — no fp32 d(ycat) and no ac_pad_rows_split pass (0.65 ms per step at B = 512) —
            # when every product of the bank runs on the plane-fed kernels (nothing then reads the fp32 form)
            planes_direct = (_LN_PLANES and not b16 and x3_mode() and need_dx and fuse_bias and _ln_sub_shape(Ncat)
                             and getattr(ctx, "xplanes", None) is not None and dy16in is None and not ctx.cat16
                             and _x3_bank_covered(B, L, Cin, Cout))
            lo16 = None
            # stage 1 (Cin = 1): its three Toeplitz weight-gradient products read (hi, lo) planes of d(ycat) too, and
            # nothing else reads the fp32 form (no input gradient, bias sums come out of this pass)
            planes1 = bool(_LN_PLANES and not b16 and x3_mode() and Cin == 1 and fuse_bias and _ln_sub_shape(Ncat)
                           and getattr(ctx, "xpl", None) is not None and dy16in is None and not ctx.cat16
                           and _WGRAD_WIN and _TOEPLITZ_RING and Cout % 16 == 0 and (L // 8) % 64 == 0 and Ncat % 8 == 0)
            if planes1:
                both = torch.empty(2, B * L, Ncat, device=dev, dtype=_H16)
                out16, lo16 = both[0], both[1]
                dy1planes = (both[0], both[1])
            elif planes_direct:
                both = torch.empty(2, B, Lpd, Ncat, device=dev, dtype=_H16)
                if Pmax > 0:
                    both[:, :, :Pmax].zero_()
                    both[:, :, Pmax + L:].zero_()
                out16, lo16, seg = both[0], both[1], (L, Lpd, Pmax)
                ctx_dyplanes = (both[0], both[1])
            elif direct16 and need_dx:
                dypad = torch.empty(B, Lpd, Ncat, device=dev, dtype=_H16)
                if Pmax > 0:
                    dypad[:, :Pmax].zero_()
                    dypad[:, Pmax + L:].zero_()
                out16, seg = dypad, (L, Lpd, Pmax)
            elif direct16:
                dyop = torch.empty(B * L, Ncat, device=dev, dtype=_H16)
                out16 = dyop
            elif ctx.tail:
                # (no plane-fed consumer for this call - e.g. an input that needs no gradient: planes, then their sum)
                both = torch.empty(2, B * L, Ncat, device=dev, dtype=_H16)
                out16, lo16 = both[0], both[1]
            else:
                dpre, out16 = torch.empty(B, L, Ncat, device=dev, dtype=torch.float32), None
            if ctx.tail:
                wth, wtl = split16_wT(pw_w)
                spectail_bwd_dx(ycat, mean, rstd, ln_gamma, ln_beta, dpool, pidx, wth, wtl, out16, lo16, seg, dgam, dbet,
                                bias_sums, B * L, Ncat, pw_w.shape[0])
                if not (planes1 or planes_direct):
                    dpre = (both[0].float() + both[1].float()).reshape(B, L, Ncat)
            else:
                _lib.check(_lib_().ac_layernorm_bwd_split(_p(dy16in if dy16in is not None else dycat), Ncat,
                                                          _p(ycat), Ncat, _p(mean), _p(rstd),
                                                          _p(ln_gamma), _p(ln_beta), _p(dpre), Ncat, _p(dgam),
                                                          _p(dbet), _p(bias_sums), B * L, Ncat, _kact(ACT_GELU),
                                                          _p(out16), _p(lo16), Ncat, seg[0], seg[1], seg[2],
                                                          1 if dy16in is not None else 0, 1 if ctx.cat16 else 0,
                                                          _stream()),
                           "ac_layernorm_bwd_split")
            dycat = dpre
            if ln_direct:
                dgam = dbet = None
                _grad_written(ctx.params[1])
                _grad_written(ctx.params[2])
        else:
            xpad, *ws = ctx.saved_tensors

        # the conv biases' gradients = slices of the column sums the LayerNorm / tail backward kernel formed: one
        # launch adds them into the biases' gradient sinks (they were nconv AccumulateGrad launches per bank)
        bias_direct = False
        if bias_sums is not None and nconv <= 4 and all(ctx.has_b):
            bsinks = [_sink(bp) for bp in ctx.bparams]
            if all(t is not None for t in bsinks):
                ptrs = [_p(t) for t in bsinks] + [None] * (4 - nconv)
                _lib.check(_lib_().ac_add_segments(_p(bias_sums), ptrs[0], ptrs[1], ptrs[2], ptrs[3], Cout, nconv,
                                                   _stream()), "ac_add_segments")
                for bp in ctx.bparams:
                    _grad_written(bp)
                bias_direct = True

        def bias_grad(j):
            if not ctx.has_b[j] or bias_direct:
                return None
            if bias_sums is not None:
                return bias_sums[j * Cout:(j + 1) * Cout]
            return colsum(_p(dycat, j * Cout), Ncat, B * L, Cout, dev)

        grads = []
        dx = None
        x3win = bool(need_dx and not b16 and x3_mode() and _CONVWIN and Ncat % 8 == 0 and Cout % 8 == 0)
        if need_dx and dypad is None and not x3win:
            # one zero-padded (bf16 in bf16 mode) copy of d(ycat) serves the input-gradient products
            # (as the window / gathered operand) and the weight-gradient products (rows Pmax..Pmax+L)
            dypad = (_pad_rows16 if b16 else _pad_rows)(dycat, B, L, Ncat, Pmax, Lpd)
        if b16 and need_dx:
            dy_mat = lambda j: mat(_p(dypad, Pmax * Ncat + j * Cout), r1=L, r2=L, s1=Lpd * Ncat, s3=Ncat)
        else:
            if dyop is None:
                dyop = cast16(dycat) if b16 else dycat  # A operand of the dW products
            dy_mat = lambda j: mat(_p(dyop, j * Cout), Ncat)
        if Cin == 1:
            Lq = L // 8
            xpl = getattr(ctx, "xpl", None)
            # split-bf16: the Toeplitz weight gradients on the LDS-window weight-gradient kernel (its Toeplitz form:
            # 64-element "taps", window rows 8 apart, blocked dy columns) over the planes of d(ycat) and of the flux
            toepw = bool(xpl is not None and _WGRAD_WIN and _TOEPLITZ_RING and Cout % 16 == 0 and Lq % 64 == 0
                         and Ncat % 8 == 0)
            dyp = None
            if toepw:
                dyp = dy1planes if dy1planes is not None else split16(dycat.reshape(B * L, Ncat))
            for j, k in enumerate(ksizes):
                base, shift, Kp, goff_c = ctx.meta[j]
                if toepw:
                    off = Pmax - k // 2
                    shift8 = off % 8
                    Kp64 = (k + 7 + shift8 + 63) // 64 * 64
                    dwexp8 = _zeros((8 * Cout, Kp64), dev)
                    xoff = off - shift8
                    if conv_wgrad(dyp[0], dyp[1], L * Ncat, 8 * Ncat, 0, j * Cout, xpl[0], xpl[1], Lp, 8, 0,
                                  (Lp - xoff - 64) // 8 + 1, B, Lq, 8 * Cout, 64, Kp64 // 64, dwexp8, tap_row_step=8,
                                  dy_block=Cout, dy_block_stride=Ncat, x_elem_off=xoff, ldw=Kp64):
                        dw = torch.empty(Cout, k, device=dev, dtype=torch.float32)
                        _lib.check(_lib_().ac_toeplitz_fold(_p(dwexp8), _p(dw), Cout, k, Kp64, shift8, _stream()),
                                   "ac_toeplitz_fold")
                        grads += [dw, bias_grad(j)]
                        continue
                    if dycat is None:
                        raise RuntimeError("split-bf16 Cin = 1 conv bank: the Toeplitz weight-gradient kernel refused "
                                           "a shape its dispatch admits")
                if dyop is None:
                    dyop = dycat
                dwexp = _zeros((8 * Cout, Kp), dev)
                gemm(AC_GEMM_TN, 8 * Cout, Kp, B * Lq, mat(_p(dyop), 8 * Ncat, goff=goff_c),
                     mat(_p(xpad, base), r1=Lq, r2=Lq, s1=Lp, s3=8), mat(_p(dwexp), Kp),
                     accumulate=2, split_k=_split_for(8 * Cout, Kp, B * Lq), math=mth)
                dw = torch.empty(Cout, k, device=dev, dtype=torch.float32)
                _lib.check(_lib_().ac_toeplitz_fold(_p(dwexp), _p(dw), Cout, k, Kp, shift,
                                                    _stream()), "ac_toeplitz_fold")
                grads += [dw, bias_grad(j)]
            if ctx.needs_input_grad[0]:
                raise NotImplementedError("input gradient of the Cin == 1 conv bank is not needed "
                                          "on the path (the flux is a network input)")
        else:
            if need_dx:
                dx = torch.empty(B, L, Cin, device=dev, dtype=torch.float32)
            dyplanes = ctx_dyplanes if planes_direct else (_pad_rows_split(dycat, B, L, Ncat, Pmax, Lpd) if x3win else None)
            # convolutions in the frequency domain that share a plan (one sequence per sample) sum their input-gradient
            # spectra and transform back once, after the loop; `dx_started` = something already wrote dx
            plans = [ctx.fft[j][2] for j in ctx.fft]
            groups = {pl: {} for pl in set(plans) if _FFT_SHARE and pl[1] == 1 and plans.count(pl) > 1}
            dx_started = False
            for j, k in enumerate(ksizes):
                p = k // 2
                off = Pmax - p
                if j in ctx.fft:
                    wsink = _sink(ctx.params[0][j])
                    dw = wsink if wsink is not None else torch.zeros(Cout, k * Cin, device=dev, dtype=torch.float32)
                    if dycat is not None:      # fp32 rows of d(ycat)
                        src = (dycat, None, 0, L * Ncat)
                    else:                      # the zero-padded (hi, lo) planes LayerNorm's backward wrote
                        src = (dyplanes[0], dyplanes[1], Pmax * Ncat, Lpd * Ncat)
                    grp = groups.get(ctx.fft[j][2])
                    fftconv_backward(ctx.fft[j], src[0], src[1], src[2], src[3], Ncat, j * Cout, B, L, Cin, Cout, k,
                                     dx if ctx.needs_input_grad[0] else None, dx_started, dw, dx_group=grp)
                    if grp is None and ctx.needs_input_grad[0]:
                        dx_started = True
                    if wsink is not None:
                        dw = None
                        _grad_written(ctx.params[0][j])
                    grads += [dw, bias_grad(j)]
                    continue
                if ctx.needs_input_grad[0] and dyplanes is not None and conv_window_x3(
                        dyplanes, Lpd * Ncat, Ncat, j * Cout, off, B, L, Cout, k, split16_wT(ctx.params[0][j]),
                        Cout, Cin * Cout, True, Cin, _p(dx), Cin, None, dx_started):
                    dx_started = True
                elif ctx.needs_input_grad[0] and planes_direct:
                    raise RuntimeError("split-bf16 conv bank: the plane-fed window kernel refused a shape that "
                                       "_x3_bank_covered() admits")
                elif ctx.needs_input_grad[0]:
                    goff = _table(("cg_dx", Ncat, Cout, j, k, Pmax),
                                  lambda j=j, k=k, p=p: [(Pmax + p - t) * Ncat + j * Cout + cb * 32
                                                         for t in range(k)
                                                         for cb in range(Cout // 32)], dev)
                    if b16:
                        # NT against the k-contiguous copy of the taps: wT[(t,ci), co]; operand row ci,
                        # inner index (t, co) through an offset table
                        wT = cast16_wT(ctx.params[0][j])  # [k*Cin, Cout]
                        if conv_window(dypad, Lpd * Ncat, Ncat, j * Cout, off, B, L, Cout, k, wT, Cout,
                                       Cin * Cout, True, Cin, _p(dx), Cin, None, dx_started):
                            wT = None
                            dx_started = True
                    if b16 and wT is not None:
                        goff_b = _table(("cg_dxw", Cin, Cout, k),
                                        lambda k=k: [t * Cin * Cout + cb * 32 for t in range(k)
                                                     for cb in range(Cout // 32)], dev)
                        gemm(AC_GEMM_NT, B * L, Cin, k * Cout,
                             mat(_p(dypad), r1=L, r2=L, s1=Lpd * Ncat, s3=Ncat, goff=goff),
                             mat(_p(wT), Cout, goff=goff_b), mat(_p(dx), Cin),
                             accumulate=1 if dx_started else 0, math=mth)
                        dx_started = True
                    elif not b16:
                        if dypad is None:
                            dypad = _pad_rows(dycat, B, L, Ncat, Pmax, Lpd)
                        gemm(AC_GEMM_NN, B * L, Cin, k * Cout,
                             mat(_p(dypad), r1=L, r2=L, s1=Lpd * Ncat, s3=Ncat, goff=goff),
                             mat(_p(ws[j]), r1=Cout, r2=Cout, s1=Cin, s3=k * Cin),
                             mat(_p(dx), Cin), accumulate=1 if dx_started else 0)
                        dx_started = True
                wsink = _sink(ctx.params[0][j])
                dw = wsink if wsink is not None else torch.zeros(Cout, k * Cin, device=dev,
                                                                  dtype=torch.float32)
                done = False
                if b16:
                    # LDS-window weight-gradient kernel: 8 taps x 64 channels share one input window
                    if dypad is not None and need_dx:
                        done = conv_wgrad(dypad, None, Lpd * Ncat, Ncat, Pmax, j * Cout, xpad, None, Lp * Cin,
                                          Cin, off, Lp, B, L, Cout, Cin, k, dw)
                    else:
                        done = conv_wgrad(dyop, None, L * Ncat, Ncat, 0, j * Cout, xpad, None, Lp * Cin, Cin,
                                          off, Lp, B, L, Cout, Cin, k, dw)
                elif x3_mode() and getattr(ctx, "xplanes", None) is not None and Ncat % 8 == 0:
                    if dyplanes is None:
                        dyplanes = _pad_rows_split(dycat, B, L, Ncat, Pmax, Lpd)
                    done = conv_wgrad(dyplanes[0], dyplanes[1], Lpd * Ncat, Ncat, Pmax, j * Cout,
                                      ctx.xplanes[0], ctx.xplanes[1], Lp * Cin, Cin, off, Lp, B, L, Cout, Cin, k, dw)
                if not done:
                    if xpad is None or planes_direct:
                        raise RuntimeError("split-bf16 conv bank: the weight-gradient kernel refused a covered shape")
                    tile_, split_ = _tn_plan(Cout, k * Cin, B * L) if b16 else (0, _split_for(Cout, k * Cin, B * L))
                    gemm(AC_GEMM_TN, Cout, k * Cin, B * L, dy_mat(j),
                         mat(_p(xpad, off * Cin), r1=L, r2=L, s1=Lp * Cin, s3=Cin),
                         mat(_p(dw), k * Cin), accumulate=2, split_k=split_, tile=tile_, math=mth)
                if wsink is not None:
                    dw = None
                    _grad_written(ctx.params[0][j])
                grads += [dw, bias_grad(j)]
        if Cin != 1 and ctx.needs_input_grad[0]:
            for grp in groups.values():
                if grp.get("dxf") is not None:
                    fftconv_finish_dx(grp, dx, dx_started, B, L, Cin)
                    dx_started = True
        return (dx, None, dgam, dbet, None, None, dpw_w, dpw_b, *grads)


_FUSED_TAIL = True   # tests / A-B: False = LayerNorm, 1x1 conv and MaxPool(4) as separate kernels everywhere


# the three tail kernels behind module-level names (bench.py brackets them with HIP events)
def spectail_fwd(ycat, gamma, beta, eps, wh, wl, bias, mean, rstd, pooled, pidx, rows, K, N):
    _lib.check(_lib_().ac_spectail_fwd(_p(ycat), _p(gamma), _p(beta), eps, _p(wh), _p(wl), _p(bias), _p(mean), _p(rstd),
                                       _p(pooled), _p(pidx), rows, K, N, _stream()), "ac_spectail_fwd")


def spectail_bwd_dx(ycat, mean, rstd, gamma, beta, dpool, pidx, wth, wtl, out_hi, out_lo, seg, dgam, dbet, dxsum, rows, K, N):
    _lib.check(_lib_().ac_spectail_bwd_dx(_p(ycat), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dpool), _p(pidx), _p(wth),
                                          _p(wtl), _p(out_hi), _p(out_lo), seg[0], seg[1], seg[2], _p(dgam), _p(dbet),
                                          _p(dxsum), rows, K, N, _stream()), "ac_spectail_bwd_dx")


def spectail_bwd_dw(ycat, mean, rstd, gamma, beta, dpool, pidx, dw, rows, K, N):
    _lib.check(_lib_().ac_spectail_bwd_dw(_p(ycat), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dpool), _p(pidx), _p(dw),
                                          rows, K, N, _stream()), "ac_spectail_bwd_dw")


def tail_covered(B: int, L: int, Cin: int, Cout: int, nk: int = 3) -> bool:
    """Whether LayerNorm + GELU + 1x1 conv + MaxPool(4) of a pooled SpectraNetBlock (spectranet.py:31-40) run as the
    fused tail kernels behind the conv bank (ac_tail.hip: ac_spectail_fwd / _bwd_dx / _bwd_dw; split-bf16 mode):
    the kernels' shapes (32-row blocks; (3 Cout, Cout) = (192, 64) or (384, 128): stages 1 and 2 of the default
    configuration, 75 % of the tails' bytes) AND the bank's gradient products all on the plane-fed kernels, because the
    backward kernel hands d(conv outputs) over as (hi, lo) planes only."""
    if not (_FUSED_TAIL and x3_mode() and _LN_PLANES and L % 32 == 0 and nk * Cout % 8 == 0):
        return False
    if not _lib_().ac_spectail_supported(B * L, nk * Cout, Cout):
        return False
    if Cin == 1:   # the Toeplitz products of stage 1 on the plane-fed kernels (_ConvGroup1d: `toep`, `planes1`)
        Lq = L // 8
        toep = Lq % 256 == 0 or (8 <= Lq < 128 and (Lq & (Lq - 1)) == 0 and (B * Lq) % 256 == 0)
        return bool(_CONVWIN and _CONVWIN_X3_FUSED and _TOEPLITZ_RING and _WGRAD_WIN and Cout % 16 == 0 and L % 8 == 0
                    and toep and Lq % 64 == 0)
    return bool(Cin % 8 == 0 and _x3_bank_covered(B, L, Cin, Cout))


def conv_group1d(x, ksizes, weights, biases, ln=None, out16_only=False, tail=None):
    """ln = (gamma, beta, eps) fuses LayerNorm + GELU over the concatenated channels.
    out16_only (bf16 math mode, with ln): the caller feeds the result straight into `linear` — the
    fp32 output is then not written at all (its bf16 copy is), and the gradient comes back in bf16.
    tail = (w [Cpw, 3 Cout], b) with ln: the 1x1 conv and MaxPool1d(4) of spectranet.py:36-40 join the same node
    (the caller checks tail_covered); the result is the pooled [B, L / 4, Cpw] tensor."""
    args = []
    for w, b in zip(weights, biases):
        args += [w, b]
    g, bt, eps = ln if ln is not None else (None, None, 0.0)
    pw_w, pw_b = tail if (tail is not None and ln is not None) else (None, None)
    return _ConvGroup1d.apply(x, tuple(int(k) for k in ksizes), g, bt, float(eps),
                              bool(out16_only and ln is not None), pw_w, pw_b, *args)


# --------------------------------------------------------------------------- photometry branch
class _Embed(Function):
    """CLS + in_proj(x) + Time2Vec(x[...,0])  (HyraxBaselineCLS.py:58-71).
    x8 [B,L,8] (7 channels + zero pad), W8 [D,8], bias [D], tw [D], tb [D], cls [D]."""

    @staticmethod
    def forward(ctx, x8, W8, bias, tw, tb, cls):
        x8 = _chk(x8, "x8")
        B, L, _ = x8.shape
        D = W8.shape[0]
        h = torch.empty(B, L + 1, D, device=x8.device, dtype=torch.float32)
        _lib.check(_lib_().ac_embed_fwd(_p(x8), _p(_chk(W8)), _p(bias), _p(tw), _p(tb), _p(cls),
                                        _p(h), B, L, D, _stream()), "ac_embed_fwd")
        ctx.save_for_backward(x8, tw, tb)
        ctx.dims = (B, L, D)
        ctx.params, ctx.cls_shape = (W8, bias, tw, tb, cls), cls.shape
        return h

    @staticmethod
    def backward(ctx, dh):
        x8, tw, tb = ctx.saved_tensors
        B, L, D = ctx.dims
        dh = _chk(dh, "dh")
        dev = dh.device
        # the kernel accumulates with atomics: parameters that own a slice of the flat gradient buffer take their sums
        # directly (no zeroed temporaries, no AccumulateGrad launches)
        sinks = [_sink(p) for p in ctx.params]
        direct = all(s is not None for s in sinks)
        if direct:
            dW, db, dtw, dtb, dcls = sinks
        else:
            dW = torch.zeros(D, 8, device=dev, dtype=torch.float32)
            db, dtw, dtb, dcls = (torch.zeros(D, device=dev, dtype=torch.float32) for _ in range(4))
        _lib.check(_lib_().ac_embed_bwd(_p(dh), _p(x8), _p(tw), _p(tb), _p(dW), _p(db), _p(dtw),
                                        _p(dtb), _p(dcls), B, L, D, _stream()), "ac_embed_bwd")
        if direct:
            for p in ctx.params:
                _grad_written(p)
            return None, None, None, None, None, None
        return None, dW, db, dtw, dtb, dcls.reshape(ctx.cls_shape)


def embed(x8, W8, bias, tw, tb, cls):
    return _Embed.apply(x8, W8, bias, tw, tb, cls)


_MHA_MFMA = True   # tests switch the matrix-core attention kernels off to compare with the scalar ones


class _MHA(Function):
    @staticmethod
    def forward(ctx, qkv, pad_u8, H, p_drop, seed):
        qkv = _chk(qkv, "qkv")
        B, T, D3 = qkv.shape
        D = D3 // 3
        Dh = D // H
        out = torch.empty(B, T, D, device=qkv.device, dtype=torch.float32)
        lse = torch.empty(B, H, T, device=qkv.device, dtype=torch.float32)
        # matrix-core kernels (ac_attn.hip) in the bf16 / split-bf16 modes; the exact fp32 mode and
        # shapes they do not cover (d_head != 16, T > 288) use the scalar fp32 kernels
        mfma = _MHA_MFMA and _MATH != _lib.MATH_F32 and Dh == 16 and T <= 288
        split = 1 if _MATH == _lib.MATH_BF16X3 else 0
        if mfma:
            _lib.check(_lib_().ac_mha_fwd_mfma(_p(qkv), _p(pad_u8), _p(out), _p(lse), B, T, H, Dh, p_drop,
                                               seed, _p(_STEP_DEV), split, _stream()), "ac_mha_fwd_mfma")
        else:
            _lib.check(_lib_().ac_mha_fwd(_p(qkv), _p(pad_u8), _p(out), _p(lse), B, T, H, Dh, p_drop,
                                          seed, _p(_STEP_DEV), _stream()), "ac_mha_fwd")
        ctx.save_for_backward(qkv, pad_u8, out, lse)
        ctx.cfg = (B, T, H, Dh, p_drop, seed)
        ctx.mfma, ctx.split = mfma, split
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, pad_u8, out, lse = ctx.saved_tensors
        B, T, H, Dh, p_drop, seed = ctx.cfg
        dout = _chk(dout, "dout")
        dqkv = torch.empty_like(qkv)
        if ctx.mfma:
            _lib.check(_lib_().ac_mha_bwd_mfma(_p(dout), _p(qkv), _p(pad_u8), _p(out), _p(lse), _p(dqkv), B,
                                               T, H, Dh, p_drop, seed, _p(_STEP_DEV), ctx.split, _stream()),
                       "ac_mha_bwd_mfma")
        else:
            _lib.check(_lib_().ac_mha_bwd(_p(dout), _p(qkv), _p(pad_u8), _p(out), _p(lse), _p(dqkv), B,
                                          T, H, Dh, p_drop, seed, _p(_STEP_DEV), _stream()), "ac_mha_bwd")
        return dqkv, None, None, None, None


def mha(qkv, pad_u8, n_heads: int, p_drop: float = 0.0, training: bool = False):
    p = float(p_drop) if training else 0.0
    return _MHA.apply(qkv, pad_u8, n_heads, p, next_seed() if p > 0 else 0)


# --------------------------------------------------------------------------- losses
class _Loss(Function):
    @staticmethod
    def forward(ctx, logits, target, kind, gamma, eps, alpha):
        logits = _chk(logits, "logits")
        B, Cn = logits.shape
        if kind in (0, 3):
            target = _chk(target, "target")
            if tuple(target.shape) != tuple(logits.shape):
                raise ValueError(f"loss target shape {tuple(target.shape)} != prediction shape {tuple(logits.shape)}")
        else:
            if target.dtype != torch.int64:
                target = target.to(torch.int64)
            target = target.contiguous()
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        dlogits = torch.empty_like(logits)
        _lib.check(_lib_().ac_loss_fwd_bwd(_p(logits), _p(target), _p(alpha), _p(loss), _p(dlogits),
                                           B, Cn, kind, gamma, eps, _stream()), "ac_loss_fwd_bwd")
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (dlogits,) = ctx.saved_tensors
        if dloss.numel() == 1 and not dloss.requires_grad:
            # scale by the (usually 1.0) upstream gradient through the add kernel: g = d*dloss
            scale = dloss.reshape(1)
            g = dlogits.clone()
            _lib.check(_lib_().ac_scale_by_dev(_p(g), g.numel(), _p(_chk(scale)), _stream()),
                       "ac_scale_by_dev")
            return g, None, None, None, None, None
        raise RuntimeError("loss backward expects a scalar upstream gradient")


def cross_entropy_soft(logits, target_prob):
    return _Loss.apply(logits, target_prob, 0, 0.0, 0.0, None)


def cross_entropy_index(logits, target_idx):
    return _Loss.apply(logits, target_idx, 1, 0.0, 0.0, None)


def focal_loss(logits, target_idx, gamma=2.0, alpha=None, eps=0.0):
    return _Loss.apply(logits, target_idx, 2, float(gamma), float(eps), alpha)


def mse_loss(pred, target):
    """nn.MSELoss() (mean): pred / target [B] or [B, C] float (SpectraNet's redshift head, spectranet.py:178-179)."""
    if pred.dim() == 1:
        return _Loss.apply(pred.reshape(-1, 1), target.reshape(-1, 1).to(torch.float32), 3, 0.0, 0.0, None)
    return _Loss.apply(pred, target.to(torch.float32), 3, 0.0, 0.0, None)


# --------------------------------------------------------------------------- optimizer kernels
def adam_flat(param, grad, exp_avg, exp_avg_sq, segs, step, grad_scale_dev=None):
    arr = (AdamSeg * len(segs))(*segs)
    _lib.check(_lib_().ac_adam_flat(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), arr,
                                    len(segs), step, _p(grad_scale_dev), _stream()),
               "ac_adam_flat")


def adam_flat_dev(param, grad, exp_avg, exp_avg_sq, segs, step_dev, grad_scale_dev=None):
    """adam_flat with the step count read from device memory (int64[1]): graph-capturable."""
    arr = (AdamSeg * len(segs))(*segs)
    _lib.check(_lib_().ac_adam_flat_dev(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), arr,
                                        len(segs), _p(step_dev), _p(grad_scale_dev), _stream()),
               "ac_adam_flat_dev")


def sgd_flat(param, grad, buf, lr, momentum, weight_decay, first_step):
    _lib.check(_lib_().ac_sgd_flat(_p(param), _p(grad), _p(buf), param.numel(), lr, momentum,
                                   weight_decay, int(first_step), _stream()), "ac_sgd_flat")


def clip_coef(grad_flat, max_norm: float):
    """Device-side clip coefficient min(1, max_norm/(||g||+1e-6)); returns (coef, sumsq)."""
    sumsq = torch.empty(1, device=grad_flat.device, dtype=torch.float32)
    coef = torch.empty(1, device=grad_flat.device, dtype=torch.float32)
    _lib.check(_lib_().ac_sumsq(_p(grad_flat), grad_flat.numel(), _p(sumsq), _stream()), "ac_sumsq")
    _lib.check(_lib_().ac_clip_coef(_p(sumsq), float(max_norm), _p(coef), _stream()),
               "ac_clip_coef")
    return coef, sumsq


# --------------------------------------------------------------------------- misc
class _Act(Function):
    @staticmethod
    def forward(ctx, x, kind):
        x = _chk(x, "x")
        y = torch.empty_like(x)
        _lib.check(_lib_().ac_act_fwd(_p(x), _p(y), x.numel(), kind, _stream()), "ac_act_fwd")
        ctx.kind = kind
        ctx.save_for_backward(x if kind in (ACT_GELU, ACT_RELU) else y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (aux,) = ctx.saved_tensors
        dy = _chk(dy, "dy")
        dx = torch.empty_like(dy)
        _lib.check(_lib_().ac_act_bwd(_p(dy), _p(aux), _p(dx), dy.numel(), ctx.kind, _stream()),
                   "ac_act_bwd")
        return dx, None


def activation(x, kind: str):
    return _Act.apply(x, ACT_CODES[kind])


class _TakeToken(Function):
    """z[:, idx, :] of a [B, T, D] tensor as a contiguous [B, D] tensor."""

    @staticmethod
    def forward(ctx, z, idx):
        z = _chk(z, "z")
        B, T, D = z.shape
        out = torch.empty(B, D, device=z.device, dtype=torch.float32)
        _lib.check(_lib_().ac_copy2d(_p(z, idx * D), T * D, _p(out), D, B, D, _stream()),
                   "ac_copy2d")
        ctx.cfg = (B, T, D, idx)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, T, D, idx = ctx.cfg
        dout = _chk(dout, "dout")
        dz = _zeros((B, T, D), dout.device)
        _lib.check(_lib_().ac_copy2d(_p(dout), D, _p(dz, idx * D), T * D, B, D, _stream()),
                   "ac_copy2d")
        return dz, None


def take_token(z, idx: int = 0):
    return _TakeToken.apply(z, int(idx))


def pad_channels(x: torch.Tensor, width: int) -> torch.Tensor:
    """[..., C] -> [..., width] zero padded (network inputs only; no gradient)."""
    x = _chk(x, "x")
    Cn = x.shape[-1]
    rows = x.numel() // Cn
    out = torch.zeros(*x.shape[:-1], width, device=x.device, dtype=torch.float32)
    _lib.check(_lib_().ac_copy2d(_p(x), Cn, _p(out), width, rows, Cn, _stream()), "ac_copy2d")
    return out


# --------------------------------------------------------------------------- masked pre-training
def mpt_mask(data: torch.Tensor, pad: torch.Tensor, mask_p: float, seed: Optional[int] = None):
    """MPTModel._mask_batch on the device (HyraxBaselineCLS.py:286-319): zeroes channels 2..6 of the
    selected tokens of `data` IN PLACE and returns the selection as a bool tensor."""
    data = _chk(data, "data")
    B, L, Cn = data.shape
    if Cn != 7:
        raise ValueError("photometry tokens must have 7 channels")
    pad_u8 = pad.to(torch.uint8).contiguous()
    masked = torch.empty(B, L, device=data.device, dtype=torch.uint8)
    _lib.check(_lib_().ac_mpt_mask(_p(data), _p(pad_u8), _p(masked), B, L, float(mask_p),
                                   next_seed() if seed is None else int(seed), _p(_STEP_DEV), _stream()),
               "ac_mpt_mask")
    return masked.bool()


class _MPTLoss(Function):
    """Three-term product loss of MPTModel.train_step (HyraxBaselineCLS.py:258-278); loss and the
    gradients of the three head outputs in one pass over the tokens."""

    @staticmethod
    def forward(ctx, f_hat, b_hat, dt_hat, data, masked_u8, lambdas):
        f_hat, b_hat, dt_hat = _chk(f_hat, "f_hat"), _chk(b_hat, "b_hat"), _chk(dt_hat, "dt_hat")
        data = _chk(data, "data")
        B, L, _ = data.shape
        dev = data.device
        if f_hat.numel() != B * (L + 1) or b_hat.numel() != 3 * B * (L + 1):
            raise ValueError("head outputs must cover the whole encoder output [B, L+1, .]")
        sums = torch.empty(4, device=dev, dtype=torch.float32)
        loss = torch.empty((), device=dev, dtype=torch.float32)
        df, db, ddt = torch.empty_like(f_hat), torch.empty_like(b_hat), torch.empty_like(dt_hat)
        _lib.check(_lib_().ac_mpt_loss_fwd_bwd(_p(f_hat), _p(b_hat), _p(dt_hat), _p(data), _p(masked_u8),
                                               _p(sums), _p(loss), _p(df), _p(db), _p(ddt), B, L,
                                               float(lambdas[0]), float(lambdas[1]), float(lambdas[2]),
                                               _stream()), "ac_mpt_loss_fwd_bwd")
        ctx.save_for_backward(df, db, ddt)
        return loss

    @staticmethod
    def backward(ctx, g):
        df, db, ddt = ctx.saved_tensors
        if g.numel() == 1 and not g.requires_grad:
            # upstream gradient of a scalar loss: scale on the device, no host sync
            gs = g.reshape(1).to(torch.float32)
            lib, st = _lib_(), _stream()
            for t in (df, db, ddt):
                _lib.check(lib.ac_scale_by_dev(_p(t), t.numel(), _p(gs), st), "ac_scale_by_dev")
        return df, db, ddt, None, None, None


def mpt_loss(f_hat, b_hat, dt_hat, data, masked: torch.Tensor, lambdas):
    return _MPTLoss.apply(f_hat, b_hat, dt_hat, data, masked.to(torch.uint8).contiguous(), tuple(lambdas))
