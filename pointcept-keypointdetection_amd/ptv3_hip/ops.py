"""Tensor-level wrappers over the C ABI (include/ptv3_hip.h).

torch supplies device memory and the current HIP stream; all arithmetic happens in libptv3_hip.so.
Every wrapper validates device / dtype / contiguity on the host before a pointer reaches a kernel.
"""
import ctypes

import torch

from .lib import lib, PTV3_F32, PTV3_BF16, ACT_NONE, ACT_GELU, ACT_RELU, ORDER_IDS  # noqa: F401

_DT = {torch.float32: PTV3_F32, torch.bfloat16: PTV3_BF16}

_raw_stream = torch._C._cuda_getCurrentRawStream if hasattr(torch._C, "_cuda_getCurrentRawStream") else None


def _stream():
    """Raw hipStream_t of torch's current stream (the fast C accessor; ~20x cheaper than current_stream())."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _chk(t, name, dtype=None, dim=None):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (the PTv3 HIP path has no CPU fallback)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: tensor must be contiguous")
    if dtype is not None and t.dtype not in (dtype if isinstance(dtype, tuple) else (dtype,)):
        raise TypeError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if dim is not None and t.dim() != dim:
        raise RuntimeError(f"{name}: {t.dim()}-d tensor, expected {dim}-d")


def k_granule(dtype):
    """K (input-channel) granularity of ptv3_gemm: 16 bytes."""
    return 4 if dtype == torch.float32 else 8


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported feature dtype {t.dtype} (float32 or bfloat16)")


# ---------------------------------------------------------------------------------------------
# serialization
# ---------------------------------------------------------------------------------------------
def sfc_encode(grid_coord, batch, depth, orders):
    """code (k, n) int64; replaces encode() of utils/serialization/default.py:9-24 for k orders."""
    _chk(grid_coord, "grid_coord", (torch.int32, torch.int64), 2)
    _chk(batch, "batch", torch.int64, 1)
    n = grid_coord.shape[0]
    ids = (ctypes.c_int * len(orders))(*[ORDER_IDS[o] for o in orders])
    code = torch.empty((len(orders), n), dtype=torch.int64, device=grid_coord.device)
    lib.check(lib.ptv3_sfc_encode(_p(grid_coord), int(grid_coord.dtype == torch.int64), _p(batch), n, int(depth),
                                  ids, len(orders), _p(code), _stream()), "ptv3_sfc_encode")
    return code


def argsort_codes(code, end_bit):
    """(order, inverse), both (k, n) int64: stable argsort of every row + its inverse permutation."""
    _chk(code, "code", torch.int64, 2)
    k, n = code.shape
    order = torch.empty_like(code)
    inverse = torch.empty_like(code)
    ws_bytes = lib.ptv3_argsort_workspace_bytes(k, n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=code.device)
    lib.check(lib.ptv3_argsort_i64(_p(code), k, n, int(end_bit), _p(order), _p(inverse), _p(ws), ws_bytes,
                                   _stream()), "ptv3_argsort_i64")
    return order, inverse


def plan_sizes(offset_host, patch):
    """(n_pad, windows, ragged, sum of len^2) of the pad plan (v3m1_base.py:114-170): a scene longer than `patch` is
    padded to a multiple of it, a shorter one stays ONE short window (ragged; only with enable_flash=True)."""
    prev, n_pad, nwin, ragged, sq = 0, 0, 0, False, 0
    for o in offset_host:
        cnt = o - prev
        prev = o
        w = (cnt + patch - 1) // patch
        nwin += w
        if cnt < patch:
            ragged = True
            n_pad += cnt
            sq += cnt * cnt
        else:
            n_pad += w * patch
            sq += w * patch * patch
    return n_pad, nwin, ragged, sq


def pad_plan(offset, offset_host, patch):
    """pad, unpad, cu_seqlens of SerializedAttention.get_padding_and_inverse (v3m1_base.py:114-170)."""
    _chk(offset, "offset", torch.int64, 1)
    n_pad, nwin, _, _ = plan_sizes(offset_host, patch)
    n = int(offset_host[-1])
    pad = torch.empty(n_pad, dtype=torch.int64, device=offset.device)
    unpad = torch.empty(n, dtype=torch.int64, device=offset.device)
    cu = torch.empty(nwin + 1, dtype=torch.int32, device=offset.device)
    lib.check(lib.ptv3_pad_plan(_p(offset), len(offset_host), n, n_pad, int(patch), _p(pad), _p(unpad), _p(cu),
                                _stream()), "ptv3_pad_plan")
    return pad, unpad, cu


def window_maps(order, inverse, pad, unpad):
    for t, nm in ((order, "order"), (inverse, "inverse"), (pad, "pad"), (unpad, "unpad")):
        _chk(t, nm, torch.int64, 1)
    n, n_pad = order.shape[0], pad.shape[0]
    wo = torch.empty(n_pad, dtype=torch.int32, device=order.device)
    wi = torch.empty(n, dtype=torch.int32, device=order.device)
    lib.check(lib.ptv3_window_maps(_p(order), _p(inverse), _p(pad), _p(unpad), n, n_pad, _p(wo), _p(wi), _stream()),
              "ptv3_window_maps")
    return wo, wi


def window_plan(order, inverse, offset, offset_host, patch, with_cu=False):
    """(win_order (k, n_pad), win_inverse (k, n)) int32 for all k orders in one launch [+ cu_seqlens (windows+1)]."""
    _chk(order, "order", torch.int64, 2)
    _chk(inverse, "inverse", torch.int64, 2)
    _chk(offset, "offset", torch.int64, 1)
    k, n = order.shape
    n_pad, nwin, _, _ = plan_sizes(offset_host, patch)
    wo = torch.empty((k, n_pad), dtype=torch.int32, device=order.device)
    wi = torch.empty((k, n), dtype=torch.int32, device=order.device)
    cu = torch.empty(nwin + 1, dtype=torch.int32, device=order.device) if with_cu else None
    lib.check(lib.ptv3_window_plan(_p(order), _p(inverse), _p(offset), len(offset_host), k, n, n_pad, int(patch),
                                   _p(wo), _p(wi), _p(cu), _stream()), "ptv3_window_plan")
    return (wo, wi, cu) if with_cu else (wo, wi)


def window_attention(qkv, win_order, win_inverse, heads, patch, scale, rpe_bias=None):
    """softmax(scale q k^T) v per window with gather/scatter fused (v3m1_base.py:188-216)."""
    _chk(qkv, "qkv", (torch.float32, torch.bfloat16), 2)
    _chk(win_order, "win_order", torch.int32, 1)
    _chk(win_inverse, "win_inverse", torch.int32, 1)
    _chk(rpe_bias, "rpe_bias", torch.float32)
    n, c3 = qkv.shape
    c = c3 // 3
    if win_inverse.shape[0] != n or c * 3 != c3:
        raise RuntimeError("window_attention: shape mismatch")
    n_pad = win_order.shape[0]
    if rpe_bias is not None and rpe_bias.numel() != (n_pad // patch) * heads * patch * patch:
        raise RuntimeError("window_attention: rpe_bias must be (n_pad/patch, heads, patch, patch)")
    out = torch.empty((n, c), dtype=qkv.dtype, device=qkv.device)
    lib.check(lib.ptv3_window_attn_fwd(_p(qkv), _p(win_order), _p(win_inverse), _p(out), n, n_pad, c, int(heads),
                                       int(patch), float(scale), _p(rpe_bias), _dt(qkv), _stream()),
              "ptv3_window_attn_fwd")
    return out


def window_attention_varlen(qkv, win_order, win_inverse, cu_seqlens, heads, max_seqlen, scale, sum_len_sq=0.0):
    """The enable_flash=True call site (v3m1_base.py:207-215): flash_attn_varlen_qkvpacked_func semantics over ragged
    windows [cu_seqlens[w], cu_seqlens[w+1]) of at most max_seqlen padded slots, gather / scatter fused."""
    _chk(qkv, "qkv", (torch.float32, torch.bfloat16), 2)
    _chk(win_order, "win_order", torch.int32, 1)
    _chk(win_inverse, "win_inverse", torch.int32, 1)
    _chk(cu_seqlens, "cu_seqlens", torch.int32, 1)
    n, c3 = qkv.shape
    c = c3 // 3
    if win_inverse.shape[0] != n or c * 3 != c3 or cu_seqlens.numel() < 2:
        raise RuntimeError("window_attention_varlen: shape mismatch")
    out = torch.empty((n, c), dtype=qkv.dtype, device=qkv.device)
    lib.check(lib.ptv3_window_attn_varlen_fwd(_p(qkv), _p(win_order), _p(win_inverse), _p(cu_seqlens),
                                              cu_seqlens.numel() - 1, _p(out), n, win_order.shape[0], c, int(heads),
                                              int(max_seqlen), float(scale), float(sum_len_sq), _dt(qkv), _stream()),
              "ptv3_window_attn_varlen_fwd")
    return out


def window_attention_any(qkv, win_order, win_inverse, heads, patch, scale, cu_seqlens=None, sum_len_sq=0.0):
    """Uniform windows (cu_seqlens None) or ragged ones (the pad plan's cu_seqlens)."""
    if cu_seqlens is None:
        return window_attention(qkv, win_order, win_inverse, heads, patch, scale)
    return window_attention_varlen(qkv, win_order, win_inverse, cu_seqlens, heads, patch, scale, sum_len_sq)


def window_attention_train(qkv, win_order, win_inverse, heads, patch, scale, cu_seqlens=None, sum_len_sq=0.0):
    """The training forward: window_attention_any() that also returns the log-sum-exp rows (n_pad, heads) fp32 the
    backward would otherwise recompute."""
    _chk(qkv, "qkv", (torch.float32, torch.bfloat16), 2)
    _chk(win_order, "win_order", torch.int32, 1)
    _chk(win_inverse, "win_inverse", torch.int32, 1)
    _chk(cu_seqlens, "cu_seqlens", torch.int32, 1)
    n, c3 = qkv.shape
    c = c3 // 3
    if win_inverse.shape[0] != n:
        raise RuntimeError("window_attention_train: shape mismatch")
    out = torch.empty((n, c), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((win_order.shape[0], int(heads)), dtype=torch.float32, device=qkv.device)
    nwin = cu_seqlens.numel() - 1 if cu_seqlens is not None else 0
    lib.check(lib.ptv3_window_attn_train_fwd(_p(qkv), _p(win_order), _p(win_inverse), _p(cu_seqlens), nwin, _p(out),
                                             _p(lse), n, win_order.shape[0], c, int(heads), int(patch), float(scale),
                                             float(sum_len_sq), _dt(qkv), _stream()), "ptv3_window_attn_train_fwd")
    return out, lse


def window_attention_train_bwd(qkv, out, dout, lse, win_order, win_inverse, heads, patch, scale, cu_seqlens=None):
    _chk(qkv, "qkv", (torch.float32, torch.bfloat16), 2)
    _chk(out, "out", qkv.dtype, 2)
    _chk(dout, "dout", qkv.dtype, 2)
    _chk(lse, "lse", torch.float32, 2)
    _chk(cu_seqlens, "cu_seqlens", torch.int32, 1)
    n, c3 = qkv.shape
    c = c3 // 3
    n_pad = win_order.shape[0]
    if tuple(lse.shape) != (n_pad, int(heads)):
        raise RuntimeError("window_attention_train_bwd: lse must be (n_pad, heads)")
    dqkv = torch.empty_like(qkv)
    nb = lib.ptv3_window_attn_bwd_workspace_bytes(n, n_pad, c, int(heads), _dt(qkv))
    ws = _ws(nb, qkv.device)
    nwin = cu_seqlens.numel() - 1 if cu_seqlens is not None else 0
    lib.check(lib.ptv3_window_attn_train_bwd(_p(qkv), _p(out), _p(dout), _p(lse), _p(win_order), _p(win_inverse),
                                             _p(cu_seqlens), nwin, _p(dqkv), n, n_pad, c, int(heads), int(patch),
                                             float(scale), _dt(qkv), _p(ws), nb, _stream()), "ptv3_window_attn_train_bwd")
    return dqkv


def window_attention_drop(qkv, win_order, win_inverse, heads, patch, scale, p_drop, seed, cu_seqlens=None):
    """Training forward with attention dropout (:203 / :211): softmax over all pairs, kept pairs / (1 - p_drop) into the
    value sum; the keep mask is a hash of (query slot, head, key slot, seed) - see drop_keep_mask()."""
    _chk(qkv, "qkv", (torch.float32, torch.bfloat16), 2)
    _chk(win_order, "win_order", torch.int32, 1)
    _chk(win_inverse, "win_inverse", torch.int32, 1)
    _chk(cu_seqlens, "cu_seqlens", torch.int32, 1)
    n, c3 = qkv.shape
    c = c3 // 3
    if win_inverse.shape[0] != n:
        raise RuntimeError("window_attention_drop: shape mismatch")
    out = torch.empty((n, c), dtype=qkv.dtype, device=qkv.device)
    nwin = cu_seqlens.numel() - 1 if cu_seqlens is not None else 0
    lib.check(lib.ptv3_window_attn_drop_fwd(_p(qkv), _p(win_order), _p(win_inverse), _p(cu_seqlens), nwin, _p(out), n,
                                            win_order.shape[0], c, int(heads), int(patch), float(scale), float(p_drop),
                                            int(seed) & 0xFFFFFFFF, _dt(qkv), _stream()), "ptv3_window_attn_drop_fwd")
    return out


def window_attention_drop_bwd(qkv, out, dout, win_order, win_inverse, heads, patch, scale, p_drop, seed, cu_seqlens=None):
    _chk(qkv, "qkv", (torch.float32, torch.bfloat16), 2)
    _chk(out, "out", qkv.dtype, 2)
    _chk(dout, "dout", qkv.dtype, 2)
    _chk(cu_seqlens, "cu_seqlens", torch.int32, 1)
    n, c3 = qkv.shape
    c = c3 // 3
    n_pad = win_order.shape[0]
    dqkv = torch.empty_like(qkv)
    nb = lib.ptv3_window_attn_bwd_workspace_bytes(n, n_pad, c, int(heads), _dt(qkv))
    ws = _ws(nb, qkv.device)
    nwin = cu_seqlens.numel() - 1 if cu_seqlens is not None else 0
    lib.check(lib.ptv3_window_attn_drop_bwd(_p(qkv), _p(out), _p(dout), _p(win_order), _p(win_inverse), _p(cu_seqlens),
                                            nwin, _p(dqkv), n, n_pad, c, int(heads), int(patch), float(scale),
                                            float(p_drop), int(seed) & 0xFFFFFFFF, _dt(qkv), _p(ws), nb, _stream()),
              "ptv3_window_attn_drop_bwd")
    return dqkv


def drop_keep_mask(slot, head, key, heads, seed, p_drop):
    """Host restatement of csrc/common.h drop_keep: numpy integer arrays (broadcastable) of padded query slots, heads and
    key slots inside the window -> boolean keep mask.  For tests and for anyone who needs the mask a call used."""
    import numpy as np
    idv = ((np.asarray(slot, np.uint64) * np.uint64(heads) + np.asarray(head, np.uint64)) << np.uint64(14)) | np.asarray(key, np.uint64)
    lo = (idv & np.uint64(0xFFFFFFFF)).astype(np.uint64)
    hi = (idv >> np.uint64(32)).astype(np.uint64)
    m32 = np.uint64(0xFFFFFFFF)
    x = ((lo * np.uint64(0x9E3779B1)) & m32) ^ ((hi * np.uint64(0x85EBCA77)) & m32) ^ np.uint64(int(seed) & 0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & m32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & m32
    x ^= x >> np.uint64(16)
    thr = np.uint64(min(4294967295, int(float(np.float32(p_drop)) * 4294967296.0)))
    return x >= thr


def window_attention_rpe(qkv, win_order, win_inverse, heads, patch, scale, grid_coord, rpe_table, pos_bnd):
    """window_attention() + the RPE bias looked up from the (3*(2*pos_bnd+1), heads) table inside the kernel.
    Returns None when the window does not fit the resident-window kernel (caller falls back to the dense bias)."""
    _chk(qkv, "qkv", (torch.float32, torch.bfloat16), 2)
    _chk(win_order, "win_order", torch.int32, 1)
    _chk(win_inverse, "win_inverse", torch.int32, 1)
    _chk(grid_coord, "grid_coord", torch.int32, 2)
    _chk(rpe_table, "rpe_table", torch.float32, 2)
    n, c3 = qkv.shape
    c = c3 // 3
    if win_inverse.shape[0] != n or tuple(grid_coord.shape) != (n, 3) or \
            tuple(rpe_table.shape) != (3 * (2 * pos_bnd + 1), heads):
        raise RuntimeError("window_attention_rpe: shape mismatch")
    out = torch.empty((n, c), dtype=qkv.dtype, device=qkv.device)
    rc = lib.ptv3_window_attn_rpe_fwd(_p(qkv), _p(win_order), _p(win_inverse), _p(out), n, win_order.shape[0], c,
                                      int(heads), int(patch), float(scale), _p(grid_coord), _p(rpe_table),
                                      int(pos_bnd), _dt(qkv), _stream())
    if rc == 3:   # PTV3_ERR_UNSUPPORTED
        return None
    lib.check(rc, "ptv3_window_attn_rpe_fwd")
    return out


# ---------------------------------------------------------------------------------------------
# sparse conv support + the implicit GEMM
# ---------------------------------------------------------------------------------------------
def subm_neighbors(indices, ksize, table=None):
    """(nbr (n, ksize^3) int32, table) for unique (n,4) int32 [b,x,y,z] sites."""
    _chk(indices, "indices", torch.int32, 2)
    n = indices.shape[0]
    if table is None:
        slots = lib.ptv3_subm_table_slots(n)
        table = torch.empty(slots * 12, dtype=torch.uint8, device=indices.device)
        lib.check(lib.ptv3_subm_build_table(_p(indices), n, _p(table), slots, _stream()), "ptv3_subm_build_table")
    slots = table.numel() // 12
    nbr = torch.empty((n, ksize ** 3), dtype=torch.int32, device=indices.device)
    lib.check(lib.ptv3_subm_neighbors(_p(indices), n, _p(table), slots, int(ksize), _p(nbr), _stream()),
              "ptv3_subm_neighbors")
    return nbr, table


def gemm(x, w, bias=None, nbr=None, kvol=1, row_order=None, bn_scale=None, bn_shift=None, act=ACT_NONE,
         res=None, res_index=None, dual=False, m=None):
    """out = epi(sum_d sum_c w[o][d][c] x[nbr[i][d]][c]); see ptv3_gemm in include/ptv3_hip.h.

    Returns out, or (out_pre_residual, out_with_residual) when dual=True."""
    _chk(x, "x", (torch.float32, torch.bfloat16), 2)
    _chk(w, "w", x.dtype)
    for t, nm in ((bias, "bias"), (bn_scale, "bn_scale"), (bn_shift, "bn_shift")):
        _chk(t, nm, torch.float32, 1)
    _chk(nbr, "nbr", torch.int32, 2)
    _chk(row_order, "row_order", torch.int32, 1)
    _chk(res, "res", x.dtype, 2)
    _chk(res_index, "res_index", torch.int32, 1)
    cin = x.shape[1]
    cout = w.shape[0]
    if w.numel() != cout * kvol * cin:
        raise RuntimeError(f"gemm: weight has {w.numel()} elements, expected {cout}x{kvol}x{cin}")
    if m is None:
        m = nbr.shape[0] if nbr is not None else x.shape[0]
    if nbr is not None and (nbr.shape[0] != m or nbr.shape[1] != kvol):
        raise RuntimeError("gemm: neighbour table shape mismatch")
    for t in (bias, bn_scale, bn_shift):
        if t is not None and t.numel() != cout:
            raise RuntimeError("gemm: epilogue vector length != cout")
    if res is not None:
        if res.shape[1] != cout or (res_index is None and res.shape[0] != m):
            raise RuntimeError("gemm: residual shape mismatch")
        if res_index is not None and res_index.shape[0] != m:
            raise RuntimeError("gemm: res_index length != m")
    if row_order is not None and row_order.shape[0] != m:
        raise RuntimeError("gemm: row_order length != m")
    out = torch.empty((m, cout), dtype=x.dtype, device=x.device)
    out2 = torch.empty_like(out) if dual else None
    dt = _dt(x)
    ws_bytes = lib.ptv3_gemm_workspace_bytes(m, cin, cout, int(kvol), dt)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None
    lib.check(lib.ptv3_gemm(_p(x), _p(w), _p(out), m, cin, cout, int(kvol), _p(nbr), _p(row_order), _p(bias),
                            _p(bn_scale), _p(bn_shift), int(act), _p(res), _p(res_index), _p(out2), dt, _p(ws),
                            ws_bytes, _stream()), "ptv3_gemm")
    return (out, out2) if dual else out


def block_fusable(c, hidden, dtype, m=0):
    """0: no fused block kernels; 1: wave-local register chain (chain_permute'd weights); 2: workgroup-cooperative
    (natural weights)."""
    return int(lib.ptv3_block_fusable(int(c), int(hidden), _DT[dtype], int(m)))


def chain_permute(w, dtype):
    """Input-channel order the register-chained GEMMs of the fused block kernels expect (bf16: inside every
    32-chunk [0-3,16-19,4-7,20-23,8-11,24-27,12-15,28-31]; fp32: unchanged)."""
    if dtype == torch.float32:
        return w
    k = w.shape[1]
    assert k % 32 == 0
    base = [(4 * g + e) if e < 4 else (16 + 4 * g + e - 4) for g in range(4) for e in range(8)]
    perm = torch.tensor([32 * q + b for q in range(k // 32) for b in base], device=w.device)
    return w.index_select(1, perm).contiguous()


def conv_slabs(x, w, nbr, kvol, row_order=None):
    """Sparse conv left as split-K fp32 slabs (no bias): (slab tensor, splits) or None when the shape does not split."""
    m, cin, cout = nbr.shape[0], x.shape[1], w.shape[0]
    dt = _dt(x)
    splits = lib.ptv3_gemm_splits(m, cin, cout, int(kvol), dt)
    if splits <= 1:
        return None
    ws_bytes = lib.ptv3_gemm_workspace_bytes(m, cin, cout, int(kvol), dt)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    lib.check(lib.ptv3_gemm(_p(x), _p(w), None, m, cin, cout, int(kvol), _p(nbr), _p(row_order), None, None, None, 0,
                            None, None, None, dt, _p(ws), ws_bytes, _stream()), "ptv3_gemm(slabs)")
    return ws, splits


def rows_linear_rows():
    """row count from which the callers prefer ptv3_rows_linear to LayerNorm + tiled GEMM launches (PTV3_ROWS_MIN)"""
    import os
    return int(os.environ.get("PTV3_ROWS_MIN", "0"))


def rows_linear_capable(c, cout, dtype, m):
    return bool(lib.ptv3_rows_linear_capable(int(c), int(cout), _DT[dtype], int(m)))


def rows_linear(x, w, bias, act=ACT_NONE, res=None, ln=None, ln0=None, shortcut=None, eps=1e-5):
    """out = act(prologue(x) @ w^T + bias) [+ res] (see ptv3_rows_linear).  ln = (gamma, beta) of the LayerNorm in front
    of the GEMM; ln0 = (gamma, beta) + `shortcut`: f1 = LayerNorm(x; ln0) + shortcut first -> returns (f1, out)."""
    _chk(x, "x", (torch.float32, torch.bfloat16), 2)
    _chk(w, "w", x.dtype, 2)
    m, c = x.shape
    cout = w.shape[0]
    out = torch.empty((m, cout), dtype=x.dtype, device=x.device)
    f1 = torch.empty_like(x) if ln0 is not None else None
    g0, b0 = ln0 if ln0 is not None else (None, None)
    g1, b1 = ln if ln is not None else (None, None)
    lib.check(lib.ptv3_rows_linear(_p(x), _p(shortcut), _p(g0), _p(b0), _p(g1), _p(b1), _p(w), _p(bias), int(act), _p(res),
                                   _p(f1), _p(out), m, c, cout, float(eps), _dt(x), _stream()), "ptv3_rows_linear")
    return (f1, out) if ln0 is not None else out


def block_head(x, slab, splits, conv_bias, shortcut, g0, b0, g1, b1, wqkv, bqkv, eps):
    """f1, qkv of the fused head (see ptv3_block_head)."""
    _chk(shortcut, "shortcut", (torch.float32, torch.bfloat16), 2)
    m, c = shortcut.shape
    f1 = torch.empty_like(shortcut)
    qkv = torch.empty((m, 3 * c), dtype=shortcut.dtype, device=shortcut.device)
    lib.check(lib.ptv3_block_head(_p(x), _p(slab), int(splits), _p(conv_bias), _p(shortcut), _p(g0), _p(b0), _p(g1),
                                  _p(b1), _p(wqkv), _p(bqkv), _p(f1), _p(qkv), m, c, float(eps), _dt(shortcut),
                                  _stream()), "ptv3_block_head")
    return f1, qkv


def block_tail(attn, f1, wproj, bproj, g2, b2, w1, bias1, w2, bias2, eps):
    _chk(attn, "attn", (torch.float32, torch.bfloat16), 2)
    _chk(f1, "f1", attn.dtype, 2)
    m, c = attn.shape
    out = torch.empty_like(attn)
    lib.check(lib.ptv3_block_tail(_p(attn), _p(f1), _p(wproj), _p(bproj), _p(g2), _p(b2), _p(w1), _p(bias1), _p(w2),
                                  _p(bias2), _p(out), m, c, int(w1.shape[0]), float(eps), _dt(attn), _stream()),
              "ptv3_block_tail")
    return out


def mlp2_fusable(cin, hidden, cout, dtype):
    return bool(lib.ptv3_mlp2_fusable(int(cin), int(hidden), int(cout), _DT[dtype]))


def mlp2_weight2(w2, dtype):
    """(cout, hidden) parameter -> the (16 | 32 | 64, hidden) chain-permuted matrix ptv3_mlp2 reads (rows beyond
    cout are zero; the kernel works on 1, 2 or 4 output tiles)."""
    w = w2.detach().to(dtype)
    rows = 16 if w.shape[0] <= 16 else 32 if w.shape[0] <= 32 else 64
    pad = rows - w.shape[0]
    if pad:
        w = torch.nn.functional.pad(w, (0, 0, 0, pad))
    return chain_permute(w.contiguous(), dtype)


def mlp2(x, w1, b1, s1, t1, act, w2p, b2, cout, out_f32=True):
    """act((x w1^T + b1) * s1 + t1) w2^T + b2 with the hidden layer in registers; w2p from mlp2_weight2."""
    _chk(x, "x", (torch.float32, torch.bfloat16), 2)
    _chk(w1, "w1", x.dtype, 2)
    _chk(w2p, "w2p", x.dtype, 2)
    for t, nm in ((b1, "b1"), (s1, "s1"), (t1, "t1"), (b2, "b2")):
        _chk(t, nm, torch.float32, 1)
    m, cin = x.shape
    hidden = w1.shape[0]
    if w1.shape[1] != cin or w2p.shape[1] != hidden or w2p.shape[0] < cout or w2p.shape[0] % 16:
        raise RuntimeError("mlp2: shape mismatch")
    out = torch.empty((m, cout), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    lib.check(lib.ptv3_mlp2(_p(x), _p(w1), _p(b1), _p(s1), _p(t1), int(act), _p(w2p), _p(b2), _p(out), int(out_f32), m,
                            cin, hidden, int(cout), _dt(x), _stream()), "ptv3_mlp2")
    return out


def layernorm(x, gamma, beta, eps=1e-5, res=None, gamma2=None, beta2=None):
    """y = LN(x)*g+b (+res); with gamma2/beta2 also returns y2 = LN(y)*g2+b2."""
    _chk(x, "x", (torch.float32, torch.bfloat16), 2)
    _chk(gamma, "gamma", torch.float32, 1)
    _chk(beta, "beta", torch.float32, 1)
    _chk(res, "res", x.dtype, 2)
    _chk(gamma2, "gamma2", torch.float32, 1)
    _chk(beta2, "beta2", torch.float32, 1)
    m, c = x.shape
    if gamma.numel() != c or beta.numel() != c or (res is not None and res.shape != x.shape):
        raise RuntimeError("layernorm: shape mismatch")
    y = torch.empty_like(x)
    y2 = torch.empty_like(x) if gamma2 is not None else None
    lib.check(lib.ptv3_layernorm(_p(x), _p(gamma), _p(beta), _p(res), _p(y), _p(gamma2), _p(beta2), _p(y2), m, c,
                                 float(eps), _dt(x), _stream()), "ptv3_layernorm")
    return (y, y2) if gamma2 is not None else y


def layernorm_slabs(slab, splits, m, c, slab_bias, dtype, gamma, beta, eps=1e-5, res=None, gamma2=None, beta2=None):
    """layernorm() whose input is still the split-K slabs of conv_slabs() (its workspace tensor, `splits` slabs of
    (m, c) fp32): x = dtype(sum slabs + slab_bias)."""
    _chk(slab, "slab", (torch.uint8, torch.float32))
    _chk(slab_bias, "slab_bias", torch.float32, 1)
    _chk(res, "res", dtype, 2)
    if slab.numel() * slab.element_size() < splits * m * c * 4:
        raise RuntimeError("layernorm_slabs: slab buffer too small")
    y = torch.empty((m, c), dtype=dtype, device=slab.device)
    y2 = torch.empty_like(y) if gamma2 is not None else None
    lib.check(lib.ptv3_layernorm_slabs(_p(slab), int(splits), _p(slab_bias), _p(gamma), _p(beta), _p(res), _p(y),
                                       _p(gamma2), _p(beta2), _p(y2), m, c, float(eps), _DT[dtype], _stream()),
              "ptv3_layernorm_slabs")
    return (y, y2) if gamma2 is not None else y


def affine_act(x, scale, shift, act):
    _chk(x, "x", (torch.float32, torch.bfloat16), 2)
    _chk(scale, "scale", torch.float32, 1)
    _chk(shift, "shift", torch.float32, 1)
    m, c = x.shape
    y = torch.empty_like(x)
    lib.check(lib.ptv3_affine_act(_p(x), _p(scale), _p(shift), int(act), _p(y), m, c, _dt(x), _stream()),
              "ptv3_affine_act")
    return y


def cast(x, dtype):
    if x.dtype == dtype:
        return x
    _chk(x, "x", (torch.float32, torch.bfloat16))
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    lib.check(lib.ptv3_cast(_p(x), _dt(x), _p(y), _DT[dtype], x.numel(), _stream()), "ptv3_cast")
    return y


# ---------------------------------------------------------------------------------------------
# serialized pooling
# ---------------------------------------------------------------------------------------------
def pool_segments(code0, order0, shift_bits, batch=None, num_scenes=0):
    """cluster (n) int64, seg_start (n_out+1) int32, n_out (python int; ONE host sync, as torch.unique).
    With batch: also the pooled Point's cumulative offsets, returned as (device tensor, host list)."""
    _chk(code0, "code0", torch.int64, 1)
    _chk(order0, "order0", torch.int64, 1)
    _chk(batch, "batch", torch.int64, 1)
    n = code0.shape[0]
    pooled_offset = torch.empty(num_scenes, dtype=torch.int64, device=code0.device) if batch is not None else None
    cluster = torch.empty(n, dtype=torch.int64, device=code0.device)
    seg_start = torch.empty(n + 1, dtype=torch.int32, device=code0.device)
    n_out = torch.empty(1, dtype=torch.int32, device=code0.device)
    ws_bytes = lib.ptv3_pool_workspace_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=code0.device)
    lib.check(lib.ptv3_pool_segments(_p(code0), _p(order0), n, int(shift_bits), _p(batch), _p(cluster),
                                     _p(seg_start), _p(n_out), _p(pooled_offset), _p(ws), ws_bytes, _stream()),
              "ptv3_pool_segments")
    if batch is not None:
        host = [int(v) for v in pooled_offset.tolist()]  # the one sync; n_out is the last entry
        return cluster, seg_start[:host[-1] + 1], host[-1], pooled_offset, host
    cnt = int(n_out.item())
    return cluster, seg_start[:cnt + 1], cnt


def pool_reduce(feat, coord, grid_coord, batch, code, order0, seg_start, n_out, pooling_depth, bn_scale=None,
                bn_shift=None, act=ACT_NONE, row_perm=None):
    _chk(feat, "feat", (torch.float32, torch.bfloat16), 2)
    _chk(coord, "coord", torch.float32, 2)
    _chk(grid_coord, "grid_coord", torch.int64, 2)
    _chk(batch, "batch", torch.int64, 1)
    _chk(code, "code", torch.int64, 2)
    _chk(order0, "order0", torch.int64, 1)
    _chk(seg_start, "seg_start", torch.int32, 1)
    n, c = feat.shape
    k = code.shape[0]
    dev = feat.device
    feat_out = torch.empty((n_out, c), dtype=feat.dtype, device=dev)
    coord_out = torch.empty((n_out, 3), dtype=torch.float32, device=dev) if coord is not None else None
    grid_out = torch.empty((n_out, 3), dtype=torch.int64, device=dev)
    batch_out = torch.empty(n_out, dtype=torch.int64, device=dev)
    code_out = torch.empty((k, n_out), dtype=torch.int64, device=dev)
    lib.check(lib.ptv3_pool_reduce(_p(feat), _p(coord), _p(grid_coord), _p(batch), _p(code), k, _p(order0),
                                   _p(seg_start), n, n_out, c, int(pooling_depth), _p(bn_scale), _p(bn_shift),
                                   int(act), (ctypes.c_int * k)(*row_perm) if row_perm is not None else None,
                                   _p(feat_out), _p(coord_out), _p(grid_out), _p(batch_out), _p(code_out),
                                   _dt(feat), _stream()), "ptv3_pool_reduce")
    return feat_out, coord_out, grid_out, batch_out, code_out


def pool_geometry(coord, grid_coord, batch, code, order0, seg_start, n_out, pooling_depth, row_perm=None):
    """The geometry half of ptv3_pool_reduce alone (no feature matrix): coord mean, grid >> depth, batch and the
    pooled codes of every cluster - what a training forward can compute for ALL pooling stages before the first
    feature kernel (the host syncs of pool_segments then fall on a near-empty queue)."""
    _chk(coord, "coord", torch.float32, 2)
    _chk(grid_coord, "grid_coord", torch.int64, 2)
    _chk(batch, "batch", torch.int64, 1)
    _chk(code, "code", torch.int64, 2)
    _chk(order0, "order0", torch.int64, 1)
    _chk(seg_start, "seg_start", torch.int32, 1)
    k, n = code.shape
    dev = code.device
    coord_out = torch.empty((n_out, 3), dtype=torch.float32, device=dev) if coord is not None else None
    grid_out = torch.empty((n_out, 3), dtype=torch.int64, device=dev)
    batch_out = torch.empty(n_out, dtype=torch.int64, device=dev)
    code_out = torch.empty((k, n_out), dtype=torch.int64, device=dev)
    lib.check(lib.ptv3_pool_reduce(None, _p(coord), _p(grid_coord), _p(batch), _p(code), k, _p(order0), _p(seg_start), n,
                                   n_out, 4, int(pooling_depth), None, None, ACT_NONE,
                                   (ctypes.c_int * k)(*row_perm) if row_perm is not None else None, None, _p(coord_out),
                                   _p(grid_out), _p(batch_out), _p(code_out), PTV3_F32, _stream()), "ptv3_pool_reduce")
    return coord_out, grid_out, batch_out, code_out


# ---------------------------------------------------------------------------------------------
# training: backward kernels
# ---------------------------------------------------------------------------------------------
_F = (torch.float32, torch.bfloat16)


def _ws(nbytes, dev):
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=dev)


def gemm_tn(dy, x, nbr=None, kvol=1, with_bias=False):
    """dW (cout, kvol*cin) fp32 = dy^T . gather(x): weight gradient of a Linear (kvol=1) or SubMConv3d.
    with_bias: -> (dW, db) with db (cout) fp32 = column sums of dy from the same pass."""
    _chk(dy, "dy", _F, 2)
    _chk(x, "x", dy.dtype, 2)
    _chk(nbr, "nbr", torch.int32, 2)
    m, cout = dy.shape
    cin = x.shape[1]
    if (nbr is None) != (kvol == 1) or (nbr is not None and tuple(nbr.shape) != (m, kvol)) or \
            (nbr is None and x.shape[0] != m):
        raise RuntimeError("gemm_tn: shape mismatch")
    dw = torch.empty((cout, kvol * cin), dtype=torch.float32, device=dy.device)
    db = torch.empty(cout, dtype=torch.float32, device=dy.device) if with_bias else None
    nb = lib.ptv3_gemm_tn_workspace_bytes(m, cout, cin, int(kvol))
    ws = _ws(nb, dy.device)
    lib.check(lib.ptv3_gemm_tn(_p(dy), _p(x), _p(nbr), _p(dw), _p(db), m, cout, cin, int(kvol), _dt(dy), _p(ws), nb,
                               _stream()), "ptv3_gemm_tn")
    return (dw, db) if with_bias else dw


def col_reduce(a, b=None, mu=None, rs=None, mode=0, mu_scale=1.0):
    """fp32 column sums over rows: mode 0 (c) sum a; 1 (2, c) sum a, sum a^2; 2 (2, c) sum a, sum a*(b-mu)*rs;
    3 (2, c) sum a, sum (a-mu)^2.  mu is multiplied by mu_scale (column sums and 1 / m instead of a mean tensor)."""
    _chk(a, "a", _F, 2)
    _chk(b, "b", a.dtype, 2)
    _chk(mu, "mu", torch.float32, 1)
    _chk(rs, "rs", torch.float32, 1)
    m, c = a.shape
    out = torch.empty((1 if mode == 0 else 2, c), dtype=torch.float32, device=a.device)
    nb = lib.ptv3_col_reduce_workspace_bytes(m, c)
    ws = _ws(nb, a.device)
    lib.check(lib.ptv3_col_reduce(_p(a), _p(b), _p(mu), _p(rs), float(mu_scale), int(mode), _p(out), m, c, _dt(a), _p(ws),
                                  nb, _stream()), "ptv3_col_reduce")
    return out[0] if mode == 0 else out


def bn_finalize(total, centred_sq, m, weight, bias, running_mean, running_var, momentum, eps):
    """-> mean, rstd, scale, shift (c) fp32 of a BatchNorm1d training forward; running buffers updated in place."""
    for t, nm in ((total, "total"), (centred_sq, "centred_sq"), (weight, "weight"), (bias, "bias"),
                  (running_mean, "running_mean"), (running_var, "running_var")):
        _chk(t, nm, torch.float32, 1)
    c = total.shape[0]
    out = torch.empty((4, c), dtype=torch.float32, device=total.device)
    lib.check(lib.ptv3_bn_finalize(_p(total), _p(centred_sq), int(m), _p(weight), _p(bias), _p(running_mean),
                                   _p(running_var), float(momentum), float(eps), _p(out[0]), _p(out[1]), _p(out[2]),
                                   _p(out[3]), c, _stream()), "ptv3_bn_finalize")
    return out[0], out[1], out[2], out[3]


def bn_bwd_coeffs(sums, m, weight, rstd, mean):
    """-> ca, cb, cc (c) fp32: BatchNorm1d input gradient dx = ca dy + cb x + cc from the two batch sums of mode 2."""
    _chk(sums, "sums", torch.float32, 2)
    c = sums.shape[1]
    out = torch.empty((3, c), dtype=torch.float32, device=sums.device)
    lib.check(lib.ptv3_bn_bwd_coeffs(_p(sums), int(m), _p(weight), _p(rstd), _p(mean), _p(out[0]), _p(out[1]), _p(out[2]),
                                     c, _stream()), "ptv3_bn_bwd_coeffs")
    return out[0], out[1], out[2]


def layernorm_bwd(x, dy, gamma, eps, add=None):
    """-> dx (m, c) [+ add], dgamma (c), dbeta (c)."""
    _chk(x, "x", _F, 2)
    _chk(dy, "dy", x.dtype, 2)
    _chk(add, "add", x.dtype, 2)
    if add is not None and add.shape != x.shape:
        raise RuntimeError("layernorm_bwd: add must have the shape of x")
    _chk(gamma, "gamma", torch.float32, 1)
    m, c = x.shape
    dx = torch.empty_like(x)
    dgb = torch.empty((2, c), dtype=torch.float32, device=x.device)
    nb = lib.ptv3_col_reduce_workspace_bytes(m, c)
    ws = _ws(nb, x.device)
    lib.check(lib.ptv3_layernorm_bwd(_p(x), _p(dy), _p(add), _p(gamma), float(eps), _p(dx), _p(dgb), m, c, _dt(x), _p(ws), nb,
                                     _stream()), "ptv3_layernorm_bwd")
    return dx, dgb[0], dgb[1]


def act_bwd(dy, x, act, scale=None, shift=None):
    """dy * act'(x*scale + shift)."""
    _chk(dy, "dy", _F, 2)
    _chk(x, "x", dy.dtype, 2)
    _chk(scale, "scale", torch.float32, 1)
    _chk(shift, "shift", torch.float32, 1)
    m, c = x.shape
    dx = torch.empty_like(x)
    lib.check(lib.ptv3_act_bwd(_p(dy), _p(x), _p(scale), _p(shift), int(act), _p(dx), m, c, _dt(x), _stream()),
              "ptv3_act_bwd")
    return dx


def affine2(dy, x, ca, cb, cc):
    """ca*dy + cb*x + cc per column."""
    _chk(dy, "dy", _F, 2)
    _chk(x, "x", dy.dtype, 2)
    for t, nm in ((ca, "ca"), (cb, "cb"), (cc, "cc")):
        _chk(t, nm, torch.float32, 1)
    m, c = x.shape
    dx = torch.empty_like(x)
    lib.check(lib.ptv3_affine2(_p(dy), _p(x), _p(ca), _p(cb), _p(cc), _p(dx), m, c, _dt(x), _stream()), "ptv3_affine2")
    return dx


def pool_max(feat, order0, seg_start, n_out):
    """segment max over the members of each cluster (the feature half of ptv3_pool_reduce, no BN / act)."""
    _chk(feat, "feat", _F, 2)
    _chk(order0, "order0", torch.int64, 1)
    _chk(seg_start, "seg_start", torch.int32, 1)
    n, c = feat.shape
    out = torch.empty((n_out, c), dtype=feat.dtype, device=feat.device)
    lib.check(lib.ptv3_pool_reduce(_p(feat), None, None, None, None, 1, _p(order0), _p(seg_start), n, n_out, c, 0, None,
                                   None, ACT_NONE, None, _p(out), None, None, None, None, _dt(feat), _stream()),
              "ptv3_pool_reduce")
    return out


def pool_max_bwd(feat, dy, order0, seg_start):
    _chk(feat, "feat", _F, 2)
    _chk(dy, "dy", feat.dtype, 2)
    n_out, c = dy.shape
    dfeat = torch.empty_like(feat)
    lib.check(lib.ptv3_pool_max_bwd(_p(feat), _p(dy), _p(order0), _p(seg_start), n_out, c, _p(dfeat), _dt(feat),
                                    _stream()), "ptv3_pool_max_bwd")
    return dfeat


def segment_sum(dy, order0, seg_start, n_out):
    _chk(dy, "dy", _F, 2)
    c = dy.shape[1]
    out = torch.empty((n_out, c), dtype=dy.dtype, device=dy.device)
    lib.check(lib.ptv3_segment_sum(_p(dy), _p(order0), _p(seg_start), n_out, c, _p(out), _dt(dy), _stream()),
              "ptv3_segment_sum")
    return out


def window_attention_bwd(qkv, out, dout, win_order, win_inverse, heads, patch, scale, cu_seqlens=None):
    _chk(qkv, "qkv", _F, 2)
    _chk(cu_seqlens, "cu_seqlens", torch.int32, 1)
    _chk(out, "out", qkv.dtype, 2)
    _chk(dout, "dout", qkv.dtype, 2)
    _chk(win_order, "win_order", torch.int32, 1)
    _chk(win_inverse, "win_inverse", torch.int32, 1)
    n, c3 = qkv.shape
    c = c3 // 3
    n_pad = win_order.shape[0]
    if out.shape != (n, c) or dout.shape != (n, c) or win_inverse.shape[0] != n:
        raise RuntimeError("window_attention_bwd: shape mismatch")
    dqkv = torch.empty_like(qkv)
    nb = lib.ptv3_window_attn_bwd_workspace_bytes(n, n_pad, c, int(heads), _dt(qkv))
    ws = _ws(nb, qkv.device)
    if cu_seqlens is not None:
        lib.check(lib.ptv3_window_attn_varlen_bwd(_p(qkv), _p(out), _p(dout), _p(win_order), _p(win_inverse),
                                                  _p(cu_seqlens), cu_seqlens.numel() - 1, _p(dqkv), n, n_pad, c,
                                                  int(heads), int(patch), float(scale), _dt(qkv), _p(ws), nb,
                                                  _stream()), "ptv3_window_attn_varlen_bwd")
        return dqkv
    lib.check(lib.ptv3_window_attn_bwd(_p(qkv), _p(out), _p(dout), _p(win_order), _p(win_inverse), _p(dqkv), n, n_pad,
                                       c, int(heads), int(patch), float(scale), _dt(qkv), _p(ws), nb, _stream()),
              "ptv3_window_attn_bwd")
    return dqkv


def window_attention_rpe_bwd(qkv, out, dout, win_order, win_inverse, heads, patch, scale, grid_coord, rpe_table,
                             pos_bnd):
    """-> (dqkv (n, 3c), dtable (3*(2*pos_bnd+1), heads) fp32) of window_attention_rpe."""
    _chk(qkv, "qkv", _F, 2)
    _chk(out, "out", qkv.dtype, 2)
    _chk(dout, "dout", qkv.dtype, 2)
    _chk(win_order, "win_order", torch.int32, 1)
    _chk(win_inverse, "win_inverse", torch.int32, 1)
    _chk(grid_coord, "grid_coord", torch.int32, 2)
    _chk(rpe_table, "rpe_table", torch.float32, 2)
    n, c3 = qkv.shape
    c = c3 // 3
    n_pad = win_order.shape[0]
    if out.shape != (n, c) or dout.shape != (n, c) or win_inverse.shape[0] != n or tuple(grid_coord.shape) != (n, 3) \
            or tuple(rpe_table.shape) != (3 * (2 * pos_bnd + 1), heads):
        raise RuntimeError("window_attention_rpe_bwd: shape mismatch")
    dqkv = torch.empty_like(qkv)
    dtable = torch.empty_like(rpe_table)
    nb = lib.ptv3_window_attn_rpe_bwd_workspace_bytes(n, n_pad, c, int(heads), int(patch), int(pos_bnd), _dt(qkv))
    ws = _ws(nb, qkv.device)
    lib.check(lib.ptv3_window_attn_rpe_bwd(_p(qkv), _p(out), _p(dout), _p(win_order), _p(win_inverse), _p(grid_coord),
                                           _p(rpe_table), int(pos_bnd), _p(dqkv), _p(dtable), n, n_pad, c, int(heads),
                                           int(patch), float(scale), _dt(qkv), _p(ws), nb, _stream()),
              "ptv3_window_attn_rpe_bwd")
    return dqkv, dtable


# ---------------------------------------------------------------------------------------------
# GridSample (before the model)
# ---------------------------------------------------------------------------------------------
HASH_FNV, HASH_RAVEL = 0, 1


def grid_hash(coord, grid_size, hash_type=HASH_FNV):
    """-> grid_coord (n,3) int64 (minimum subtracted), min_max (6) int64, key (n) int64 holding the uint64 hash."""
    _chk(coord, "coord", torch.float32, 2)
    n = coord.shape[0]
    grid = torch.empty((n, 3), dtype=torch.int64, device=coord.device)
    mm = torch.empty(6, dtype=torch.int64, device=coord.device)
    key = torch.empty(n, dtype=torch.int64, device=coord.device)
    lib.check(lib.ptv3_grid_hash(_p(coord), n, float(grid_size), int(hash_type), _p(grid), _p(mm), _p(key), _stream()),
              "ptv3_grid_hash")
    return grid, mm, key


def voxel_unique(key):
    """np.argsort + np.unique(return_inverse, return_counts) of the voxel keys on the device:
    idx_sort (n) int64 (stable), inverse (n) int64 in ORIGINAL point order, seg_start (nvox+1) int32, nvox."""
    _chk(key, "key", torch.int64, 1)
    order, _ = argsort_codes(key.view(1, -1), 64)
    order = order[0].contiguous()
    cluster, seg_start, nvox = pool_segments(key, order, 0)
    return order, cluster, seg_start, nvox


# ---------------------------------------------------------------------------------------------
# Swin3D window partition + cRSE attention (row A19, parity unpinned: see oracle/swin3d.py)
# ---------------------------------------------------------------------------------------------
def swin_window_mapping(coords, stride, window_size, shift=0):
    """BasicLayer.get_window_mapping (swin3d_layers.py:746-795) of voxels `coords` (n,4) int32 [batch,x,y,z] at tensor
    stride `stride`, shifted by `shift` voxels (:826-840).  -> w_w_id (n) int64 and w_w_xyz (n,3) int64 in sorted
    order, w_sizes (W) int64, sort_idx (n) int64, inv_sort_idx (n) int64, w_start (W+1) int32.  One host sync (W)."""
    _chk(coords, "coords", torch.int32, 2)
    n = coords.shape[0]
    key = torch.empty(n, dtype=torch.int64, device=coords.device)
    bad = torch.empty(1, dtype=torch.int32, device=coords.device)
    lib.check(lib.ptv3_swin_window_keys(_p(coords), n, int(stride), int(window_size), int(shift), _p(key), _p(bad),
                                        _stream()), "ptv3_swin_window_keys")
    order, inverse = argsort_codes(key.view(1, -1), 61)
    order, inverse = order[0].contiguous(), inverse[0].contiguous()
    _, w_start, nwin = pool_segments(key, order, 9)
    if int(bad.item()):
        raise ValueError("swin_window_mapping: batch index outside 0..4095 or window coordinate outside -4096..4095")
    w_w_id = key[order] & 511
    ws = int(window_size)
    w_w_xyz = torch.stack([w_w_id // ws // ws, w_w_id // ws % ws, w_w_id % ws], dim=-1)
    w_sizes = (w_start[1:] - w_start[:-1]).long()
    return w_w_id, w_w_xyz, w_sizes, order, inverse, w_start


def swin_attention(q, k, v, q_table, k_table, v_table, table_offsets, n2n, w_start, n_crse, max_tokens):
    """SelfAttnAIOFunction forward (swin3d_layers.py:556-569): q (pre-scaled), k, v (n, H, D) in original voxel order,
    concatenated fp32 tables with `table_offsets` elements per signal axis, n2n / w_start / n_crse as
    swin_window_mapping and WindowAttention.forward build them.  -> (n, H, D) in original order."""
    _chk(q, "q", _F, 3)
    for name, t in (("k", k), ("v", v)):
        _chk(t, name, q.dtype, 3)
        if t.shape != q.shape:
            raise ValueError(f"swin_attention: {name} {tuple(t.shape)} vs q {tuple(q.shape)}")
    for name, t in (("q_table", q_table), ("k_table", k_table), ("v_table", v_table)):
        _chk(t, name, torch.float32, 1)
    _chk(n2n, "n2n", torch.int64, 1)
    _chk(w_start, "w_start", torch.int32, 1)
    _chk(n_crse, "n_crse", torch.float32, 2)
    n, heads, hd = q.shape
    axes = len(table_offsets)
    total = int(sum(int(t) for t in table_offsets))
    if n_crse.shape != (n, axes) or n2n.shape[0] != n:
        raise ValueError(f"swin_attention: n_crse {tuple(n_crse.shape)} / n2n {tuple(n2n.shape)} for {n} voxels, {axes} axes")
    if min(q_table.numel(), k_table.numel(), v_table.numel()) < total:
        raise ValueError("swin_attention: tables are shorter than sum(table_offsets)")
    out = torch.empty_like(q)
    offs = (ctypes.c_int32 * axes)(*[int(t) for t in table_offsets])
    if _PROFILING:   # the pair count is device data: read back only while the launch is being bracketed
        lens = (w_start[1:] - w_start[:-1]).double()
        lib.ptv3_profile_hint_flops(float((lens * lens).sum().item()) * heads * (1 + 3 * axes) * 2 * hd)
    lib.check(lib.ptv3_swin_attn_fwd(_p(q), _p(k), _p(v), _p(q_table), _p(k_table), _p(v_table), offs, axes, _p(n2n),
                                     _p(w_start), w_start.shape[0] - 1, _p(n_crse), _p(out), n, heads, hd,
                                     int(max_tokens), _dt(q), _stream()), "ptv3_swin_attn_fwd")
    return out


def swin_attention_bwd(q, k, v, dout, q_table, k_table, v_table, table_offsets, n2n, w_start, n_crse, max_tokens):
    """-> dq, dk, dv (n, H, D) in the dtype of q and dq_table, dk_table, dv_table (fp32, flat like the tables)."""
    _chk(q, "q", _F, 3)
    for name, t in (("k", k), ("v", v), ("dout", dout)):
        _chk(t, name, q.dtype, 3)
        if t.shape != q.shape:
            raise ValueError(f"swin_attention_bwd: {name} {tuple(t.shape)} vs q {tuple(q.shape)}")
    for name, t in (("q_table", q_table), ("k_table", k_table), ("v_table", v_table)):
        _chk(t, name, torch.float32, 1)
    _chk(n2n, "n2n", torch.int64, 1)
    _chk(w_start, "w_start", torch.int32, 1)
    _chk(n_crse, "n_crse", torch.float32, 2)
    n, heads, hd = q.shape
    axes = len(table_offsets)
    total = int(sum(int(t) for t in table_offsets))
    if n_crse.shape != (n, axes) or n2n.shape[0] != n or min(q_table.numel(), k_table.numel(), v_table.numel()) < total:
        raise ValueError("swin_attention_bwd: shapes of n_crse / n2n / tables do not fit")
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    dtab = torch.empty((3, total), dtype=torch.float32, device=q.device)
    offs = (ctypes.c_int32 * axes)(*[int(t) for t in table_offsets])
    lib.check(lib.ptv3_swin_attn_bwd(_p(q), _p(k), _p(v), _p(dout), _p(q_table), _p(k_table), _p(v_table), offs, axes,
                                     _p(n2n), _p(w_start), w_start.shape[0] - 1, _p(n_crse), _p(dq), _p(dk), _p(dv),
                                     _p(dtab[0]), _p(dtab[1]), _p(dtab[2]), n, heads, hd, int(max_tokens), _dt(q),
                                     _stream()), "ptv3_swin_attn_bwd")
    return dq, dk, dv, dtab[0], dtab[1], dtab[2]


# ---------------------------------------------------------------------------------------------
# keypoint aggregation (after the model)
# ---------------------------------------------------------------------------------------------
KP_ARGMAX, KP_WEIGHTED, KP_GT_MEAN, KP_GT_FIRST = 0, 1, 2, 3


def keypoint_aggregate(coord, pred, offset, mode, scale=None, centroid=None, thresh=0.5):
    """(B, K, 3) keypoints + (B, K) int32 aux per scene and keypoint; see ptv3_keypoint_aggregate."""
    _chk(coord, "coord", torch.float32, 2)
    _chk(pred, "pred", torch.float32, 3)
    _chk(offset, "offset", torch.int64, 1)
    _chk(scale, "scale", torch.float32, 1)
    _chk(centroid, "centroid", torch.float32, 2)
    n, k, four = pred.shape
    b = offset.shape[0]
    if four != 4 or coord.shape != (n, 3) or (scale is not None and scale.shape[0] != b) or \
            (centroid is not None and tuple(centroid.shape) != (b, 3)):
        raise RuntimeError("keypoint_aggregate: shape mismatch")
    kp = torch.empty((b, k, 3), dtype=torch.float32, device=coord.device)
    aux = torch.empty((b, k), dtype=torch.int32, device=coord.device)
    lib.check(lib.ptv3_keypoint_aggregate(_p(coord), _p(pred), _p(offset), b, k, _p(scale), _p(centroid), int(mode),
                                          float(thresh), _p(kp), _p(aux), _stream()), "ptv3_keypoint_aggregate")
    return kp, aux


# ---------------------------------------------------------------------------------------------
# pointops
# ---------------------------------------------------------------------------------------------
def knn_query(nsample, xyz, offset, new_xyz, new_offset):
    _chk(xyz, "xyz", torch.float32, 2)
    _chk(new_xyz, "new_xyz", torch.float32, 2)
    _chk(offset, "offset", torch.int32, 1)
    _chk(new_offset, "new_offset", torch.int32, 1)
    m = new_xyz.shape[0]
    _check_offsets("knn_query", offset, xyz.shape[0], new_offset, m)
    idx = torch.zeros((m, nsample), dtype=torch.int32, device=xyz.device)
    dist2 = torch.zeros((m, nsample), dtype=torch.float32, device=xyz.device)
    lib.check(lib.ptv3_knn_query(m, int(nsample), _p(xyz), _p(new_xyz), _p(offset), _p(new_offset),
                                 offset.shape[0], _p(idx), _p(dist2), _stream()), "ptv3_knn_query")
    return idx, dist2


def _scene_ids(offset, n):
    counts = torch.diff(offset.long(), prepend=offset.new_zeros(1).long())
    return torch.repeat_interleave(torch.arange(offset.shape[0], device=offset.device), counts, output_size=n)


def _check_offsets(what, offset, n, new_offset, m):
    """The kernels trust the scene ends: a wrong last entry walks past the coordinate arrays."""
    ends = torch.stack([offset[-1], new_offset[-1]]).tolist()
    if ends != [n, m] or offset.shape != new_offset.shape:
        raise ValueError(f"{what}: offsets end at {ends} for {n} candidates / {m} queries "
                         f"({offset.shape[0]} / {new_offset.shape[0]} scenes)")


def knn_query_cells(nsample, xyz, offset, new_xyz, new_offset, cell=None):
    """ptv3_knn_query_cells: the neighbours of knn_query by a walk over a uniform grid of edge `cell` (same unit as xyz;
    default: an estimate of the nsample-th neighbour distance from the candidates' bounding box, so that one or two
    shells of cells settle a query).  Rows ascend in (distance, index).  Host syncs: offsets check + grid extent, number
    of occupied cells."""
    _chk(xyz, "xyz", torch.float32, 2)
    _chk(new_xyz, "new_xyz", torch.float32, 2)
    _chk(offset, "offset", torch.int32, 1)
    _chk(new_offset, "new_offset", torch.int32, 1)
    n, m = xyz.shape[0], new_xyz.shape[0]
    dev = xyz.device
    idx = torch.full((m, nsample), -1, dtype=torch.int32, device=dev)
    dist2 = torch.full((m, nsample), 1e10, dtype=torch.float32, device=dev)
    if m == 0 or n == 0:
        return idx, dist2
    _check_offsets("knn_query_cells", offset, n, new_offset, m)
    lo_f = torch.minimum(xyz.amin(0), new_xyz.amin(0))
    hi_f = torch.maximum(xyz.amax(0), new_xyz.amax(0))
    if cell is None:
        # distance to the nsample-th neighbour if the candidates fill their bounding box (r_vol) or lie on a sheet across
        # its largest face (r_surf): a sheet in a thick box makes r_vol too large (cells hold more candidates than
        # needed, still correct), a filled box makes r_surf too small (many shells), hence the clamp
        ext = torch.sort((xyz.amax(0) - xyz.amin(0)).clamp_min(1e-12)).values.tolist()
        r_surf = (nsample * ext[1] * ext[2] / (3.14159 * n)) ** 0.5
        r_vol = (3.0 * nsample * ext[0] * ext[1] * ext[2] / (4 * 3.14159 * n)) ** (1.0 / 3.0)
        cell = max(r_surf, min(r_vol, 4.0 * r_surf), 1e-9)
    cell = float(cell)
    cell_t = torch.full((), cell, dtype=torch.float32, device=dev)
    # cells relative to the bounding box's corner: the quotient stays below the grid extent wherever the scene sits
    # (absolute coordinates far from the origin lose the cell boundary to fp32 rounding: |x / cell| 2^-24 cells)
    csrc = torch.floor((xyz - lo_f) / cell_t).int()
    cq = torch.floor((new_xyz - lo_f) / cell_t).int()
    extent = int(torch.maximum(csrc.amax(), cq.amax()))
    if extent >= 65536:
        raise ValueError(f"knn_query_cells: {extent + 1} cells of edge {cell} along one axis (limit 65536): use a larger cell")
    src_cells = torch.cat([_scene_ids(offset, n).int().unsqueeze(1), csrc], dim=1).contiguous()
    q_cells = torch.cat([_scene_ids(new_offset, m).int().unsqueeze(1), cq], dim=1).contiguous()
    c = src_cells.long()
    key = (((c[:, 0] << 16 | c[:, 1]) << 16 | c[:, 2]) << 16 | c[:, 3]).contiguous()
    order, _, seg_start, ncell = voxel_unique(key)
    uniq = src_cells[order[seg_start[:-1].long()]].contiguous()
    slots = lib.ptv3_subm_table_slots(ncell)
    table = torch.empty(slots * 12, dtype=torch.uint8, device=dev)
    lib.check(lib.ptv3_subm_build_table(_p(uniq), ncell, _p(table), slots, _stream()), "ptv3_subm_build_table")
    lib.check(lib.ptv3_knn_query_cells(m, int(nsample), _p(xyz), _p(new_xyz), _p(q_cells), _p(table), slots, _p(order),
                                       _p(seg_start), _p(offset), cell, _p(idx), _p(dist2), _stream()),
              "ptv3_knn_query_cells")
    return idx, dist2


# ---------------------------------------------------------------------------------------------
# measurement
# ---------------------------------------------------------------------------------------------
FAMILIES = ("linear", "subm_conv", "window_attn", "backward")


_PROFILING = False


def profile_enable(on=True):
    global _PROFILING
    _PROFILING = bool(on)
    lib.check(lib.ptv3_profile_enable(int(on)), "ptv3_profile_enable")


def profile_collect_kernels():
    """{kernel name: dict(ms, flops, bytes, launches)} per KERNEL (one launch per bracket); call before
    profile_collect(), which resets the records."""
    n = int(lib.ptv3_profile_kernel_count())
    ms, fl, by = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_double * n)()
    la = (ctypes.c_int64 * n)()
    lib.check(lib.ptv3_profile_collect_kernels(ms, fl, by, la), "ptv3_profile_collect_kernels")
    return {lib.ptv3_profile_kernel_name(i).decode(): dict(ms=ms[i], flops=fl[i], bytes=by[i], launches=int(la[i]))
            for i in range(n) if la[i]}


def profile_collect():
    """{family: dict(ms, flops, bytes, launches)} of the device time bracketed by HIP events since enable."""
    n = len(FAMILIES)
    ms, fl, by = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_double * n)()
    la = (ctypes.c_int64 * n)()
    lib.check(lib.ptv3_profile_collect(ms, fl, by, la), "ptv3_profile_collect")
    return {FAMILIES[i]: dict(ms=ms[i], flops=fl[i], bytes=by[i], launches=int(la[i])) for i in range(n)}
