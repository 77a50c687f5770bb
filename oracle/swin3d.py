"""CPU restatement of the Swin3D window partition and cRSE window attention.

TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product path.

PARITY UNPINNED.  The Python side of the reference (pointcept/models/swin3d/swin3d_layers.py) is restated line by
line where it is plain torch (window ids, sort order, the index tuples, the table layout).  The arithmetic of the
attention itself lives in two libraries that are NOT in the reference tree and cannot be fetched here:
  * MinkowskiEngine (README.md:851 names v0.5.4; `MinkowskiEngine/` is an empty directory): window origins come from
    `MinkowskiMaxPooling(kernel_size=ws, stride=ws)` (swin3d_layers.py:703-705), restated here as
    floor(coordinate / (ws * tensor_stride)) per axis, the published behaviour of a strided pooling coordinate map.
    ME numbers the pooled coordinates in hash-map order, which is unspecified; this restatement numbers windows in
    lexicographic (batch, x, y, z) order.  The attention result per voxel does not depend on the numbering.
  * microsoft/Swin3D `Swin3D.sparse_dl.attn.attn_coff.SelfAttnAIOFunction` (unpinned, uv_requirements.txt:155; call
    site swin3d_layers.py:556-569 with PosEmb.SEPARATE, TableDims.D0, IndexMode.INDIRECT), restated from the Swin3D
    paper's contextual relative signal encoding:
        e_ij   = q_i . k_j + sum_c ( q_i . T_K[c][idx_c(i,j)] + k_j . T_Q[c][idx_c(i,j)] )
        out_i  = sum_j softmax_j(e_ij) ( v_j + sum_c T_V[c][idx_c(i,j)] )
        idx_c  = floor( s_i[c] - s_j[c] + L_c ),  L_c = half the table length of signal axis c
    where s = n_cRSE (window-local position / colour / normal, each times its quantisation, :505-528), c runs over the
    3, 6 or 9 signal axes, and the caller has already multiplied q by head_dim**-0.5 (:499).  idx is clamped to
    [0, 2 L_c - 1]: colour or normal differences of exactly +2 (white vs black) would otherwise index one row past
    the table.
No golden vector exists for either library (the reference holds no test or fixture for this path), so the HIP kernel
is checked against THIS restatement only, and every test that does so says "parity unpinned".
"""
import numpy as np


def window_mapping(coords, stride, window_size, shift=0):
    """get_window_mapping (swin3d_layers.py:746-795) of a sparse tensor with coordinates `coords` (N,4) int
    [batch, x, y, z] (multiples of `stride`, the tensor stride), optionally shifted by `shift` voxels as
    get_shifted_sp does (:826-840: C[:, 1:] += shift_size * stride).
    -> w_w_id (N,), w_w_xyz (N,3), nempty_num (W,), sort_idx (N,), inv_sort_idx (N,)   (the first two in sorted order)."""
    c = np.asarray(coords, np.int64)
    ws = int(window_size)
    vox = np.floor_divide(c[:, 1:], stride) + shift
    win = np.floor_divide(vox, ws)                       # pooled (window) coordinate per axis
    loc = vox - win * ws                                 # position inside the window, 0 .. ws-1
    wid_local = (loc[:, 0] * ws + loc[:, 1]) * ws + loc[:, 2]      # row of local_window (meshgrid x,y,z, :728-734)
    wkey = np.concatenate([c[:, :1], win], axis=1)
    _, w_id = np.unique(wkey, axis=0, return_inverse=True)         # lexicographic window numbering
    w_id = w_id.reshape(-1)
    n_w = ws ** 3
    in_map = w_id * n_w + wid_local                      # index into all_windows (:735-743)
    sort_idx = np.argsort(in_map, kind="stable")         # :752-755 (voxels are unique, so no ties)
    in_sorted = in_map[sort_idx]
    inv = np.empty_like(sort_idx)
    inv[sort_idx] = np.arange(sort_idx.shape[0])         # :756-759
    w_w_id = in_sorted % n_w                             # :768-776
    nempty = np.bincount(in_sorted // n_w, minlength=int(w_id.max()) + 1 if len(w_id) else 0)   # :777-778
    w_w_xyz = np.stack([w_w_id // ws // ws, w_w_id // ws % ws, w_w_id % ws], axis=-1)           # :781-788
    return w_w_id, w_w_xyz, nempty, sort_idx, inv


def sparse_self_attention(w_sizes):
    """protocol "v2" of swin3d_layers.py:78-152: x_offset / y_offset enumerate, window by window, every (row, column)
    pair of the window's token block in sorted order; m2w maps a pair to its window, w2n / w2m are the windows'
    offsets in token / pair space."""
    w_sizes = np.asarray(w_sizes, np.int64)
    w2n = np.concatenate([[0], np.cumsum(w_sizes)[:-1]]).astype(np.int64)
    sq = w_sizes ** 2
    w2m = np.concatenate([[0], np.cumsum(sq)[:-1]]).astype(np.int64)
    m2w = np.repeat(np.arange(len(w_sizes)), sq)
    m_off = np.arange(int(sq.sum())) - w2m[m2w]
    y_off = w2n[m2w] + m_off % w_sizes[m2w]
    x_off = w2n[m2w] + m_off // w_sizes[m2w]
    return x_off, y_off, m2w, w_sizes, w2n, w2m


def n_coords(w_w_xyz, local_xyz, signals, sort_idx):
    """get_index01 (:797-812): window-local voxel index + sub-voxel offset, then the other signals (colour, normal),
    all in sorted order."""
    xyz = w_w_xyz.astype(np.float32) + np.asarray(local_xyz, np.float32)[sort_idx]
    return np.concatenate([xyz, np.asarray(signals, np.float32)[sort_idx]], axis=1)


def n_crse(ncoords, quant_size, crse="XYZ_RGB_NORM"):
    """:505-528: each signal group times its quantisation (xyz: quant, colour and normal: 2 * quant)."""
    out, col = [], 0
    for name, q in (("XYZ", quant_size), ("RGB", quant_size * 2), ("NORM", quant_size * 2)):
        if name in crse:
            out.append(ncoords[:, col:col + 3] * np.float32(q))
            col += 3
    return np.concatenate(out, axis=1).astype(np.float32)


def table_lengths(window_size, quant_size, crse="XYZ_RGB_NORM"):
    """Rows (2 L) of each signal group's tables (:433-469): xyz 2*ws*quant, colour / normal 2*2*(2*quant)."""
    out = []
    if "XYZ" in crse:
        out.append(2 * window_size * quant_size)
    if "RGB" in crse:
        out.append(2 * 2 * quant_size * 2)
    if "NORM" in crse:
        out.append(2 * 2 * quant_size * 2)
    return out


def crse_attention(q, k, v, q_table, k_table, v_table, table_offsets, w_sizes, w2n, n2n, ncrse):
    """SelfAttnAIOFunction forward (see the header for the formula and its provenance).
    q, k, v: (N, H, D) in ORIGINAL voxel order, q already scaled; *_table: flat fp32, one slab of table_offsets[c]
    = rows_c * H * D elements per signal axis c (the reference's `table_offsets`, :441,452,464); n2n: sorted position
    -> original row (IndexMode.INDIRECT); ncrse: (N, S) in SORTED order."""
    q = np.asarray(q, np.float32)
    k = np.asarray(k, np.float32)
    v = np.asarray(v, np.float32)
    n, h, d = q.shape
    s = ncrse.shape[1]
    starts = np.concatenate([[0], np.cumsum(table_offsets)]).astype(np.int64)
    tabs = []
    for c in range(s):
        rows = int(table_offsets[c]) // (h * d)
        sl = slice(int(starts[c]), int(starts[c + 1]))
        tabs.append((rows, np.asarray(q_table, np.float32)[sl].reshape(rows, h, d),
                     np.asarray(k_table, np.float32)[sl].reshape(rows, h, d),
                     np.asarray(v_table, np.float32)[sl].reshape(rows, h, d)))
    out = np.zeros_like(q)
    for w in range(len(w_sizes)):
        m = int(w_sizes[w])
        if m == 0:
            continue
        s0 = int(w2n[w])
        rows = np.asarray(n2n[s0:s0 + m], np.int64)
        cr = ncrse[s0:s0 + m]
        qq, kk, vv = (a[rows].astype(np.float64) for a in (q, k, v))
        logit = np.einsum("ihd,jhd->hij", qq, kk)
        vsum = np.zeros((m, m, h, d), np.float64)
        for c in range(s):
            nrow, tq, tk, tv = tabs[c]
            half = np.float32(nrow // 2)
            idx = np.floor((cr[:, None, c] - cr[None, :, c]) + half).astype(np.int64)   # fp32 arithmetic
            idx = np.clip(idx, 0, nrow - 1)
            logit += np.einsum("ihd,ijhd->hij", qq, tk[idx].astype(np.float64))
            logit += np.einsum("jhd,ijhd->hij", kk, tq[idx].astype(np.float64))
            vsum += tv[idx]
        logit -= logit.max(axis=-1, keepdims=True)
        p = np.exp(logit)
        p /= p.sum(axis=-1, keepdims=True)
        o = np.einsum("hij,jhd->ihd", p, vv) + np.einsum("hij,ijhd->ihd", p, vsum)
        out[rows] = o.astype(np.float32)
    return out


def window_attention_forward(params, feats, attn_args, num_heads, quant_size, crse="XYZ_RGB_NORM"):
    """WindowAttention.forward (:482-577).  params: dict with qkv.weight/bias, proj.weight/bias and the
    {query,key,value}_{xyz,rgb,norm}_table arrays under the reference's parameter names."""
    (_, _, _, w_sizes, w2n, n2n, _, ncoords) = attn_args
    feats = np.asarray(feats, np.float32)
    nv, dim = feats.shape
    hd = dim // num_heads
    qkv = feats @ params["qkv.weight"].T + params["qkv.bias"]
    qkv = qkv.reshape(nv, 3, num_heads, hd)
    q, k, v = qkv[:, 0] * np.float32(hd ** -0.5), qkv[:, 1], qkv[:, 2]
    names = [g for g in ("xyz", "rgb", "norm") if g.upper() in crse]
    qt = np.concatenate([params[f"query_{g}_table"].reshape(-1) for g in names])
    kt = np.concatenate([params[f"key_{g}_table"].reshape(-1) for g in names])
    vt = np.concatenate([params[f"value_{g}_table"].reshape(-1) for g in names])
    offs = []
    for g in names:
        shp = params[f"query_{g}_table"].shape
        offs += [int(np.prod(shp[1:]))] * 3
    o = crse_attention(q, k, v, qt, kt, vt, offs, w_sizes, w2n, n2n, n_crse(ncoords, quant_size, crse))
    return o.reshape(nv, dim) @ params["proj.weight"].T + params["proj.bias"]
