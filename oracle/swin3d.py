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


# ---------------------------------------------------------------------------------------------------------------
# the whole Swin3DUNet forward (swin3d_v1m1_base.py:149-244), eval mode, state_dict driven
# ---------------------------------------------------------------------------------------------------------------
def _ln(x, g, b, eps=1e-5):
    x = x.astype(np.float64)
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def _gelu(x):
    from scipy.special import erf
    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


def _bn(x, sd, pre, eps=1e-5):
    return (x - sd[pre + "running_mean"]) / np.sqrt(sd[pre + "running_var"].astype(np.float64) + eps) * \
        sd[pre + "weight"] + sd[pre + "bias"]


def _unique_cells(batch, cell):
    """Sorted distinct (batch, cell) rows -> (first member of each in sorted order, cell id of every row, counts)."""
    rows = np.concatenate([batch[:, None], cell], axis=1)
    uniq, inverse, counts = np.unique(rows, axis=0, return_inverse=True, return_counts=True)
    return uniq, inverse.reshape(-1), counts


def _offset_of(batch, nb):
    return np.cumsum(np.bincount(batch, minlength=nb)).astype(np.int32)


class Swin3DOracle:
    """Functional restatement of Swin3DUNet.forward (knn_down True | False: GridKNNDownsample | GridDownsample,
    swin3d_layers.py:246-318; stem_transformer True | False: MinkConvBNRelu | + MinkResBlock and a downsample in front
    of the first attention stage, swin3d_v1m1_base.py:69-85, 213-216).  Sparse tensors are
    (coords int (n,4) at a tensor stride, feat (n,C), cfeat (n, 4 + signals)) triples numbered in sorted
    (batch, x, y, z) order; see the header of this file and of the product's swin3d_v1m1_base.py for the choices the
    absent libraries leave open (tap order of the stem kernel, Euclidean KNN distance, the GridCoordsDown tie rule)."""

    def __init__(self, state_dict, cfg):
        self.sd = {k: np.asarray(v, np.float32) if np.asarray(v).dtype.kind == "f" else np.asarray(v)
                   for k, v in state_dict.items()}
        self.cfg = cfg

    # ---- pieces -------------------------------------------------------------------------------------------
    def _lin(self, x, pre):
        y = x.astype(np.float64) @ self.sd[pre + "weight"].T.astype(np.float64)
        return y + self.sd[pre + "bias"] if pre + "bias" in self.sd else y

    def _stage(self, pre, feat, coords, stride, cfeat, depth, heads, ws):
        quant, crse = self.cfg["quant_size"], self.cfg["cRSE"]
        local = (cfeat[:, 1:4] - coords[:, 1:].astype(np.float32)) / np.float32(stride)
        sig = cfeat[:, 4:]
        x = feat.astype(np.float64)
        for i in range(depth):
            shift = 0 if i % 2 == 0 else ws // 2
            _, w_w_xyz, nempty, sort_idx, _ = window_mapping(coords, stride, ws, shift)
            nc = n_coords(w_w_xyz, local, sig, sort_idx)
            _, _, _, w_sizes, w2n, _ = sparse_self_attention(nempty)
            args = (None, None, None, w_sizes, w2n, sort_idx, None, nc)
            b = f"{pre}blocks.{i}."
            params = {k[len(b) + 5:]: v for k, v in self.sd.items() if k.startswith(b + "attn.")}
            h = _ln(x, self.sd[b + "norm1.weight"], self.sd[b + "norm1.bias"]).astype(np.float32)
            x = x + window_attention_forward(params, h, args, heads, quant, crse)
            h = _ln(x, self.sd[b + "norm2.weight"], self.sd[b + "norm2.bias"])
            x = x + self._lin(_gelu(self._lin(h, b + "mlp.fc1.")), b + "mlp.fc2.")
        return x.astype(np.float32)

    def _down(self, pre, coords, stride, feat, cfeat, offset, down):
        from . import pointops as OP
        new_stride = stride * down
        uniq, cell, counts = _unique_cells(coords[:, 0], coords[:, 1:] // new_stride)
        m = len(uniq)
        new_coords = np.concatenate([uniq[:, :1], uniq[:, 1:] * new_stride], axis=1).astype(np.int64)
        mean = np.zeros((m, cfeat.shape[1]), np.float64)
        np.add.at(mean, cell, cfeat.astype(np.float64))
        mean /= counts[:, None]
        dist = np.sqrt(((mean[cell] - cfeat) ** 2).sum(1))
        best = np.full(m, np.inf)
        np.minimum.at(best, cell, dist)
        near = dist <= best[cell] * (1 + 1e-4) + 1e-12
        pick = np.full(m, len(dist), np.int64)
        np.minimum.at(pick, cell, np.where(near, np.arange(len(dist)), len(dist)))
        new_cfeat = cfeat[pick]
        new_offset = _offset_of(new_coords[:, 0], len(offset))
        if not self.cfg.get("knn_down", True):
            # GridDownsample (:246-272): LayerNorm -> Linear (no bias) -> maximum over the voxels of each cell
            y = self._lin(_ln(feat, self.sd[pre + "norm.norm.weight"], self.sd[pre + "norm.norm.bias"]), pre + "linear.linear.")
            new_feat = np.full((m, y.shape[1]), -np.inf)
            np.maximum.at(new_feat, cell, y)
            return new_coords, new_stride, new_feat.astype(np.float32), new_cfeat, new_offset
        y = self._lin(_ln(feat, self.sd[pre + "norm.weight"], self.sd[pre + "norm.bias"]), pre + "linear.")
        idx, _ = OP.knn_query(16, np.ascontiguousarray(cfeat[:, 1:4]), offset,
                              np.ascontiguousarray(new_cfeat[:, 1:4]), new_offset)
        idx = np.where(idx < 0, idx[:, :1], idx)
        new_feat = y[idx].max(axis=1).astype(np.float32)
        return new_coords, new_stride, new_feat, new_cfeat, new_offset

    def _conv3(self, x, coords, key):
        """3x3x3 convolution on the occupied voxels, kernel[t] at offset t = (dx+1) + 3 (dy+1) + 9 (dz+1), no bias"""
        import torch
        from .ptv3 import subm_conv3d
        kern = self.sd[key]
        cin, cout = kern.shape[1:]
        w = np.zeros((cout, 3, 3, 3, cin), np.float32)
        for a in range(3):
            for b in range(3):
                for c in range(3):
                    w[:, a, b, c, :] = kern[a + 3 * b + 9 * c].T
        return subm_conv3d(torch.from_numpy(np.ascontiguousarray(x, np.float32)), torch.from_numpy(coords),
                           torch.from_numpy(w)).numpy()

    def _up(self, pre, deep, shallow, heads, ws):
        from . import pointops as OP
        (dc, ds, dfeat, dcf, doff), (sc, ss, sfeat, scf, soff) = deep, shallow
        y2 = self._lin(_ln(dfeat, self.sd[pre + "linear2.0.weight"], self.sd[pre + "linear2.0.bias"]), pre + "linear2.1.")
        idx, dist = OP.knn_query(self.cfg["up_k"], np.ascontiguousarray(dcf[:, 1:4]), doff,
                                 np.ascontiguousarray(scf[:, 1:4]), soff)
        w = 1.0 / (np.sqrt(dist.astype(np.float32)) + np.float32(1e-8))   # Euclidean, as libs/pointops returns it
        w = w / w.sum(1, keepdims=True)
        carried = (y2[idx] * w[:, :, None]).sum(1)
        feat = self._lin(_ln(sfeat, self.sd[pre + "linear1.0.weight"], self.sd[pre + "linear1.0.bias"]),
                         pre + "linear1.1.") + carried
        feat = feat.astype(np.float32)
        if "attn" in self.cfg["upsample"] and ws > 0:
            feat = self._stage(pre + "block.", feat, sc, ss, scf, 1, heads, ws)
        return sc, ss, feat, scf, soff

    # ---- forward -------------------------------------------------------------------------------------------
    def forward(self, data, trace=None):
        import torch
        trace = {} if trace is None else trace
        from .ptv3 import subm_conv3d
        cfg, sd = self.cfg, self.sd
        coord, feat, coord_feat = (np.asarray(data[k], np.float32) for k in ("coord", "feat", "coord_feat"))
        grid = np.asarray(data["grid_coord"], np.int64)
        offset = np.asarray(data["offset"], np.int64)
        batch = np.repeat(np.arange(len(offset)), np.diff(np.concatenate([[0], offset])))
        rows = np.concatenate([batch[:, None].astype(np.float32), coord / np.float32(cfg["base_grid_size"]),
                               coord_feat / np.float32(1.001), feat], axis=1)
        uniq, p2v, counts = _unique_cells(batch, grid)
        mean = np.zeros((len(uniq), rows.shape[1]), np.float64)
        np.add.at(mean, p2v, rows.astype(np.float64))
        mean = (mean / counts[:, None]).astype(np.float32)
        ncf = coord_feat.shape[1] + 4
        coords, stride, cfeat, x = uniq.astype(np.int64), 1, mean[:, :ncf], mean[:, ncf:]
        off = _offset_of(coords[:, 0], len(offset))
        stem_tr = cfg.get("stem_transformer", True)
        sp = "stem_layer." if stem_tr else "stem_layer.0."
        x = np.maximum(_bn(self._conv3(x, coords, sp + "conv_layers.0.kernel"), sd, sp + "conv_layers.1.bn.", 1e-5), 0)
        x = x.astype(np.float32)
        skips = []
        nl = cfg["num_layers"]
        start = 0
        if not stem_tr:
            # MinkResBlock (mink_layers.py:115-155), then the model's own downsample in front of stage 1 (:213-216)
            r = "stem_layer.1."
            y = np.maximum(_bn(self._conv3(x, coords, r + "conv1.kernel"), sd, r + "norm1.bn.", 1e-5), 0)
            y = _bn(self._conv3(y, coords, r + "conv2.kernel"), sd, r + "norm2.bn.", 1e-5)
            x = np.maximum(y + x, 0).astype(np.float32)
            skips.append((coords, stride, x, cfeat, off))
            coords, stride, x, cfeat, off = self._down("downsample.", coords, stride, x, cfeat, off, cfg["down_stride"])
            start = 1
        trace["stem"] = x
        for i in range(start, nl):
            pre = f"layers.{i - start}."
            x = self._stage(pre, x, coords, stride, cfeat, cfg["depths"][i], cfg["num_heads"][i], cfg["window_sizes"][i])
            skips.append((coords, stride, x, cfeat, off))
            trace[f"layer{i}"] = x
            if i < nl - 1:
                coords, stride, x, cfeat, off = self._down(pre + "downsample.", coords, stride, x, cfeat, off,
                                                           cfg["down_stride"] if i == 0 else 2)
                trace[f"down{i}"], trace[f"down{i}_cfeat"] = x, cfeat
        level = skips.pop()
        for j, i in enumerate(range(nl - 1, 0, -1)):
            level = self._up(f"upsamples.{j}.", level, skips.pop(), cfg["num_heads"][i - 1], cfg["window_sizes"][i - 1])
            trace[f"up{j}"] = level[2]
        h = self._lin(level[2], "classifier.0.")
        h = np.maximum(_bn(h, sd, "classifier.1.", 1e-5), 0)
        return self._lin(h, "classifier.3.")[p2v].astype(np.float32)
