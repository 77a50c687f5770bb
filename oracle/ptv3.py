"""Oracle (test infrastructure): plain torch-CPU fp32 restatement of the PTv3 forward.

Functional (state_dict driven) so that it shares no code with the product modules.
Follows, in eval mode, with upcast_*=False / enable_rpe optional; enable_flash=False (vanilla branch) or
enable_flash=True (fixed K, ragged windows through varlen_attention()):
  * PointTransformerV3.forward           point_transformer_v3m1_base.py:699-714
  * Embedding                            :485-515   (SubMConv3d k=5 -> BN(eps 1e-3) -> GELU)
  * Block.forward                        :318-338   (xCPE, pre-norm attn, pre-norm MLP)
  * SerializedAttention.forward          :172-222   (vanilla path 190-206) + RPE :29-48
  * SerializedPooling.forward            :371-444
  * SerializedUnpooling.forward          :471-482
  * OffsetKeypointPTv3 head/loss         offset_keypoint_ptv3.py:26-31,37-107
  * DefaultSegmentorV2 head              default.py:41-95 (Linear(C, num_classes))
Third-party arithmetic restated here (absent from /root/reference):
  * spconv 2.3.6 SubMConv3d  -> subm_conv3d()  [weight (O,k0,k1,k2,I); parity unpinned]
  * torch_scatter.segment_csr(max|mean) -> segment_reduce()
  * flash-attn 2.6.3 flash_attn_varlen_qkvpacked_func -> varlen_attention()  [published semantics: exact
    softmax(scale q k^T) v per sequence and head; parity unpinned against the CUDA wheel]
Pinned by tests/golden/ptv3_*.npz (reference code run with these stubs).
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

from . import sfc


# ----------------------------------------------------------------------------
# third-party restatements
# ----------------------------------------------------------------------------
def subm_conv3d(feat, indices, weight, bias=None):
    """Submanifold sparse conv: out[i] = b + sum_{d in kernel, site(p_i + d) active} W[:,d,:] @ x[site].

    feat (N, I) fp32; indices (N, 4) int [batch, x, y, z] (unique rows);
    weight (O, k0, k1, k2, I) (spconv 2.x layout, kernel index (k0,k1,k2) pairs with offset
    (k0-c, k1-c, k2-c) on (x, y, z), c = k//2; correlation form like torch.nn.Conv3d).
    """
    N = feat.shape[0]
    O, k0, k1, k2, I = weight.shape
    idx = indices.detach().cpu().numpy().astype(np.int64)
    S = int(idx[:, 1:].max()) + 1 + 2 * max(k0, k1, k2)
    pad = max(k0, k1, k2)

    def key_of(b, x, y, z):
        return ((b * S + (x + pad)) * S + (y + pad)) * S + (z + pad)

    keys = key_of(idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3])
    sorter = np.argsort(keys, kind="stable")
    skeys = keys[sorter]
    out = torch.zeros(N, O, dtype=feat.dtype)
    for a in range(k0):
        for b_ in range(k1):
            for c in range(k2):
                da, db, dc = a - k0 // 2, b_ - k1 // 2, c - k2 // 2
                q = key_of(idx[:, 0], idx[:, 1] + da, idx[:, 2] + db, idx[:, 3] + dc)
                pos = np.searchsorted(skeys, q)
                pos = np.minimum(pos, N - 1)
                hit = skeys[pos] == q
                if not hit.any():
                    continue
                dst = torch.from_numpy(np.nonzero(hit)[0])
                src = torch.from_numpy(sorter[pos[hit]])
                out.index_add_(0, dst, feat[src] @ weight[:, a, b_, c, :].t())
    if bias is not None:
        out = out + bias
    return out


def segment_reduce(src, idx_ptr, reduce):
    """torch_scatter.segment_csr(src, indptr, reduce) for non-empty segments."""
    lengths = (idx_ptr[1:] - idx_ptr[:-1])
    return torch.segment_reduce(src, reduce, lengths=lengths, axis=0, unsafe=True)


# ----------------------------------------------------------------------------
# pieces of the path
# ----------------------------------------------------------------------------
def rpe_bias(table, grid_coord_sorted, K, patch_size_cfg, H):
    """RPE.forward, point_transformer_v3m1_base.py:29-48; get_rel_pos :104-112."""
    pos_bnd = int((4 * patch_size_cfg) ** (1 / 3) * 2)
    rpe_num = 2 * pos_bnd + 1
    gc = grid_coord_sorted.reshape(-1, K, 3)
    rel = gc.unsqueeze(2) - gc.unsqueeze(1)
    idx = rel.clamp(-pos_bnd, pos_bnd) + pos_bnd + torch.arange(3) * rpe_num
    out = table.index_select(0, idx.reshape(-1).long())
    out = out.view(idx.shape + (-1,)).sum(3)
    return out.permute(0, 3, 1, 2)


def window_attention(feat, qkv_w, qkv_b, proj_w, proj_b, order, inverse, pad, unpad, H, K,
                     rpe_table=None, grid_coord=None, patch_size_cfg=None):
    """SerializedAttention.forward vanilla path (:172-222)."""
    C = feat.shape[1]
    scale = (C // H) ** -0.5
    o = order[pad]
    inv = unpad[inverse]
    qkv = F.linear(feat, qkv_w, qkv_b)[o]
    q, k, v = qkv.reshape(-1, K, 3, H, C // H).permute(2, 0, 3, 1, 4).unbind(dim=0)
    attn = (q * scale) @ k.transpose(-2, -1)
    if rpe_table is not None:
        attn = attn + rpe_bias(rpe_table, grid_coord[o], K, patch_size_cfg, H)
    attn = torch.softmax(attn, dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(-1, C)
    out = out[inv]
    return F.linear(out, proj_w, proj_b)


def varlen_attention(qkv, cu_seqlens, H, scale, bf16_io=False):
    """flash_attn.flash_attn_varlen_qkvpacked_func(qkv (T, 3, H, D), cu_seqlens, max_seqlen, softmax_scale) restated:
    every sequence [cu[w], cu[w+1]) attends within itself, per head.  flash-attn 2.6.3 (uv_requirements.txt:36) is a
    CUDA-only wheel outside the reference tree; this is its documented function, evaluated in fp32.
    bf16_io=True adds what the call site itself does to the data (v3m1_base.py:209: qkv.to(torch.bfloat16)) and what
    the library returns (its input dtype): inputs and output rounded to bf16."""
    T = qkv.shape[0]
    D = qkv.shape[-1]
    x = qkv.reshape(T, 3, H, D)
    if bf16_io:
        x = x.to(torch.bfloat16)
    x = x.float()
    out = torch.empty(T, H, D, dtype=torch.float32)
    cu = [int(v) for v in cu_seqlens]
    lens = [b - a for a, b in zip(cu[:-1], cu[1:])]
    # sequences of equal length are evaluated together (a scene's full windows); short ones one by one
    by_len = {}
    for w, ln in enumerate(lens):
        by_len.setdefault(ln, []).append(cu[w])
    for ln, starts in by_len.items():
        idx = (torch.tensor(starts).unsqueeze(1) + torch.arange(ln).unsqueeze(0)).reshape(-1)
        blk = x[idx].reshape(len(starts), ln, 3, H, D).permute(2, 0, 3, 1, 4)   # (3, W, H, ln, D)
        attn = torch.softmax((blk[0] * scale) @ blk[1].transpose(-2, -1), dim=-1)
        out[idx] = (attn @ blk[2]).transpose(1, 2).reshape(-1, H, D)
    if bf16_io:
        out = out.to(torch.bfloat16)
    return out


def window_attention_flash(feat, qkv_w, qkv_b, proj_w, proj_b, order, inverse, pad, unpad, cu_seqlens, H,
                           bf16_io=False):
    """SerializedAttention.forward, enable_flash=True branch (:172-222 with :207-215)."""
    C = feat.shape[1]
    scale = (C // H) ** -0.5
    o = order[pad]
    inv = unpad[inverse]
    qkv = F.linear(feat, qkv_w, qkv_b)[o]
    out = varlen_attention(qkv.reshape(-1, 3, H, C // H), cu_seqlens, H, scale, bf16_io).reshape(-1, C)
    out = out.to(qkv.dtype)[inv]
    return F.linear(out, proj_w, proj_b)


def window_attention_core(qkv, order, inverse, pad, unpad, H, K):
    """Only the gather -> softmax(QK^T)V -> scatter part of :184-216 (what the HIP kernel fuses)."""
    C = qkv.shape[1] // 3
    scale = (C // H) ** -0.5
    o = order[pad]
    inv = unpad[inverse]
    x = qkv[o]
    q, k, v = x.reshape(-1, K, 3, H, C // H).permute(2, 0, 3, 1, 4).unbind(dim=0)
    attn = torch.softmax((q * scale) @ k.transpose(-2, -1), dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(-1, C)
    return out[inv]


def _leafify(sd):
    """training oracle: parameters become autograd leaves, buffers private copies."""
    out = {}
    for k, v in sd.items():
        v = v.clone()
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            v.requires_grad_(True)
        out[k] = v
    return out


class PTv3Oracle:
    """Functional forward of "PT-v3m1" (+ optional offset-keypoint / segmentor head).  Eval mode by default;
    training=True: BatchNorm uses batch statistics and updates (private copies of) the running buffers
    (momentum 0.01, v3m1_base.py:576), parameters are autograd leaves so that torch autograd over this
    restatement yields the reference gradients (the reference trains through torch autograd,
    engines/train.py:184-213).  DropPath is stochastic per point (RNG stream differs CPU vs device):
    the training oracle requires drop_path = 0."""

    def __init__(self, cfg, state_dict, prefix="", training=False):
        self.cfg = dict(cfg)
        self.training = training
        self.sd = {k[len(prefix):]: v.detach().float().cpu() for k, v in state_dict.items()
                   if k.startswith(prefix)}
        if training:
            assert self.cfg.get("drop_path", 0.3) == 0.0, "training oracle: set drop_path=0 (stochastic depth RNG)"
            self.sd = _leafify(self.sd)
        c = self.cfg
        self.order = [c["order"]] if isinstance(c["order"], str) else list(c["order"])
        self.enc_depths = c.get("enc_depths", (2, 2, 2, 6, 2))
        self.enc_channels = c.get("enc_channels", (32, 64, 128, 256, 512))
        self.enc_num_head = c.get("enc_num_head", (2, 4, 8, 16, 32))
        self.enc_patch_size = c.get("enc_patch_size", (48,) * 5)
        self.dec_depths = c.get("dec_depths", (2, 2, 2, 2))
        self.dec_channels = c.get("dec_channels", (64, 64, 128, 256))
        self.dec_num_head = c.get("dec_num_head", (4, 4, 8, 16))
        self.dec_patch_size = c.get("dec_patch_size", (48,) * 4)
        self.stride = c.get("stride", (2, 2, 2, 2))
        self.shuffle_orders = c.get("shuffle_orders", True)
        self.enable_rpe = c.get("enable_rpe", False)
        self.enable_flash = c.get("enable_flash", True)   # constructor default of the reference (:549)
        self.flash_bf16_io = False   # True: also apply the call site's bf16 cast of qkv / bf16 result (:209-214)
        self.enc_mode = c.get("enc_mode", False)
        self.num_stages = len(self.enc_depths)
        self.trace = {}

    # -- helpers ------------------------------------------------------------
    def _bn(self, x, name, eps=1e-3):
        sd = self.sd
        return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                            sd[name + ".weight"], sd[name + ".bias"], self.training, 0.01, eps)

    def _ln(self, x, name):
        return F.layer_norm(x, (x.shape[1],), self.sd[name + ".weight"], self.sd[name + ".bias"], 1e-5)

    def _lin(self, x, name):
        return F.linear(x, self.sd[name + ".weight"], self.sd[name + ".bias"])

    # -- Point.serialization (structure.py:52-109) ---------------------------
    def _serialize(self, P):
        gc = P["grid_coord"].numpy()
        code, order, inverse, depth = sfc.serialization(gc, P["batch"].numpy(), self.order)
        code, order, inverse = map(torch.from_numpy, (code, order, inverse))
        if self.shuffle_orders:
            perm = torch.randperm(code.shape[0])
            code, order, inverse = code[perm], order[perm], inverse[perm]
        P.update(serialized_code=code, serialized_order=order, serialized_inverse=inverse,
                 serialized_depth=depth)

    @staticmethod
    def _indices(P):
        return torch.cat([P["batch"].unsqueeze(-1).int(), P["grid_coord"].int()], dim=1)

    # -- Block (v3m1_base.py:318-338) ---------------------------------------
    def _block(self, P, name, C, H, patch_size, order_index):
        sd = self.sd
        feat = P["feat"]
        shortcut = feat
        # The conv reads point.sparse_conv_feat.features, which SerializedUnpooling does NOT refresh
        # after `parent.feat = parent.feat + point.feat[inverse]` (:478; modules.py:97-103 only
        # refreshes it inside PointSequential) -> first decoder block convolves the skip branch only.
        conv_in = P.pop("sparse_conv_feat", feat)
        x = subm_conv3d(conv_in, self._indices(P), sd[name + ".cpe.0.weight"], sd[name + ".cpe.0.bias"])
        x = self._lin(x, name + ".cpe.1")
        x = self._ln(x, name + ".cpe.2")
        feat = shortcut + x
        shortcut = feat
        x = self._ln(feat, name + ".norm1.0")
        # attention (:172-222)
        offset = P["offset"].numpy()
        # enable_flash=False: K shrinks to the smallest scene (:173-176); True: K is the configured patch
        K = int(patch_size) if self.enable_flash else sfc.patch_size_for(offset, patch_size)
        key = ("pad", K)
        if key not in P:
            pad, unpad, cu = sfc.pad_plan(offset, K)
            P[key] = (torch.from_numpy(pad), torch.from_numpy(unpad), torch.from_numpy(cu))
        pad, unpad, cu = P[key]
        rpe = sd.get(name + ".attn.rpe.rpe_table") if self.enable_rpe else None
        if self.enable_flash:
            x = window_attention_flash(x, sd[name + ".attn.qkv.weight"], sd[name + ".attn.qkv.bias"],
                                       sd[name + ".attn.proj.weight"], sd[name + ".attn.proj.bias"],
                                       P["serialized_order"][order_index], P["serialized_inverse"][order_index],
                                       pad, unpad, cu, H, bf16_io=self.flash_bf16_io)
        else:
            x = window_attention(x, sd[name + ".attn.qkv.weight"], sd[name + ".attn.qkv.bias"],
                                 sd[name + ".attn.proj.weight"], sd[name + ".attn.proj.bias"],
                                 P["serialized_order"][order_index], P["serialized_inverse"][order_index],
                                 pad, unpad, H, K, rpe_table=rpe, grid_coord=P["grid_coord"],
                                 patch_size_cfg=patch_size)
        if name + ".ls1.0.gamma" in sd:     # LayerScale of "PT-v3m2" (v3m2_sonata.py:26-38, 349)
            x = x * sd[name + ".ls1.0.gamma"]
        feat = shortcut + x
        shortcut = feat
        x = self._ln(feat, name + ".norm2.0")
        x = self._lin(x, name + ".mlp.0.fc1")
        x = F.gelu(x)
        x = self._lin(x, name + ".mlp.0.fc2")
        if name + ".ls2.0.gamma" in sd:
            x = x * sd[name + ".ls2.0.gamma"]
        P["feat"] = shortcut + x
        return P

    # -- SerializedPooling (v3m1_base.py:371-444) ---------------------------
    def _pool(self, P, name, stride):
        pooling_depth = (math.ceil(stride) - 1).bit_length()
        if pooling_depth > P["serialized_depth"]:
            pooling_depth = 0
        code = P["serialized_code"] >> pooling_depth * 3
        code_, cluster, counts = torch.unique(code[0], sorted=True, return_inverse=True, return_counts=True)
        _, indices = torch.sort(cluster, stable=True)
        idx_ptr = torch.cat([counts.new_zeros(1), torch.cumsum(counts, dim=0)])
        head_indices = indices[idx_ptr[:-1]]
        code = code[:, head_indices]
        order = torch.argsort(code, stable=True)
        inverse = torch.zeros_like(order).scatter_(
            dim=1, index=order,
            src=torch.arange(0, code.shape[1]).repeat(code.shape[0], 1))
        if self.shuffle_orders:
            perm = torch.randperm(code.shape[0])
            code, order, inverse = code[perm], order[perm], inverse[perm]
        feat = segment_reduce(self._lin(P["feat"], name + ".proj")[indices], idx_ptr, "max")
        Q = dict(
            feat=feat,
            coord=segment_reduce(P["coord"][indices], idx_ptr, "mean"),
            grid_coord=P["grid_coord"][head_indices] >> pooling_depth,
            serialized_code=code, serialized_order=order, serialized_inverse=inverse,
            serialized_depth=P["serialized_depth"] - pooling_depth,
            batch=P["batch"][head_indices],
            pooling_inverse=cluster, pooling_parent=P,
        )
        Q["offset"] = torch.cumsum(torch.bincount(Q["batch"]), dim=0).long()
        Q["feat"] = F.gelu(self._bn(Q["feat"], name + ".norm.0"))
        return Q

    # -- SerializedUnpooling (v3m1_base.py:471-482) -------------------------
    def _unpool(self, P, name):
        parent = P.pop("pooling_parent")
        inverse = P.pop("pooling_inverse")
        x = F.gelu(self._bn(self._lin(P["feat"], name + ".proj.0"), name + ".proj.1"))
        y = F.gelu(self._bn(self._lin(parent["feat"], name + ".proj_skip.0"), name + ".proj_skip.1"))
        parent["feat"] = y + x[inverse]
        parent["sparse_conv_feat"] = y  # stale on purpose, see _block
        return parent

    # -- PointTransformerV3.forward (:699-714) ------------------------------
    def backbone(self, data):
        P = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in data.items()}
        if "batch" not in P:
            bincount = torch.diff(P["offset"], prepend=torch.zeros(1, dtype=P["offset"].dtype))
            P["batch"] = torch.arange(len(bincount)).repeat_interleave(bincount)
        elif "offset" not in P:
            P["offset"] = torch.cumsum(P["batch"].bincount(), dim=0).long()
        P["offset"] = P["offset"].long()
        P["batch"] = P["batch"].long()
        P["feat"] = P["feat"].float()
        self._serialize(P)
        self.trace["serialized_code"] = P["serialized_code"].clone()
        self.trace["serialized_order"] = P["serialized_order"].clone()
        # embedding (:485-515)
        x = subm_conv3d(P["feat"], self._indices(P), self.sd["embedding.stem.conv.weight"], None)
        P["feat"] = F.gelu(self._bn(x, "embedding.stem.norm"))
        self.trace["embedding"] = P["feat"].clone()
        k = len(self.order)
        for s in range(self.num_stages):
            if s > 0:
                P = self._pool(P, f"enc.enc{s}.down", self.stride[s - 1])
            for i in range(self.enc_depths[s]):
                P = self._block(P, f"enc.enc{s}.block{i}", self.enc_channels[s], self.enc_num_head[s],
                                self.enc_patch_size[s], i % k)
            self.trace[f"enc{s}"] = P["feat"].clone()
            self.trace[f"n{s}"] = P["feat"].shape[0]
        if not self.enc_mode:
            for s in reversed(range(self.num_stages - 1)):
                P = self._unpool(P, f"dec.dec{s}.up")
                for i in range(self.dec_depths[s]):
                    P = self._block(P, f"dec.dec{s}.block{i}", self.dec_channels[s], self.dec_num_head[s],
                                    self.dec_patch_size[s], i % k)
                self.trace[f"dec{s}"] = P["feat"].clone()
        return P


class PTv3m2Oracle(PTv3Oracle):
    """"PT-v3m2" (point_transformer_v3m2_sonata.py:545-732): Linear+LayerNorm+GELU stem, serialization after the
    stem, GridPooling / GridUnpooling (LayerNorm, re-serialization of every pooled level), LayerScale."""

    # -- GridPooling (:402-470) ------------------------------------------------
    def _pool(self, P, name, stride):
        gc = torch.div(P["grid_coord"], stride, rounding_mode="trunc")
        gc = gc | (P["batch"].view(-1, 1) << 48)
        gc, cluster, counts = torch.unique(gc, sorted=True, return_inverse=True, return_counts=True, dim=0)
        gc = gc & ((1 << 48) - 1)
        _, indices = torch.sort(cluster, stable=True)
        idx_ptr = torch.cat([counts.new_zeros(1), torch.cumsum(counts, dim=0)])
        head_indices = indices[idx_ptr[:-1]]
        Q = dict(
            feat=segment_reduce(self._lin(P["feat"], name + ".proj")[indices], idx_ptr, "max"),
            coord=segment_reduce(P["coord"][indices], idx_ptr, "mean"),
            grid_coord=gc, batch=P["batch"][head_indices],
            pooling_inverse=cluster, pooling_parent=P,
        )
        Q["offset"] = torch.cumsum(torch.bincount(Q["batch"]), dim=0).long()
        Q["feat"] = F.gelu(self._ln(Q["feat"], name + ".norm.0"))
        self._serialize(Q)
        return Q

    # -- GridUnpooling (:497-512): sparse_conv_feat IS refreshed here, no stale-skip quirk ---------
    def _unpool(self, P, name):
        parent = P.pop("pooling_parent")
        inverse = P["pooling_inverse"]
        y = F.gelu(self._ln(self._lin(parent["feat"], name + ".proj_skip.0"), name + ".proj_skip.1"))
        x = F.gelu(self._ln(self._lin(P["feat"], name + ".proj.0"), name + ".proj.1"))
        parent["feat"] = y + x[inverse]
        return parent

    def backbone(self, data):
        P = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in data.items()}
        if "batch" not in P:
            bincount = torch.diff(P["offset"], prepend=torch.zeros(1, dtype=P["offset"].dtype))
            P["batch"] = torch.arange(len(bincount)).repeat_interleave(bincount)
        elif "offset" not in P:
            P["offset"] = torch.cumsum(P["batch"].bincount(), dim=0).long()
        P["offset"], P["batch"], P["feat"] = P["offset"].long(), P["batch"].long(), P["feat"].float()
        # Embedding (:505-540): stem on the raw features, THEN serialization (:722-725)
        x = self._lin(P["feat"], "embedding.stem.linear")
        P["feat"] = F.gelu(self._ln(x, "embedding.stem.norm"))
        self.trace["embedding"] = P["feat"].clone()
        self._serialize(P)
        k = len(self.order)
        for s in range(self.num_stages):
            if s > 0:
                P = self._pool(P, f"enc.enc{s}.down", self.stride[s - 1])
            for i in range(self.enc_depths[s]):
                P = self._block(P, f"enc.enc{s}.block{i}", self.enc_channels[s], self.enc_num_head[s],
                                self.enc_patch_size[s], i % k)
            self.trace[f"enc{s}"] = P["feat"].clone()
        if not self.enc_mode:
            for s in reversed(range(self.num_stages - 1)):
                P = self._unpool(P, f"dec.dec{s}.up")
                for i in range(self.dec_depths[s]):
                    P = self._block(P, f"dec.dec{s}.block{i}", self.dec_channels[s], self.dec_num_head[s],
                                    self.dec_patch_size[s], i % k)
                self.trace[f"dec{s}"] = P["feat"].clone()
        return P


class OffsetKeypointOracle:
    """OffsetKeypointPTv3 (offset_keypoint_ptv3.py:6-107); training=True as PTv3Oracle."""

    def __init__(self, backbone_conf, state_dict, num_keypoints=6, training=False):
        self.backbone = PTv3Oracle(backbone_conf, state_dict, prefix="backbone.", training=training)
        self.training = training
        self.sd = {k: v.detach().float().cpu() for k, v in state_dict.items() if k.startswith("head.")}
        if training:
            self.sd = _leafify(self.sd)
        self.K = num_keypoints

    def named_parameters(self):
        for k, v in self.backbone.sd.items():
            if v.requires_grad:
                yield "backbone." + k, v
        for k, v in self.sd.items():
            if v.requires_grad:
                yield k, v

    def named_buffers(self):
        for k, v in self.backbone.sd.items():
            if k.endswith(("running_mean", "running_var")):
                yield "backbone." + k, v
        for k, v in self.sd.items():
            if k.endswith(("running_mean", "running_var")):
                yield k, v

    def forward(self, data):
        P = self.backbone.backbone(data)
        sd = self.sd
        x = F.linear(P["feat"], sd["head.0.weight"], sd["head.0.bias"])
        x = F.batch_norm(x, sd["head.1.running_mean"], sd["head.1.running_var"],
                         sd["head.1.weight"], sd["head.1.bias"], self.training, 0.1, 1e-5)
        x = F.relu(x)
        x = F.linear(x, sd["head.3.weight"], sd["head.3.bias"])
        pred = x.view(-1, self.K, 4)
        out = {}
        if "target" in data:
            out.update(offset_keypoint_loss(pred, data["target"].float().cpu()))
        final = pred.clone()
        final[..., 3] = torch.sigmoid(pred[..., 3])
        out["pred"] = final
        out["logits"] = pred
        return out


class SegmentorOracle:
    """DefaultSegmentorV2 (pointcept/models/default.py:41-95) around "PT-v3m1" or "PT-v3m2": backbone -> (enc_mode:
    concatenate the pooled levels back up, :70-75) -> seg_head Linear -> criteria.  Only CrossEntropyLoss
    (models/losses/misc.py: nn.CrossEntropyLoss(ignore_index) * loss_weight) is restated for the loss value."""

    def __init__(self, backbone_conf, state_dict, variant="PT-v3m1", ignore_index=-1, loss_weight=1.0):
        cls = PTv3m2Oracle if variant == "PT-v3m2" else PTv3Oracle
        self.backbone = cls(backbone_conf, state_dict, prefix="backbone.")
        self.sd = {k: v.detach().float().cpu() for k, v in state_dict.items() if k.startswith("seg_head.")}
        self.ignore_index, self.loss_weight = ignore_index, loss_weight

    def forward(self, data):
        P = self.backbone.backbone(data)
        while "pooling_parent" in P:
            parent = P.pop("pooling_parent")
            inverse = P.pop("pooling_inverse")
            parent["feat"] = torch.cat([parent["feat"], P["feat"][inverse]], dim=-1)
            P = parent
        feat = P["feat"]
        logits = F.linear(feat, self.sd["seg_head.weight"], self.sd["seg_head.bias"]) if self.sd else feat
        out = {"seg_logits": logits}
        if "segment" in data:
            out["loss"] = F.cross_entropy(logits, data["segment"].long().cpu(),
                                          ignore_index=self.ignore_index) * self.loss_weight
        return out


def offset_keypoint_loss(pred, target):
    """offset_keypoint_ptv3.py:50-98."""
    offset_gt, mask_gt = target[..., :3], target[..., 3]
    offset_pred, mask_logits = pred[..., :3], pred[..., 3]
    cls_loss = F.binary_cross_entropy_with_logits(mask_logits, mask_gt, reduction="none").mean()
    valid = (mask_gt > 0.5).float().unsqueeze(-1)
    reg_loss = ((offset_pred - offset_gt).abs() * valid).sum() / (valid.sum() * 3 + 1e-6)
    return {"loss": cls_loss + reg_loss * 2.0, "cls_loss": cls_loss, "reg_loss": reg_loss}
