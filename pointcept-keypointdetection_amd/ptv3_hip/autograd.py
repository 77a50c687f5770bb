"""torch.autograd.Function per layer of the PTv3 path: forward AND backward run in libptv3_hip.so.

The reference trains through torch autograd over torch / spconv / torch_scatter ops
(engines/train.py:184-213); here every differentiable layer of the path is one Function whose two halves call
the C ABI (include/ptv3_hip.h, "training: backward kernels").  torch supplies the tape, device memory and the
parameter tensors (fp32 masters; activations may be bf16: weights are cast per call, weight gradients are
accumulated in fp32 by the kernels and returned in the parameter's dtype).
"""
import ctypes
import os

import torch
from torch.autograd import Function

from . import ops
from .lib import lib


def _wmat(w, dtype, cin_pad=None):
    """(cout, k..., cin) parameter -> (cout, kvol*cin_pad) K-contiguous matrix in `dtype`."""
    w = w.detach()
    if cin_pad is not None and cin_pad != w.shape[-1]:
        w = torch.nn.functional.pad(w, (0, cin_pad - w.shape[-1]))
    return w.reshape(w.shape[0], -1).to(dtype).contiguous()


def _weights(weight, dtype, cin_pad=None):
    """(natural (cout, kvol*cin) matrix, transposed / tap-mirrored (cin, kvol*cout) matrix or None) in `dtype`.
    With a FusedAdamW(shadow_dtype=...) attached both come from the shadows its step kernel keeps current; otherwise
    the natural one is cast here and the transposed one is built in backward."""
    if cin_pad is None or cin_pad == weight.shape[-1]:
        from .optim import weight_shadow
        sh = weight_shadow(weight, dtype)
        if sh is not None:
            return sh["nat"], sh["t"]
    return _wmat(weight, dtype, cin_pad), None


def _pad_cols(x, gran):
    pad = (-x.shape[1]) % gran
    return torch.nn.functional.pad(x, (0, pad)).contiguous() if pad else x


class LinearFn(Function):
    """y = x W^T + b   (nn.Linear; point_transformer_v3m1_base.py:188, 219, 232-244)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        gran = ops.k_granule(x.dtype)
        xp = _pad_cols(x.contiguous(), gran)
        w, wt = _weights(weight, x.dtype, xp.shape[1])
        ctx.save_for_backward(xp, weight)
        ctx.has_bias = bias is not None
        ctx.cin = x.shape[1]
        ctx.w, ctx.wt = w, wt   # the cast copy serves the backward too (one cast per step, not two)
        return ops.gemm(xp, w, bias=None if bias is None else bias.detach().float().contiguous())

    @staticmethod
    def backward(ctx, dy):
        xp, weight = ctx.saved_tensors
        dy = dy.contiguous()
        gran = ops.k_granule(dy.dtype)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # dx = dy W : the same GEMM with the transposed weight (cin, cout); cout padded to the K granule
            dyp = _pad_cols(dy, gran)
            wt = ctx.wt if ctx.wt is not None else _pad_cols(ctx.w[:, :ctx.cin].t().contiguous(), gran)
            dx = ops.gemm(dyp, wt)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            # the kernel wants cout in multiples of 4 (a class count like 13 or 19 is not): zero columns, dropped again
            dy4 = _pad_cols(dy, 4)
            dw = ops.gemm_tn(dy4, xp, with_bias=want_b)     # the bias gradient rides on the same pass over dy
            if want_b:
                dw, db = dw[0], dw[1][:dy.shape[1]]
            dw = dw[:dy.shape[1], :ctx.cin].to(weight.dtype)
        elif want_b:
            db = ops.col_reduce(dy)
        return dx, dw, db


class SubMConvFn(Function):
    """Submanifold sparse conv as implicit GEMM (spconv SubMConv3d; :277-284, 499-506).
    weight (cout, k, k, k, cin); nbr (n, kvol) int32; row_order optional visiting order."""

    @staticmethod
    def forward(ctx, x, weight, bias, nbr, row_order):
        gran = ops.k_granule(x.dtype)
        xp = _pad_cols(x.contiguous(), gran)
        kvol = nbr.shape[1]
        w, ctx.wt = _weights(weight, x.dtype, xp.shape[1])
        ctx.save_for_backward(xp, weight, nbr, row_order)
        ctx.has_bias = bias is not None
        ctx.cin = x.shape[1]
        return ops.gemm(xp, w, bias=None if bias is None else bias.detach().float().contiguous(), nbr=nbr,
                        kvol=kvol, row_order=row_order)

    @staticmethod
    def backward(ctx, dy):
        xp, weight, nbr, row_order = ctx.saved_tensors
        dy = dy.contiguous()
        kvol = nbr.shape[1]
        cout, cin = weight.shape[0], ctx.cin
        gran = ops.k_granule(dy.dtype)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # neighbour maps of a submanifold conv are symmetric: nbr[i][t] = j  <=>  nbr[j][kvol-1-t] = i, so
            # dx[j] = sum_t dy[nbr[j][t]] . W[:, kvol-1-t, :]  -- the forward kernel on mirrored, transposed taps
            dyp = _pad_cols(dy, gran)
            wt = ctx.wt
            if wt is None:
                w = weight.detach().reshape(cout, kvol, cin).flip(1).permute(2, 1, 0)  # (cin, kvol, cout)
                if dyp.shape[1] != cout:
                    w = torch.nn.functional.pad(w, (0, dyp.shape[1] - cout))
                wt = w.reshape(cin, -1).to(dy.dtype).contiguous()
            dx = ops.gemm(dyp, wt, nbr=nbr, kvol=kvol, row_order=row_order)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            dw = ops.gemm_tn(dy, xp, nbr, kvol, with_bias=want_b)
            if want_b:
                dw, db = dw
            dw = dw.view(cout, kvol, xp.shape[1])[:, :, :cin]
            dw = dw.reshape(weight.shape).to(weight.dtype)
        elif want_b:
            db = ops.col_reduce(dy)
        return dx, dw, db, None, None


class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x = x.contiguous()
        g, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        ctx.save_for_backward(x, g)
        ctx.eps = eps
        ctx.pdtype = weight.dtype
        return ops.layernorm(x, g, b, eps)

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        dx, dg, db = ops.layernorm_bwd(x, dy.contiguous(), g, ctx.eps)
        return dx, dg.to(ctx.pdtype), db.to(ctx.pdtype), None


def _all_reduce(t, group):
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


class BatchNormActFn(Function):
    """act(BatchNorm1d(x)) with batch statistics (training) or running statistics (eval)
    (:439-442, 508-511; offset_keypoint_ptv3.py head).  Running buffers are updated in place like torch
    (momentum, unbiased variance).  `sync` (a process group, or True for the default one) makes the statistics those
    of the batch over ALL ranks, as torch.nn.SyncBatchNorm does after engines/train.py:256-257 converted the model:
    column sums and the row count are all-reduced in forward, the two gradient sums in backward; weight / bias
    gradients stay local sums (DistributedDataParallel averages them)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, act, sync=None):
        x = x.contiguous()
        m = x.shape[0]
        group = None if sync is True else sync
        ctx.sync, ctx.group = bool(sync) and training, group
        plain = training and not ctx.sync and weight.dtype == torch.float32 and bias.dtype == torch.float32 and \
            (running_mean is None or (running_mean.dtype == torch.float32 and running_var.dtype == torch.float32))
        if plain:
            # the per-channel arithmetic in one launch (ptv3_bn_finalize) instead of ~14 element-wise ones
            total = ops.col_reduce(x)
            sq = ops.col_reduce(x, mu=total, mode=3, mu_scale=1.0 / m)[1].contiguous()
            with torch.no_grad():
                mean, rstd, scale, shift = ops.bn_finalize(total, sq, m, weight.detach(), bias.detach(), running_mean,
                                                           running_var, momentum if momentum is not None else 0.0, eps)
            gamma = weight.detach()
            ctx.save_for_backward(x, scale, shift, mean, rstd, gamma)
            ctx.training, ctx.act, ctx.pdtype, ctx.count, ctx.plain = training, act, weight.dtype, m, True
            return ops.affine_act(x, scale, shift, act)
        ctx.plain = False
        if training:
            if ctx.sync:
                head = torch.cat([ops.col_reduce(x).reshape(-1), x.new_full((1,), float(m), dtype=torch.float32)])
                head = _all_reduce(head, group)
                m = int(round(head[-1].item()))
                mean = head[:-1] / m
                var = _all_reduce(ops.col_reduce(x, mu=mean.contiguous(), mode=3)[1].contiguous(), group) / m
            else:
                mean = ops.col_reduce(x) / m
                var = ops.col_reduce(x, mu=mean.contiguous(), mode=3)[1] / m  # centred second pass (biased variance)
            if running_mean is not None:
                with torch.no_grad():
                    running_mean.mul_(1 - momentum).add_(mean.to(running_mean.dtype), alpha=momentum)
                    unbiased = var * (m / max(m - 1, 1))
                    running_var.mul_(1 - momentum).add_(unbiased.to(running_var.dtype), alpha=momentum)
        else:
            mean, var = running_mean.float(), running_var.float()
        rstd = torch.rsqrt(var + eps)
        gamma = weight.detach().float()
        scale = (gamma * rstd).contiguous()
        shift = (bias.detach().float() - mean * scale).contiguous()
        ctx.save_for_backward(x, scale, shift, mean.contiguous(), rstd.contiguous(), gamma)
        ctx.training, ctx.act, ctx.pdtype, ctx.count = training, act, weight.dtype, m
        return ops.affine_act(x, scale, shift, act)

    @staticmethod
    def backward(ctx, dy):
        x, scale, shift, mean, rstd, gamma = ctx.saved_tensors
        dy = dy.contiguous()
        m = ctx.count
        dpre = ops.act_bwd(dy, x, ctx.act, scale, shift) if ctx.act != ops.ACT_NONE else dy
        sums = ops.col_reduce(dpre, x, mean, rstd, mode=2)  # sum dpre, sum dpre * xhat
        dbeta, dgamma = sums[0], sums[1]
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.plain:
                ca, cb, cc = ops.bn_bwd_coeffs(sums, m, gamma, rstd, mean)
            elif ctx.training:
                tot = _all_reduce(sums.clone(), ctx.group) if ctx.sync else sums
                k1, k2 = tot[0] / m, tot[1] / m
                ca = gamma * rstd
                cb = -ca * rstd * k2
                cc = ca * (mean * rstd * k2 - k1)
            else:
                ca, cb, cc = scale, torch.zeros_like(scale), torch.zeros_like(scale)
            dx = ops.affine2(dpre, x, ca.contiguous(), cb.contiguous(), cc.contiguous())
        return dx, dgamma.to(ctx.pdtype), dbeta.to(ctx.pdtype), None, None, None, None, None, None, None


class ActFn(Function):
    @staticmethod
    def forward(ctx, x, act):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.act = act
        return ops.affine_act(x, None, None, act)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.act_bwd(dy.contiguous(), x, ctx.act), None


class WindowAttentionFn(Function):
    """softmax(scale q k^T) v per serialized window, gather / scatter fused (:188-216)."""

    @staticmethod
    def forward(ctx, qkv, win_order, win_inverse, heads, patch, scale, cu=None):
        qkv = qkv.contiguous()
        out, lse = ops.window_attention_train(qkv, win_order, win_inverse, heads, patch, scale, cu)
        ctx.save_for_backward(qkv, out, lse, win_order, win_inverse, cu)
        ctx.cfg = (heads, patch, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, win_order, win_inverse, cu = ctx.saved_tensors
        heads, patch, scale = ctx.cfg
        return (ops.window_attention_train_bwd(qkv, out, dout.contiguous(), lse, win_order, win_inverse, heads, patch,
                                               scale, cu), None, None, None, None, None, None)


class WindowAttentionDropFn(Function):
    """WindowAttentionFn with attention dropout: the mask is regenerated in the backward from (p_drop, seed)."""

    @staticmethod
    def forward(ctx, qkv, win_order, win_inverse, heads, patch, scale, p_drop, seed, cu=None):
        qkv = qkv.contiguous()
        out = ops.window_attention_drop(qkv, win_order, win_inverse, heads, patch, scale, p_drop, seed, cu)
        ctx.save_for_backward(qkv, out, win_order, win_inverse, cu)
        ctx.cfg = (heads, patch, scale, p_drop, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, win_order, win_inverse, cu = ctx.saved_tensors
        heads, patch, scale, p_drop, seed = ctx.cfg
        return (ops.window_attention_drop_bwd(qkv, out, dout.contiguous(), win_order, win_inverse, heads, patch, scale,
                                              p_drop, seed, cu), None, None, None, None, None, None, None, None)


class WindowAttentionRpeFn(Function):
    """Window attention with the RPE bias looked up in the kernel (:29-48, :196-204): gradients for qkv AND the bias
    table (each (query, key) pair's dS goes to the three table entries it read)."""

    @staticmethod
    def forward(ctx, qkv, rpe_table, win_order, win_inverse, grid_coord, heads, patch, scale, pos_bnd):
        qkv = qkv.contiguous()
        table = rpe_table.detach().float().contiguous()
        out = ops.window_attention_rpe(qkv, win_order, win_inverse, heads, patch, scale, grid_coord, table, pos_bnd)
        if out is None:
            raise NotImplementedError("training with enable_rpe: the window does not fit the resident-window kernel")
        ctx.save_for_backward(qkv, out, win_order, win_inverse, grid_coord, table)
        ctx.cfg = (heads, patch, scale, pos_bnd, rpe_table.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, win_order, win_inverse, grid_coord, table = ctx.saved_tensors
        heads, patch, scale, pos_bnd, pdtype = ctx.cfg
        dqkv, dtable = ops.window_attention_rpe_bwd(qkv, out, dout.contiguous(), win_order, win_inverse, heads, patch,
                                                    scale, grid_coord, table, pos_bnd)
        return dqkv, dtable.to(pdtype), None, None, None, None, None, None, None


class SwinAttentionFn(Function):
    """Swin3D cRSE window attention (ops.swin_attention; swin3d_layers.py:556-569) with its HIP backward: gradients of
    q, k, v and of the three concatenated tables."""

    @staticmethod
    def forward(ctx, q, k, v, q_table, k_table, v_table, table_offsets, n2n, w_start, n_crse, max_tokens):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        tabs = [t.detach().float().contiguous() for t in (q_table, k_table, v_table)]
        out = ops.swin_attention(q, k, v, *tabs, table_offsets, n2n, w_start, n_crse, max_tokens)
        ctx.save_for_backward(q, k, v, *tabs, n2n, w_start, n_crse)
        ctx.cfg = (tuple(int(t) for t in table_offsets), int(max_tokens), q_table.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, qt, kt, vt, n2n, w_start, n_crse = ctx.saved_tensors
        offs, max_tokens, tdt = ctx.cfg
        dq, dk, dv, dqt, dkt, dvt = ops.swin_attention_bwd(q, k, v, dout.contiguous(), qt, kt, vt, offs, n2n, w_start,
                                                           n_crse, max_tokens)
        return dq, dk, dv, dqt.to(tdt), dkt.to(tdt), dvt.to(tdt), None, None, None, None, None


def swin_attention(q, k, v, q_table, k_table, v_table, table_offsets, n2n, w_start, n_crse, max_tokens):
    return SwinAttentionFn.apply(q, k, v, q_table, k_table, v_table, table_offsets, n2n, w_start, n_crse, max_tokens)


class SegmentMaxFn(Function):
    """torch_scatter.segment_csr(feat[indices], idx_ptr, reduce="max") (:416-421) over serialized-order runs."""

    @staticmethod
    def forward(ctx, feat, order0, seg_start, n_out):
        feat = feat.contiguous()
        ctx.save_for_backward(feat, order0, seg_start)
        return ops.pool_max(feat, order0, seg_start, n_out)

    @staticmethod
    def backward(ctx, dy):
        feat, order0, seg_start = ctx.saved_tensors
        return ops.pool_max_bwd(feat, dy.contiguous(), order0, seg_start), None, None, None


class ClusterGatherFn(Function):
    """point.feat[inverse] of SerializedUnpooling (:478): gather forward (index plumbing), deterministic
    segment-sum backward."""

    @staticmethod
    def forward(ctx, feat, cluster, order0, seg_start):
        ctx.save_for_backward(order0, seg_start)
        ctx.n_out = feat.shape[0]
        return feat.index_select(0, cluster)

    @staticmethod
    def backward(ctx, dy):
        order0, seg_start = ctx.saved_tensors
        return ops.segment_sum(dy.contiguous(), order0, seg_start, ctx.n_out), None, None, None


def linear(x, weight, bias=None):
    return LinearFn.apply(x, weight, bias)


def subm_conv(x, weight, bias, nbr, row_order=None):
    return SubMConvFn.apply(x, weight, bias, nbr, row_order)


def layer_norm(x, weight, bias, eps):
    return LayerNormFn.apply(x, weight, bias, eps)


def batch_norm_act(x, bn, act=ops.ACT_NONE):
    sync = getattr(bn, "sync_group", None)
    if sync is not None:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(None if sync is True else sync) > 1):
            sync = None
    return BatchNormActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training,
                                bn.momentum if bn.momentum is not None else 0.1, bn.eps, act, sync)


def activation(x, act):
    return ActFn.apply(x, act)


def window_attention(qkv, win_order, win_inverse, heads, patch, scale, cu=None):
    return WindowAttentionFn.apply(qkv, win_order, win_inverse, heads, patch, scale, cu)


def window_attention_drop(qkv, win_order, win_inverse, heads, patch, scale, p_drop, seed, cu=None):
    return WindowAttentionDropFn.apply(qkv, win_order, win_inverse, heads, patch, scale, float(p_drop), int(seed), cu)


def window_attention_rpe(qkv, rpe_table, win_order, win_inverse, grid_coord, heads, patch, scale, pos_bnd):
    return WindowAttentionRpeFn.apply(qkv, rpe_table, win_order, win_inverse, grid_coord, heads, patch, scale, pos_bnd)


def segment_max(feat, order0, seg_start, n_out):
    return SegmentMaxFn.apply(feat, order0, seg_start, n_out)


def cluster_gather(feat, cluster, order0, seg_start):
    return ClusterGatherFn.apply(feat, cluster, order0, seg_start)


# -------------------------------------------------------------------------------------------------
# One Function per Block for training: the same kernels as the per-layer Functions above, issued back to back
# without going through the autograd engine between them (13 Function round trips per block become one).
# -------------------------------------------------------------------------------------------------
def _lin_fwd(x, w, b, res=None):
    return ops.gemm(x, w, bias=None if b is None else b.detach().float().contiguous(), res=res)


def _branch_add(skip, branch_in, w, b, mask):
    """skip + mask * linear(branch_in): without DropPath the residual rides in the GEMM epilogue, with it the
    per-point factor and the sum are one element-wise pass."""
    if mask is None:
        return _lin_fwd(branch_in, w, b, res=skip)
    return torch.addcmul(skip, _lin_fwd(branch_in, w, b), mask)


def _lin_bwd(dy, x, w_pair, gran):
    """-> dx, dW (fp32, (cout, cin)), db (fp32) of y = x w^T + b; w_pair = _weights(...) of the layer."""
    w_cast, wt = w_pair
    if wt is None:
        wt = _pad_cols(w_cast.t().contiguous(), gran)
    dx = ops.gemm(_pad_cols(dy, gran), wt)
    dw, db = ops.gemm_tn(dy, x, with_bias=True)
    return dx, dw, db


class BlockFn(Function):
    """Block.forward (point_transformer_v3m1_base.py:318-338, pre-norm form) with LayerNorm / GELU layers:
      c  = LN0(lin(conv(conv_feat)));  f1 = feat + c
      f2 = f1 + mask1 * proj(attn(qkv(LN1(f1))))
      out = f2 + mask2 * fc2(GELU(fc1(LN2(f2))))
    mask1 / mask2 are the per-point DropPath factors (None when drop_path = 0).  `conv_feat` is the tensor the
    xCPE conv reads: `feat` itself, except in the first decoder block, where the reference leaves the sparse tensor
    on the skip branch (see SerializedUnpooling)."""

    @staticmethod
    def forward(ctx, feat, conv_feat, conv_w, conv_b, lin_w, lin_b, ln0_g, ln0_b, n1_g, n1_b, qkv_w, qkv_b,
                proj_w, proj_b, n2_g, n2_b, fc1_w, fc1_b, fc2_w, fc2_b, nbr, row_order, wo, wi, heads, patch, scale,
                mask1, mask2, eps, cu=None):
        dt = feat.dtype
        feat = feat.contiguous()
        same = conv_feat is None
        xin = feat if same else conv_feat.contiguous()
        kvol = nbr.shape[1]
        f32 = lambda t: t.detach().float().contiguous()  # noqa: E731
        w_conv = _weights(conv_w, dt)
        w_lin, w_qkv, w_proj = _weights(lin_w, dt), _weights(qkv_w, dt), _weights(proj_w, dt)
        w_fc1, w_fc2 = _weights(fc1_w, dt), _weights(fc2_w, dt)
        c1 = ops.gemm(xin, w_conv[0], bias=f32(conv_b), nbr=nbr, kvol=kvol, row_order=row_order)
        c2 = _lin_fwd(c1, w_lin[0], lin_b)
        f1 = ops.layernorm(c2, f32(ln0_g), f32(ln0_b), eps, res=feat)
        t3 = ops.layernorm(f1, f32(n1_g), f32(n1_b), eps)
        qkv = _lin_fwd(t3, w_qkv[0], qkv_b)
        a, lse = ops.window_attention_train(qkv, wo, wi, heads, patch, scale, cu)
        f2 = _branch_add(f1, a, w_proj[0], proj_b, mask1)
        t5 = ops.layernorm(f2, f32(n2_g), f32(n2_b), eps)
        h0 = _lin_fwd(t5, w_fc1[0], fc1_b)
        h = ops.affine_act(h0, None, None, ops.ACT_GELU)
        out = _branch_add(f2, h, w_fc2[0], fc2_b, mask2)
        ctx.save_for_backward(xin, c1, c2, f1, t3, qkv, a, f2, t5, h0, h, nbr, row_order, wo, wi, mask1, mask2,
                              conv_w, ln0_g, n1_g, n2_g, cu, lse)
        ctx.cast = (w_conv, w_lin, w_qkv, w_proj, w_fc1, w_fc2)
        ctx.cfg = (heads, patch, scale, eps, same)
        return out

    @staticmethod
    def backward(ctx, dout):
        (xin, c1, c2, f1, t3, qkv, a, f2, t5, h0, h, nbr, row_order, wo, wi, mask1, mask2, conv_w, ln0_g, n1_g,
         n2_g, cu, lse) = ctx.saved_tensors
        w_conv, w_lin, w_qkv, w_proj, w_fc1, w_fc2 = ctx.cast
        heads, patch, scale, eps, same = ctx.cfg
        dt = dout.dtype
        gran = ops.k_granule(dt)
        pd = conv_w.dtype
        dout = dout.contiguous()
        f32 = lambda t: t.detach().float().contiguous()  # noqa: E731
        # ---- MLP branch
        dm = dout if mask2 is None else dout * mask2
        dh, dW_fc2, db_fc2 = _lin_bwd(dm, h, w_fc2, gran)
        dh0 = ops.act_bwd(dh, h0, ops.ACT_GELU)
        dt5, dW_fc1, db_fc1 = _lin_bwd(dh0, t5, w_fc1, gran)
        df2, dg2, db2 = ops.layernorm_bwd(f2, dt5, f32(n2_g), eps, add=dout)
        # ---- attention branch
        dp = df2 if mask1 is None else df2 * mask1
        da, dW_proj, db_proj = _lin_bwd(dp, a, w_proj, gran)
        dqkv = ops.window_attention_train_bwd(qkv, a, da.contiguous(), lse, wo, wi, heads, patch, scale, cu)
        dt3, dW_qkv, db_qkv = _lin_bwd(dqkv, t3, w_qkv, gran)
        df1, dg1, db1 = ops.layernorm_bwd(f1, dt3, f32(n1_g), eps, add=df2)
        # ---- xCPE branch
        dc2, dg0, db0 = ops.layernorm_bwd(c2, df1, f32(ln0_g), eps)
        dc1, dW_lin, db_lin = _lin_bwd(dc2, c1, w_lin, gran)
        kvol = nbr.shape[1]
        cout, cin = conv_w.shape[0], conv_w.shape[-1]
        wt = w_conv[1]
        if wt is None:
            wt = w_conv[0].view(cout, kvol, cin).flip(1).permute(2, 1, 0).reshape(cin, -1).contiguous()
        # same: the conv read `feat` itself, its input gradient lands on df1 in the GEMM epilogue
        dxin = ops.gemm(_pad_cols(dc1, gran), wt, nbr=nbr, kvol=kvol, row_order=row_order, res=df1 if same else None)
        dW_conv, db_conv = ops.gemm_tn(dc1, xin, nbr, kvol, with_bias=True)
        dW_conv = dW_conv.view(conv_w.shape)
        dfeat = dxin if same else df1
        c = lambda t: t.to(pd)  # noqa: E731
        return (dfeat, None if same else dxin, c(dW_conv), c(db_conv), c(dW_lin), c(db_lin), c(dg0), c(db0), c(dg1),
                c(db1), c(dW_qkv), c(db_qkv), c(dW_proj), c(db_proj), c(dg2), c(db2), c(dW_fc1), c(db_fc1), c(dW_fc2),
                c(db_fc2), None, None, None, None, None, None, None, None, None, None, None)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _f32_ptr(p, name):
    if p.dtype != torch.float32 or not p.is_contiguous():
        raise TypeError(f"BlockNativeFn: {name} must be a contiguous fp32 parameter")
    return p.data_ptr()


def _transposed(pair, kvol):
    """(natural, transposed) weight pair of _weights(); the transposed one is built here when no optimizer keeps it"""
    nat, t = pair
    if t is None:
        if kvol == 1:
            t = nat.t().contiguous()
        else:
            cout = nat.shape[0]
            t = nat.view(cout, kvol, -1).flip(1).permute(2, 1, 0).reshape(nat.shape[1] // kvol, -1).contiguous()
    return nat, t


class BlockNativeFn(Function):
    """BlockFn with the forward and the backward each issued by ONE native call (ptv3_block_train_fwd / _bwd, csrc/
    block_train.hip): same kernels, same order, same results - the ~12 + ~30 launches per block no longer pass through
    Python wrappers one by one (a training step of the fork model was host-bound on them)."""

    @staticmethod
    def forward(ctx, feat, conv_feat, conv_w, conv_b, lin_w, lin_b, ln0_g, ln0_b, n1_g, n1_b, qkv_w, qkv_b,
                proj_w, proj_b, n2_g, n2_b, fc1_w, fc1_b, fc2_w, fc2_b, nbr, row_order, wo, wi, heads, patch, scale,
                mask1, mask2, eps, cu=None, keep1=1.0, keep2=1.0):
        from .lib import BlockTrain
        dt = feat.dtype
        feat = feat.contiguous()
        xin = None if conv_feat is None else conv_feat.contiguous()
        n, c = feat.shape
        hidden = fc1_w.shape[0]
        kvol = nbr.shape[1]
        dev = feat.device
        taped = any(ctx.needs_input_grad)      # the transposed weights are only read by the backward
        ws = [_transposed(_weights(w, dt), kvol if i == 0 else 1) if taped else (_weights(w, dt)[0], None)
              for i, w in enumerate((conv_w, lin_w, qkv_w, proj_w, fc1_w, fc2_w))]
        for name, u in (("mask1", mask1), ("mask2", mask2)):    # uniform draws per point (fp32), see drop_factor
            if u is not None and (u.dtype != torch.float32 or u.numel() != n or not u.is_contiguous()):
                raise TypeError(f"BlockNativeFn: {name} must be a contiguous fp32 draw per point")
        # activations the backward needs: ONE allocation; the descriptor gets addresses inside it (only `out`, which
        # leaves this Function, becomes a tensor view)
        esz = feat.element_size()
        flat = torch.empty(n * (9 * c + 3 * c + 2 * hidden), dtype=dt, device=dev)
        base = flat.data_ptr()
        nc, nh = n * c * esz, n * hidden * esz
        act = [base + i * nc for i in range(8)]                                  # c1 c2 f1 t3 a f2 t5 out
        qkv_p, h0_p, h_p = base + 8 * nc, base + 11 * nc, base + 11 * nc + nh
        out = flat[7 * n * c:8 * n * c].view(n, c)
        vec = [_f32_ptr(p, nm) for nm, p in (("b_conv", conv_b), ("b_lin", lin_b), ("b_qkv", qkv_b), ("b_proj", proj_b),
                                            ("b_fc1", fc1_b), ("b_fc2", fc2_b), ("g0", ln0_g), ("b0", ln0_b),
                                            ("g1", n1_g), ("b1", n1_b), ("g2", n2_g), ("b2", n2_b))]
        wp = [_ptr(w) for w, _ in ws] + [_ptr(t) for _, t in ws]
        # positional construction in the field order of struct ptv3_block_train (one call instead of ~70 attribute sets)
        b = BlockTrain(n, wo.shape[0], c, hidden, int(heads), int(patch), kvol, 0 if cu is None else cu.numel() - 1,
                       ops._dt(feat), 0, float(scale), float(eps), float(keep1), float(keep2), 0.0,
                       _ptr(nbr), _ptr(row_order), _ptr(wo), _ptr(wi), _ptr(cu), feat.data_ptr(), _ptr(xin),
                       *wp, *vec, _ptr(mask1), _ptr(mask2),
                       act[0], act[1], act[2], act[3], qkv_p, act[4], act[5], act[6], h0_p, h_p, act[7])
        nb = lib.ptv3_block_train_workspace_bytes(ctypes.byref(b), 0)
        wsb = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
        b.workspace, b.workspace_bytes = wsb.data_ptr(), nb
        lse = torch.empty(wo.shape[0] * int(heads), dtype=torch.float32, device=dev)   # the attention's log-sum-exp rows
        b.attn_lse = lse.data_ptr()
        lib.check(lib.ptv3_block_train_fwd(ctypes.byref(b), ops._stream()), "ptv3_block_train_fwd")
        ctx.save_for_backward(feat, xin, nbr, row_order, wo, wi, cu, mask1, mask2)
        ctx.lse = lse
        ctx.block, ctx.flat, ctx.keep = b, flat, (ws, conv_w, (conv_b, lin_b, qkv_b, proj_b, fc1_b, fc2_b, ln0_g, ln0_b,
                                                                n1_g, n1_b, n2_g, n2_b))
        ctx.shapes = (n, c, hidden, kvol, conv_w.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        b, (ws, conv_w, _) = ctx.block, ctx.keep
        n, c, hidden, kvol, conv_shape = ctx.shapes
        same = ctx.saved_tensors[1] is None
        dout = dout.contiguous()
        dev, dt = dout.device, dout.dtype
        f32 = dict(dtype=torch.float32, device=dev)
        dfeat = torch.empty((n, c), dtype=dt, device=dev)
        dconv = None if same else torch.empty((n, c), dtype=dt, device=dev)
        # parameter gradients: one fp32 allocation (weights, biases, the three LayerNorm (2, c) pairs)
        sizes = (c * kvol * c, c * c, 3 * c * c, c * c, hidden * c, c * hidden, c, c, 3 * c, c, hidden, c, 2 * c, 2 * c, 2 * c)
        flat = torch.empty(sum(sizes), **f32)
        parts, at = [], 0
        for sz in sizes:
            parts.append(flat[at:at + sz])
            at += sz
        for name, t in zip(("dw_conv", "dw_lin", "dw_qkv", "dw_proj", "dw_fc1", "dw_fc2", "db_conv", "db_lin", "db_qkv",
                            "db_proj", "db_fc1", "db_fc2", "dln0", "dln1", "dln2"), parts):
            setattr(b, name, t.data_ptr())
        b.dout, b.dfeat, b.dconv_feat = dout.data_ptr(), dfeat.data_ptr(), _ptr(dconv)
        nb = lib.ptv3_block_train_workspace_bytes(ctypes.byref(b), 1)
        wsb = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
        b.workspace, b.workspace_bytes = wsb.data_ptr(), nb
        lib.check(lib.ptv3_block_train_bwd(ctypes.byref(b), ops._stream()), "ptv3_block_train_bwd")
        (dw_conv, dw_lin, dw_qkv, dw_proj, dw_fc1, dw_fc2, db_conv, db_lin, db_qkv, db_proj, db_fc1, db_fc2, dln0, dln1,
         dln2) = parts
        ctx.flat = ctx.lse = None
        return (dfeat, dconv, dw_conv.view(conv_shape), db_conv, dw_lin.view(c, c), db_lin, dln0[:c], dln0[c:], dln1[:c],
                dln1[c:], dw_qkv.view(3 * c, c), db_qkv, dw_proj.view(c, c), db_proj, dln2[:c], dln2[c:],
                dw_fc1.view(hidden, c), db_fc1, dw_fc2.view(c, hidden), db_fc2, None, None, None, None, None, None, None,
                None, None, None, None, None, None)


_NATIVE_BLOCK = os.environ.get("PTV3_BLOCK_NATIVE", "1") != "0"


def drop_factor(drop, dtype):
    """(n, 1) DropPath factor in `dtype` from a (uniform draw (n) fp32, keep probability) pair, or None"""
    if drop is None:
        return None
    u, keep = drop
    return ((u < keep).to(torch.float32) * (1.0 / keep if keep > 0.0 else 0.0)).to(dtype).unsqueeze(1)


def block(feat, conv_feat, blk_params, nbr, row_order, wo, wi, heads, patch, scale, drop1, drop2, eps, cu=None):
    """drop1 / drop2: None or (u, keep) - a uniform draw per point (fp32, contiguous) and the keep probability of the
    attention / MLP branch's DropPath (factor u < keep ? 1 / keep : 0, as timm's drop_path with scale_by_keep)."""
    if _NATIVE_BLOCK and blk_params[0].dtype == torch.float32:
        u1, k1 = drop1 if drop1 is not None else (None, 1.0)
        u2, k2 = drop2 if drop2 is not None else (None, 1.0)
        return BlockNativeFn.apply(feat, conv_feat, *blk_params, nbr, row_order, wo, wi, heads, patch, scale, u1, u2, eps,
                                   cu, float(k1), float(k2))
    return BlockFn.apply(feat, conv_feat, *blk_params, nbr, row_order, wo, wi, heads, patch, scale,
                         drop_factor(drop1, feat.dtype), drop_factor(drop2, feat.dtype), eps, cu)
