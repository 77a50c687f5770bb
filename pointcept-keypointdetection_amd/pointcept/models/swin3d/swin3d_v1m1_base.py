"""Swin3D U-Net on MI355X: registry name "Swin3D-v1m1" (SURVEY.md section 8, rows A19 / f4).

Counterpart of the reference's pointcept/models/swin3d/swin3d_v1m1_base.py (`Swin3DUNet`, constructor keywords of
:30-49, module tree and parameter names of :71-146) without MinkowskiEngine: a sparse tensor is a `_Level` (int32
voxel coordinates at a tensor stride, fp32 features, and the cRSE carrier `cfeat` = the reference's `coords_sp.F`),
the coordinate manager's jobs are done by this package's device primitives:

  reference (MinkowskiEngine / Swin3D.sparse_dl)               here
  ME.TensorField(...).sparse(), UNWEIGHTED_AVERAGE (:162-181)   key sort + segments (ptv3_argsort_i64, ptv3_pool_segments),
                                                                ptv3_segment_sum / count
  MinkowskiConvolution k=3 + BN + ReLU (mink_layers.py:49-80)   ptv3_subm_neighbors + ptv3_gemm (gathered, BN and ReLU folded)
  BasicLayer window maps (swin3d_layers.py:715-824)             ptv3_swin_window_keys + sort + segments (swin3d_layers.window_attn_args)
  SelfAttnAIOFunction (:556-569)                                ptv3_swin_attn_fwd
  MinkowskiMaxPooling coordinates + GridCoordsDown (:180-231)   key sort + segments, segment mean, nearest-to-mean member
  KNN k=16 + LayerNorm + Linear + MaxPool1d (:274-314)          LayerNorm + Linear per source voxel, then ptv3_knn_query_cells and a
                                                                segment max over the 16 gathered rows (same result: both are per-row)
  Upsample: Linear + 3-NN inverse-distance blend (:320-378)     ptv3_gemm + ptv3_knn_query_cells + ptv3_interpolation_forward
  sp.slice(in_field) (:244)                                     row gather by the voxel id of every input point

PARITY UNPINNED (oracle/swin3d.py header): MinkowskiEngine and microsoft/Swin3D are not in the reference tree.  Choices
the reference leaves to those libraries and this file fixes: voxels are numbered in sorted (batch, x, y, z) order;
the stem kernel's 27 taps are read x-fastest (`kernel[t]`, t = (dx+1) + 3 (dy+1) + 9 (dz+1)); KNN distances are
Euclidean (as libs/pointops, from which Swin3D's KNN descends); GridCoordsDown keeps the LOWEST-numbered member among
those within 1e-4 (relative) of the smallest distance to the cell mean (the reference keeps whichever equal-distance
member's write lands last - in a two-voxel cell both members are equidistant by construction).
Training: the same forward with every feature layer taped (Functions of ptv3_hip/autograd.py, the cRSE attention's own
backward kernel); geometry (voxel ids, window maps, neighbour indices, the signal carriers) carries no gradient.
"""
import torch
import torch.nn as nn

import pointops
from ptv3_hip import ops
from ptv3_hip import autograd as A
from pointcept.models.builder import MODELS
from pointcept.models.utils.hip_layers import Linear, LayerNorm, BatchNorm1d
from .swin3d_layers import WindowStage


class _Level:
    """One resolution of the sparse tensor pair (sp, coords_sp) the reference threads through its layers."""

    def __init__(self, coords, stride, feat, cfeat, offset):
        self.coords, self.stride, self.feat, self.cfeat, self.offset = coords, stride, feat, cfeat, offset

    @property
    def xyz(self):
        return self.cfeat[:, 1:4].contiguous()

    def local_xyz(self):
        """swin3d_layers.py:855-858: sub-voxel position of the voxel's signal carrier, in units of this level's voxel."""
        return _true_div(self.cfeat[:, 1:4] - self.coords[:, 1:].float(), self.stride)

    def signals(self):
        return self.cfeat[:, 4:].contiguous()


def _true_div(x, s):
    """x / s as an IEEE division.  A Python-scalar divisor lets torch multiply by the rounded reciprocal on the device,
    which moves sub-voxel offsets by an ulp - enough to flip floor() in the cRSE table index of a few pairs per scene."""
    return x / torch.full((), float(s), dtype=x.dtype, device=x.device)


def _key(coords, div):
    c = coords.long()
    if int(c.min()) < 0 or int(c[:, 1:].max()) >= 65536 * div:
        raise ValueError("Swin3D: voxel coordinates must lie in [0, 65536)")
    return (((c[:, 0] << 16 | c[:, 1] // div) << 16 | c[:, 2] // div) << 16 | c[:, 3] // div).contiguous()


def _segments(key):
    """order (n), cluster (n), seg_start (m+1), m of the sorted unique keys (np.unique(return_inverse))."""
    order, cluster, seg_start, m = ops.voxel_unique(key)
    return order, cluster, seg_start, m


def _offsets(batch_col, nb):
    return torch.cumsum(torch.bincount(batch_col.long(), minlength=nb), 0).int()


def _segment_mean(x, order, seg_start, m):
    cnt = (seg_start[1:] - seg_start[:-1]).float().unsqueeze(1)
    return ops.segment_sum(x.contiguous(), order, seg_start, m) / cnt


class _ConvBNRelu(nn.Module):
    """mink_layers.MinkConvBNRelu (:49-80): `conv_layers` = [MinkowskiConvolution (`kernel` (27, cin, cout), no bias),
    MinkowskiBatchNorm (`bn` = BatchNorm1d), ReLU]."""

    class _Conv(nn.Module):
        def __init__(self, cin, cout, k):
            super().__init__()
            self.in_channels, self.out_channels, self.kernel_size = cin, cout, k
            self.kernel = nn.Parameter(torch.empty(k ** 3, cin, cout))
            nn.init.kaiming_uniform_(self.kernel, a=5 ** 0.5)

    class _BN(nn.Module):
        def __init__(self, c):
            super().__init__()
            self.bn = BatchNorm1d(c)

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1):
        super().__init__()
        assert kernel_size == 3 and stride == 1, "only the stem configuration of Swin3DUNet"
        self.conv_layers = nn.Sequential(self._Conv(in_channels, out_channels, kernel_size), self._BN(out_channels),
                                         nn.ReLU(inplace=True))

    def forward(self, level):
        level.feat = _conv3_bn(level.coords, level.feat, self.conv_layers[0], self.conv_layers[1].bn, ops.ACT_RELU,
                               self.training)
        return level


def _conv3_bn(coords, x, conv, bn, act, training, res=None):
    """MinkowskiConvolution (kernel (k^3, cin, cout), no bias) -> MinkowskiBatchNorm [-> act] [+ res] on the occupied
    voxels `coords`: ptv3_subm_neighbors + one gathered ptv3_gemm with the BatchNorm folded (eval), or the taped
    Functions with batch statistics (training)."""
    gran = ops.k_granule(x.dtype)
    cin_pad = (conv.in_channels + gran - 1) // gran * gran
    k = conv.kernel_size
    # this package's gather order is tap d = (a k + b) k + c with (a, b, c) the x, y, z offsets; the kernel is read
    # x-fastest (see the module docstring)
    a, b, c = torch.meshgrid(torch.arange(k), torch.arange(k), torch.arange(k), indexing="ij")
    t_me = (a + k * b + k * k * c).reshape(-1).to(conv.kernel.device)
    nbr, _ = ops.subm_neighbors(coords, k)
    if training:
        # taped: the kernel re-indexed to this package's tap order as a differentiable view, batch-statistic BN
        w5 = conv.kernel[t_me].permute(2, 0, 1).reshape(conv.out_channels, k, k, k, conv.in_channels)
        y = bn(A.subm_conv(x, w5, None, nbr), act=act)
        return y if res is None else y + res
    if cin_pad != conv.in_channels:
        x = torch.nn.functional.pad(x, (0, cin_pad - conv.in_channels)).contiguous()
    w = conv.kernel.detach()[t_me]                                   # (27 taps in gather order, cin, cout)
    w = torch.nn.functional.pad(w, (0, 0, 0, cin_pad - conv.in_channels))
    w = w.permute(2, 0, 1).reshape(conv.out_channels, -1).to(x.dtype).contiguous()
    scale, shift = bn.folded()
    return ops.gemm(x, w, nbr=nbr, kvol=k ** 3, bn_scale=scale, bn_shift=shift, act=act, res=res)


class MinkResBlock(nn.Module):
    """mink_layers.MinkResBlock (:115-155): conv1 -> norm1 -> ReLU -> conv2 -> norm2 -> (+ input) -> ReLU, both
    convolutions 3^3 on the occupied voxels without bias (the `stem_transformer=False` stem, swin3d_v1m1_base.py:69-85)."""

    def __init__(self, in_channels, out_channels, stride=1, dilation=1):
        super().__init__()
        assert stride == 1 and dilation == 1 and in_channels == out_channels, "only the stem configuration of Swin3DUNet"
        self.conv1 = _ConvBNRelu._Conv(in_channels, out_channels, 3)
        self.norm1 = _ConvBNRelu._BN(out_channels)
        self.conv2 = _ConvBNRelu._Conv(out_channels, out_channels, 3)
        self.norm2 = _ConvBNRelu._BN(out_channels)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, level):
        x = level.feat
        y = _conv3_bn(level.coords, x, self.conv1, self.norm1.bn, ops.ACT_RELU, self.training)
        y = _conv3_bn(level.coords, y, self.conv2, self.norm2.bn, ops.ACT_NONE, self.training, res=x)   # out += residual
        if self.training:
            level.feat = A.activation(y, ops.ACT_RELU)
        else:
            c = y.shape[1]
            one, zero = torch.ones(c, device=y.device), torch.zeros(c, device=y.device)
            level.feat = ops.affine_act(y, one, zero, ops.ACT_RELU)
        return level


def _pool_cells(level, stride):
    """Cells of `stride` voxels per axis (MinkowskiMaxPooling's output coordinates) and GridCoordsDown (:180-231): the
    member nearest (over ALL carried columns, :203-206) to its cell's mean carries the cell's signals."""
    new_stride = level.stride * stride
    order, cluster, seg_start, m = _segments(_key(level.coords, new_stride))
    head = order[seg_start[:-1].long()]
    coords = (level.coords[head] // new_stride * new_stride).clone()
    coords[:, 0] = level.coords[head, 0]
    mean = _segment_mean(level.cfeat, order, seg_start, m)
    dist = (mean[cluster] - level.cfeat).pow(2).sum(1).sqrt()
    n = dist.shape[0]
    best = torch.full((m,), float("inf"), device=dist.device).scatter_reduce(0, cluster, dist, "amin")
    near = dist <= best[cluster] * (1 + 1e-4) + 1e-12
    ids = torch.where(near, torch.arange(n, device=dist.device), torch.full_like(cluster, n))
    pick = torch.full((m,), n, device=dist.device, dtype=torch.long).scatter_reduce(0, cluster, ids, "amin")
    cfeat = level.cfeat[pick].contiguous()
    offset = _offsets(coords[:, 0], level.offset.shape[0])
    return coords.int().contiguous(), new_stride, cfeat, offset, order, seg_start, m


class GridDownsample(nn.Module):
    """swin3d_layers.py:246-272 (`knn_down=False`): SparseTensorLayerNorm -> SparseTensorLinear (no bias) ->
    MinkowskiMaxPooling(kernel_size = stride): the maximum over the voxels of each cell; coordinates and signal carrier
    as GridKNNDownsample."""

    class _Norm(nn.Module):
        def __init__(self, dim):
            super().__init__()
            self.norm = LayerNorm(dim)

    class _Linear(nn.Module):
        def __init__(self, cin, cout):
            super().__init__()
            self.linear = Linear(cin, cout, bias=False)

    def __init__(self, in_channels, out_channels, kernel_size=2, stride=2):
        super().__init__()
        assert kernel_size == stride, "Swin3DUNet builds its downsample with kernel_size = stride"
        self.kernel_size, self.stride, self.in_channels, self.out_channels = kernel_size, stride, in_channels, out_channels
        self.norm = self._Norm(in_channels)
        self.linear = self._Linear(in_channels, out_channels)

    def forward(self, level):
        coords, new_stride, cfeat, offset, order, seg_start, m = _pool_cells(level, self.stride)
        y = self.linear.linear(self.norm.norm(level.feat))
        feat = A.segment_max(y, order, seg_start, m) if self.training else ops.pool_max(y, order, seg_start, m)
        return _Level(coords, new_stride, feat, cfeat, offset)


class GridKNNDownsample(nn.Module):
    """swin3d_layers.py:274-318 with GridCoordsDown (:180-231)."""

    def __init__(self, in_channels, out_channels, kernel_size=2, stride=2):
        super().__init__()
        self.stride, self.in_channels, self.out_channels, self.k = stride, in_channels, out_channels, 16
        self.norm = LayerNorm(in_channels)
        self.linear = Linear(in_channels, out_channels, bias=False)

    def forward(self, level):
        coords, new_stride, cfeat, offset, _, _, m = _pool_cells(level, self.stride)
        # LayerNorm and Linear act row by row, so they run once per source voxel instead of once per gathered copy
        y = self.linear(self.norm(level.feat))
        idx, _ = pointops.knn_query(self.k, level.xyz, level.offset, cfeat[:, 1:4].contiguous(), offset,
                                     cell=float(level.stride))
        idx = torch.where(idx < 0, idx[:, :1], idx).long()
        if self.training:
            # a source voxel is a neighbour of SEVERAL coarse voxels: its gradient is a sum over them, which the
            # segment-max backward of the pooling path (every row in exactly one segment) does not form - gather and
            # max through torch's tape here
            feat = y[idx].max(dim=1).values
        else:
            starts = torch.arange(0, (m + 1) * self.k, self.k, device=idx.device, dtype=torch.int32)
            feat = ops.pool_max(y, idx.reshape(-1).contiguous(), starts, m)
        return _Level(coords, new_stride, feat, cfeat, offset)


class BasicLayer(WindowStage):
    """swin3d_layers.py:634-876: the window-attention blocks (WindowStage) and, when given, the downsample after them."""

    def __init__(self, dim, depth, num_heads, window_size, quant_size, out_channels=None, mlp_ratio=4.0, qkv_bias=True,
                 qk_scale=None, drop_path=0.0, norm_layer=LayerNorm, downsample=None, down_stride=2, cRSE="XYZ_RGB",
                 fp16_mode=0):
        super().__init__(dim, depth, num_heads, window_size, quant_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                         qk_scale=qk_scale, drop_path=drop_path, norm_layer=norm_layer, cRSE=cRSE, fp16_mode=fp16_mode)
        self.downsample = None
        if downsample is not None:
            self.downsample = downsample(dim, out_channels if out_channels is not None else dim * 2,
                                         kernel_size=down_stride, stride=down_stride)

    def forward(self, level):
        level.feat = super().forward(level.feat, level.coords, level.stride, level.local_xyz().contiguous(),
                                     level.signals())
        return level, (self.downsample(level) if self.downsample is not None else level)


class Upsample(nn.Module):
    """swin3d_layers.py:320-381."""

    def __init__(self, in_channels, out_channels, num_heads, window_size, quant_size, attn=True, up_k=3,
                 cRSE="XYZ_RGB", fp16_mode=0):
        super().__init__()
        self.in_channels, self.out_channels, self.up_k = in_channels, out_channels, up_k
        self.linear1 = nn.Sequential(LayerNorm(out_channels), Linear(out_channels, out_channels))
        self.linear2 = nn.Sequential(LayerNorm(in_channels), Linear(in_channels, out_channels))
        self.attn = attn and window_size > 0
        if self.attn:
            self.block = BasicLayer(dim=out_channels, depth=1, num_heads=num_heads, window_size=window_size,
                                    quant_size=quant_size, drop_path=0.1, downsample=None, out_channels=None,
                                    cRSE=cRSE, fp16_mode=fp16_mode)

    def forward(self, deep, shallow):
        # pointops works in fp32 (its C API, libs/pointops: float only); under compute_dtype = bf16 the blend is fp32 too
        carried = pointops.interpolation(deep.xyz, shallow.xyz, self.linear2(deep.feat).float().contiguous(), deep.offset,
                                         shallow.offset, k=self.up_k, cell=float(deep.stride))
        shallow.feat = self.linear1(shallow.feat) + carried.to(shallow.feat.dtype)
        if self.attn:
            shallow, _ = self.block(shallow)
        return shallow


@MODELS.register_module("Swin3D-v1m1")
class Swin3DUNet(nn.Module):
    def __init__(self, in_channels, num_classes, base_grid_size, depths, channels, num_heads, window_sizes, quant_size,
                 drop_path_rate=0.2, up_k=3, num_layers=5, stem_transformer=True, down_stride=2, upsample="linear",
                 knn_down=True, cRSE="XYZ_RGB", fp16_mode=0):
        super().__init__()
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        downsample = GridKNNDownsample if knn_down else GridDownsample
        self.cRSE = cRSE
        if stem_transformer:
            self.stem_layer = _ConvBNRelu(in_channels, channels[0], kernel_size=3, stride=1)
            self.layer_start = 0
        else:
            # :69-85: a residual stem, its own downsample, and no attention stage at the finest level
            self.stem_layer = nn.Sequential(_ConvBNRelu(in_channels, channels[0], kernel_size=3, stride=1),
                                            MinkResBlock(in_channels=channels[0], out_channels=channels[0]))
            self.downsample = downsample(channels[0], channels[1], kernel_size=down_stride, stride=down_stride)
            self.layer_start = 1
        self.layers = nn.ModuleList([
            BasicLayer(dim=channels[i], depth=depths[i], num_heads=num_heads[i], window_size=window_sizes[i],
                       quant_size=quant_size, drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])],
                       downsample=downsample if i < num_layers - 1 else None,
                       down_stride=down_stride if i == 0 else 2,
                       out_channels=channels[i + 1] if i < num_layers - 1 else None, cRSE=cRSE, fp16_mode=fp16_mode)
            for i in range(self.layer_start, num_layers)])
        self.upsamples = nn.ModuleList([
            Upsample(channels[i], channels[i - 1], num_heads[i - 1], window_sizes[i - 1], quant_size,
                     attn="attn" in upsample, up_k=up_k, cRSE=cRSE, fp16_mode=fp16_mode)
            for i in range(num_layers - 1, 0, -1)])
        self.classifier = nn.Sequential(Linear(channels[0], channels[0]), BatchNorm1d(channels[0]),
                                        nn.ReLU(inplace=True), Linear(channels[0], num_classes))
        self.num_classes, self.base_grid_size = num_classes, base_grid_size
        self.init_weights()

    def init_weights(self):
        def _init(m):
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.LayerNorm, nn.BatchNorm1d)):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        self.apply(_init)

    def voxelize(self, data_dict):
        """:149-194: one voxel per distinct (batch, grid_coord), its features the plain mean of its points' rows
        [batch, coord / base_grid_size, coord_feat / 1.001, feat].  -> level 0 and the voxel id of every point."""
        grid_coord, feat, coord_feat = data_dict["grid_coord"], data_dict["feat"], data_dict["coord_feat"]
        coord, offset = data_dict["coord"], data_dict["offset"]
        n = coord.shape[0]
        counts = torch.diff(offset.long(), prepend=offset.new_zeros(1).long())
        batch = torch.repeat_interleave(torch.arange(offset.shape[0], device=coord.device), counts, output_size=n)
        rows = torch.cat([batch.unsqueeze(1).float(), _true_div(coord.float(), self.base_grid_size),
                          _true_div(coord_feat.float(), 1.001), feat.float()], dim=1)
        coords = torch.cat([batch.unsqueeze(1), grid_coord.long()], dim=1)
        order, cluster, seg_start, m = _segments(_key(coords, 1))
        mean = _segment_mean(rows, order, seg_start, m)
        vox = coords[order[seg_start[:-1].long()]].int().contiguous()
        ncf = coord_feat.shape[1] + 4
        level = _Level(vox, 1, mean[:, ncf:].contiguous(), mean[:, :ncf].contiguous(),
                       _offsets(vox[:, 0], offset.shape[0]))
        return level, cluster

    #: eval only.  None / torch.float32: fp32 features (default).  torch.bfloat16: features, Linear / conv operands and
    #: q, k, v in bf16 with fp32 accumulation, LayerNorm statistics, softmax, cRSE tables, signals, kNN and interpolation
    #: in fp32 - what the reference's S3DIS configs run under `enable_amp = True` (fp16 autocast there; bf16 here, as
    #: for PTv3).  Logits come back in fp32.
    compute_dtype = None

    def forward(self, data_dict):
        level, point2voxel = self.voxelize(data_dict)
        if not self.training and self.compute_dtype not in (None, torch.float32):
            level.feat = level.feat.to(self.compute_dtype)
        level = self.stem_layer(level)
        skips = []
        if self.layer_start > 0:          # :213-216
            skips.append(level)
            level = self.downsample(level)
        for layer in self.layers:
            kept, level = layer(level)
            skips.append(kept)
        level = skips.pop()
        for up in self.upsamples:
            level = up(level, skips.pop())
        x = self.classifier[1](self.classifier[0](level.feat), act=ops.ACT_RELU)   # BN: batch statistics in training
        return self.classifier[3](x).float()[point2voxel]
