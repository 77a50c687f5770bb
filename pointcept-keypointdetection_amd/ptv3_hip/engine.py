"""Python side of ptv3_forward (include/ptv3_hip.h): packs a PT-v3m1 module tree into the flat
parameter table + model description the native executor walks, and runs one forward with it."""
import ctypes
from ctypes import c_float, c_int32, c_int64, c_void_p, POINTER

import torch

from .lib import lib, PTV3_F32, PTV3_BF16, ORDER_IDS
from . import ops


class ModelDesc(ctypes.Structure):
    _fields_ = [("dtype", c_int32), ("in_channels", c_int32), ("num_stages", c_int32), ("num_orders", c_int32),
                ("enc_depths", c_int32 * 8), ("enc_channels", c_int32 * 8), ("enc_heads", c_int32 * 8),
                ("enc_patch", c_int32 * 8), ("dec_depths", c_int32 * 8), ("dec_channels", c_int32 * 8),
                ("dec_heads", c_int32 * 8), ("dec_patch", c_int32 * 8), ("stride", c_int32 * 8),
                ("enc_mode", c_int32), ("enable_flash", c_int32), ("head_hidden", c_int32), ("head_out", c_int32),
                ("mlp_ratio", c_float), ("ln_eps", c_float), ("qk_scale", c_float)]


class ForwardIO(ctypes.Structure):
    _fields_ = [("grid_coord", c_void_p), ("coord_is_i64", c_int32), ("feat", c_void_p), ("batch", c_void_p),
                ("offset", c_void_p), ("offset_host", POINTER(c_int64)), ("b", c_int32), ("n", c_int64),
                ("depth", c_int32), ("order_ids_host", POINTER(c_int32)), ("pool_perm_host", POINTER(c_int32)),
                ("code", c_void_p), ("order", c_void_p), ("inverse", c_void_p), ("out_feat", c_void_p),
                ("out_head", c_void_p), ("stage_points_host", POINTER(c_int64)), ("depth_out", POINTER(c_int32)),
                ("batch_out", c_void_p), ("inputs_resident", c_int32), ("overlap_calls", c_int32),
                ("raw_feat", c_void_p), ("raw_feat_channels", c_int32), ("raw_feat_dtype", c_int32),
                ("arena_n", c_int64), ("arena_b", c_int32), ("executor", c_void_p)]


def declare(dll):
    dll.ptv3_forward_workspace_bytes.restype = ctypes.c_size_t
    dll.ptv3_forward_workspace_bytes.argtypes = [POINTER(ModelDesc), c_int64, ctypes.c_int]
    dll.ptv3_forward.restype = ctypes.c_int
    dll.ptv3_forward.argtypes = [POINTER(ModelDesc), POINTER(c_void_p), ctypes.c_int, POINTER(ForwardIO), c_void_p,
                                 ctypes.c_size_t, c_void_p]


class Executor:
    """Handle of one native executor (ptv3_executor_create): the internal streams, events and call parity of
    ptv3_forward.  One per (backbone, device), so two models - or one model replicated on two devices - never share
    streams or the alternating arena halves."""

    def __init__(self):
        self._dll = lib.load()
        self.handle = self._dll.ptv3_executor_create()
        if not self.handle:
            raise RuntimeError("ptv3_executor_create failed")

    def __deepcopy__(self, memo):   # copy.deepcopy(model): the copy gets an executor of its own
        return Executor()

    def __reduce__(self):
        return (Executor, ())

    def __del__(self):
        try:
            if self.handle:
                self._dll.ptv3_executor_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def _bn(bn):
    scale = bn.weight.detach().float() * torch.rsqrt(bn.running_var.float() + bn.eps)
    shift = bn.bias.detach().float() - bn.running_mean.float() * scale
    return [scale.contiguous(), shift.contiguous()]


def _f32(p):
    return p.detach().float().contiguous()


class Packed:
    """Flat parameter table of one (backbone, head, dtype); rebuilt when any parameter version changes."""

    def __init__(self, backbone, head, dtype):
        self.backbone, self.head, self.dtype = backbone, head, dtype
        self.sources = [t for t in list(backbone.parameters()) + list(backbone.buffers())]
        if head is not None:
            self.sources += list(head.parameters()) + list(head.buffers())
        self.versions = None
        self.tensors = None
        self.repacked = False

    def fresh(self):
        v = [t._version for t in self.sources]
        if v != self.versions:
            if self.tensors is not None:
                # earlier forwards may still be reading the old packed copies (inputs_resident / overlap_calls run
                # ahead of the caller's stream): drain the device before they are released and their memory reused
                torch.cuda.synchronize()
            with torch.no_grad():
                self.tensors = self._pack()
            self.table = (c_void_p * len(self.tensors))(*[t.data_ptr() for t in self.tensors])
            self.versions = v
            self.repacked = True
        return self

    def _mat(self, w, cin_pad=None):
        w = w.detach()
        if cin_pad is not None and w.shape[-1] != cin_pad:
            w = torch.nn.functional.pad(w, (0, cin_pad - w.shape[-1]))
        return w.reshape(w.shape[0], -1).to(self.dtype).contiguous()

    def _block(self, blk):
        cpe = blk.cpe
        wf, bf = blk.folded_cpe(self.dtype)   # conv with the cpe Linear folded in
        mlp = blk.mlp[0]
        if ops.block_fusable(blk.channels, mlp.fc1.out_features, self.dtype):
            wqkv, wproj, w1, w2 = blk.chain_weights(self.dtype)  # register-chained GEMMs: permuted inputs
        else:
            wqkv, wproj, w1, w2 = (self._mat(blk.attn.qkv.weight), self._mat(blk.attn.proj.weight),
                                   self._mat(mlp.fc1.weight), self._mat(mlp.fc2.weight))
        return [wf, bf, _f32(cpe[2].weight), _f32(cpe[2].bias), _f32(blk.norm1[0].weight), _f32(blk.norm1[0].bias),
                wqkv, _f32(blk.attn.qkv.bias), wproj, _f32(blk.attn.proj.bias), _f32(blk.norm2[0].weight),
                _f32(blk.norm2[0].bias), w1, _f32(mlp.fc1.bias), w2, _f32(mlp.fc2.bias)]

    def _pack(self):
        bb = self.backbone
        gran = ops.k_granule(self.dtype)
        stem = bb.embedding.stem
        self.cin_pad = (bb.embedding.in_channels + gran - 1) // gran * gran
        t = [self._mat(stem.conv.weight, self.cin_pad)] + _bn(stem.norm)
        for s in range(bb.num_stages):
            enc = getattr(bb.enc, f"enc{s}")
            if s > 0:
                t += [self._mat(enc.down.proj.weight), _f32(enc.down.proj.bias)] + _bn(enc.down.norm[0])
            i = 0
            while hasattr(enc, f"block{i}"):
                t += self._block(getattr(enc, f"block{i}"))
                i += 1
        if not bb.enc_mode:
            for s in reversed(range(bb.num_stages - 1)):
                dec = getattr(bb.dec, f"dec{s}")
                up = dec.up
                t += [self._mat(up.proj[0].weight), _f32(up.proj[0].bias)] + _bn(up.proj[1])
                t += [self._mat(up.proj_skip[0].weight), _f32(up.proj_skip[0].bias)] + _bn(up.proj_skip[1])
                i = 0
                while hasattr(dec, f"block{i}"):
                    t += self._block(getattr(dec, f"block{i}"))
                    i += 1
        if self.head is not None:
            h = self.head
            if ops.mlp2_fusable(h[0].in_features, h[0].out_features, h[3].out_features, self.dtype):
                w2 = ops.mlp2_weight2(h[3].weight, self.dtype)   # hidden layer stays in registers: ptv3_mlp2
            else:
                w2 = self._mat(h[3].weight)
            t += [self._mat(h[0].weight), _f32(h[0].bias)] + _bn(h[1]) + [w2, _f32(h[3].bias)]
        return t


def eligible(backbone, head=None):
    """True when the module tree is the standard eval-mode PT-v3m1 the native executor implements."""
    from pointcept.models.utils.hip_layers import Linear, LayerNorm, BatchNorm1d
    import torch.nn as nn
    if backbone.training or backbone.enc_mode:
        return False
    cached = backbone.__dict__.get("_engine_static_ok")
    if cached is None:
        ok = True
        for m in backbone.modules():
            n = type(m).__name__
            if n == "Block":
                ok &= m.pre_norm and isinstance(m.cpe[2], LayerNorm) and isinstance(m.norm1[0], LayerNorm) \
                    and isinstance(m.norm2[0], LayerNorm) and isinstance(m.mlp[0].act, nn.GELU) \
                    and not m.attn.enable_rpe
            elif n == "SerializedPooling":
                ok &= m.norm is not None and isinstance(m.norm[0], BatchNorm1d) and m.act is not None \
                    and isinstance(m.act[0], nn.GELU) and m.shuffle_orders
            elif n == "SerializedUnpooling":
                ok &= len(m.proj) == 3 and len(m.proj_skip) == 3 and isinstance(m.proj[1], BatchNorm1d) \
                    and isinstance(m.proj[2], nn.GELU)
            elif n == "Embedding":
                ok &= isinstance(m.stem._modules.get("norm"), BatchNorm1d) and isinstance(m.stem._modules.get("act"), nn.GELU)
        if head is not None:
            ok &= (len(head) == 4 and isinstance(head[0], Linear) and isinstance(head[1], BatchNorm1d)
                   and isinstance(head[2], nn.ReLU) and isinstance(head[3], Linear))
        backbone.__dict__["_engine_static_ok"] = cached = bool(ok)
        backbone.__dict__["_engine_modules"] = list(backbone.modules()) + (list(head.modules()) if head is not None else [])
    if not cached:
        return False
    # forward hooks (debug taps, profilers) need the module-by-module path
    for m in backbone.__dict__["_engine_modules"]:
        if m._forward_hooks or m._forward_pre_hooks:
            return False
    return True


def _desc(bb, head, dtype, cin_pad):
    d = ModelDesc()
    d.dtype = PTV3_F32 if dtype == torch.float32 else PTV3_BF16
    d.in_channels = cin_pad
    d.num_stages = bb.num_stages
    d.num_orders = len(bb.order)
    blk0 = bb.enc.enc0.block0
    for s in range(bb.num_stages):
        enc = getattr(bb.enc, f"enc{s}")
        depth = sum(1 for k in enc._modules if k.startswith("block"))
        b = enc.block0
        d.enc_depths[s], d.enc_channels[s], d.enc_heads[s] = depth, b.channels, b.attn.num_heads
        d.enc_patch[s] = b.attn.patch_size if b.attn.enable_flash else b.attn.patch_size_max
        if s > 0:
            d.stride[s - 1] = enc.down.stride
    for s in range(bb.num_stages - 1):
        dec = getattr(bb.dec, f"dec{s}")
        depth = sum(1 for k in dec._modules if k.startswith("block"))
        b = dec.block0
        d.dec_depths[s], d.dec_channels[s], d.dec_heads[s] = depth, b.channels, b.attn.num_heads
        d.dec_patch[s] = b.attn.patch_size if b.attn.enable_flash else b.attn.patch_size_max
    d.enc_mode = 0
    d.enable_flash = int(blk0.attn.enable_flash)
    d.mlp_ratio = blk0.mlp[0].fc1.out_features / blk0.channels
    d.ln_eps = blk0.norm1[0].eps
    default_scale = (blk0.channels // blk0.attn.num_heads) ** -0.5
    d.qk_scale = 0.0 if abs(blk0.attn.scale - default_scale) < 1e-12 else blk0.attn.scale
    if head is not None:
        d.head_hidden, d.head_out = head[0].out_features, head[3].out_features
    return d


def forward(backbone, point, dtype, head=None):
    """Runs the native executor.  `point` is a Point with feat (already `dtype`), grid_coord, batch, offset.
    Fills the Point like the module path does and returns (point, head_out or None)."""
    dll = lib.load()
    if not hasattr(dll, "_engine_declared"):
        declare(dll)
        dll._engine_declared = True
    cache = backbone.__dict__.setdefault("_engine_packed", {})
    key = (dtype, id(head))
    if key not in cache:
        cache[key] = Packed(backbone, head, dtype)
    pk = cache[key].fresh()
    if getattr(pk, "desc", None) is None:
        pk.desc = _desc(backbone, head, dtype, pk.cin_pad)
    desc = pk.desc

    k = len(backbone.order)
    S = backbone.num_stages
    # the reference's CPU-RNG draws, in its order: serialization (structure.py:101-105), then one per pooling
    orders = list(backbone.order)
    if backbone.shuffle_orders:
        orders = [orders[p] for p in torch.randperm(k).tolist()]
    perms = []
    for _ in range(S - 1):
        perms += torch.randperm(k).tolist()
    order_ids = (c_int32 * k)(*[ORDER_IDS[o] for o in orders])
    perm_arr = (c_int32 * max(1, len(perms)))(*perms)

    feat = point.feat
    n = feat.shape[0]
    # overlap mode (backbone.overlap_calls + inputs_resident): nothing may be produced on the caller's stream for
    # this call, so the pad / cast of the features moves into the executor and the outputs come from a ring
    overlap = bool(getattr(backbone, "overlap_calls", False)) and bool(getattr(backbone, "inputs_resident", False))
    raw = None
    if overlap:
        raw = feat
        if pk.repacked:                      # parameter tables were (re)built on this stream: settle them once
            torch.cuda.current_stream().synchronize()
    elif feat.shape[1] != pk.cin_pad:
        feat = torch.nn.functional.pad(feat, (0, pk.cin_pad - feat.shape[1])).contiguous()
    pk.repacked = False
    gc = point.grid_coord
    if gc.dtype not in (torch.int32, torch.int64):
        gc = gc.long()
    gc = gc.contiguous()
    offset = point.offset.long().contiguous()
    derive_batch = point.get("_batch_pending", False)
    batch = point.batch if derive_batch else point.batch.long().contiguous()
    nb = int(offset.shape[0])   # scene offsets and the coordinate maximum are read back by the executor on its
    dev = feat.device           # geometry stream: no synchronisation of the caller's stream here
    resident = bool(getattr(backbone, "inputs_resident", False))
    if resident:
        # Everything the executor's own streams WRITE must come from memory torch's caching allocator will not hand
        # out again behind their back: the allocator only orders re-use within the caller's stream, while the geometry
        # (and, with overlap_calls, the feature) pipeline runs ahead of that stream.  A fresh torch.empty() here could
        # be a block that still has caller-stream work pending on it (e.g. the sigmoid temporary of the previous
        # call's head, queued behind that call's completion wait) - which would then scribble over this call's
        # codes / orders while the geometry kernels are reading them.  Hence a persistent ring of three generations
        # per shape: call i's tensors stay valid until call i+3 is issued.
        # The ring is sized by CAPACITY (grow-only, x1.25 like the arena), not by this call's n: real datasets have a
        # different point count in almost every batch, and a ring per distinct n would grow without bound and drain the
        # device on nearly every call.  A call receives views [:n] of its generation's flat buffers.  The returned
        # point.feat / serialized_* tensors are such views: they are overwritten three calls later.
        ring = backbone.__dict__.get("_engine_out_ring")
        c_out, c_head = desc.dec_channels[0], (desc.head_out if head is not None else 0)
        rkey = (k, c_out, dtype, c_head, dev)
        if ring is None or ring["key"] != rkey or ring["cap"] < n:
            cap_n = max(n, int((ring["cap"] if ring is not None and ring["key"] == rkey else 0) * 1.25))

            def gen():
                return dict(code=torch.empty(k * cap_n, dtype=torch.int64, device=dev),
                            order=torch.empty(k * cap_n, dtype=torch.int64, device=dev),
                            inverse=torch.empty(k * cap_n, dtype=torch.int64, device=dev),
                            batch=torch.empty(cap_n, dtype=torch.int64, device=dev),
                            out_feat=torch.empty(cap_n * c_out, dtype=dtype, device=dev),
                            out_head=(torch.empty(cap_n * c_head, dtype=torch.float32, device=dev)
                                      if head is not None else None))
            torch.cuda.synchronize(dev)        # nothing pending anywhere on the blocks these allocations receive
            backbone.__dict__["_engine_out_ring"] = None   # release the old generations before allocating
            ring = {"key": rkey, "cap": cap_n, "i": 0, "bufs": [gen() for _ in range(3)]}
            backbone.__dict__["_engine_out_ring"] = ring
            torch.cuda.synchronize(dev)
        cur = ring["bufs"][ring["i"] % 3]
        ring["i"] += 1
        code, order, inverse = (cur[q][:k * n].view(k, n) for q in ("code", "order", "inverse"))
        out_feat = cur["out_feat"][:n * c_out].view(n, c_out)
        out_head = cur["out_head"][:n * c_head].view(n, c_head) if head is not None else None
        if derive_batch:
            batch = cur["batch"][:n]
            point["batch"] = batch
        # inputs that had to be converted just now were produced on the caller's stream, which the executor's
        # streams do not wait for in these modes: settle them, and keep them alive for the ring generation
        converted = [t for t, src in ((gc, point.grid_coord), (offset, point.offset)) if t is not src]
        if not derive_batch and batch is not point.batch:
            converted.append(batch)
        if converted:
            torch.cuda.current_stream().synchronize()
        cur["keep"] = (gc, offset, batch, raw)
    else:
        code = torch.empty((k, n), dtype=torch.int64, device=dev)
        order = torch.empty_like(code)
        inverse = torch.empty_like(code)
        out_feat = torch.empty((n, desc.dec_channels[0]), dtype=dtype, device=dev)
        out_head = torch.empty((n, desc.head_out), dtype=torch.float32, device=dev) if head is not None else None
    # grow-only arena with a FIXED internal layout (arena_n / arena_b): earlier calls may still be executing out of
    # it (inputs_resident / overlap_calls), so its parts must not move with the scene size, and it is only replaced
    # after the device has drained
    arena = backbone.__dict__.get("_engine_arena")
    cap = backbone.__dict__.get("_engine_arena_cap", (0, 0, None, None))
    if arena is None or cap[0] < n or cap[1] < nb or cap[2] != dtype or cap[3] != dev:
        cap = (max(n, int(cap[0] * 1.25)), max(nb, cap[1]), dtype, dev)
        ws_bytes = dll.ptv3_forward_workspace_bytes(ctypes.byref(desc), cap[0], cap[1])
        torch.cuda.synchronize(dev)
        backbone.__dict__["_engine_arena"] = None
        arena = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        backbone.__dict__["_engine_arena"], backbone.__dict__["_engine_arena_cap"] = arena, cap
    execs = backbone.__dict__.setdefault("_engine_exec", {})
    if dev not in execs:
        with torch.cuda.device(dev):   # an executor binds to the device that is current when it is created
            execs[dev] = Executor()
    stage_pts = (c_int64 * 8)()
    io = ForwardIO()
    io.executor = execs[dev].handle
    io.grid_coord, io.coord_is_i64 = gc.data_ptr(), int(gc.dtype == torch.int64)
    io.feat, io.offset = feat.data_ptr(), offset.data_ptr()
    io.overlap_calls = int(overlap)
    io.arena_n, io.arena_b = cap[0], cap[1]
    if raw is not None:
        io.raw_feat, io.raw_feat_channels = raw.data_ptr(), raw.shape[1]
        io.raw_feat_dtype = PTV3_F32 if raw.dtype == torch.float32 else PTV3_BF16
    else:
        io.raw_feat = None
    if derive_batch:   # the executor fills point.batch itself (geometry stream)
        io.batch, io.batch_out = None, batch.data_ptr()
        del point["_batch_pending"]
    else:
        io.batch, io.batch_out = batch.data_ptr(), None
    depth_out = c_int32(0)
    io.offset_host, io.b, io.n, io.depth = None, nb, n, 0
    io.depth_out = ctypes.pointer(depth_out)
    io.inputs_resident = int(bool(getattr(backbone, "inputs_resident", False)))
    io.order_ids_host, io.pool_perm_host = order_ids, perm_arr
    io.code, io.order, io.inverse = code.data_ptr(), order.data_ptr(), inverse.data_ptr()
    io.out_feat = out_feat.data_ptr()
    io.out_head = out_head.data_ptr() if out_head is not None else None
    io.stage_points_host = stage_pts
    rc = dll.ptv3_forward(ctypes.byref(desc), pk.table, len(pk.tensors), ctypes.byref(io), arena.data_ptr(),
                          arena.numel(), ops._stream())
    lib.check(rc, "ptv3_forward")
    depth = int(depth_out.value)
    assert depth * 3 + nb.bit_length() <= 63
    point["order"] = list(backbone.order)
    point["serialized_depth"] = depth
    point["serialized_code"], point["serialized_order"], point["serialized_inverse"] = code, order, inverse
    point["feat"] = out_feat
    point["_stage_points"] = [int(stage_pts[i]) for i in range(S)]
    return point, out_head
