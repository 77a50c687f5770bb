"""Oracle (test infrastructure): space-filling-curve codes, argsort, pad plan.

Integer restatement (numpy int64) of
  * pointcept/models/utils/serialization/default.py:9-24   (encode, "-trans" swap, batch bits)
  * pointcept/models/utils/serialization/z_order.py:40-50,66-101  (Morton bit interleave)
  * pointcept/models/utils/serialization/hilbert.py:91-198  (Skilling transform + Gray->binary)
  * pointcept/models/utils/structure.py:52-109  (Point.serialization: depth, argsort, inverse)
  * pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:114-170 (pad plan)
Pinned by tests/golden/sfc_*.npz and padplan_*.npz (generated from the reference code).
"""
import numpy as np

ORDERS = ("z", "z-trans", "hilbert", "hilbert-trans")


def morton3(x, y, z, depth):
    """z_order.py:40-50: bit i of x -> 3i+2, y -> 3i+1, z -> 3i."""
    x = x.astype(np.int64)
    y = y.astype(np.int64)
    z = z.astype(np.int64)
    key = np.zeros_like(x)
    for i in range(depth):
        key |= ((x >> i) & 1) << (3 * i + 2)
        key |= ((y >> i) & 1) << (3 * i + 1)
        key |= ((z >> i) & 1) << (3 * i)
    return key


def hilbert3(x, y, z, depth):
    """hilbert.py:91-198 in integer form (Skilling AxesToTranspose).

    The reference walks MSB-first bit planes; for bit plane ``bit`` (Q = 1 << (depth-1-bit))
    and each dim: if the bit is set, invert the lower bits of dim 0; otherwise exchange
    the lower bits of dim 0 and dim ``dim`` where they differ (hilbert.py:156-175).
    Then interleave as (bit, dim) MSB-first (hilbert.py:178) and Gray->binary by a
    prefix xor over the 3*depth bits (hilbert.py:69-88,181).
    """
    X = [x.astype(np.int64).copy(), y.astype(np.int64).copy(), z.astype(np.int64).copy()]
    for bit in range(depth):
        Q = np.int64(1) << (depth - 1 - bit)
        P = Q - 1
        for i in range(3):
            on = (X[i] & Q) != 0
            # on: invert low bits of dim 0
            X0_inv = X[0] ^ P
            # off: exchange differing low bits of dim 0 and dim i
            t = (X[0] ^ X[i]) & P
            X0_new = np.where(on, X0_inv, X[0] ^ t)
            Xi_new = np.where(on, X[i], X[i] ^ t)
            if i == 0:
                # dim 0 against itself: t == 0, so "off" is a no-op; "on" inverts
                X[0] = np.where(on, X0_inv, X[0])
            else:
                X[0] = X0_new
                X[i] = Xi_new
    g = np.zeros_like(X[0])
    for b in range(depth):  # b = bit position from LSB
        for d in range(3):
            g |= ((X[d] >> b) & 1) << (3 * b + (2 - d))
    # Gray -> binary: prefix xor from the MSB
    key = g.copy()
    shift = 1
    while shift < 3 * depth:
        key ^= key >> shift
        shift <<= 1
    return key


def encode(grid_coord, batch, depth, order):
    """default.py:9-24."""
    assert order in ORDERS
    gc = np.asarray(grid_coord).astype(np.int64)
    x, y, z = gc[:, 0], gc[:, 1], gc[:, 2]
    if order.endswith("-trans"):
        x, y = y, x
    if order.startswith("z"):
        code = morton3(x, y, z, depth)
    else:
        code = hilbert3(x, y, z, depth)
    if batch is not None:
        code = (np.asarray(batch).astype(np.int64) << (depth * 3)) | code
    return code


def serialized_depth(grid_coord):
    """structure.py:73."""
    return int(int(np.asarray(grid_coord).max()) + 1).bit_length()


def serialization(grid_coord, batch, orders, depth=None):
    """structure.py:52-99 without the shuffle: returns code, order, inverse (k, N) int64.

    argsort ties cannot occur when (batch, grid_coord) rows are unique (GridSample output).
    """
    if depth is None:
        depth = serialized_depth(grid_coord)
    assert depth <= 16
    code = np.stack([encode(grid_coord, batch, depth, o) for o in orders])
    order = np.argsort(code, axis=1, kind="stable")
    inverse = np.zeros_like(order)
    k, n = code.shape
    for r in range(k):
        inverse[r, order[r]] = np.arange(n, dtype=np.int64)
    return code, order, inverse, depth


def pad_plan(offset, patch_size):
    """point_transformer_v3m1_base.py:114-170 (get_padding_and_inverse).

    offset: cumulative scene ends (B,). Returns pad (N',), unpad (N,), cu_seqlens (int32).
    """
    offset = np.asarray(offset).astype(np.int64)
    K = int(patch_size)
    bincount = np.diff(offset, prepend=0)
    bincount_pad = (bincount + K - 1) // K * K
    mask_pad = bincount > K
    bincount_pad = np.where(mask_pad, bincount_pad, bincount)
    _offset = np.concatenate([[0], offset])
    _offset_pad = np.concatenate([[0], np.cumsum(bincount_pad)])
    pad = np.arange(_offset_pad[-1], dtype=np.int64)
    unpad = np.arange(_offset[-1], dtype=np.int64)
    cu = []
    for i in range(len(offset)):
        unpad[_offset[i]:_offset[i + 1]] += _offset_pad[i] - _offset[i]
        if bincount[i] != bincount_pad[i]:
            r = bincount[i] % K
            pad[_offset_pad[i + 1] - K + r:_offset_pad[i + 1]] = \
                pad[_offset_pad[i + 1] - 2 * K + r:_offset_pad[i + 1] - K]
        pad[_offset_pad[i]:_offset_pad[i + 1]] -= _offset_pad[i] - _offset[i]
        cu.append(np.arange(_offset_pad[i], _offset_pad[i + 1], K, dtype=np.int32))
    cu_seqlens = np.concatenate(cu + [np.array([_offset_pad[-1]], dtype=np.int32)]).astype(np.int32)
    return pad, unpad, cu_seqlens


def patch_size_for(offset, patch_size_max):
    """point_transformer_v3m1_base.py:173-176 (enable_flash=False)."""
    bincount = np.diff(np.asarray(offset).astype(np.int64), prepend=0)
    return int(min(int(bincount.min()), int(patch_size_max)))
