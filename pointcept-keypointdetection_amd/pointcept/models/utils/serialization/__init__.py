from .default import encode, z_order_encode, hilbert_encode  # noqa: F401
