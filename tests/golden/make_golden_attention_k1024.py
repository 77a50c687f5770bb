"""tests/golden/attention_k1024.npz: the reference's SerializedAttention (enable_flash=False, the vanilla path of
point_transformer_v3m1_base.py:172-222) at the two (C, H, K) = (64, 4, 1024) and (512, 32, 1024) shapes SURVEY 8(c)
names - the head count and width of the fork's level-1 and level-4 blocks at the full 1024-point patch.

    python tests/golden/make_golden_attention_k1024.py          (build container only: imports /root/reference)

Data only: seeded inputs / weights and the reference module's output.  qkv is NOT stored (the test recomputes it from
`feat` and the stored qkv weights in fp32), which keeps the fixture under 10 MB.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pointcept-keypointdetection_amd"))

import ref_loader  # noqa: E402
import ptv3_scenes as S  # noqa: E402
from make_golden_cfg import ORDERS  # noqa: E402


def main():
    ns = ref_loader.load()
    out = {}
    cases = [(64, 4, 1024, [1300], 64), (512, 32, 1024, [1100], 64)]   # C, H, patch, scene sizes, extent
    for i, (C, H, pmax, sizes, extent) in enumerate(cases):
        torch.manual_seed(300 + i)
        m = ns.v3m1.SerializedAttention(C, H, pmax, order_index=i % 4, enable_rpe=False, enable_flash=False,
                                        upcast_attention=False, upcast_softmax=False).eval()
        data = S.make_batch(sizes, in_channels=C, extent=extent, seed=70 + i)
        P = ns.Point(data)
        P.serialization(order=ORDERS, shuffle_orders=False)
        feat_in = P.feat.clone()
        with torch.no_grad():
            o = m(P).feat
        assert m.patch_size == 1024
        t = f"a{i}_"
        out[t + "cfg"] = np.array([C, H, pmax, i % 4, 0, m.patch_size])
        out[t + "grid_coord"], out[t + "offset"] = data["grid_coord"].numpy(), data["offset"].numpy()
        out[t + "feat"], out[t + "out"] = feat_in.numpy(), o.numpy()
        for k, v in m.state_dict().items():
            out[t + "w_" + k] = v.numpy()
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "attention_k1024.npz"), **out)
    print("attention_k1024.npz", os.path.getsize(os.path.join(HERE, "attention_k1024.npz")))


if __name__ == "__main__":
    main()
