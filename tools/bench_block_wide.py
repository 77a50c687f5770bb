"""The weight-streaming block halves (csrc/block_wide.hip) against the launches they replace, on the level shapes of
BASELINE configs[2] (120k-point LiDAR scan): correctness against torch fp32 first, then time per call.
usage: python tools/bench_block_wide.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointcept-keypointdetection_amd"))
import torch
from ptv3_hip import ops
dev = torch.device("cuda:0")
F = torch.nn.functional


def timeit(f, it=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(it): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / it * 1e6


def make(c, m, seed=0):
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    p = dict(x=rnd(m, c), shortcut=rnd(m, c), attn=rnd(m, c))
    for k in ("g0", "b0", "g1", "b1", "g2", "b2", "bproj", "bias2"):
        p[k] = rnd(c)
    p.update(wqkv=rnd(3 * c, c) / c ** 0.5, bqkv=rnd(3 * c), wproj=rnd(c, c) / c ** 0.5, w1=rnd(4 * c, c) / c ** 0.5,
             bias1=rnd(4 * c), w2=rnd(c, 4 * c) / (4 * c) ** 0.5)
    return p


def reference(p, c):
    f1 = F.layer_norm(p["x"], (c,), p["g0"], p["b0"], 1e-5) + p["shortcut"]
    qkv = F.linear(F.layer_norm(f1, (c,), p["g1"], p["b1"], 1e-5), p["wqkv"], p["bqkv"])
    f2 = F.linear(p["attn"], p["wproj"], p["bproj"]) + f1
    out = f2 + F.linear(F.gelu(F.linear(F.layer_norm(f2, (c,), p["g2"], p["b2"], 1e-5), p["w1"], p["bias1"])), p["w2"], p["bias2"])
    return f1, qkv, out


def unfused_head(q):
    f1, t3 = ops.layernorm(q["x"], q["g0"], q["b0"], 1e-5, res=q["shortcut"], gamma2=q["g1"], beta2=q["b1"])
    return f1, ops.gemm(t3, q["wqkv"], bias=q["bqkv"])


def unfused_tail(q, f1):
    f2 = ops.gemm(q["attn"], q["wproj"], bias=q["bproj"], res=f1)
    t5 = ops.layernorm(f2, q["g2"], q["b2"], 1e-5)
    t6 = ops.gemm(t5, q["w1"], bias=q["bias1"], act=ops.ACT_GELU)
    return ops.gemm(t6, q["w2"], bias=q["bias2"], res=f2)


def main():
    check_m = int(os.environ.get("WIDE_CHECK_M", "24653"))
    for c in (() if os.environ.get("WIDE_NO_CHECK") else tuple(int(v) for v in os.environ.get("WIDE_CHECK_CS", "128,256").split(","))):
        p = make(c, check_m, c)
        f1_ref, qkv_ref, out_ref = reference(p, c)
        for dtype, tol in ((torch.float32, 1e-4), (torch.bfloat16, None)):
            q = {k: (v.to(dev).to(dtype) if v.dim() == 2 else v.to(dev)).contiguous() for k, v in p.items()}
            assert ops.block_fusable(c, 4 * c, dtype, check_m) == 3, "wide variant not selected"
            f1, qkv = ops.block_head(q["x"], None, 0, None, q["shortcut"], q["g0"], q["b0"], q["g1"], q["b1"], q["wqkv"], q["bqkv"], 1e-5)
            out = ops.block_tail(q["attn"], f1_ref.to(dev).to(dtype), q["wproj"], q["bproj"], q["g2"], q["b2"], q["w1"], q["bias1"],
                                 q["w2"], q["bias2"], 1e-5)
            torch.cuda.synchronize()
            errs = [(a.float().cpu() - b).abs().max().item() for a, b in ((f1, f1_ref), (qkv, qkv_ref), (out, out_ref))]
            line = f"check c={c} m={check_m} {str(dtype):15s} max|err| f1 {errs[0]:.3e} qkv {errs[1]:.3e} out {errs[2]:.3e}"
            if dtype == torch.bfloat16:   # against the launches it replaces, same dtype
                f1u, qkvu = unfused_head(q)
                outu = unfused_tail(q, f1_ref.to(dev).to(dtype))
                d = [(a.float() - b.float()).abs().max().item() for a, b in ((f1, f1u), (qkv, qkvu), (out, outu))]
                line += f" | vs unfused bf16: f1 {d[0]:.3e} qkv {d[1]:.3e} out {d[2]:.3e}"
            print(line, flush=True)
            if tol:
                assert max(errs) < tol, errs
    print(f"{'shape':24s}{'head wide':>10s}{'LN+qkv':>10s}{'tail wide':>10s}{'proj+LN+fc1+fc2':>17s}   (us, bf16)")
    shapes = ((256, 56504), (128, 80168), (256, 27743), (128, 120000))
    if os.environ.get("WIDE_SHAPES"):
        shapes = tuple(tuple(int(v) for v in s.split("x")) for s in os.environ["WIDE_SHAPES"].split(","))
    for c, m in shapes:
        p = make(c, m, 1)
        q = {k: (v.to(dev).bfloat16() if v.dim() == 2 else v.to(dev)).contiguous() for k, v in p.items()}
        f1 = q["shortcut"]
        th = timeit(lambda: ops.block_head(q["x"], None, 0, None, q["shortcut"], q["g0"], q["b0"], q["g1"], q["b1"], q["wqkv"], q["bqkv"], 1e-5))
        tu = timeit(lambda: unfused_head(q))
        tt = timeit(lambda: ops.block_tail(q["attn"], f1, q["wproj"], q["bproj"], q["g2"], q["b2"], q["w1"], q["bias1"], q["w2"], q["bias2"], 1e-5))
        tv = timeit(lambda: unfused_tail(q, f1))
        fl_t = 2.0 * m * c * 9 * c
        print(f"C={c:<4d}M={m:<8d}      {th:10.1f}{tu:10.1f}{tt:10.1f}{tv:17.1f}   tail {fl_t / tt / 1e6:6.0f} TFLOP/s", flush=True)


def rows_bench():
    """ptv3_rows_linear on the C = 512 level of the LiDAR scan (27 743 rows) against LayerNorm + tiled GEMM launches"""
    print(f"{'rows_linear':24s}{'LN0+LN1+qkv':>12s}{'unfused':>9s}{'proj+res':>10s}{'unfused':>9s}{'LN+fc1+GELU':>13s}{'unfused':>9s}   (us, bf16)")
    for c, m in ((512, 27743), (256, 56504), (128, 80168), (256, 1388), (512, 245)):
        p = make(c, m, 2)
        q = {k: (v.to(dev).bfloat16() if v.dim() == 2 else v.to(dev)).contiguous() for k, v in p.items()}
        t = [timeit(lambda: ops.rows_linear(q["x"], q["wqkv"], q["bqkv"], ln=(q["g1"], q["b1"]), ln0=(q["g0"], q["b0"]), shortcut=q["shortcut"])),
             timeit(lambda: unfused_head(q)),
             timeit(lambda: ops.rows_linear(q["attn"], q["wproj"], q["bproj"], res=q["shortcut"])),
             timeit(lambda: ops.gemm(q["attn"], q["wproj"], bias=q["bproj"], res=q["shortcut"])),
             timeit(lambda: ops.rows_linear(q["x"], q["w1"], q["bias1"], act=ops.ACT_GELU, ln=(q["g2"], q["b2"]))),
             timeit(lambda: ops.gemm(ops.layernorm(q["x"], q["g2"], q["b2"], 1e-5), q["w1"], bias=q["bias1"], act=ops.ACT_GELU))]
        print(f"C={c:<4d}M={m:<8d}      " + "".join(f"{v:11.1f}" for v in t), flush=True)


if __name__ == "__main__":
    if os.environ.get("WIDE_ROWS_ONLY"):
        rows_bench()
    else:
        main()
        rows_bench()
