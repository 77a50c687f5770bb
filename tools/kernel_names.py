"""One readable label per kernel for the profile summaries (rocprofv3 prints some names demangled, some not).
Labels match `roofline.kernels` of bench.py where a kernel is bracketed there."""
import re


def label(name):
    n = name
    m = re.search(r"gemm_big_kernelI(DF16b|f)Li(\d)ELi(\d)ELi(\d+)ELb([01])", n)
    if m:
        return "gemm_big_kernel<%s,%sch> %s" % ("bf16" if m.group(1) == "DF16b" else "f32", m.group(4),
                                                 "gather (sparse conv)" if m.group(5) == "1" else "dense")
    m = re.search(r"conv_tile_kernelI(DF16b|f)Li(\d+)E", n)
    if m:
        return "conv_tile_kernel<%s,%sch> gather (sparse conv)" % ("bf16" if m.group(1) == "DF16b" else "f32", m.group(2))
    m = re.search(r"gemm_kernelI(DF16b|f)Li(\d+)ELb([01])(?:ELi(\d))?", n)
    if m:
        pd = ",pd%s" % m.group(4) if m.group(4) and m.group(4) != "1" else ""
        return "gemm_kernel<%s,%dch%s> %s" % ("bf16" if m.group(1) == "DF16b" else "f32", 16 * int(m.group(2)), pd,
                                              "gather (sparse conv)" if m.group(3) == "1" else "dense")
    if "gemm_kernel<bool _Accum" in n:   # rocprofv3's demangler on gemm_kernel<__bf16, 2, ...>: the 32-channel tile
        return "gemm_kernel<bf16,32ch>"
    if "gemm_tn_group_kernel" in n:
        return "gemm_tn_group_kernel"
    m = re.search(r"(rows_linear|block_head_wide|block_tail_wide|block_head_coop|block_tail_coop|block_head|block_tail|mlp2|layernorm|splitk_reduce|pool_feat|"
                  r"gemm_tn|attn_bwd_dq|attn_bwd_dkv)_kernel", n)
    if m:
        return m.group(1) + "_kernel"
    m = re.search(r"window_attn_full_kernel<[^>]*?(\d), (\d)>", n)
    if m:
        return "window_attn_full_kernel<rpe%s,qt%s>" % (m.group(1), m.group(2))
    for k in ("window_attn_full_kernel", "window_attn_kernel", "ht_neighbors_kernel", "ht_insert_kernel",
              "radix_scatter_kernel", "radix_hist_kernel", "knn_query_kernel", "swin_attn_bwd_kernel", "swin_attn_mfma_kernel", "swin_attn_kernel",
              "conv_tile_kernel"):
        if k in n:
            return k
    return n.split("(")[0].replace("void ", "").replace("ptv3::", "")[:48]


def is_gather(lbl):
    """kernels whose dominant reads are 64-128 byte row gathers (FETCH_SIZE x2 correction of wide streaming reads
    does not apply to them)"""
    return "gather" in lbl or lbl.startswith(("window_attn", "ht_neighbors"))
