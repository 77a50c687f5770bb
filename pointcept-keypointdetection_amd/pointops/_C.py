"""The `pointops._C` entry points with the reference's pybind signatures
(libs/pointops/src/pointops_api.cpp:15-23): caller-allocated, contiguous tensors, current stream."""
import ctypes

import torch

from ptv3_hip.lib import lib


def _s():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t, dtype):
    if not (t.is_cuda and t.is_contiguous() and t.dtype == dtype):
        raise RuntimeError(f"pointops._C: expected a contiguous GPU tensor of {dtype}, got {t.dtype} on {t.device}")
    return ctypes.c_void_p(t.data_ptr())


def knn_query_cuda(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2):
    lib.check(lib.ptv3_knn_query(m, nsample, _p(xyz, torch.float32), _p(new_xyz, torch.float32),
                                 _p(offset, torch.int32), _p(new_offset, torch.int32), offset.shape[0],
                                 _p(idx, torch.int32), _p(dist2, torch.float32), _s()), "knn_query")


def grouping_forward_cuda(m, nsample, c, input, idx, output):
    lib.check(lib.ptv3_grouping_forward(m, nsample, c, _p(input, torch.float32), _p(idx, torch.int32),
                                        _p(output, torch.float32), _s()), "grouping_forward")


def grouping_backward_cuda(m, nsample, c, grad_output, idx, grad_input):
    lib.check(lib.ptv3_grouping_backward(m, nsample, c, _p(grad_output, torch.float32), _p(idx, torch.int32),
                                         _p(grad_input, torch.float32), _s()), "grouping_backward")


def interpolation_forward_cuda(n, c, k, input, idx, weight, output):
    lib.check(lib.ptv3_interpolation_forward(n, c, k, _p(input, torch.float32), _p(idx, torch.int32),
                                             _p(weight, torch.float32), _p(output, torch.float32), _s()),
              "interpolation_forward")


def interpolation_backward_cuda(n, c, k, grad_output, idx, weight, grad_input):
    lib.check(lib.ptv3_interpolation_backward(n, c, k, _p(grad_output, torch.float32), _p(idx, torch.int32),
                                              _p(weight, torch.float32), _p(grad_input, torch.float32), _s()),
              "interpolation_backward")
