/* Oracle (test infrastructure): plain-C restatement of libs/pointops knn_query / grouping /
 * interpolation, one loop iteration per CUDA thread of the reference.
 *   knn:    libs/pointops/src/knn_query/knn_query_cuda_kernel.cu:15-104 (max-heap of nsample, reheap,
 *           heap_sort ascending, -1 / 1e10 padding, batch lookup through new_offset)
 *   group:  libs/pointops/src/grouping/grouping_cuda_kernel.cu:5-25
 *   interp: libs/pointops/src/interpolation/interpolation_cuda_kernel.cu:5-33
 * The reference .cu files include torch / CUDA headers (cuda_utils.h, ATen) and cannot be compiled
 * with gcc in this image, so the algorithm is restated; build with -ffp-contract=off so the distance
 * expression rounds exactly as written.  Never linked into the product. */
#include <stdint.h>

static void swap_f(float* x, float* y) { float t = *x; *x = *y; *y = t; }
static void swap_i(int* x, int* y) { int t = *x; *x = *y; *y = t; }

static void reheap(float* dist, int* idx, int k) {
  int root = 0, child = 1;
  while (child < k) {
    if (child + 1 < k && dist[child + 1] > dist[child]) child++;
    if (dist[root] > dist[child]) return;
    swap_f(&dist[root], &dist[child]);
    swap_i(&idx[root], &idx[child]);
    root = child;
    child = root * 2 + 1;
  }
}

static void heap_sort(float* dist, int* idx, int k) {
  for (int i = k - 1; i > 0; i--) {
    swap_f(&dist[0], &dist[i]);
    swap_i(&idx[0], &idx[i]);
    reheap(dist, idx, i);
  }
}

void oracle_knn_query(int m, int nsample, const float* xyz, const float* new_xyz, const int* offset,
                      const int* new_offset, int* idx, float* dist2) {
  for (int pt = 0; pt < m; ++pt) {
    int bt = 0;
    while (!(pt < new_offset[bt])) bt++;
    int start = bt == 0 ? 0 : offset[bt - 1], end = offset[bt];
    float nx = new_xyz[pt * 3], ny = new_xyz[pt * 3 + 1], nz = new_xyz[pt * 3 + 2];
    float best_dist[128];
    int best_idx[128];
    for (int i = 0; i < nsample; i++) { best_dist[i] = 1e10f; best_idx[i] = -1; }
    for (int i = start; i < end; i++) {
      float x = xyz[i * 3], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
      float d2 = (nx - x) * (nx - x) + (ny - y) * (ny - y) + (nz - z) * (nz - z);
      if (d2 < best_dist[0]) {
        best_dist[0] = d2;
        best_idx[0] = i;
        reheap(best_dist, best_idx, nsample);
      }
    }
    heap_sort(best_dist, best_idx, nsample);
    for (int i = 0; i < nsample; i++) {
      idx[(int64_t)pt * nsample + i] = best_idx[i];
      dist2[(int64_t)pt * nsample + i] = best_dist[i];
    }
  }
}

void oracle_grouping_forward(int m, int nsample, int c, const float* input, const int* idx, float* output) {
  for (int64_t index = 0; index < (int64_t)m * nsample * c; ++index) {
    int c_idx = index % c;
    int ns = (index / c) % nsample;
    int64_t mi = index / nsample / c;
    output[index] = input[(int64_t)idx[mi * nsample + ns] * c + c_idx];
  }
}

void oracle_interpolation_forward(int n, int c, int k, const float* input, const int* idx, const float* weight,
                                  float* output) {
  for (int64_t index = 0; index < (int64_t)n * c; ++index) {
    int c_idx = index % c;
    int64_t ni = index / c;
    for (int i = 0; i < k; i++)
      output[index] += input[(int64_t)idx[ni * k + i] * c + c_idx] * weight[ni * k + i];
  }
}

/* grouping_backward_cuda_kernel, libs/pointops/src/grouping/grouping_cuda_kernel.cu:16-25: one atomicAdd per
 * (m, nsample, c) element into grad_in[idx]; here the threads run in index order (the CUDA order is unspecified:
 * fp32 sums agree with a GPU run up to the order of additions). */
void oracle_grouping_backward(int m, int nsample, int c, const float* grad_out, const int* idx, float* grad_in) {
  for (int64_t index = 0; index < (int64_t)m * nsample * c; ++index) {
    int c_idx = index % c;
    int ns = (index / c) % nsample;
    int64_t mi = index / nsample / c;
    grad_in[(int64_t)idx[mi * nsample + ns] * c + c_idx] += grad_out[index];
  }
}

/* interpolation_backward_cuda_kernel, libs/pointops/src/interpolation/interpolation_cuda_kernel.cu:20-33 */
void oracle_interpolation_backward(int n, int c, int k, const float* grad_out, const int* idx, const float* weight,
                                   float* grad_in) {
  for (int64_t index = 0; index < (int64_t)n * c; ++index) {
    int c_idx = index % c;
    int64_t ni = index / c;
    for (int i = 0; i < k; i++)
      grad_in[(int64_t)idx[ni * k + i] * c + c_idx] += grad_out[index] * weight[ni * k + i];
  }
}
