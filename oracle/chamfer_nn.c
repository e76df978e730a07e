/* oracle/chamfer_nn.c — TEST INFRASTRUCTURE ONLY (CPU checker, never shipped, never on the product path).
 *
 * Plain-C brute-force K=1 nearest neighbour under squared L2, the arithmetic PyTorch3D 0.5.0's
 * knn_points(K=1) performs for pytorch3d.loss.chamfer_distance, which the reference calls at
 * pterotactyl/utility/utils.py:207,212 (PyTorch3D is not vendored in the reference: parity unpinned,
 * see oracle/__init__.py).  d(i,j) = (x-y)^2 summed over the 3 coordinates in float32; the first
 * minimum in j order wins ties.
 *
 * oracle_nn_sqdist_fma is the same search with the three products contracted exactly as the device compiler
 * contracts them, d = fma(dz, dz, fma(dy, dy, dx*dx)) (csrc/chamfer.hip): the distances, and therefore the first
 * arg-min under near-ties, are then comparable BIT FOR BIT with the HIP kernels (tests at the named sizes).
 * PyTorch3D's CUDA kernel leaves the contraction to nvcc, so neither form is "more" the reference.
 *
 * Build: gcc -O3 -ffp-contract=off -mfma -fopenmp -shared -fPIC chamfer_nn.c -o libchamfer_nn.so -lm
 */
#include <float.h>
#include <math.h>
#include <omp.h>

/* number of OpenMP threads of the searches below (the cpu_baseline leg of bench.py times them on all cores and on 1) */
void oracle_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int oracle_get_threads(void) { return omp_get_max_threads(); }

void oracle_nn_sqdist(const float *x, int nx, const float *y, int ny, float *dist, int *idx) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < nx; ++i) {
    const float xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2];
    float best = FLT_MAX;
    int bj = 0;
    for (int j = 0; j < ny; ++j) {
      const float dx = xi - y[3 * j], dy = yi - y[3 * j + 1], dz = zi - y[3 * j + 2];
      const float d = dx * dx + dy * dy + dz * dz;
      if (d < best) {
        best = d;
        bj = j;
      }
    }
    dist[i] = best;
    idx[i] = bj;
  }
}

void oracle_nn_sqdist_fma(const float *x, int nx, const float *y, int ny, float *dist, int *idx) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < nx; ++i) {
    const float xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2];
    float best = FLT_MAX;
    int bj = 0;
    for (int j = 0; j < ny; ++j) {
      const float dx = xi - y[3 * j], dy = yi - y[3 * j + 1], dz = zi - y[3 * j + 2];
      const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
      if (d < best) {
        best = d;
        bj = j;
      }
    }
    dist[i] = best;
    idx[i] = bj;
  }
}
