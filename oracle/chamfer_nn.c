/* oracle/chamfer_nn.c — TEST INFRASTRUCTURE ONLY (CPU checker, never shipped, never on the product path).
 *
 * Plain-C brute-force K=1 nearest neighbour under squared L2, the arithmetic PyTorch3D 0.5.0's
 * knn_points(K=1) performs for pytorch3d.loss.chamfer_distance, which the reference calls at
 * pterotactyl/utility/utils.py:207,212 (PyTorch3D is not vendored in the reference: parity unpinned,
 * see oracle/__init__.py).  d(i,j) = (x-y)^2 summed over the 3 coordinates in float32; the first
 * minimum in j order wins ties.
 *
 * Build: gcc -O3 -ffp-contract=off -fopenmp -shared -fPIC chamfer_nn.c -o libchamfer_nn.so
 */
#include <float.h>

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
