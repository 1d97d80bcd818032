/* Oracle helper (test infrastructure only): TimeEncode of the reference, model/time_encoding.py:17-25.
 * torch-CPU evaluates Linear(1, D) on a scalar input as ONE fp32 FMA; C99 fmaf() is exactly that. */
#include <math.h>
#include <stdint.h>

void oracle_time_encode(const float* t, int64_t n, const float* w, const float* b, int D, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i)
    for (int d = 0; d < D; ++d) out[i * D + d] = cosf(fmaf(t[i], w[d], b[d]));
}

/* dw[d] = sum_i g[i,d] * -sin(arg) * t[i],  db[d] = sum_i g[i,d] * -sin(arg)   (fp64 accumulation) */
void oracle_time_encode_bwd(const float* t, int64_t n, const float* w, const float* b, int D, const float* g,
                            double* dw, double* db) {
  for (int d = 0; d < D; ++d) { dw[d] = 0.0; db[d] = 0.0; }
#pragma omp parallel
  {
    double lw[512], lb[512];
    for (int d = 0; d < D; ++d) { lw[d] = 0.0; lb[d] = 0.0; }
#pragma omp for schedule(static)
    for (int64_t i = 0; i < n; ++i)
      for (int d = 0; d < D; ++d) {
        const float s = -sinf(fmaf(t[i], w[d], b[d])) * g[i * D + d];
        lw[d] += (double)(s * t[i]);
        lb[d] += (double)s;
      }
#pragma omp critical
    for (int d = 0; d < D; ++d) { dw[d] += lw[d]; db[d] += lb[d]; }
  }
}
