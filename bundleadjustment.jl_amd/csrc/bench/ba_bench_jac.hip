// Bench-only entry point of the Jacobian kernel variants (tools/bench_jac.py): built into tools/libba_bench.so, NOT
// into libba_hip.so.  Includes the product source so that the ablation variants of k_jac_coord are instantiated here only.
#include "../ba_model_kernels.hip"

extern "C" int ba_debug_jac_bench(ba_problem *p, const double *d_x, double *d_vals, int variant, int reps, double *ms_out) {
  hipEvent_t e0, e1;
  BA_HIP_CHECK(hipSetDevice(p->device));
  BA_HIP_CHECK(hipEventCreate(&e0));
  BA_HIP_CHECK(hipEventCreate(&e1));
  hipStream_t st = p->stream;
  double *cpad = nullptr;
  BA_CHECK(ba_scratch(p, 3, (size_t)(p->ncams * CPAD + 2) * sizeof(double), (void **)&cpad));
  auto launch = [&]() {
    dim3 b(BLK);
    hipLaunchKernelGGL(k_cam_pre<double>, dim3(grid_for(p->ncams, BLK)), dim3(BLK), 0, st, p->ncams, d_x + 3 * p->npnts,
                       cpad);
#define JV(v) case v: hipLaunchKernelGGL((k_jac_coord<double, v>), dim3(jac_grid(p, k_jac_coord<double, v>, p->nobs)), b, 0, st, p->nobs, p->npnts, p->cam0, p->pnt0, d_x, (const double *)cpad, d_vals); break;
    switch (variant) {
      JV(1) JV(2) JV(3) JV(5) JV(13) JV(9) JV(25) JV(41) JV(57) JV(64) JV(77)
      default: hipLaunchKernelGGL((k_jac_coord<double, 0>), dim3(jac_grid(p, k_jac_coord<double, 0>, p->nobs)), b, 0, st, p->nobs, p->npnts, p->cam0, p->pnt0, d_x, (const double *)cpad, d_vals);
    }
#undef JV
  };
  launch();
  BA_HIP_CHECK(hipStreamSynchronize(st));
  BA_HIP_CHECK(hipEventRecord(e0, st));
  for (int r = 0; r < reps; r++) launch();
  BA_HIP_CHECK(hipEventRecord(e1, st));
  BA_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0;
  BA_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / reps;
  return BA_OK;
}

