/* ABI check of include/ba_hip.h from plain C (gcc -std=c99): the header must compile as C, the library must link from C,
 * and the struct layouts a foreign-function binding (ctypes in _lib.py, the Julia structs of julia/BALHIP.jl) mirrors by
 * hand are printed so that tests/test_host.py can compare them field by field.
 *
 *   gcc -std=c99 -Wall -Werror -pedantic -I include tests/c_abi/abi_check.c -L bundleadjustment.jl_amd -lba_hip -o abi_check
 *
 * Calls only entries that need no GPU: ba_last_error, ba_device_count (may fail without a device; must not crash),
 * ba_read_bal_header / ba_read_bal on the file given as argv[1], ba_problem_dims on a null handle (must be refused). */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>

#include "ba_hip.h"

#define OFF(T, f) printf("offsetof %s %s %zu\n", #T, #f, offsetof(T, f))

int main(int argc, char **argv) {
  printf("sizeof ba_lm_opts %zu\n", sizeof(ba_lm_opts));
  OFF(ba_lm_opts, variant); OFF(ba_lm_opts, facto); OFF(ba_lm_opts, normalize); OFF(ba_lm_opts, linesearch);
  OFF(ba_lm_opts, facto_type); OFF(ba_lm_opts, ite_max); OFF(ba_lm_opts, verbose); OFF(ba_lm_opts, x_f32);
  OFF(ba_lm_opts, restol); OFF(ba_lm_opts, satol); OFF(ba_lm_opts, srtol); OFF(ba_lm_opts, oatol);
  OFF(ba_lm_opts, ortol); OFF(ba_lm_opts, atol); OFF(ba_lm_opts, rtol); OFF(ba_lm_opts, nu_d); OFF(ba_lm_opts, nu_m);
  OFF(ba_lm_opts, lambda); OFF(ba_lm_opts, delta_d); OFF(ba_lm_opts, max_time); OFF(ba_lm_opts, pcg_tol); OFF(ba_lm_opts, pcg_max_iter); OFF(ba_lm_opts, perm);
  printf("sizeof ba_lm_stats %zu\n", sizeof(ba_lm_stats));
  OFF(ba_lm_stats, status); OFF(ba_lm_stats, iter); OFF(ba_lm_stats, n_accepted); OFF(ba_lm_stats, n_rejected);
  OFF(ba_lm_stats, n_residual); OFF(ba_lm_stats, n_jacobian); OFF(ba_lm_stats, n_factor); OFF(ba_lm_stats, n_cg);
  OFF(ba_lm_stats, objective); OFF(ba_lm_stats, dual_feas); OFF(ba_lm_stats, lambda_final);
  OFF(ba_lm_stats, elapsed_s); OFF(ba_lm_stats, loop_s);
  printf("enum BA_OK %d BA_ERR_ARG %d BA_ERR_HIP %d BA_ERR_IO %d BA_ERR_ZERO_PIVOT %d BA_ERR_NAN_STEP %d BA_ERR_COMM %d\n",
         BA_OK, BA_ERR_ARG, BA_ERR_HIP, BA_ERR_IO, BA_ERR_ZERO_PIVOT, BA_ERR_NAN_STEP, BA_ERR_COMM);
  printf("enum BA_ST_SMALL_STEP %d BA_ST_FIRST_ORDER %d BA_ST_SMALL_RESIDUAL %d BA_ST_ACCEPTABLE %d BA_ST_NEG_PRED %d "
         "BA_ST_EXCEPTION %d BA_ST_MAX_ITER %d BA_ST_UNKNOWN %d\n",
         BA_ST_SMALL_STEP, BA_ST_FIRST_ORDER, BA_ST_SMALL_RESIDUAL, BA_ST_ACCEPTABLE, BA_ST_NEG_PRED, BA_ST_EXCEPTION,
         BA_ST_MAX_ITER, BA_ST_UNKNOWN);

  int ndev = -1;
  int rc = ba_device_count(&ndev);
  printf("ba_device_count rc %d n %d\n", rc, ndev);

  int64_t d[6] = {0, 0, 0, 0, 0, 0};
  rc = ba_problem_dims(NULL, &d[0], &d[1], &d[2], &d[3], &d[4], &d[5]);
  printf("ba_problem_dims(NULL) rc %d\n", rc);
  if (rc == BA_OK) return 2;

  if (argc > 1) {
    int64_t ncams = 0, npnts = 0, nobs = 0;
    rc = ba_read_bal_header(argv[1], &ncams, &npnts, &nobs);
    printf("ba_read_bal_header rc %d ncams %lld npnts %lld nobs %lld\n", rc, (long long)ncams, (long long)npnts, (long long)nobs);
    if (rc != BA_OK) {
      printf("error: %s\n", ba_last_error());
      return 3;
    }
    int64_t *cam = (int64_t *)malloc((size_t)nobs * sizeof(int64_t)), *pnt = (int64_t *)malloc((size_t)nobs * sizeof(int64_t));
    double *pt2d = (double *)malloc((size_t)(2 * nobs) * sizeof(double));
    double *x0 = (double *)malloc((size_t)(9 * ncams + 3 * npnts) * sizeof(double));
    rc = ba_read_bal(argv[1], ncams, npnts, nobs, cam, pnt, pt2d, x0);
    printf("ba_read_bal rc %d cam[0] %lld pnt[last] %lld x0[last] %.17g\n", rc, (long long)cam[0], (long long)pnt[nobs - 1],
           x0[9 * ncams + 3 * npnts - 1]);
    free(cam); free(pnt); free(pt2d); free(x0);
    if (rc != BA_OK) return 4;
  }
  rc = ba_read_bal_header("/nonexistent/problem-1-1-pre.txt", &d[0], &d[1], &d[2]);
  printf("ba_read_bal_header(missing) rc %d msg %s\n", rc, ba_last_error());
  return rc == BA_ERR_IO ? 0 : 5;
}
