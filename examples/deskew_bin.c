/* deskew_bin.c -- the C-ABI from plain C99 (no C++, no Python): deskew one KITTI velodyne .bin with two scan poses.
 *
 *   deskew_bin <in.bin> <out.bin> <tx ty tz  rx ry rz>      (relative motion T_start -> T_end as an se(3) twist)
 *
 * Build:  gcc -std=c99 -Iinclude examples/deskew_bin.c -Lkitti_motion_compensation_amd/lib -lkmc_hip \
 *             -Wl,-rpath,$PWD/kitti_motion_compensation_amd/lib -Wl,-rpath,/opt/rocm/lib -o deskew_bin
 * With no arguments it only checks the library and the host pre-step (usable on a box without a GPU).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kmc_hip.h"

int main(int argc, char** argv) {
  /* T_start = identity, T_end = translation (1.3, 0, 0): f must come back as rho = (1.3, 0, 0), phi = 0 */
  double T0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  double T1[12] = {1, 0, 0, 1.3, 0, 1, 0, 0, 0, 0, 1, 0};
  kmc_frame_params p;
  int rc = kmc_frame_params_from_poses(T0, T1, 10.0, 10.1, 10.05, &p);
  if (rc != KMC_OK || fabs(p.twist[0] - 1.3) > 1e-12 || fabs(p.x_req - 0.5) > 1e-12) {
    fprintf(stderr, "host pre-step failed: %s\n", kmc_status_string(rc));
    return 1;
  }
  printf("kmc ABI %d, host pre-step ok (rho_x = %.3f, x_req = %.2f)\n", kmc_abi_version(), p.twist[0], p.x_req);
  if (argc < 9) return 0;

  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  fseek(f, 0, SEEK_END);
  long bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  size_t n = (size_t)bytes / 16;
  /* page-locked, device-addressable buffers from the library's pool when there is a HIP device: kmc_hip_deskew_f32 then works on
   * them IN PLACE over the link (one kernel, no staging copies); ordinary memory otherwise (staged three-stream pipeline) */
  float* in = NULL;
  float* out = NULL;
  int pooled = kmc_host_pool_alloc(n * 16 + 16, (void**)&in) == KMC_OK && kmc_host_pool_alloc(n * 16 + 16, (void**)&out) == KMC_OK;
  if (!pooled) {
    if (in) kmc_host_pool_free(in);
    in = out = NULL;
    if (posix_memalign((void**)&in, 64, n * 16 + 16) || posix_memalign((void**)&out, 64, n * 16 + 16)) return 1;
  }
  if (fread(in, 16, n, f) != n) { fprintf(stderr, "short read\n"); return 1; }
  fclose(f);

  for (int i = 0; i < 6; ++i) p.twist[i] = atof(argv[3 + i]);
  p.x_req = 0.5;
  kmc_ctx* ctx = NULL;
  rc = kmc_hip_create(&ctx, 0);
  if (rc != KMC_OK) { fprintf(stderr, "%s\n", kmc_status_string(rc)); return 1; }
  kmc_stats st;
  rc = kmc_hip_deskew_f32(ctx, in, out, (uint64_t)n, &p, KMC_MEM_HOST, &st);
  if (rc != KMC_OK) { fprintf(stderr, "%s: %s\n", kmc_status_string(rc), kmc_hip_last_error(ctx)); return 1; }
  f = fopen(argv[2], "wb");
  if (!f) { perror(argv[2]); return 1; }
  fwrite(out, 16, n, f);
  fclose(f);
  printf("deskewed %llu points in %u launch(es), tier %u, %s\n", (unsigned long long)st.n_points, st.n_launches, st.variant,
         pooled ? "in place on page-locked buffers" : "through staged copies");
  kmc_hip_destroy(ctx);
  if (pooled) {
    kmc_host_pool_free(in);
    kmc_host_pool_free(out);
  } else {
    free(in);
    free(out);
  }
  return 0;
}
