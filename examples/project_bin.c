/* project_bin.c -- the projection entry point of the C-ABI from plain C99: where do the points of a KITTI velodyne .bin land in
 * the four rectified cameras (the arithmetic of the reference's camera_model.cpp; the drawing stays with the caller)?
 *
 *   project_bin <in.bin> <calib_dir> <out.uv> [max_range=15]
 *     calib_dir holds calib_velo_to_cam.txt and calib_cam_to_cam.txt; out.uv receives int32 uv[n][4][2] then uint8 bgrv[n][4]
 *
 * Build like deskew_bin.c.  The calibration text files are parsed here the way the reference does (data_io.cpp:168-210, :321-406):
 * "name: v1 v2 ..." lines, the values after the first token.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kmc_hip.h"

static int values_after_name(const char* line, double* out, int want) {
  const char* p = strchr(line, ':');
  int got = 0;
  if (!p) return 0;
  ++p;
  while (got < want) {
    char* end;
    double v = strtod(p, &end);
    if (end == p) break;
    out[got++] = v;
    p = end;
  }
  return got;
}

static int find_line(const char* path, const char* name, double* out, int want) {
  char line[1024];
  FILE* f = fopen(path, "r");
  int ok = 0;
  if (!f) { perror(path); return 0; }
  while (fgets(line, sizeof(line), f))
    if (strncmp(line, name, strlen(name)) == 0 && line[strlen(name)] == ':') { ok = values_after_name(line, out, want) == want; break; }
  fclose(f);
  return ok;
}

int main(int argc, char** argv) {
  char path[1024], name[32];
  double R[9], T[3];
  kmc_camera_rig rig;
  if (argc < 4) { fprintf(stderr, "usage: project_bin <in.bin> <calib_dir> <out.uv> [max_range]\n"); return 2; }
  snprintf(path, sizeof(path), "%s/calib_velo_to_cam.txt", argv[2]);
  if (!find_line(path, "R", R, 9) || !find_line(path, "T", T, 3)) { fprintf(stderr, "bad %s\n", path); return 1; }
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) rig.tf_c00_lo[4 * i + j] = R[3 * i + j];
    rig.tf_c00_lo[4 * i + 3] = T[i];
  }
  snprintf(path, sizeof(path), "%s/calib_cam_to_cam.txt", argv[2]);
  if (!find_line(path, "R_rect_00", rig.R_rect_00, 9)) { fprintf(stderr, "bad %s\n", path); return 1; }
  for (int c = 0; c < 4; ++c) {
    snprintf(name, sizeof(name), "P_rect_%02d", c);
    if (!find_line(path, name, rig.P_rect[c], 12)) { fprintf(stderr, "bad %s\n", path); return 1; }
  }
  rig.max_range = argc > 4 ? atof(argv[4]) : 15.0;

  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  fseek(f, 0, SEEK_END);
  long bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  size_t n = (size_t)bytes / 16;
  float* in = NULL;
  int32_t* uv = NULL;
  uint8_t* bgrv = NULL;
  if (posix_memalign((void**)&in, 64, n * 16 + 16) || posix_memalign((void**)&uv, 64, n * 32 + 32) || posix_memalign((void**)&bgrv, 64, n * 4 + 4)) return 1;
  if (fread(in, 16, n, f) != n) { fprintf(stderr, "short read\n"); return 1; }
  fclose(f);

  kmc_ctx* ctx = NULL;
  int rc = kmc_hip_create(&ctx, 0);
  if (rc != KMC_OK) { fprintf(stderr, "kmc_hip_create: %s\n", kmc_status_string(rc)); return 1; }
  rc = kmc_hip_project_f32(ctx, in, (uint64_t)n, &rig, NULL, NULL, uv, bgrv, KMC_MEM_HOST, NULL);
  if (rc != KMC_OK) { fprintf(stderr, "kmc_hip_project_f32: %s %s\n", kmc_status_string(rc), kmc_hip_last_error(ctx)); return 1; }
  kmc_hip_destroy(ctx);

  size_t drawn = 0;
  for (size_t i = 0; i < n; ++i) drawn += bgrv[4 * i + 3];
  printf("%zu points, %zu drawn within %.1f m\n", n, drawn, rig.max_range);
  f = fopen(argv[3], "wb");
  if (!f) { perror(argv[3]); return 1; }
  fwrite(uv, 32, n, f);
  fwrite(bgrv, 4, n, f);
  fclose(f);
  free(in); free(uv); free(bgrv);
  return 0;
}
