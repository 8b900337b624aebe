/* A host WITHOUT Python: runs a model exported by deephar_amd.Model.export_plan through the plan / execute pair of
 * include/deephar_hip.h -- what a C replacement of `model.predict(x, batch_size=...)` (exp/common/mpii_tools.py:86,
 * exp/pennaction/eval_speed2d.py:70-77) looks like.  Plain C, no HIP headers:
 *     gcc -O2 -Iinclude tools/c_host/predict_plan.c -Ldeephar_amd/csrc -ldeephar_hip -Wl,-rpath,$PWD/deephar_amd/csrc -o predict_plan
 *     ./predict_plan model.dhplan frames.f32 out_prefix        (frames.f32: raw float32 [m, 256, 256, 3])
 * writes out_prefix.<k>.f32 per model output and prints the frames/s of a second, timed call. */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "deephar_hip.h"

static void* slurp(const char* path, size_t* n) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END);
  *n = (size_t)ftell(f);
  fseek(f, 0, SEEK_SET);
  void* p = malloc(*n);
  if (fread(p, 1, *n, f) != *n) { perror("read"); exit(2); }
  fclose(f);
  return p;
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s plan.dhplan frames.f32 out_prefix\n", argv[0]); return 2; }
  size_t nb, nx;
  void* blob = slurp(argv[1], &nb);
  float* x = (float*)slurp(argv[2], &nx);
  dh_plan* plan = NULL;
  int rc = dh_plan_create(blob, nb, &plan);
  if (rc != DH_OK) { fprintf(stderr, "dh_plan_create: %s\n", dh_error_string(rc)); return 1; }
  free(blob);
  const int nout = dh_plan_num_outputs(plan);
  const int m = (int)(nx / 4 / (size_t)dh_plan_input_items(plan, 0));
  if (dh_plan_num_inputs(plan) != 1 || m < 1 || m > dh_plan_batch(plan)) { fprintf(stderr, "bad input size\n"); return 1; }
  float** outs = (float**)calloc((size_t)nout, sizeof(float*));
  for (int k = 0; k < nout; ++k) outs[k] = (float*)malloc((size_t)dh_plan_output_items(plan, k) * 4 * (size_t)m);
  const float* ins[1] = {x};
  rc = dh_forward_host(plan, ins, m, outs);
  if (rc != DH_OK) { fprintf(stderr, "dh_forward_host: %s\n", dh_error_string(rc)); return 1; }
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  rc = dh_forward_host(plan, ins, m, outs);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (rc != DH_OK) return 1;
  for (int k = 0; k < nout; ++k) {
    char path[1024];
    snprintf(path, sizeof path, "%s.%d.f32", argv[3], k);
    FILE* f = fopen(path, "wb");
    fwrite(outs[k], 4, (size_t)dh_plan_output_items(plan, k) * (size_t)m, f);
    fclose(f);
  }
  const double dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  printf("%d frames, %d outputs, %.3f ms (%.1f frames/s incl. H2D / D2H)\n", m, nout, 1e3 * dt, m / dt);
  dh_plan_destroy(plan);
  return 0;
}
