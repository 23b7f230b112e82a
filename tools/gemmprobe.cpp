#include <hip/hip_runtime.h>
#include <cstdio>
#include "../include/pgmuvi_hip.h"
extern "C" int pgm_debug_gemm_probe(pgm_ws*, int, int, int, int, double*);
int main() {
  pgm_ws* ws; if (pgm_workspace_create(&ws, 0, 4096, 4, 1, 1)) return 1;
  for (int cfg = 0; cfg < 2; ++cfg)
    for (int spread : {0, 1})
      for (int blocks : {256, 512, 1024, 2048})
        for (int nkb : {1, 4, 16}) {
          double tf = 0; pgm_debug_gemm_probe(ws, cfg, blocks, nkb, spread, &tf);
          printf("cfg %s spread %d blocks %4d nkb %2d : %6.1f TFLOP/s\n", cfg ? "64x64 " : "128x128", spread, blocks, nkb, tf);
        }
  return 0;
}
