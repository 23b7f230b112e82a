// tools/tileprobe.cpp -- what one trailing-update tile product (C tile read, nkb k-blocks of 128, C tile written) costs per
// tile configuration and occupancy, `tiles` of them in one launch (pgm_debug_tile_probe).  Chip time per product =
// launch time / (tiles * nkb).
#include <cstdio>
#include <cstdlib>
#include "../include/pgmuvi_hip.h"
extern "C" int pgm_debug_tile_probe(pgm_ws* ws, int cfg, int tiles, int nkb, int reps, double* us_host);
int main() {
  pgm_ws* ws = nullptr;
  if (pgm_workspace_create(&ws, 0, 4096, 4, 1, 1) != 0) { printf("no workspace\n"); return 1; }
  const char* names[] = {"16 waves 128x128 (32x32 each), 1/CU, PF1", "16 waves 128x128, PF2", "8 waves 128x128 (64x32 each), 2/CU, PF1",
                         "4 waves 128x128 (64x64 each), 2/CU, PF2", "8 waves 64x64 sub-tiles (32x16 each)", "4 waves 64x64 sub-tiles (32x32 each)",
                         "8 waves 128x128 (64x32 each), 1/CU by registers, PF2", "16 waves 128x128, PF2, C tile not read",
                         "16 waves 128x128, PF2, C tile not written", "16 waves 128x128, PF2, C tile neither read nor written",
                         "16 waves 128x128, PF2, C tile negated late", "4 waves 128x128 (64x64 each), 2/CU, C tile negated late",
                         "8 waves 64x64 sub-tiles, C tile negated late", "16 waves 128x128, PF2, 32-row chunks, negated late",
                         "16 waves 128x128, PF4, 16-row chunks, negated late", "16 waves 128x128, PF4, 8-row chunks, negated late"};
  // fewer tiles than CUs: is the C tile's round trip a matter of how many workgroups ask at once?
  for (int tiles : {16, 32, 64, 128, 192, 255})
    for (int nkb : {1, 2}) {
      double us = 0.0;
      const int rc = pgm_debug_tile_probe(ws, 10, tiles, nkb, 20, &us);
      printf("%-58s tiles %3d nkb %d: %7.2f us per launch%s\n", "16 waves 128x128, PF2, negated late, partly filled chip", tiles, nkb, us, rc ? "  (FAILED)" : "");
    }
  for (int cfg = 0; cfg < 16; ++cfg)
    for (int tiles : {255, 510})
      for (int nkb : {1, 2}) {
        double us = 0.0;
        const int rc = pgm_debug_tile_probe(ws, cfg, tiles, nkb, 20, &us);
        printf("%-58s tiles %3d nkb %d: %7.2f us per launch, %.4f us of chip time per product, %.1f TFLOP/s%s\n", names[cfg], tiles, nkb, us,
               us / (tiles * nkb), 2.0 * 128 * 128 * 128 * nkb * tiles / us * 1e-6, rc ? "  (FAILED)" : "");
      }
  pgm_workspace_destroy(ws);
  return 0;
}
