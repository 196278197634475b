// Host-side property check of the persistent-tile schedule shared by mlp_fwd_kernel / mlp_bwd_data_kernel
// (plenoctree_amd/csrc/pxo_common.h: num_tiles, tile_sched, mask_slots).  Compiled and run by
// tests/test_host_cpu.py::test_tile_schedule_properties (hipcc, host code only - no device is touched).
#include <cstdint>
#include <cstdio>
#include <vector>

#include "pxo_common.h"

using namespace pxo;

static long g_bad = 0;
#define CHECK(cond)                                                                                          \
  do {                                                                                                       \
    if (!(cond)) {                                                                                           \
      if (g_bad < 10) std::printf("FAIL %s  (M=%lld grid=%lld n_full=%lld n_half=%lld half_row0=%lld)\n", #cond, \
                                  (long long)M, (long long)grid, (long long)t.n_full, (long long)t.n_half,    \
                                  (long long)t.half_row0);                                                    \
      ++g_bad;                                                                                               \
    }                                                                                                        \
  } while (0)

static void check(int64_t M, int64_t grid) {
  const TileSched t = tile_sched(M, grid);
  const int64_t H = kTM / 2;
  CHECK(t.n_full >= 0 && t.n_half >= 0);
  CHECK(t.half_row0 == t.n_full * kTM);                                   // half tiles start where the full ones end
  if (t.n_half == 0) {                                                    // full tiles alone cover [0, M), minimally
    CHECK(t.n_full == num_tiles(M));
    CHECK(t.n_full * kTM >= M && (t.n_full == 0 || (t.n_full - 1) * kTM < M));
  } else {
    CHECK(t.n_full % grid == 0 && t.n_full > 0);                          // whole rounds of full tiles ...
    CHECK(t.n_half <= grid);                                              // ... then ONE round of half tiles
    CHECK(t.n_full * kTM < M);
    CHECK(t.n_full * kTM + t.n_half * H >= M && t.n_full * kTM + (t.n_half - 1) * H < M);
    // and it is a saving: the same rows as full tiles would need a round of their own with more than half of it idle
    CHECK(num_tiles(M) > t.n_full && (M - t.n_full * kTM) <= grid * H);
  }
  CHECK(t.n_full + t.n_half <= mask_slots(M) || grid > kMaxMlpGrid);      // every tile has a relu-mask / partial slot
}

int main() {
  long cases = 0;
  const int64_t grids[] = {1, 2, 3, 7, 8, 64, 104, 255, 256, 304, kMaxMlpGrid};
  for (int64_t grid : grids) {
    if (grid > kMaxMlpGrid) continue;
    for (int64_t M = 0; M <= 5000; ++M, ++cases) check(M, grid);
    for (int64_t k = 1; k <= 40; ++k)
      for (int64_t d = -130; d <= 130; ++d) {
        const int64_t M = k * kTM * grid / 2 + d;
        if (M >= 0) { check(M, grid); ++cases; }
      }
    const int64_t sizes[] = {262144, 786432, 796432, 32768, 98304, 108304, 65536, 206608, 134217728, 46000003, 75600000};
    for (int64_t M : sizes) { check(M, grid); ++cases; }
  }
  std::printf("cases %ld bad %ld kTM %d kMaxMlpGrid %d\n", cases, g_bad, kTM, (int)kMaxMlpGrid);
  return g_bad ? 1 : 0;
}
