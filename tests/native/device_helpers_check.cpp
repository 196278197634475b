// Device helper functions of the kernels, compiled as HOST code (hipcc -x c++: the __device__ qualifiers are inert) and
// checked on the CPU against reference-pinned data: run by tests/test_host_cpu.py::test_device_helpers_on_the_host.
//   sh_basis<4>      (pxo_sh.h, used by the shading kernels and the octree renderer) against tests/golden/sh_proj.npz =
//                    the reference's octree/nerf/sh_proj.py:EvalSH, passed in as a raw float64 file [n][3] + [n][25]
//   parameter layout (pxo_common.h) against the reference's parameter counts (505,649 / 512,588 per MLP)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "pxo_common.h"
#include "pxo_sh.h"

using namespace pxo;

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<double> buf;
  double v;
  while (std::fread(&v, sizeof(double), 1, f) == 1) buf.push_back(v);
  std::fclose(f);
  const size_t n = buf.size() / 28;
  if (n == 0 || buf.size() != n * 28) return 2;
  double worst = 0.0;
  for (size_t i = 0; i < n; ++i) {
    float Y4[25], Y3[16], Y2[9], Y1[4], Y0[1];
    const float x = (float)buf[i * 3], y = (float)buf[i * 3 + 1], z = (float)buf[i * 3 + 2];
    sh_basis<4>(x, y, z, Y4); sh_basis<3>(x, y, z, Y3); sh_basis<2>(x, y, z, Y2); sh_basis<1>(x, y, z, Y1); sh_basis<0>(x, y, z, Y0);
    for (int k = 0; k < 25; ++k) {
      const double want = buf[n * 3 + i * 25 + k];
      worst = std::fmax(worst, std::fabs((double)Y4[k] - want));
      if (k < 16) worst = std::fmax(worst, std::fabs((double)Y3[k] - want));   // lower degrees: prefixes of the same list
      if (k < 9) worst = std::fmax(worst, std::fabs((double)Y2[k] - want));
      if (k < 4) worst = std::fmax(worst, std::fabs((double)Y1[k] - want));
      if (k < 1) worst = std::fmax(worst, std::fabs((double)Y0[k] - want));
    }
  }
  long bad = 0;
  bad += mlp_param_count(3) != 505649;
  bad += mlp_param_count(4) != 512588;
  for (int deg = 0; deg <= 4; ++deg) {
    int64_t off = 0;
    for (int l = 0; l < 10; ++l) {                                      // Dense_l kernel [in, out] then bias [out], l = 0..9
      const int in = l == 0 ? 63 : (l == 5 ? 319 : 256), out = l < 8 ? 256 : (l == 8 ? 1 : 3 * (deg + 1) * (deg + 1));
      bad += layer_in(l) != in || layer_out(l, deg) != out;
      bad += leaf_kernel_off(l, deg) != off;
      off += (int64_t)in * out;
      bad += leaf_bias_off(l, deg) != off;
      off += out;
    }
    bad += mlp_param_count(deg) != off;
    bad += head_blocks(deg) != (3 * (deg + 1) * (deg + 1) + 1 + 31) / 32;
  }
  std::printf("dirs %zu sh_max_err %.3e layout_bad %ld\n", n, worst, bad);
  return (worst < 2e-6 && bad == 0) ? 0 : 1;
}
