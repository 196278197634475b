// Real spherical-harmonics basis shared by the NeRF-SH shading kernels and the octree renderer.
#pragma once
#include <hip/hip_runtime.h>

namespace pxo {

// ------------------------------------------------------------------------------------------
// SH basis (nerf_sh/nerf/sh.py:24-52, :72-108): multipliers of sh[..., k]
// ------------------------------------------------------------------------------------------
template <int DEG>
__device__ __forceinline__ void sh_basis(float x, float y, float z, float (&Y)[(DEG + 1) * (DEG + 1)]) {
  Y[0] = 0.28209479177387814f;
  if constexpr (DEG > 0) {
    Y[1] = -0.4886025119029199f * y;
    Y[2] = 0.4886025119029199f * z;
    Y[3] = -0.4886025119029199f * x;
  }
  if constexpr (DEG > 1) {
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    Y[4] = 1.0925484305920792f * xy;
    Y[5] = -1.0925484305920792f * yz;
    Y[6] = 0.31539156525252005f * (2.0f * zz - xx - yy);
    Y[7] = -1.0925484305920792f * xz;
    Y[8] = 0.5462742152960396f * (xx - yy);
    if constexpr (DEG > 2) {
      Y[9] = -0.5900435899266435f * y * (3.f * xx - yy);
      Y[10] = 2.890611442640554f * xy * z;
      Y[11] = -0.4570457994644658f * y * (4.f * zz - xx - yy);
      Y[12] = 0.3731763325901154f * z * (2.f * zz - 3.f * xx - 3.f * yy);
      Y[13] = -0.4570457994644658f * x * (4.f * zz - xx - yy);
      Y[14] = 1.445305721320277f * z * (xx - yy);
      Y[15] = -0.5900435899266435f * x * (xx - 3.f * yy);
    }
    if constexpr (DEG > 3) {
      Y[16] = 2.5033429417967046f * xy * (xx - yy);
      Y[17] = -1.7701307697799304f * yz * (3.f * xx - yy);
      Y[18] = 0.9461746957575601f * xy * (7.f * zz - 1.f);
      Y[19] = -0.6690465435572892f * yz * (7.f * zz - 3.f);
      Y[20] = 0.10578554691520431f * (zz * (35.f * zz - 30.f) + 3.f);
      Y[21] = -0.6690465435572892f * xz * (7.f * zz - 3.f);
      Y[22] = 0.47308734787878004f * (xx - yy) * (7.f * zz - 1.f);
      Y[23] = -1.7701307697799304f * xz * (xx - 3.f * yy);
      Y[24] = 0.6258357354491761f * (xx * (xx - 3.f * yy) - yy * (3.f * xx - yy));
    }
  }
}

}  // namespace pxo
