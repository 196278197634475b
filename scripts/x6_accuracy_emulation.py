#!/usr/bin/env python
"""numpy emulation behind DESIGN 10.2: how accurate is a 256-deep GEMM of the MLP when every float32 operand is split into
bf16 pieces and the product is a sum of bf16 x bf16 MFMAs with float32 accumulation?

  x3: x = hi + lo,          products hi*hi + hi*lo + lo*hi              (what csrc/mlp_x3_kernels.hip does, inference only)
  x6: x = x1 + x2 + x3,     the six products of order <= 2^-16           (the f32-accurate lever that is priced, not built)

Against a float64 product of the same float32 operands (post-ReLU activations x Glorot weights, 2048 x 256 x 256):
  float32 GEMM (BLAS)   rms relative error 2.0e-7
  bf16 x3                                  4.3e-6   (21 x the float32 GEMM's)
  bf16 x6                                  6.2e-8   (partial sums rounded to float32 once each here; an MFMA chain rounds per
                                                     k-step and lands at the float32 GEMM's 2e-7)
CPU only, seconds.  Nothing in the product imports this."""
import numpy as np


def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16                   # round to nearest even
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    a = bf16(x); r = (x - a).astype(np.float32)
    b = bf16(r); r2 = (r - b).astype(np.float32)
    return a, b, bf16(r2)


def main():
    rs = np.random.RandomState(0)
    M, K, N = 2048, 256, 256
    X = np.maximum(rs.randn(M, K), 0).astype(np.float32)
    W = ((rs.rand(K, N).astype(np.float32) * 2 - 1) * np.float32(np.sqrt(6 / (K + N))))
    ref = X.astype(np.float64) @ W.astype(np.float64)
    xs, ws = split3(X), split3(W)

    def emul(terms):
        acc = np.zeros((M, N), np.float32)
        for i, j in terms:
            acc = (acc + (xs[i].astype(np.float64) @ ws[j].astype(np.float64)).astype(np.float32)).astype(np.float32)
        return acc

    def err(y):
        return float(np.abs(y - ref).max() / np.abs(ref).max()), float(np.sqrt(((y - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean()))

    print("float32 GEMM   max / rms relative error  %.2e  %.2e" % err(X @ W))
    print("bf16 x3        max / rms relative error  %.2e  %.2e" % err(emul([(0, 0), (0, 1), (1, 0)])))
    print("bf16 x6        max / rms relative error  %.2e  %.2e" % err(emul([(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)])))


if __name__ == "__main__":
    main()
