#!/usr/bin/env python3
"""VERDICT r2 item 10, numerical part only (no kernel): what would a 3-way split-bf16 correlation lose?

f32 x is written x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (8 + 8 + 8 significant bits), and a
product a*b is replaced by the six bf16 x bf16 products whose weight is >= 2^-16 of it (hh, hm, mh, hl, lh, mm) -- each exact
in f32 -- accumulated in f32 (what v_mfma_f32_16x16x32_bf16 delivers; the order inside the instruction is not modelled,
terms are added in channel order).  Compared with the reference's f32 FMA chain (pointwise_correlation_cuda.cu:105-107)
and with the exact (f64) dot product, on the metric's input distribution U[0,1) and on zero-mean data.

    python lab/tools/split_bf16_study.py            # CPU, numpy, a few seconds
"""
import numpy as np


def bf16(x):
    u = x.astype(np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)


def split3(x):
    h = bf16(x)
    m = bf16(x - h)
    l = bf16(x - h - m)
    return h, m, l


def study(C, n, rng, signed):
    a = (rng.standard_normal((n, C)) if signed else rng.random((n, C))).astype(np.float32)
    b = (rng.standard_normal((n, C)) if signed else rng.random((n, C))).astype(np.float32)
    exact = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
    chain = np.zeros(n, np.float32)
    for c in range(C):                                    # one rounding per term: an fmaf chain
        chain = (chain.astype(np.float64) + a[:, c].astype(np.float64) * b[:, c].astype(np.float64)).astype(np.float32)
    ah, am, al = split3(a)
    bh, bm, bl = split3(b)
    acc = np.zeros(n, np.float32)
    for c in range(C):
        for x, y in ((ah, bh), (ah, bm), (am, bh), (ah, bl), (al, bh), (am, bm)):
            acc = acc + x[:, c] * y[:, c]                 # bf16 x bf16 is exact in f32; f32 accumulate
    mag = (np.abs(a.astype(np.float64) * b.astype(np.float64))).sum(1)
    return (np.abs(chain - exact) / mag).max(), (np.abs(acc - exact) / mag).max(), (np.abs(acc.astype(np.float64) - chain) / mag).max()


def main():
    rng = np.random.default_rng(0)
    print("max over 20000 dot products of |error| / sum|terms|      (contract: 1e-5)")
    print(f"{'C':>6} {'data':>8} {'f32 chain vs exact':>20} {'split-bf16 vs exact':>20} {'split-bf16 vs chain':>20}")
    for C in (256, 512, 2048):
        for signed in (False, True):
            e = study(C, 20000, rng, signed)
            print(f"{C:6d} {'N(0,1)' if signed else 'U[0,1)':>8} {e[0]:20.2e} {e[1]:20.2e} {e[2]:20.2e}")


if __name__ == "__main__":
    main()
