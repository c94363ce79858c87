"""What a bf16-split product would cost in accuracy (CPU emulation, numpy): dot products of length K with fp32 accumulation, operands split into
2 or 3 bf16 pieces (truncation-free: round-to-nearest-even at every split), compared with a float64 result.
  x3: a_hi*b_hi + a_hi*b_lo + a_lo*b_hi           (2-way split, lo*lo dropped)
  x6: 3-way split, the six products of weight >= 2^-16
  h3: two FP16 pieces per operand (h = fp16(x), m = fp16((x - h) * 2^11) / 2^11: 22 bits), the three products h*h + h*m + m*h — the form the
      opt-in kernels ship (conv_sp3.hip, conv_pw.hip SPLIT 2); the weights are scaled to max |w| in [2^14, 2^15) first, the activations by 2^-4
Context (DESIGN section 3 "Split products"): the fp32 MFMA (v_mfma_f32_32x32x2_f32) runs 32 MAC/cycle/SIMD, v_mfma_f32_32x32x16_bf16 512 — a 6-product split would
still be 2.7x the fp32 instruction rate.  This file only answers whether such a path could meet the parity bar; nothing in the package uses it."""
import numpy as np

def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)

def split(x, n):
    parts, rest = [], x.astype(np.float32)
    for _ in range(n):
        p = bf16(rest); parts.append(p); rest = (rest - p).astype(np.float32)
    return parts

def split_f16(x, scale):            # two fp16 pieces of x * scale; the residual is scaled by 2^11 before it is rounded (it would be subnormal otherwise)
    xs = (x.astype(np.float32) * np.float32(scale)).astype(np.float32)
    h = xs.astype(np.float16).astype(np.float32)
    m = ((xs - h) * np.float32(2048.0)).astype(np.float16).astype(np.float32) / np.float32(2048.0)
    return h, m

def dot32(a, b):                      # fp32 accumulation, sequential over K in chunks of 16 (an MFMA's K) then across chunks
    acc = np.zeros(a.shape[0], np.float32)
    for k in range(0, a.shape[1], 16):
        acc = (acc + (a[:, k:k + 16].astype(np.float32) * b[:, k:k + 16].astype(np.float32)).sum(1, dtype=np.float32)).astype(np.float32)
    return acc

rng = np.random.default_rng(0)
for K in (1152, 2304, 9216):
    a = rng.standard_normal((4096, K)).astype(np.float32); b = (rng.standard_normal((4096, K)) * (2.0 / K) ** 0.5).astype(np.float32)
    ref = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
    e32 = np.abs(dot32(a, b) - ref)
    a2, b2 = split(a, 2), split(b, 2)
    x3 = dot32(a2[0], b2[0]) + dot32(a2[0], b2[1]) + dot32(a2[1], b2[0])
    a3, b3 = split(a, 3), split(b, 3)
    x6 = sum(dot32(a3[i], b3[j]) for i in range(3) for j in range(3) if i + j <= 2)
    sw = 2.0 ** (14 - np.floor(np.log2(np.abs(b).max())))
    (ah, am), (bh, bm) = split_f16(a, 2.0 ** -4), split_f16(b, sw)
    h3 = (dot32(am, bh) + dot32(ah, bm) + dot32(ah, bh)) * np.float32(16.0 / sw)
    print("K %5d  |ref| rms %.2f   fp32: max %.1e rms %.1e   bf16x3: max %.1e rms %.1e   bf16x6: max %.1e rms %.1e   fp16x3: max %.1e rms %.1e" % (
        K, float(np.sqrt((ref ** 2).mean())), e32.max(), np.sqrt((e32 ** 2).mean()), np.abs(x3 - ref).max(), np.sqrt(((x3 - ref) ** 2).mean()),
        np.abs(x6 - ref).max(), np.sqrt(((x6 - ref) ** 2).mean()), np.abs(h3 - ref).max(), np.sqrt(((h3 - ref) ** 2).mean())))
