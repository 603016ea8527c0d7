"""Jump-ahead polynomials of MT19937 for the parallel fold generator (nvt_util.hip).

The generator's word sequence satisfies w[k+624] = w[k+397] ^ twist(w[k], w[k+1]): a linear map T
over GF(2) on the 19937-bit state (window w[k..k+623], only the top bit of w[k] counts) with
characteristic polynomial phi of degree 19937.  With g(x) = x^J mod phi(x), the window at k + J is
    W'[m] = XOR over {i : g_i = 1} of w[k + i + m],   m = 0 .. 623
(g(T) applied to the window: T^i shifts the window by i words; Haramoto, Matsumoto, Nishimura,
Panneton, L'Ecuyer, "Efficient jump ahead for F2-linear random number generators", 2008, in the
form "evaluate g at the sequence").  This script
  1. finds phi with Berlekamp-Massey on one output bit of numpy's own MT19937,
  2. computes p_k = x^(B * 2^k) mod phi for k = 0 .. K-1 (B = 2^18 words per chunk: chunk c starts
     at word c * B, reached from the seed window by the p_k of the set bits of c),
  3. checks a jump against numpy stepped word by word,
  4. writes nvtabular_amd/csrc/nvt_mt_jump_polys.inc.
Pure Python big-integer arithmetic (polynomials over GF(2) as ints): ~1 minute."""
import os
import sys
import time

import numpy as np

N, M = 624, 397
UPPER, LOWER, MATRIX_A = 0x80000000, 0x7FFFFFFF, 0x9908B0DF
B_LOG2, K = 18, 14
DEG = 19937


def words(seed, count):
    """w[0 .. count) of numpy's legacy RandomState(seed): w[0..623] = init_genrand(seed)."""
    key = np.random.RandomState(seed).get_state()[1].astype(np.uint64)
    w = np.zeros(count, dtype=np.uint64)
    w[:N] = key
    for k in range(0, count - N):   # word by word: the reference the block forms are checked against
        y = (w[k] & UPPER) | (w[k + 1] & LOWER)
        w[k + N] = w[k + M] ^ (y >> np.uint64(1)) ^ (MATRIX_A if (int(y) & 1) else 0)
    return w


def words_fast(seed, count):
    """Same, 227 words at a time (the dependency distance of the recurrence)."""
    key = np.random.RandomState(seed).get_state()[1].astype(np.uint64)
    w = np.zeros(count + N, dtype=np.uint64)
    w[:N] = key
    k = 0
    while k + N < count:
        step = min(227, count - N - k)
        a, b, c = w[k:k + step], w[k + 1:k + 1 + step], w[k + M:k + M + step]
        y = (a & np.uint64(UPPER)) | (b & np.uint64(LOWER))
        w[k + N:k + N + step] = c ^ (y >> np.uint64(1)) ^ np.where(y & np.uint64(1), np.uint64(MATRIX_A), np.uint64(0))
        k += step
    return w[:count]


def berlekamp_massey(bits):
    """Connection polynomial C (int, bit j = c_j) and length L with s_n = sum_{j=1..L} c_j s_{n-j}."""
    C, Bp, L, m = 1, 1, 0, 1
    rev = 0   # bit j = s_{n-j}
    for n, s in enumerate(bits):
        rev = (rev << 1) | int(s)          # now bit 0 = s_n, bit j = s_{n-j}
        d = (C & rev).bit_count() & 1
        if d:
            T = C
            C ^= Bp << m
            if 2 * L <= n:
                L, Bp, m = n + 1 - L, T, 1
            else:
                m += 1
        else:
            m += 1
    return C, L


def main():
    t0 = time.time()
    w = words_fast(12345, N + 2 * DEG + 200)
    assert (w[:N + 2000] == words(12345, N + 2000)).all()
    bits = (w[N:] & np.uint64(1)).astype(np.uint8)
    C, L = berlekamp_massey(bits)
    assert L == DEG, L
    phi = 0
    for j in range(L + 1):
        if (C >> j) & 1:
            phi |= 1 << (L - j)
    assert phi >> DEG == 1
    print(f"phi: degree {L}, weight {phi.bit_count()}  ({time.time() - t0:.1f} s)")

    def mulmod(a, b):
        r = 0
        for i in reversed(range(b.bit_length())):
            r <<= 1
            if (r >> DEG) & 1:
                r ^= phi
            if (b >> i) & 1:
                r ^= a
        return r

    polys = []
    g = 2   # x
    for _ in range(B_LOG2):
        g = mulmod(g, g)
    polys.append(g)
    for _ in range(K - 1):
        polys.append(mulmod(polys[-1], polys[-1]))
    print(f"{K} polynomials x^(2^{B_LOG2} * 2^k) mod phi  ({time.time() - t0:.1f} s)")

    # ---- check: jump by B (and by 3 B = p_0 then p_1) from another seed against numpy stepped ----
    Bw = 1 << B_LOG2
    seed = 42
    ref = words_fast(seed, 3 * Bw + 2 * N)

    def jump(window_words_from, g):
        """window at the position reached by g from the word sequence starting at the window."""
        seq = window_words_from      # w[k .. k + DEG + N)
        out = np.zeros(N, dtype=np.uint64)
        for i in range(DEG):
            if (g >> i) & 1:
                out ^= seq[i:i + N]
        return out

    def forward(window, count):
        w2 = np.zeros(count, dtype=np.uint64)
        w2[:N] = window
        k = 0
        while k + N < count:
            step = min(227, count - N - k)
            a, b, c = w2[k:k + step], w2[k + 1:k + 1 + step], w2[k + M:k + M + step]
            y = (a & np.uint64(UPPER)) | (b & np.uint64(LOWER))
            w2[k + N:k + N + step] = c ^ (y >> np.uint64(1)) ^ np.where(y & np.uint64(1), np.uint64(MATRIX_A), np.uint64(0))
            k += step
        return w2

    w1 = jump(ref[:DEG + N], polys[0])
    exp = ref[Bw:Bw + N]
    assert (w1[1:] == exp[1:]).all() and (int(w1[0]) ^ int(exp[0])) & UPPER == 0, "jump by B"
    w3 = jump(forward(w1, DEG + N), polys[1])
    exp = ref[3 * Bw:3 * Bw + N]
    assert (w3[1:] == exp[1:]).all() and (int(w3[0]) ^ int(exp[0])) & UPPER == 0, "jump by 3 B"
    print(f"jumps check against numpy's sequence  ({time.time() - t0:.1f} s)")

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "nvtabular_amd", "csrc", "nvt_mt_jump_polys.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/mt_jump_polys.py -- do not edit.\n")
        f.write("// kMtJumpPoly[k] = coefficients of x^(2^%d * 2^k) mod phi(x), phi = the characteristic\n" % B_LOG2)
        f.write("// polynomial of MT19937 (degree 19937); bit i of the bit string (word i / 32, bit i %% 32).\n")
        f.write("constexpr int kMtJumpLog2 = %d;   // words per chunk = 2^%d\n" % (B_LOG2, B_LOG2))
        f.write("constexpr int kMtJumpPolys = %d;\n" % K)
        f.write("constexpr int kMtJumpWords = 624;   // 19937 coefficient bits in 624 words\n")
        f.write("__device__ const uint32_t kMtJumpPoly[kMtJumpPolys][kMtJumpWords] = {\n")
        for g in polys:
            ws = [(g >> (32 * i)) & 0xFFFFFFFF for i in range(624)]
            f.write("  {")
            for i in range(0, 624, 8):
                f.write("\n    " + ", ".join("0x%08Xu" % x for x in ws[i:i + 8]) + ",")
            f.write("\n  },\n")
        f.write("};\n")
    print("wrote", path)


if __name__ == "__main__":
    sys.exit(main())
