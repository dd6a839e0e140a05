"""The register-staged FFT of csrc/mrstft.hip addresses its exchange buffer and twiddle table as (lane part) + (compile-time
constant), so that every LDS access is one instruction with an immediate offset.  This script checks those separable forms
against the plain index algebra (position -> padded slot, butterfly -> twiddle step) for every lane and index, N = 512 / 1024 / 2048.
    python tools/probe/check_fft_addr.py
"""


def out_pos4(P, j, r):
    Ns = 1 << (2 * P)
    k = j & (Ns - 1)
    return ((j - k) << 2) + k + r * Ns


def pad1(q):
    return q + (q >> 4)


def pad2(q):
    return q + 16 * (q >> 8)


def step(N, P, j):
    Ns = 1 << (2 * P)
    return (j & (Ns - 1)) * (N // (Ns * 4))


for N in (512, 1024, 2048):
    L = 32 if N == 512 else 64
    E = N // L
    NB = E // 4
    NBQ = NB // 4
    for a in range(L):
        k = a & 15
        # ---- exchange after passes (0, 1)
        for r in range(4):
            for be in range(NBQ):
                jn = out_pos4(0, a, r) + 4 * L * be
                assert step(N, 1, jn) == r * (N // 16)                                   # pass-1 twiddles: lane independent
                for r2 in range(4):
                    assert pad1(out_pos4(1, jn, r2)) == 17 * a + (17 * L * be + r + 4 * r2)
        for b in range(NB):
            for c in range(4):
                q = a + L * b + (N // 4) * c
                assert pad1(q) == (a + (a >> 4)) + (L * b + (N // 4) * c + ((L * b + (N // 4) * c) >> 4))
                assert pad2(q) == a + (L * b + (N // 4) * c + 16 * ((L * b + (N // 4) * c) >> 8)), (N, a, b, c)
        # ---- passes (2, 3)
        for b in range(NB):
            assert step(N, 2, a + L * b) == k * (N // 64)                                # pass-2 twiddles: lane part only
        for r in range(4):
            for be in range(NBQ):
                jn = out_pos4(2, a, r) + 4 * L * be
                assert step(N, 3, jn) == k * (N // 256) + r * (N // 16)
                for r2 in range(4):
                    assert pad2(out_pos4(3, jn, r2)) == (272 * (a >> 4) + k) + (17 * L * be + 16 * r + 64 * r2), (N, a, r, be, r2)
        # ---- last passes
        if N == 1024:
            for b in range(NB):
                assert step(N, 4, a + L * b) == a + 64 * b
        if N == 2048:
            for b in range(NB):
                assert step(N, 4, a + L * b) == 2 * a + 128 * (b & 3)
    print(N, "ok")
