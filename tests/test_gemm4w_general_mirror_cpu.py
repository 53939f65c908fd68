"""Host-side mirror of the GENERAL form of the 4-wave GEMM (opendwm_amd/csrc/gemm_bf16_4w.hip, template parameter GEN; written
without a GPU at hand, DWM_GEMM4W=2): the index arithmetic that form adds to the validated kernel - restated in numpy, line by
line, and held against a direct evaluation of the implicit convolution / ragged product:

  * dwm_gemm4w_try (host): the A resource starts `a_base_rows` = the smallest tap shift before p.A, `tap_off[t]` = (shift_t -
    smallest) * row pitch in bytes, `steps_per_tap` = k_per_tap / 64;
  * the kernel: per request row `va = u32(map_row4(min(m0 + row, M - 1))) * u32(lda * 2)`, `vw = u32(min(n0 + row, N - 1)) *
    u32(K * 2)`; the scalar K walk of A (`walk_next`: + 128 bytes inside a tap, the next tap's offset at a boundary); W walks
    kt * 128 bytes; stores guarded by m < M, n < N.
The arguments are the ones the product passes (ops.PaddedGrid / ops.TimeGrid fill the row map and the tap shifts).  What this does
NOT cover: the machine code (LDS image, MFMA fragments, epilogues) - shared with the validated fast form - and the asm request
sequence; that is tests/test_unvalidated_gpu.py on a GPU."""
import numpy as np
import pytest

from opendwm_amd import _lib, ops

BM = BN = 256
BK = 64
U32 = 1 << 32


def _map_row4(rm, m):
    if rm is None:
        return m
    q, x = m // rm.rw, m % rm.rw
    i, y = q // rm.rh, q % rm.rh
    return i * rm.ipitch + y * rm.rpitch + x * (rm.xstep if rm.xstep > 0 else 1) + rm.origin


def gen4w_mirror(A, lda, W, M, N, K, rm=None, taps=None, kpt=None, front_rows=0):
    """A: flat float64 array holding `front_rows` rows in front of p.A; returns C[M, N]"""
    ntaps = len(taps) if taps else 1
    kpt = kpt if taps else K
    assert K % BK == 0 and K >= 2 * BK and kpt % BK == 0 and kpt * ntaps == K and lda >= kpt
    steps_per_tap = kpt // BK
    smin = min([0] + list(taps)) if taps else 0
    tap_off = [((t - smin) * lda * 2) % U32 for t in taps] if taps else [0]
    a_base = (front_rows + smin) * lda                               # element index of the A resource's base (may lie before p.A)
    Wf = W.reshape(-1)
    C = np.full((M, N), np.nan)
    rows = np.arange(256)
    for tm in range((M + BM - 1) // BM):
        for tn in range((N + BN - 1) // BN):
            m0, n0 = tm * BM, tn * BN
            gmr, gnr = np.minimum(m0 + rows, M - 1), np.minimum(n0 + rows, N - 1)
            va = ((_map_row4(rm, gmr) % U32) * ((lda * 2) % U32)) % U32
            vw = ((gnr % U32) * ((K * 2) % U32)) % U32
            acc = np.zeros((256, 256))
            walk_left, walk_tap, walk_a = steps_per_tap, 0, tap_off[0]
            for kt in range(K // BK):
                if kt > 0:                                           # walk_next()
                    walk_left -= 1
                    if walk_left == 0:
                        walk_tap += 1
                        walk_left = steps_per_tap
                        walk_a = tap_off[walk_tap]
                    else:
                        walk_a = (walk_a + BK * 2) % U32
                ai = a_base + (va + walk_a) // 2                     # voffset + soffset, in elements
                wi = (vw + kt * BK * 2) // 2
                assert ai.min() >= 0 and ai.max() + BK <= A.size, "every request stays inside the A allocation"
                acc += A[ai[:, None] + np.arange(BK)] @ Wf[wi[:, None] + np.arange(BK)].T
            mm, nn = min(BM, M - m0), min(BN, N - n0)
            C[m0:m0 + mm, n0:n0 + nn] = acc[:mm, :nn]
    return C


def _direct(Ap, lda, W, M, rm, taps, kpt):
    """C[m] = sum_t A[map(m) + shift_t, :kpt] . W[:, t kpt : (t + 1) kpt]^T"""
    base = _map_row4(rm, np.arange(M))
    out = 0.0
    for t, sh in enumerate(taps if taps else [0]):
        out = out + Ap[base + sh][:, :kpt] @ W[:, t * kpt:(t + 1) * kpt].T
    return out


@pytest.mark.parametrize("I,h,w,Cin,N,mode", [(2, 4, 6, 64, 192, "dense"), (3, 16, 28, 128, 320, "dense"), (7, 9, 5, 64, 264, "dense"),
                                              (2, 8, 12, 192, 256, "stride2"), (2, 8, 12, 64, 40, "stride2_sym")])
def test_general_form_mirror_implicit_conv3x3(I, h, w, Cin, N, mode):
    rng = np.random.default_rng(0)
    grid = ops.PaddedGrid(I, h, w)
    rm = _lib.RowMap2D()
    if mode == "dense":
        grid.fill(rm)
        taps, M = grid.tap_shifts(), grid.pixels
    else:
        (grid.fill_stride2 if mode == "stride2" else grid.fill_stride2_sym)(rm)
        taps, M = grid.tap_shifts_stride2(), grid.pixels // 4
    Ap = np.zeros((grid.rows + grid.w + 3, Cin))                      # (the stride-2 taps of the last pixel reach past the grid's last row)
    Ap[:grid.rows][grid.interior_index().numpy()] = rng.standard_normal((grid.pixels, Cin))
    W = rng.standard_normal((N, 9 * Cin))
    got = gen4w_mirror(Ap.reshape(-1), Cin, W, M, N, 9 * Cin, rm=rm, taps=taps, kpt=Cin)
    want = _direct(Ap, Cin, W, M, rm, taps, Cin)
    assert not np.isnan(got).any() and np.allclose(got, want, rtol=1e-10, atol=1e-10)


def test_general_form_mirror_temporal_taps_and_plain_ragged():
    rng = np.random.default_rng(1)
    tg = ops.TimeGrid(2, 5, 36)
    rm = _lib.RowMap2D()
    tg.fill(rm)
    Cin, N = 128, 328
    Ap = rng.standard_normal((tg.rows, Cin))
    W = rng.standard_normal((N, 3 * Cin))
    got = gen4w_mirror(Ap.reshape(-1), Cin, W, tg.pixels, N, 3 * Cin, rm=rm, taps=tg.tap_shifts(), kpt=Cin)
    assert np.allclose(got, _direct(Ap, Cin, W, tg.pixels, rm, tg.tap_shifts(), Cin), rtol=1e-10, atol=1e-10)
    # no map, no taps: ragged M and N, lda > K (a column slice of a wider matrix)
    M, N, K, lda = 300, 264, 192, 256
    A = rng.standard_normal((M, lda))
    W = rng.standard_normal((N, K))
    assert np.allclose(gen4w_mirror(A.reshape(-1), lda, W, M, N, K), A[:, :K] @ W.T, rtol=1e-10, atol=1e-10)
