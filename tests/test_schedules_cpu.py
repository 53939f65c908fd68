"""Host-side mirrors of index arithmetic that lives in the HIP sources - checked exhaustively on the CPU, because a schedule bug
(a tile computed twice, a tile dropped) is cheap to find here and expensive to find on the GPU.

  * attn_res2_kernel's tile schedule (opendwm_amd/csrc/attention.hip, res2_unit_of): full rounds of 16 query tiles (wave w takes
    tiles 16 r + 2 w and + 1), then the remaining rem < 16 tiles as rem / 8 + (w < rem % 8) adjacent tiles per wave."""


def res2_unit_of(r, wave, nfull, rem):
    """mirror of res2_unit_of in attention.hip: (first tile, tile count) of wave `wave` in round `r`"""
    if r < nfull:
        return r * 16 + 2 * wave, 2
    q, x = rem >> 3, rem & 7
    return nfull * 16 + wave * q + (wave if wave < x else x), q + (1 if wave < x else 0)


def test_paired_resident_attention_schedule_covers_every_query_tile_once():
    for qend in range(1, 1300):                       # (the kernel serves 64 <= L <= 608; the schedule itself holds for any length)
        nqt = (qend + 31) >> 5
        nfull = nqt >> 4
        rem = nqt - (nfull << 4)
        rounds = nfull + (1 if rem > 0 else 0)
        seen = []
        for w in range(8):
            idle = False
            for r in range(rounds):
                t0, cnt = res2_unit_of(r, w, nfull, rem)
                assert cnt in (0, 1, 2)
                assert not (idle and cnt), "a wave's units are a prefix of its rounds (the Q prefetch chain relies on it)"
                idle = cnt == 0
                seen += list(range(t0, t0 + cnt))
        assert sorted(seen) == list(range(nqt)), (qend, nqt)


def test_paired_resident_attention_schedule_simd_loads_of_the_headline_shapes():
    """waves w and w + 4 share a SIMD: the joint attention (L = 602: 19 tiles) and the dual / row-wise temporal attention
    (L = 448: 14 tiles) load the four SIMDs as evenly as whole tiles allow"""
    for nqt, want in ((19, [5, 5, 5, 4]), (14, [4, 4, 3, 3])):
        nfull, rem = nqt >> 4, nqt & 15
        loads = [0] * 4
        for w in range(8):
            for r in range(nfull + (1 if rem else 0)):
                loads[w % 4] += res2_unit_of(r, w, nfull, rem)[1]
        assert loads == want
