"""Host-side launch logic that needs no GPU."""
from applecider_amd import hipops as H


def test_split_k_never_leaves_an_empty_piece():
    """the library gives each of `split` K pieces ceil(nkt / split) tiles; a piece that starts
    beyond the last tile stores nothing and, in slab form, ac_splitk_reduce would add its unwritten slab.  The host
    picks the largest split whose last piece still owns a tile (and ac_gemm returns AC_EINVAL otherwise)."""
    assert H._exact_split(130, 16) == 15           # the advisor's example: K = 4160, M, N <= 128
    for nkt in range(1, 400):
        for want in range(1, 70):
            s = H._exact_split(nkt, want)
            per = -(-nkt // s)
            assert 1 <= s <= max(1, min(want, nkt))
            assert (s - 1) * per < nkt                # the last piece starts inside the reduction
            assert s * per >= nkt                     # the pieces cover it
    for M, N, K in [(128, 128, 4160), (512, 768, 3072), (4608, 384, 1536), (512, 384, 3072), (300, 100, 1024),
                    (128, 96, 32 * 137), (4, 4, 32 * 4099)]:
        s = H._small_grid_split(M, N, K)
        nkt = K // 32
        assert s == 1 or (s - 1) * -(-nkt // s) < nkt, (M, N, K, s)
