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


def _hash32(seed, idx):
    """ac_hash32 of csrc/ac_common.h in numpy (uint32 wrap-around arithmetic)."""
    import numpy as np
    idx = np.asarray(idx, dtype=np.uint64)
    m = np.uint64(0xFFFFFFFF)
    h = ((idx & m) * np.uint64(0x9E3779B1) + (idx >> np.uint64(32)) * np.uint64(0x85EBCA77)) & m
    h ^= np.uint64(seed & 0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & m
    h = (h + np.uint64((seed >> 32) & 0xFFFFFFFF)) & m
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & m
    h ^= h >> np.uint64(16)
    return h


def test_attention_dropout_scheme_statistics():
    """ac_att_keep (csrc/ac_common.h, round 4): keep(q, k) = mix(word_q ^ word_k) >= p 2^32 with one 32-bit multiply
    per score element.  Restated in numpy: the keep rate is 1 - p, rows and columns of the mask are uncorrelated (also
    between neighbouring queries, whose words differ by a constant XOR), and the diagonal is not special."""
    import numpy as np
    m = np.uint64(0xFFFFFFFF)
    for seed, p in ((12345, 0.4), (0x9E3779B97F4A7C15, 0.1), (7, 0.5)):
        T, heads = 129, 64
        keep = np.empty((heads, T, T), dtype=bool)
        for bh in range(heads):
            tok = np.arange(T, dtype=np.uint64) + np.uint64(bh * T)
            wq = _hash32(seed, tok)
            wk = _hash32((seed + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF, tok)
            h = wq[:, None] ^ wk[None, :]
            h ^= h >> np.uint64(16)
            h = (h * np.uint64(0x85EBCA6B)) & m
            h ^= h >> np.uint64(13)
            keep[bh] = h >= np.uint64(int(p * 4294967296.0))
        n = keep.size
        rate = keep.mean()
        assert abs(rate - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5 + 1e-4, (seed, p, rate)
        x = keep.astype(np.float64) - (1 - p)
        var = p * (1 - p)
        # neighbouring queries (rows), neighbouring keys (columns), diagonal, and the same (q, k) in the next head
        for name, c in (("rows", (x[:, :-1] * x[:, 1:]).mean() / var), ("cols", (x[:, :, :-1] * x[:, :, 1:]).mean() / var),
                        ("heads", (x[:-1] * x[1:]).mean() / var)):
            assert abs(c) < 5.0 / (n ** 0.5), (seed, p, name, c)
        diag = keep[:, np.arange(T), np.arange(T)].mean()
        assert abs(diag - (1 - p)) < 5 * (var / (heads * T)) ** 0.5, (seed, p, diag)
        # every row / column keeps about (1 - p) T entries: no dead or fully kept rows
        assert keep.sum(2).min() > (1 - p) * T - 6 * (var * T) ** 0.5 and keep.sum(1).min() > (1 - p) * T - 6 * (var * T) ** 0.5
