// In-LDS FFT core of the frequency-domain convolutions (ac_fft.hip).  Host-compilable on purpose: the CPU test
// harness (tests/fft_core_harness.cpp) runs exactly these functions, work item by work item, against a direct DFT,
// so that the index algebra is proven before a kernel ever runs on the GPU.
//
// One complex sequence of N = 2^logn points lives in a padded image: element i at phys(i) = i + (i >> 3) (one spare
// 8-byte slot per 8 elements), sequences seq_pitch(logn, nseq) slots apart.  With that padding every access
// pattern below is (near) bank-conflict free for 8-byte LDS accesses:
//   * passes with a large stride: the lanes of a wave walk consecutive elements;
//   * the pass with stride 8: element blk*64 + i0 + 8m -> slot 72 blk + i0 + 9m: 4 blocks x 8 i0 = 32 distinct slots;
//   * the pass with stride 1 (a lane owns 8 consecutive elements): slot 9u + m, 9 odd: 32 distinct slots.
// Forward = decimation in frequency (natural order in, bit-reversed order out), inverse = decimation in time
// (bit-reversed in, natural out): neither direction ever permutes.  A pass performs R <= 3 radix-2 stages on 2^R
// points held in registers, so a 2048-point transform is 4 LDS round trips instead of 11.  Per pass and work item only
// R twiddles come from the table (w, w^2, w^4 of the item's base index); the other factors are 8th roots of unity.
#pragma once
#ifdef __HIPCC__
#define AC_FFT_HD __host__ __device__ __forceinline__
#else
#define AC_FFT_HD inline
#endif

typedef float ac_c2 __attribute__((ext_vector_type(2)));   // (re, im)

namespace acfft {

AC_FFT_HD int phys(int i) { return i + (i >> 3); }
// slots between the sequences of a workgroup: = 4 mod 32 when 8 sequences share it (lanes = 8 sequences x 4
// consecutive elements), = 8 mod 32 for 4 sequences (x 8 elements), = 1 mod 32 for 32 sequences: distinct banks
AC_FFT_HD int seq_pitch_n(int n, int nseq) {
    const int body = n + (n >> 3);
    return ((body + 31) & ~31) + (nseq > 8 ? 1 : nseq == 4 ? 8 : 4);
}
AC_FFT_HD int seq_pitch(int logn, int nseq = 8) { return seq_pitch_n(1 << logn, nseq); }

AC_FFT_HD ac_c2 cmul(ac_c2 a, ac_c2 w) { return ac_c2{a[0] * w[0] - a[1] * w[1], a[0] * w[1] + a[1] * w[0]}; }
AC_FFT_HD ac_c2 conj(ac_c2 a) { return ac_c2{a[0], -a[1]}; }

// t * exp(-+ 2 pi i K / 8), K in 0..3 (sign -: forward, SIGN_PLUS: inverse)
template <bool SIGN_PLUS>
AC_FFT_HD ac_c2 rot8(ac_c2 t, int K) {
    const float h = 0.70710678118654752440f;
    switch (K & 3) {
        case 0: return t;
        case 1: return SIGN_PLUS ? ac_c2{(t[0] - t[1]) * h, (t[0] + t[1]) * h} : ac_c2{(t[0] + t[1]) * h, (t[1] - t[0]) * h};
        case 2: return SIGN_PLUS ? ac_c2{-t[1], t[0]} : ac_c2{t[1], -t[0]};
        default: return SIGN_PLUS ? ac_c2{-(t[0] + t[1]) * h, (t[0] - t[1]) * h} : ac_c2{(t[1] - t[0]) * h, -(t[0] + t[1]) * h};
    }
}

// A pass = R radix-2 stages on P = 2^R points of one work item u < N >> R, in four pieces (address, load, twiddles +
// butterflies, store) so that a kernel thread can keep several work items in flight: all their LDS reads and twiddle
// loads are issued before the first butterfly.
// tw(e, j) = exp(-2 pi i j / (N >> e)), j < N >> (e + 1): the table of the transform of size N >> e.  The kernels keep
// all levels back to back (level e at element N - (N >> e)), so that the lanes of a wave read consecutive entries.
struct PassItem {
    int base, lstride, i0;    // element index of point 0, log2 of the distance between points, twiddle index
};
// forward (decimation in frequency), stages s0 .. s0 + R - 1 (block size N >> s0)
template <int R>
AC_FFT_HD PassItem dif_item(int logn, int s0, int u) {
    const int lq = logn - s0 - R;
    const int blk = u >> lq, i0 = u & ((1 << lq) - 1);
    return PassItem{(blk << (logn - s0)) + i0, lq, i0};
}
// inverse (decimation in time), halves 1 << lh0 .. 1 << (lh0 + R - 1)
template <int R>
AC_FFT_HD PassItem dit_item(int lh0, int u) {
    const int blk = u >> lh0, i0 = u & ((1 << lh0) - 1);
    return PassItem{(blk << (lh0 + R)) + i0, lh0, i0};
}
template <int R>
AC_FFT_HD void pass_load(const ac_c2 *seq, const PassItem &it, ac_c2 (&v)[1 << R]) {
#pragma unroll
    for (int m = 0; m < (1 << R); ++m) v[m] = seq[phys(it.base + (m << it.lstride))];
}
template <int R>
AC_FFT_HD void pass_store(ac_c2 *seq, const PassItem &it, const ac_c2 (&v)[1 << R]) {
#pragma unroll
    for (int m = 0; m < (1 << R); ++m) seq[phys(it.base + (m << it.lstride))] = v[m];
}
template <int R, typename TW>
AC_FFT_HD void dif_twiddles(TW tw, int s0, int i0, ac_c2 (&w)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) w[r] = tw(s0 + r, i0);
}
template <int R, typename TW>
AC_FFT_HD void dit_twiddles(TW tw, int logn, int lh0, int i0, ac_c2 (&w)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) w[r] = conj(tw(logn - 1 - lh0 - r, i0));
}
template <int R>
AC_FFT_HD void dif_butterflies(ac_c2 (&v)[1 << R], const ac_c2 (&w)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int d = 1 << (R - 1 - r);
#pragma unroll
        for (int m = 0; m < (1 << R); ++m) {
            if (m & d) continue;
            const ac_c2 a = v[m], b = v[m + d];
            v[m] = a + b;
            v[m + d] = cmul(rot8<false>(a - b, (m & (d - 1)) * (4 / d)), w[r]);
        }
    }
}
template <int R>
AC_FFT_HD void dit_butterflies(ac_c2 (&v)[1 << R], const ac_c2 (&w)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int d = 1 << r;
#pragma unroll
        for (int m = 0; m < (1 << R); ++m) {
            if (m & d) continue;
            const ac_c2 a = v[m], b = rot8<true>(cmul(v[m + d], w[r]), (m & (d - 1)) * (4 / d));
            v[m] = a + b;
            v[m + d] = a - b;
        }
    }
}
template <int R, typename TW>
AC_FFT_HD void dif_pass(ac_c2 *seq, TW tw, int logn, int s0, int u) {
    const PassItem it = dif_item<R>(logn, s0, u);
    ac_c2 v[1 << R], w[R];
    pass_load<R>(seq, it, v);
    dif_twiddles<R>(tw, s0, it.i0, w);
    dif_butterflies<R>(v, w);
    pass_store<R>(seq, it, v);
}
template <int R, typename TW>
AC_FFT_HD void dit_pass(ac_c2 *seq, TW tw, int logn, int lh0, int u) {
    const PassItem it = dit_item<R>(lh0, u);
    ac_c2 v[1 << R], w[R];
    pass_load<R>(seq, it, v);
    dit_twiddles<R>(tw, logn, lh0, it.i0, w);
    dit_butterflies<R>(v, w);
    pass_store<R>(seq, it, v);
}

// Pass plan: the first DIF pass takes logn % 3 stages (none when 0), every other pass 3; the inverse mirrors it
// (its LAST pass takes logn % 3).  Both therefore meet the stride-1 data with an R = 3 pass.
AC_FFT_HD int first_r(int logn) { return logn % 3; }

// position in the bit-reversed image of the conjugate partner N - f of the frequency f = brev(i)
AC_FFT_HD int partner(int i) {
    if (i <= 1) return i;
    const int k = 31 - __builtin_clz((unsigned)i);   // i in [2^k, 2^(k+1)): the partner is its mirror image there
    return 3 * (1 << k) - 1 - i;
}
AC_FFT_HD int brev(int i, int logn) {
    unsigned v = (unsigned)i;
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    v = (v >> 16) | (v << 16);
    return (int)(v >> (32 - logn));
}

// ---------------------------------------------------------------------------------------------------------------
// N = T M, T = 3^a (a = 1, 2), M = 2^logm >= 8 (a 'same' convolution rarely needs a power of two: stage 2's k = 251
// needs 1149 points — 1152 = 9 * 128 instead of 2048 is 44 % less of everything).  A radix-3 stage splits a sequence
// of length 3 S into thirds with X[3 q + r] = FFT_S( (x[j] + w^r x[j + S] + w^2r x[j + 2S]) W_3S^(r j) )[q],
// w = exp(-2 pi i / 3); a third lives at the logical positions r S .. r S + S - 1 (phys(r S + i) = phys(r S) + phys(i):
// its image is an ordinary S-point image), and is split again (a = 2) or transformed in place by the power-of-two
// passes.  After both stages the M-point piece at slot 3 r1 + r2 holds the frequencies 9 q + 3 r2 + r1.
// tw3(t) = exp(-2 pi i t / N), t < 2 N / 3.
// ---------------------------------------------------------------------------------------------------------------
AC_FFT_HD int pow3(int a) { return a == 2 ? 9 : (a == 1 ? 3 : 1); }
AC_FFT_HD int third_base(int r, int logm) { return r * ((1 << logm) + (1 << (logm - 3))); }   // = phys(r << logm)

// one radix-3 butterfly of a stage whose thirds are `s3` elements long (s3 % 8 == 0), at the image `seq` of that
// 3 * s3 point sequence; twiddles W_(3 s3)^j = tw3(j * tstride)
template <typename TW3>
AC_FFT_HD void dif3_item(ac_c2 *seq, TW3 tw3, int s3, int tstride, int j) {
    const int sb = phys(s3), pj = phys(j);
    const ac_c2 a = seq[pj], b = seq[sb + pj], c = seq[2 * sb + pj];
    const ac_c2 t1 = b + c, t2 = a - t1 * 0.5f, d = (b - c) * 0.86602540378443864676f;
    const ac_c2 rot = {d[1], -d[0]};                       // -i d
    seq[pj] = a + t1;
    seq[sb + pj] = cmul(t2 + rot, tw3(j * tstride));
    seq[2 * sb + pj] = cmul(t2 - rot, tw3(2 * j * tstride));
}
template <typename TW3>
AC_FFT_HD void dit3_item(ac_c2 *seq, TW3 tw3, int s3, int tstride, int j) {
    const int sb = phys(s3), pj = phys(j);
    const ac_c2 u0 = seq[pj], u1 = cmul(seq[sb + pj], conj(tw3(j * tstride))), u2 = cmul(seq[2 * sb + pj], conj(tw3(2 * j * tstride)));
    const ac_c2 t1 = u1 + u2, t2 = u0 - t1 * 0.5f, d = (u1 - u2) * 0.86602540378443864676f;
    const ac_c2 rot = {-d[1], d[0]};                       // +i d
    seq[pj] = u0 + t1;
    seq[sb + pj] = t2 + rot;
    seq[2 * sb + pj] = t2 - rot;
}

// The e-th entry (e <= N / 2) of the half spectrum of a real sequence: its logical position, the position of its
// conjugate partner N - f, the frequency f itself, and whether the partner is another position (f != 0, N / 2).
// Power of two: the even positions are the frequencies below N / 2, position 1 is N / 2.  N = T M: the residue 0 piece
// by the same rule (M / 2 + 1 entries), then for every residue s = 1 .. T - 1 the even positions of its piece; the
// partner of (s, i) is (T - s, M - 1 - i).  Residue s sits at slot s (T = 3) or 3 (s % 3) + s / 3 (T = 9).
AC_FFT_HD void half_entry(int e, int logm, int radix3, int &pos, int &ppos, int &f, bool &pair) {
    const int M = 1 << logm, hm = M >> 1, T = pow3(radix3);
    if (e <= hm) {
        pos = e == hm ? 1 : 2 * e;
        ppos = partner(pos);
        pair = pos > 1;
        f = T * brev(pos, logm);
        return;
    }
    const int e2 = e - (hm + 1), s = 1 + e2 / hm, i = 2 * (e2 - (s - 1) * hm), sp = T - s;
    const int slot = T == 9 ? 3 * (s % 3) + s / 3 : s, pslot = T == 9 ? 3 * (sp % 3) + sp / 3 : sp;
    pos = slot * M + i;
    ppos = pslot * M + (M - 1 - i);
    pair = true;
    f = T * brev(i, logm) + s;
}

// Two real sequences travel as one complex one, z = x1 + i x2.  From Z[f] and Z[N - f]:
AC_FFT_HD void untangle(ac_c2 zf, ac_c2 zn, ac_c2 &x1, ac_c2 &x2) {
    const ac_c2 zc = conj(zn);
    x1 = (zf + zc) * 0.5f;
    const ac_c2 d = (zf - zc) * 0.5f;                 // X2 = d / i
    x2 = ac_c2{d[1], -d[0]};
}
// and back: Z[f] = Y1[f] + i Y2[f], Z[N - f] = conj(Y1[f]) + i conj(Y2[f])
AC_FFT_HD void tangle(ac_c2 y1, ac_c2 y2, ac_c2 &zf, ac_c2 &zn) {
    zf = ac_c2{y1[0] - y2[1], y1[1] + y2[0]};
    zn = ac_c2{y1[0] + y2[1], y2[0] - y1[1]};
}

}  // namespace acfft
