// CPU harness of applecider_amd/csrc/ac_fft_core.h: runs the pass functions work item by work item (what the
// 512 threads of a workgroup do between two barriers) and checks them against a direct fp64 DFT.
// Built and run by tests/test_fft_core.py with the host compiler; no GPU involved.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../applecider_amd/csrc/ac_fft_core.h"

using namespace acfft;

static std::vector<ac_c2> table(int logn) {
    const int N = 1 << logn;
    std::vector<ac_c2> tw(N / 2);
    for (int t = 0; t < N / 2; ++t) {
        const double a = -2.0 * M_PI * t / N;
        tw[t] = ac_c2{(float)cos(a), (float)sin(a)};
    }
    return tw;
}

template <int R>
static void run_dif(ac_c2 *seq, const std::vector<ac_c2> &tw, int logn, int s0) {
    for (int u = 0; u < (1 << (logn - R)); ++u) dif_pass<R>(seq, [&](int e, int j) { return tw[j << e]; }, logn, s0, u);
}
template <int R>
static void run_dit(ac_c2 *seq, const std::vector<ac_c2> &tw, int logn, int lh0) {
    for (int u = 0; u < (1 << (logn - R)); ++u) dit_pass<R>(seq, [&](int e, int j) { return tw[j << e]; }, logn, lh0, u);
}
static void forward(ac_c2 *seq, const std::vector<ac_c2> &tw, int logn) {
    int s = 0;
    const int r0 = first_r(logn);
    if (r0 == 1) run_dif<1>(seq, tw, logn, 0);
    if (r0 == 2) run_dif<2>(seq, tw, logn, 0);
    for (s = r0; s < logn; s += 3) run_dif<3>(seq, tw, logn, s);
}
static void inverse(ac_c2 *seq, const std::vector<ac_c2> &tw, int logn) {
    const int r0 = first_r(logn);
    int lh = 0;
    for (; lh + 3 <= logn - r0; lh += 3) run_dit<3>(seq, tw, logn, lh);
    if (r0 == 1) run_dit<1>(seq, tw, logn, lh);
    if (r0 == 2) run_dit<2>(seq, tw, logn, lh);
}

int main() {
    double worst = 0.0;
    for (int logn = 3; logn <= 11; ++logn) {
        const int N = 1 << logn;
        const auto tw = table(logn);
        std::vector<double> x1(N), x2(N);
        srand(1234 + logn);
        for (int n = 0; n < N; ++n) {
            x1[n] = rand() / (double)RAND_MAX - 0.5;
            x2[n] = rand() / (double)RAND_MAX - 0.5;
        }
        std::vector<ac_c2> seq(seq_pitch(logn), ac_c2{0.f, 0.f});
        for (int n = 0; n < N; ++n) seq[phys(n)] = ac_c2{(float)x1[n], (float)x2[n]};
        forward(seq.data(), tw, logn);
        // direct DFT of z, x1, x2
        std::vector<double> zr(N), zi(N), ar(N), ai(N), br(N), bi(N);
        for (int f = 0; f < N; ++f) {
            double s1r = 0, s1i = 0, s2r = 0, s2i = 0;
            for (int n = 0; n < N; ++n) {
                const double a = -2.0 * M_PI * ((long long)f * n % N) / N, c = cos(a), s = sin(a);
                s1r += x1[n] * c; s1i += x1[n] * s; s2r += x2[n] * c; s2i += x2[n] * s;
            }
            ar[f] = s1r; ai[f] = s1i; br[f] = s2r; bi[f] = s2i;
            zr[f] = s1r - s2i; zi[f] = s1i + s2r;
        }
        double scale = sqrt((double)N), e_fft = 0, e_part = 0, e_unt = 0, e_inv = 0;
        for (int i = 0; i < N; ++i) {
            const int f = brev(i, logn);
            e_fft = fmax(e_fft, fmax(fabs(seq[phys(i)][0] - zr[f]), fabs(seq[phys(i)][1] - zi[f])) / scale);
            if (brev(partner(i), logn) != ((N - f) & (N - 1))) e_part = 1;
        }
        // half spectra of the two real sequences from the positions (even i, and i = 1 for f = N / 2)
        std::vector<ac_c2> y1(N / 2 + 1), y2(N / 2 + 1);
        for (int idx = 0; idx <= N / 2; ++idx) {
            const int i = idx == N / 2 ? 1 : 2 * idx, f = brev(i, logn);
            ac_c2 a, b;
            untangle(seq[phys(i)], seq[phys(partner(i))], a, b);
            y1[f] = a; y2[f] = b;
            e_unt = fmax(e_unt, fmax(fmax(fabs(a[0] - ar[f]), fabs(a[1] - ai[f])), fmax(fabs(b[0] - br[f]), fabs(b[1] - bi[f]))) / scale);
        }
        // back: tangle into the bit-reversed image, inverse passes, compare with N * x
        std::vector<ac_c2> inv(seq_pitch(logn), ac_c2{0.f, 0.f});
        for (int idx = 0; idx <= N / 2; ++idx) {
            const int i = idx == N / 2 ? 1 : 2 * idx, f = brev(i, logn);
            ac_c2 zf, zn;
            tangle(y1[f], y2[f], zf, zn);
            inv[phys(i)] = zf;
            if (i > 1) inv[phys(partner(i))] = zn;
        }
        inverse(inv.data(), tw, logn);
        for (int n = 0; n < N; ++n)
            e_inv = fmax(e_inv, fmax(fabs(inv[phys(n)][0] / N - x1[n]), fabs(inv[phys(n)][1] / N - x2[n])));
        printf("logn %2d  fft %.2e  partner %s  untangle %.2e  roundtrip %.2e\n", logn, e_fft, e_part ? "BAD" : "ok", e_unt, e_inv);
        worst = fmax(worst, fmax(fmax(e_fft, e_unt), fmax(e_inv, e_part)));
    }
    printf("worst %.3e\n", worst);
    return worst < 2e-6 ? 0 : 1;
}
