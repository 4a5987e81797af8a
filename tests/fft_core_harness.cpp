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

static int run_size(int logm, int radix3) {
    const int M = 1 << logm, T = pow3(radix3), N = T * M;
    const auto tw = table(logm);
    std::vector<ac_c2> tw3(2 * N / 3 + 1);
    for (int t = 0; t < (int)tw3.size(); ++t) {
        const double a = -2.0 * M_PI * t / N;
        tw3[t] = ac_c2{(float)cos(a), (float)sin(a)};
    }
    auto t3 = [&](int t) { return tw3[t]; };
    std::vector<double> x1(N), x2(N);
    srand(1234 + logm + 77 * radix3);
    for (int n = 0; n < N; ++n) {
        x1[n] = rand() / (double)RAND_MAX - 0.5;
        x2[n] = rand() / (double)RAND_MAX - 0.5;
    }
    const int pitch = seq_pitch_n(N, 8);
    std::vector<ac_c2> seq(pitch, ac_c2{0.f, 0.f});
    for (int n = 0; n < N; ++n) seq[phys(n)] = ac_c2{(float)x1[n], (float)x2[n]};
    // radix-3 stages (what fft_radix3_all does), then the power-of-two passes on every M-point piece
    for (int st = 0; st < radix3; ++st) {
        const int s3 = N / (st == 0 ? 3 : 9), pieces = st == 0 ? 1 : 3, tstride = st == 0 ? 1 : 3;
        for (int pc = 0; pc < pieces; ++pc)
            for (int j = 0; j < s3; ++j) dif3_item(seq.data() + phys(pc * 3 * s3), t3, s3, tstride, j);
    }
    for (int r = 0; r < T; ++r) forward(seq.data() + third_base(r, logm), tw, logm);
    std::vector<double> ar(N), ai(N), br(N), bi(N);
    for (int f = 0; f < N; ++f) {
        double s1r = 0, s1i = 0, s2r = 0, s2i = 0;
        for (int n = 0; n < N; ++n) {
            const double a = -2.0 * M_PI * ((long long)f * n % N) / N, c = cos(a), s = sin(a);
            s1r += x1[n] * c; s1i += x1[n] * s; s2r += x2[n] * c; s2i += x2[n] * s;
        }
        ar[f] = s1r; ai[f] = s1i; br[f] = s2r; bi[f] = s2i;
    }
    const double scale = sqrt((double)N);
    double e_unt = 0, e_inv = 0;
    int bad = 0;
    std::vector<ac_c2> y1(N / 2 + 1), y2(N / 2 + 1);
    std::vector<int> seen(N / 2 + 1, 0);
    for (int e = 0; e <= N / 2; ++e) {
        int pos, ppos, f;
        bool pair;
        half_entry(e, logm, radix3, pos, ppos, f, pair);
        if (f < 0 || f > N / 2 || seen[f]++) { bad = 1; continue; }
        ac_c2 a, b;
        untangle(seq[phys(pos)], seq[phys(ppos)], a, b);
        y1[f] = a; y2[f] = b;
        e_unt = fmax(e_unt, fmax(fmax(fabs(a[0] - ar[f]), fabs(a[1] - ai[f])), fmax(fabs(b[0] - br[f]), fabs(b[1] - bi[f]))) / scale);
    }
    std::vector<ac_c2> inv(pitch, ac_c2{NAN, NAN});
    for (int e = 0; e <= N / 2; ++e) {
        int pos, ppos, f;
        bool pair;
        half_entry(e, logm, radix3, pos, ppos, f, pair);
        ac_c2 zf, zn;
        tangle(y1[f], y2[f], zf, zn);
        inv[phys(pos)] = zf;
        if (pair) inv[phys(ppos)] = zn;
    }
    for (int r = 0; r < T; ++r) inverse(inv.data() + third_base(r, logm), tw, logm);
    for (int st = radix3 - 1; st >= 0; --st) {
        const int s3 = N / (st == 0 ? 3 : 9), pieces = st == 0 ? 1 : 3, tstride = st == 0 ? 1 : 3;
        for (int pc = 0; pc < pieces; ++pc)
            for (int j = 0; j < s3; ++j) dit3_item(inv.data() + phys(pc * 3 * s3), t3, s3, tstride, j);
    }
    for (int n = 0; n < N; ++n)
        e_inv = fmax(e_inv, fmax(fabs(inv[phys(n)][0] / N - x1[n]), fabs(inv[phys(n)][1] / N - x2[n])));
    if (!(e_inv == e_inv)) bad = 1;    // a position of the image was never written
    printf("N %4d (%s)  half spectrum %.2e  roundtrip %.2e  %s\n", N, radix3 == 2 ? "9 x 2^m" : radix3 ? "3 x 2^m" : "2^m", e_unt, e_inv, bad ? "BAD" : "ok");
    return (bad || e_unt > 2e-6 || e_inv > 2e-6) ? 1 : 0;
}

int main() {
    double worst = 0.0;
    int fails = 0;
    for (int logm = 3; logm <= 7; ++logm) fails += run_size(logm, 2);      // N = 72 ... 1152
    for (int logm = 3; logm <= 9; ++logm) fails += run_size(logm, 1);      // N = 24 ... 1536
    for (int logm = 3; logm <= 11; ++logm) fails += run_size(logm, 0);
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
        std::vector<double> zr(N), zi(N);
        for (int f = 0; f < N; ++f) {
            double s1r = 0, s1i = 0, s2r = 0, s2i = 0;
            for (int n = 0; n < N; ++n) {
                const double a = -2.0 * M_PI * ((long long)f * n % N) / N, c = cos(a), s = sin(a);
                s1r += x1[n] * c; s1i += x1[n] * s; s2r += x2[n] * c; s2i += x2[n] * s;
            }
            zr[f] = s1r - s2i; zi[f] = s1i + s2r;
        }
        double scale = sqrt((double)N), e_fft = 0, e_part = 0;
        for (int i = 0; i < N; ++i) {
            const int f = brev(i, logn);
            e_fft = fmax(e_fft, fmax(fabs(seq[phys(i)][0] - zr[f]), fabs(seq[phys(i)][1] - zi[f])) / scale);
            if (brev(partner(i), logn) != ((N - f) & (N - 1))) e_part = 1;
        }
        printf("logn %2d  fft %.2e  partner %s\n", logn, e_fft, e_part ? "BAD" : "ok");
        worst = fmax(worst, fmax(e_fft, e_part));
    }
    printf("worst %.3e fails %d\n", worst, fails);
    return (worst < 2e-6 && fails == 0) ? 0 : 1;
}
