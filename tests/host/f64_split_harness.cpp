// Host emulation of the float64 first-pass scheme for 64x64 and 128x128 windows: runs the per-thread functions of
// torchpiv_amd/csrc/xcorr_f64_split.hpp for all threads of a workgroup, phase by phase (a barrier on the device =
// the end of a loop here), on windows read from stdin and prints the correlation maps (corr - min + 1e-7, fftshift
// layout) and the 8-double records.  Built with g++ by tests/test_host_logic.py.
//   argv[1]: window edge W (64 or 128)
//   stdin : int32 n_windows, then per window W*W bytes of frame a and W*W bytes of frame b (row-major)
//   stdout: per window W*W doubles (map) + 8 doubles (record)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../torchpiv_amd/csrc/xcorr_f64_split.hpp"

using namespace tpiv;
using namespace tpiv::f64s;

template <int W>
int run() {
    using S = Split<W>;
    constexpr int M = S::M, PL = S::PL, NDW = S::NDW, NT = S::NT;
    struct Thread {
        cd x[M];
        cd u[M];
        cd t[M];
        cd Y[M + 1];
        double c[M];
        double mre[M], mim[M];
        uint32_t da[NDW], db[NDW];
    };
    int32_t n = 0;
    if (fread(&n, 4, 1, stdin) != 1) return 1;
    std::vector<uint8_t> a(W * W), b(W * W);
    std::vector<double> plane(W * PL);
    std::vector<Thread> T(NT);
    auto line = [](int t) { return t % W; };
    auto half = [](int t) { return t / W; };
    for (int w = 0; w < n; ++w) {
        if (fread(a.data(), 1, W * W, stdin) != (size_t)(W * W) || fread(b.data(), 1, W * W, stdin) != (size_t)(W * W)) return 2;
        unsigned long long sa = 0, sb = 0;
        for (int i = 0; i < W * W; ++i) sa += a[i], sb += b[i];
        const bool dead = sa == 0 || sb == 0;
        const double map_scale = dead ? 0.0 : ((double)(W * W) * 0.25) / ((double)sa * (double)sb);
        // R
        for (int t = 0; t < NT; ++t) {
            const int y = line(t), h = half(t);
            for (int q = 0; q < NDW; ++q) {
                const uint8_t* pa = &a[y * W + 4 * q];
                const uint8_t* pb = &b[y * W + 4 * q];
                T[t].da[q] = pa[0] | (pa[1] << 8) | (pa[2] << 16) | ((uint32_t)pa[3] << 24);
                T[t].db[q] = pb[0] | (pb[1] << 8) | (pb[2] << 16) | ((uint32_t)pb[3] << 24);
            }
            S::rows_forward(T[t].da, T[t].db, h, T[t].x);
        }
        // T1, one component at a time; column stages: k = line, g = 1 - half
        for (int t = 0; t < NT; ++t) S::template t1_write<0>(T[t].x, line(t), half(t), plane.data());
        for (int t = 0; t < NT; ++t) S::template t1_read<0>(T[t].u, line(t), 1 - half(t), plane.data());
        for (int t = 0; t < NT; ++t) S::template t1_write<1>(T[t].x, line(t), half(t), plane.data());
        for (int t = 0; t < NT; ++t) S::template t1_read<1>(T[t].u, line(t), 1 - half(t), plane.data());
        // C
        for (int t = 0; t < NT; ++t) S::cols_forward(T[t].u, 1 - half(t));
        // X
        if (W == 64) {
            // partner = lane (64 - k) % 64 of the same wave; snapshot first (the device shuffles read the old values)
            std::vector<Thread> Sn = T;
            for (int t = 0; t < NT; ++t) {
                const int k = line(t), hv = half(t), g = 1 - hv;
                const int partner = ((W - k) % W) + W * hv;
                auto sh = [&](double, int reg, int comp, int pt) { return comp ? Sn[pt].u[reg].y : Sn[pt].u[reg].x; };
                if (g == 0) S::template cross_spectrum_g<0>(T[t].u, partner, sh);
                else S::template cross_spectrum_g<1>(T[t].u, partner, sh);
            }
        } else {
            for (int t = 0; t < NT; ++t) S::template cross_write<0>(T[t].u, line(t), 1 - half(t), plane.data());
            for (int t = 0; t < NT; ++t) S::cross_read(T[t].mre, line(t), 1 - half(t), plane.data());
            for (int t = 0; t < NT; ++t) S::template cross_write<1>(T[t].u, line(t), 1 - half(t), plane.data());
            for (int t = 0; t < NT; ++t) S::cross_read(T[t].mim, line(t), 1 - half(t), plane.data());
            for (int t = 0; t < NT; ++t) S::cross_finish(T[t].u, T[t].mre, T[t].mim);
        }
        // Ci
        for (int t = 0; t < NT; ++t) S::cols_inverse(T[t].u, 1 - half(t), T[t].t);
        // T2
        for (int t = 0; t < NT; ++t) S::template t2_write<0>(T[t].t, line(t), 1 - half(t), plane.data());
        for (int t = 0; t < NT; ++t) S::template t2_read<0>(T[t].Y, line(t), plane.data());
        for (int t = 0; t < NT; ++t) S::template t2_write<1>(T[t].t, line(t), 1 - half(t), plane.data());
        for (int t = 0; t < NT; ++t) S::template t2_read<1>(T[t].Y, line(t), plane.data());
        // Ri
        for (int t = 0; t < NT; ++t) S::rows_inverse(T[t].Y, half(t), T[t].c);
        // P
        double cmin = 1.7e308, graw = -1.7e308;
        std::vector<double> rraw(NT);
        for (int t = 0; t < NT; ++t) {
            double mn;
            S::peak_local_minmax(T[t].c, mn, rraw[t]);
            cmin = mn < cmin ? mn : cmin;
            graw = rraw[t] > graw ? rraw[t] : graw;
        }
        const double gmax = peak_shifted(graw, cmin, map_scale);
        std::vector<double> rmax(NT);
        for (int t = 0; t < NT; ++t) {
            S::peak_shift_and_write(T[t].c, cmin, map_scale, line(t), half(t), plane.data());
            rmax[t] = peak_shifted(rraw[t], cmin, map_scale);
        }
        int ywin = W - 1;
        for (int t = 0; t < NT; ++t) {
            const int fy = (line(t) + W / 2) & (W - 1);
            if (rmax[t] == gmax && fy < ywin) ywin = fy;
        }
        int xwin = W - 1;
        for (int x = W - 1; x >= 0; --x)
            if (plane[ywin * PL + x] == gmax) xwin = x;
        const int m = ywin * W + xwin;
        double sv = -1.0;
        for (int t = 0; t < NT; ++t) {
            const double s = S::peak_second_local(T[t].c, line(t), half(t), m, 3);
            sv = s > sv ? s : sv;
        }
        std::vector<double> map(W * W);
        for (int y = 0; y < W; ++y)
            for (int x = 0; x < W; ++x) map[y * W + x] = plane[y * PL + x];
        fwrite(map.data(), 8, W * W, stdout);
        double rec[8];
        for (int s = 0; s < 8; ++s) rec[s] = S::peak_record_slot(s, m, sv, dead, plane.data());
        fwrite(rec, 8, 8, stdout);
    }
    return 0;
}

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 64;
    return W == 128 ? run<128>() : run<64>();
}
