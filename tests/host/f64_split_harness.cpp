// Host emulation of the float64 first-pass scheme for 64x64 windows: runs the per-thread functions of
// torchpiv_amd/csrc/xcorr_f64_split.hpp for all 128 "threads" of a workgroup, phase by phase (a barrier on
// the device = the end of a loop here), on windows read from stdin and prints the correlation maps
// (corr - min + 1e-7, fftshift layout) and the 8-double records.  Built with g++ by tests/test_host_logic.py.
//   stdin : int32 n_windows, then per window 4096 bytes of frame a and 4096 bytes of frame b (row-major 64x64)
//   stdout: per window 4096 doubles (map) + 8 doubles (record)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../torchpiv_amd/csrc/xcorr_f64_split.hpp"

using namespace tpiv;
using namespace tpiv::f64s;

struct Thread {
    cd x[M];
    cd u[M];
    cd t[M];
    cd Y[M + 1];
    double c[M];
    uint32_t da[NDW], db[NDW];
};

int main() {
    int32_t n = 0;
    if (fread(&n, 4, 1, stdin) != 1) return 1;
    std::vector<uint8_t> a(4096), b(4096);
    std::vector<double> plane(WS * PL);
    std::vector<Thread> T(128);
    for (int w = 0; w < n; ++w) {
        if (fread(a.data(), 1, 4096, stdin) != 4096 || fread(b.data(), 1, 4096, stdin) != 4096) return 2;
        unsigned long long sa = 0, sb = 0;
        for (int i = 0; i < 4096; ++i) sa += a[i], sb += b[i];
        const bool dead = sa == 0 || sb == 0;
        const double map_scale = dead ? 0.0 : ((double)(WS * WS) * 0.25) / ((double)sa * (double)sb);
        // R
        for (int t = 0; t < 128; ++t) {
            const int y = t & 63, h = t >> 6;
            for (int q = 0; q < NDW; ++q) {
                T[t].da[q] = a[y * 64 + 4 * q] | (a[y * 64 + 4 * q + 1] << 8) | (a[y * 64 + 4 * q + 2] << 16) | ((uint32_t)a[y * 64 + 4 * q + 3] << 24);
                T[t].db[q] = b[y * 64 + 4 * q] | (b[y * 64 + 4 * q + 1] << 8) | (b[y * 64 + 4 * q + 2] << 16) | ((uint32_t)b[y * 64 + 4 * q + 3] << 24);
            }
            rows_forward(T[t].da, T[t].db, h, T[t].x);
        }
        // T1, one component at a time
        for (int t = 0; t < 128; ++t) t1_write<0>(T[t].x, t & 63, t >> 6, plane.data());
        for (int t = 0; t < 128; ++t) t1_read<0>(T[t].u, t & 63, 1 - (t >> 6), plane.data());
        for (int t = 0; t < 128; ++t) t1_write<1>(T[t].x, t & 63, t >> 6, plane.data());
        for (int t = 0; t < 128; ++t) t1_read<1>(T[t].u, t & 63, 1 - (t >> 6), plane.data());
        // C
        for (int t = 0; t < 128; ++t) cols_forward(T[t].u, 1 - (t >> 6));
        // X: partner = lane (64 - k) % 64 of the same wave; snapshot first (the device shuffles read the old values
        // because every register is read before it is overwritten inside one thread, and threads run in lockstep)
        {
            std::vector<Thread> S = T;
            for (int t = 0; t < 128; ++t) {
                const int k = t & 63, wv = t >> 6, g = 1 - wv;
                const int partner = ((64 - k) & 63) + 64 * wv;
                auto sh = [&](double, int reg, int comp, int pt) { return comp ? S[pt].u[reg].y : S[pt].u[reg].x; };
                if (g == 0) cross_spectrum_g<0>(T[t].u, partner, sh);
                else cross_spectrum_g<1>(T[t].u, partner, sh);
            }
        }
        // Ci
        for (int t = 0; t < 128; ++t) cols_inverse(T[t].u, 1 - (t >> 6), T[t].t);
        // T2
        for (int t = 0; t < 128; ++t) t2_write<0>(T[t].t, t & 63, 1 - (t >> 6), plane.data());
        for (int t = 0; t < 128; ++t) t2_read<0>(T[t].Y, t & 63, plane.data());
        for (int t = 0; t < 128; ++t) t2_write<1>(T[t].t, t & 63, 1 - (t >> 6), plane.data());
        for (int t = 0; t < 128; ++t) t2_read<1>(T[t].Y, t & 63, plane.data());
        // Ri
        for (int t = 0; t < 128; ++t) rows_inverse(T[t].Y, t >> 6, T[t].c);
        // P
        double cmin = 1.7e308, graw = -1.7e308;
        std::vector<double> rraw(128);
        for (int t = 0; t < 128; ++t) {
            double mn;
            peak_local_minmax(T[t].c, mn, rraw[t]);
            cmin = mn < cmin ? mn : cmin;
            graw = rraw[t] > graw ? rraw[t] : graw;
        }
        const double gmax = peak_shifted(graw, cmin, map_scale);
        std::vector<double> rmax(128);
        for (int t = 0; t < 128; ++t) {
            peak_shift_and_write(T[t].c, cmin, map_scale, t & 63, t >> 6, plane.data());
            rmax[t] = peak_shifted(rraw[t], cmin, map_scale);
        }
        int ywin = WS - 1;
        for (int t = 0; t < 128; ++t) {
            const int fy = ((t & 63) + WS / 2) & (WS - 1);
            if (rmax[t] == gmax && fy < ywin) ywin = fy;
        }
        int xwin = WS - 1;
        for (int x = WS - 1; x >= 0; --x)
            if (plane[ywin * PL + x] == gmax) xwin = x;
        const int m = ywin * WS + xwin;
        double sv = -1.0;
        for (int t = 0; t < 128; ++t) {
            const double s = peak_second_local(T[t].c, t & 63, t >> 6, m, 3);
            sv = s > sv ? s : sv;
        }
        std::vector<double> map(4096);
        for (int y = 0; y < 64; ++y)
            for (int x = 0; x < 64; ++x) map[y * 64 + x] = plane[y * PL + x];
        fwrite(map.data(), 8, 4096, stdout);
        double rec[8];
        for (int s = 0; s < 8; ++s) rec[s] = peak_record_slot(s, m, sv, dead, plane.data());
        fwrite(rec, 8, 8, stdout);
    }
    return 0;
}
