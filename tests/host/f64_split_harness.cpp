// Host emulation of the float64 first-pass scheme for 64x64 and 128x128 windows: runs the per-thread functions of
// torchpiv_amd/csrc/xcorr_f64_split.hpp for all threads of a workgroup, phase by phase (a barrier on the device =
// the end of a loop here), on windows read from stdin and prints the correlation maps (corr - min + 1e-7, fftshift
// layout) and the 8-double records.  Built with g++ by tests/test_host_logic.py.  Second mode ("maps"): the peak stage
// alone on hand-made maps.
//   argv[1]: window edge W (64 or 128)
//   stdin : int32 n_windows, then per window W*W bytes of frame a and W*W bytes of frame b (row-major)
//   stdout: per window W*W doubles (map) + 8 doubles (record)
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../torchpiv_amd/csrc/xcorr_f64_split.hpp"

using namespace tpiv;
using namespace tpiv::f64s;

// the peak stage of one window: T[t].c holds thread t's raw cells.  Mirrors the kernel's steps (xcorr_f64.hip): exchange 1
// ((min, max) per thread / row maxima) -> first row of the maximum -> zone rows -> first column -> second peak -> record.
template <int W, typename ThreadVec>
void peak_stage(ThreadVec& T, double map_scale, bool dead, int wv, std::vector<double>& map_out, double (&rec)[8]) {
    using S = Split<W>;
    constexpr int NT = S::NT;
    auto line = [](int t) { return t % W; };
    auto half = [](int t) { return t / W; };
    double cmin = 1.7e308, graw = -1.7e308;
    std::vector<double> rmx(NT);
    for (int t = 0; t < NT; ++t) {
        double mn;
        S::peak_local_minmax(T[t].c, mn, rmx[t]);
        cmin = mn < cmin ? mn : cmin;
        graw = rmx[t] > graw ? rmx[t] : graw;
    }
    const double gmax = peak_shifted(graw, cmin, map_scale);
    std::vector<double> rm(W);
    int ywin = W - 1;
    for (int r = 0; r < W; ++r) {
        rm[r] = rmx[r] > rmx[W + r] ? rmx[r] : rmx[W + r];
        const int fy = S::frow(r);
        if (peak_shifted(rm[r], cmin, map_scale) == gmax && fy < ywin) ywin = fy;
    }
    const int nz = 2 * wv + 3, zlo = ywin - wv - 1;
    std::vector<double> zone((size_t)(nz > S::ZR ? S::PLANE : S::ZR * S::ZP), -7.0e300);
    for (int t = 0; t < NT; ++t) S::peak_zone_write(T[t].c, line(t), half(t), zlo, nz, zone.data());
    int xwin = W - 1;
    for (int x = W - 1; x >= 0; --x)
        if (peak_shifted(zone[(wv + 1) * S::ZP + x], cmin, map_scale) == gmax) xwin = x;
    const int m = ywin * W + xwin;
    double sv = PEAK_NONE;
    for (int r = 0; r < W; ++r) {
        const int zr = S::frow(r) - zlo;
        if ((zr < 0 || zr >= nz) && rm[r] > sv) sv = rm[r];
    }
    const bool clamp_lo = m - wv - wv * W <= 0, clamp_hi = m + wv + wv * W >= W * W - 1;
    for (int zr = 0; zr < nz; ++zr)
        for (int fx = 0; fx < W; ++fx) {
            const int fy = zlo + zr;
            if (fy < 0 || fy >= W) continue;
            int q;
            const bool colok = S::peak_col_ok(fx, xwin, wv, q);
            const bool ex = S::peak_excluded_zr(colok, q, zr, fy, fx, wv, clamp_lo, clamp_hi);      // the kernel's form of the test
            if (ex != (S::peak_excluded(fy * W + fx, m, wv) != 0)) {
                fprintf(stderr, "exclusion test mismatch: m %d cell (%d, %d)\n", m, fy, fx);
                exit(4);
            }
            if (ex) continue;
            const double raw = zone[zr * S::ZP + fx];
            if (raw > sv) sv = raw;
        }
    // the shifted map (what the second-generation kernel kept in LDS), from the threads' cells
    map_out.assign((size_t)W * W, 0.0);
    for (int t = 0; t < NT; ++t)
        for (int i = 0; i < S::M; ++i)
            map_out[(size_t)S::frow(line(t)) * W + S::fx0(i) + half(t)] = peak_shifted(T[t].c[i], cmin, map_scale);
    for (int s = 0; s < 8; ++s) rec[s] = S::peak_record_slot(s, m, sv, dead, zone.data(), zlo, cmin, map_scale);
}

template <int W>
struct Thread {
    using S = Split<W>;
    cd x[S::M];
    cd u[S::M];
    cd pz[S::M];
    cd t[S::M];
    cd Y[S::M + 1];
    double c[S::M];
    uint32_t da[S::NDW], db[S::NDW];
};

// argv[2] == "maps": stdin = int32 n, int32 wv, then per map W*W doubles (raw cells in fftshift layout, scale 1) -> the
// peak stage alone; stdout per map the 8-double record
template <int W>
int run_maps() {
    using S = Split<W>;
    int32_t n = 0, wv = 3;
    if (fread(&n, 4, 1, stdin) != 1 || fread(&wv, 4, 1, stdin) != 1) return 1;
    std::vector<double> mp((size_t)W * W), map_out;
    std::vector<Thread<W>> T(S::NT);
    for (int w = 0; w < n; ++w) {
        if (fread(mp.data(), 8, (size_t)W * W, stdin) != (size_t)W * W) return 2;
        for (int t = 0; t < S::NT; ++t)
            for (int i = 0; i < S::M; ++i) T[t].c[i] = mp[(size_t)S::frow(t % W) * W + S::fx0(i) + t / W];
        double rec[8];
        peak_stage<W>(T, 1.0, false, wv, map_out, rec);
        fwrite(rec, 8, 8, stdout);
    }
    return 0;
}

template <int W>
int run() {
    using S = Split<W>;
    constexpr int NDW = S::NDW, NT = S::NT;
    int32_t n = 0;
    if (fread(&n, 4, 1, stdin) != 1) return 1;
    std::vector<uint8_t> a(W * W), b(W * W);
    std::vector<double> plane(S::PLANE);
    std::vector<Thread<W>> T(NT);
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
        // T1, one component at a time; column stages: position L = line, g = 1 - half
        auto t1w = [&](auto comp) {
            constexpr int COMP = decltype(comp)::value;
            for (int t = 0; t < NT; ++t) {
                if (half(t)) S::template t1_write<COMP, 1>(T[t].x, line(t), plane.data());
                else S::template t1_write<COMP, 0>(T[t].x, line(t), plane.data());
            }
        };
        t1w(std::integral_constant<int, 0>{});
        for (int t = 0; t < NT; ++t) S::template t1_read<0>(T[t].u, line(t), 1 - half(t), plane.data());
        t1w(std::integral_constant<int, 1>{});
        for (int t = 0; t < NT; ++t) S::template t1_read<1>(T[t].u, line(t), 1 - half(t), plane.data());
        // C
        for (int t = 0; t < NT; ++t) S::cols_forward(T[t].u, 1 - half(t));
        // X: the mirror thread is the neighbouring lane (line ^ 1), lines 0 and 1 are their own; snapshot first (the device
        // moves read the old values)
        {
            std::vector<Thread<W>> Sn = T;
            for (int t = 0; t < NT; ++t) {
                const int L = line(t), g = 1 - half(t);
                const int mirror = L < 2 ? t : (t ^ 1);
                if (S::col_of(L) != (W - S::col_of(line(mirror))) % W) return 3;          // the lane order pairs mirrors
                auto sh = [&](double, int reg, int comp) { return comp ? Sn[mirror].u[reg].y : Sn[mirror].u[reg].x; };
                if (g == 0) S::template cross_spectrum_g<0>(T[t].u, T[t].pz, sh);
                else S::template cross_spectrum_g<1>(T[t].u, T[t].pz, sh);
            }
        }
        // Ci
        for (int t = 0; t < NT; ++t) S::cols_inverse(T[t].pz, 1 - half(t), T[t].t);
        // T2: both components at once
        for (auto& v : plane) v = -3.0e300;
        for (int t = 0; t < NT; ++t) S::t2_write(T[t].t, line(t), 1 - half(t), plane.data());
        for (int t = 0; t < NT; ++t) S::t2_read(T[t].Y, line(t), plane.data());
        // Ri
        for (int t = 0; t < NT; ++t) S::rows_inverse(T[t].Y, half(t), T[t].c);
        // P
        std::vector<double> map;
        double rec[8];
        peak_stage<W>(T, map_scale, dead, 3, map, rec);
        fwrite(map.data(), 8, W * W, stdout);
        fwrite(rec, 8, 8, stdout);
    }
    return 0;
}

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 64;
    if (argc > 2 && std::string(argv[2]) == "maps") return W == 128 ? run_maps<128>() : run_maps<64>();
    return W == 128 ? run<128>() : run<64>();
}
