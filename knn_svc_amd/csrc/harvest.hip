// Harvest f0 estimation on the GPU.  The reference calls pyworld.harvest(x.double(), fs = 16000,
// f0_floor = 65, f0_ceil = 1047, frame_period = 20) and zeroes values below 80 Hz (ddsp_prematch_dataset.py:121-128)
// whenever `<stem>_f0.npy` is missing (:376-379).  pyworld (WORLD, M. Morise) is a third-party dependency; the algorithm
// restated here is the published one (Morise, Interspeech 2017) as pinned by oracle/f0_ref.py against the two harvest tracks
// the reference ships.  Everything is fp64 like the library: the stages make hard decisions (zero crossings, 10 % / 5 % /
// 0.8 % gates, score comparisons) on the values.
//
// Stages (one stream, no host synchronisation; all intermediate arrays live in a caller-provided workspace):
//   1  decimate 16 -> 8 kHz: reflect-extend by 9, third-order IIR forward then backward, every second sample.  The recursion is
//      cut into 64-sample chunks, each warmed up over the 192 samples before it (pole radius 0.65: 0.65^192 = 1e-36).
//   2  subtract the mean (one block, fixed summation order).
//   3  band-pass bank: 40 channels per octave, Nuttall-windowed cosine of four periods, as a direct FIR out of LDS
//      (the library multiplies FFTs; the taps are short — 29 .. 539 — and fp64 FMA is cheap here).     [nch][ylen] fp64
//   4  per channel, four event lists (negative / positive going zero crossings, peaks, dips) with the sub-sample
//      position of each event, in time order (one block per list, ballot compaction).
//   5  per channel and 1 ms frame: interpolate the four interval-f0 tracks at the frame time, average, gate to +-10 % of
//      the channel frequency.                                                                           [nch][nfr] fp64
//   6  per frame: every run of >= 10 neighbouring voiced channels gives a candidate (its mean).         [nfr][16]
//   7  overlap with the candidates of the +-3 neighbouring frames and refine each by the instantaneous frequency of up to six
//      harmonics: one wave per (frame, neighbour group); windowed DFTs at the harmonic bins only.        [nfr][112] x 2
//   8  drop candidates without a neighbour-frame candidate within 5 %.
//   9  contour: best-score base, 0.8 % jump removal, sections shorter than 6 frames removed, sections extended along
//      candidates within 18 % (one wave per section), merged by score, gaps below 9 frames bridged — single-block
//      kernels, the lists are a few hundred entries.
//  10  zero-phase second-order Butterworth smoothing per section (one thread per section), sampling at the frame period.
#include "common.h"

namespace {

constexpr int HV_NC = 16;                // candidate slots per frame before the overlap
constexpr int HV_NS = HV_NC * 7;         // ... after it
constexpr int HV_MARG = 104;             // frames a section can grow on either side (100 + the write one past the limit)
constexpr int HV_SMOOTH_MARG = 600;      // 0.875^600 = 1.6e-35: the smoother's constant extension beyond this is invisible
constexpr int HV_YPAD = 288;             // zeros either side of the decimated signal (longest filter: 269 + 1)
constexpr double HV_PI = 3.14159265358979323846;

__device__ __forceinline__ long hv_round(double x) { return x > 0 ? (long)(x + 0.5) : (long)(x - 0.5); }

// ---------------------------------------------------------------------------------------------- 1: decimation
__device__ __forceinline__ double hv_ext(const float* __restrict__ x, long L, long i) {
    if (i < 9) return 2.0 * (double)x[0] - (double)x[9 - i];
    if (i < 9 + L) return (double)x[i - 9];
    return 2.0 * (double)x[L - 1] - (double)x[L - 2 - (i - 9 - L)];
}

template <bool FIRST>
__global__ __launch_bounds__(256) void hv_iir_kernel(const float* __restrict__ x, const double* __restrict__ in, double* __restrict__ out,
                                                    long n, long L) {
    constexpr int C = 64, W = 192;
    const double a0 = 0.041156734567757189, a1 = -0.42599112459189636, a2 = 0.041037215479961225;
    const double b0 = 0.16797464681802227, b1 = 0.50392394045406674;
    const long c0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * C;
    if (c0 >= n) return;
    const long start = c0 - W > 0 ? c0 - W : 0, stop = c0 + C < n ? c0 + C : n;
    double w0 = 0, w1 = 0, w2 = 0;
    for (long p = start; p < stop; ++p) {
        const long i = FIRST ? p : n - 1 - p;
        const double xi = FIRST ? hv_ext(x, L, i) : in[i];
        const double wt = xi + a0 * w0 + a1 * w1 + a2 * w2;
        const double yv = b0 * wt + b1 * w0 + b1 * w1 + b0 * w2;
        w2 = w1; w1 = w0; w0 = wt;
        if (p >= c0) out[i] = yv;
    }
}

// ---------------------------------------------------------------------------------------------- 2: mean
__global__ __launch_bounds__(1024) void hv_mean_kernel(const double* __restrict__ t2, long nbeg, long ylen, double* __restrict__ mean) {
    __shared__ double part[1024];
    const long per = (ylen + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < ylen ? lo + per : ylen;
    double s = 0;
    for (long k = lo; k < hi; ++k) s += t2[nbeg + 2 * k + 8];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) *mean = part[0] / (double)ylen;
}

// y (padded by HV_YPAD zeros either side) = decimated - mean; also the channel table
__global__ void hv_center_kernel(const double* __restrict__ t2, long nbeg, long ylen, const double* __restrict__ mean,
                                 double* __restrict__ ypad, long ypad_len) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ypad_len) return;
    const long k = j - HV_YPAD;
    ypad[j] = (k >= 0 && k < ylen) ? t2[nbeg + 2 * k + 8] - *mean : 0.0;
}

__global__ void hv_channels_kernel(double adj_floor, int nch, double fs, double* __restrict__ bf0, int* __restrict__ hlen) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nch) return;
    const double b = adj_floor * pow(2.0, (double)(c + 1) / 40.0);
    bf0[c] = b;
    hlen[c] = (int)hv_round(fs / b * 2.0);
}

// ---------------------------------------------------------------------------------------------- 3: band-pass bank
constexpr int HV_TILE = 1024, HV_HMAX = 272;
__global__ __launch_bounds__(256) void hv_bank_kernel(const double* __restrict__ ypad, long ylen, const double* __restrict__ bf0,
                                                     const int* __restrict__ hlen, double fs, double* __restrict__ filt, long ld) {
    __shared__ double ys[HV_TILE + 2 * HV_HMAX + 2];
    __shared__ double wt[2 * HV_HMAX + 2];
    const int c = blockIdx.y, h = hlen[c], n = 2 * h + 1, tid = threadIdx.x;
    const long tile0 = (long)blockIdx.x * HV_TILE;
    const double b = bf0[c];
    for (int k = tid; k < n; k += 256) {
        const double u = (double)k / (double)(n - 1);
        const double nut = 0.355768 - 0.487396 * cos(2.0 * HV_PI * u) + 0.144232 * cos(4.0 * HV_PI * u) - 0.012604 * cos(6.0 * HV_PI * u);
        wt[k] = nut * cos(2.0 * HV_PI * b * (double)(k - h) / fs);
    }
    // ys[j] = y[tile0 - h + j]
    for (int j = tid; j < HV_TILE + 2 * h + 2; j += 256) {
        const long p = tile0 - h + j + HV_YPAD;
        ys[j] = (p >= 0 && p < ylen + 2 * HV_YPAD) ? ypad[p] : 0.0;
    }
    __syncthreads();
    double acc[4] = {0, 0, 0, 0};
    // filt[i] = sum_k wt[k] y[i + h + 1 - k]  ->  ys index (i - tile0) + 2h + 1 - k
    for (int k = 0; k < n; ++k) {
        const double w = wt[k];
        const int o = 2 * h + 1 - k + tid;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = fma(w, ys[o + 256 * r], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long i = tile0 + tid + 256 * r;
        if (i < ylen) filt[(long)c * ld + i] = acc[r];
    }
}

// ---------------------------------------------------------------------------------------------- 4: events
// exclusive prefix of v over the block in thread order; *total = block sum.  sh: >= blockDim / 64 ints
__device__ int hv_block_scan(int v, int* total, int* sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    __syncthreads();
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < nw; ++i) { const int s = sh[i]; if (i < w) base += s; tot += s; }
    *total = tot;
    return base + inc - v;
}

// kind 0: negative-going zero crossings of the filtered signal, 1: positive-going, 2: peaks, 3: dips (crossings of the
// differenced, sign-flipped signal).  events[(c * 4 + kind) * ecap + r] = 1-based sub-sample position of the r-th event
__global__ __launch_bounds__(256) void hv_events_kernel(const double* __restrict__ filt, long ld, long ylen, double* __restrict__ events,
                                                       long ecap, int* __restrict__ ecount) {
    __shared__ int sh[4];
    const int c = blockIdx.x >> 2, kind = blockIdx.x & 3;
    const double* f = filt + (long)c * ld;
    double* ev = events + (long)blockIdx.x * ecap;
    const long lim = kind < 2 ? ylen - 1 : ylen - 2;
    const double sg = (kind & 1) ? -1.0 : 1.0;
    long base = 0;
    for (long c0 = 0; c0 < lim; c0 += 256) {
        const long i = c0 + threadIdx.x;
        bool hit = false; double fine = 0;
        if (i < lim) {
            double s0, s1;
            if (kind < 2) { s0 = sg * f[i]; s1 = sg * f[i + 1]; }
            else { const double a = f[i], b = f[i + 1], d = f[i + 2]; s0 = sg * (b - a); s1 = sg * (d - b); }
            hit = s0 > 0 && s1 <= 0;
            if (hit) fine = (double)(i + 1) - s0 / (s1 - s0);
        }
        int tot;
        const int r = hv_block_scan(hit ? 1 : 0, &tot, sh);
        if (hit && base + r < ecap) ev[base + r] = fine;
        base += tot;
    }
    if (threadIdx.x == 0) ecount[blockIdx.x] = (int)(base < ecap ? base : ecap);
}

// ---------------------------------------------------------------------------------------------- 5: raw per-channel f0
__global__ __launch_bounds__(256) void hv_raw_kernel(const double* __restrict__ events, long ecap, const int* __restrict__ ecount,
                                                    const double* __restrict__ bf0, double fs, double f0_floor, double f0_ceil,
                                                    long nfr, double* __restrict__ raw) {
    const int c = blockIdx.y;
    const long fr = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (fr >= nfr) return;
    const double t = (double)fr / 1000.0;
    double sum = 0; bool ok = true;
    for (int kind = 0; kind < 4; ++kind) {
        const int nint = ecount[c * 4 + kind] - 1;
        if (nint < 3) { ok = false; break; }
        const double* e = events + (long)(c * 4 + kind) * ecap;
        // k = #{j : loc_j <= t}, clipped to [1, nint - 1];  loc_j = (e_j + e_{j+1}) / 2 / fs
        int lo = 0, hi = nint;
        while (lo < hi) { const int m = (lo + hi) >> 1; if ((e[m] + e[m + 1]) / 2.0 / fs <= t) lo = m + 1; else hi = m; }
        int k = lo < 1 ? 1 : (lo > nint - 1 ? nint - 1 : lo);
        const double x0 = (e[k - 1] + e[k]) / 2.0 / fs, x1 = (e[k] + e[k + 1]) / 2.0 / fs;
        const double y0 = fs / (e[k] - e[k - 1]), y1 = fs / (e[k + 1] - e[k]);
        const double s = (t - x0) / (x1 - x0);
        sum += y0 + s * (y1 - y0);
    }
    double v = 0;
    if (ok) {
        v = sum / 4.0;
        const double b = bf0[c];
        if (v > b * 1.1 || v < b * 0.9 || v > f0_ceil || v < f0_floor) v = 0;
    }
    raw[(long)c * nfr + fr] = v;
}

// ---------------------------------------------------------------------------------------------- 6: candidates
__global__ __launch_bounds__(256) void hv_detect_kernel(const double* __restrict__ raw, int nch, long nfr, double* __restrict__ cand0,
                                                       int* __restrict__ info) {
    const long fr = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (fr >= nfr) return;
    double* out = cand0 + fr * HV_NC;
    int cnt = 0, st = -1; bool prev = false; double run = 0;
    for (int c = 1; c < nch; ++c) {
        const double v = c < nch - 1 ? raw[(long)c * nfr + fr] : 0.0;       // first and last channel never count as voiced
        const bool cur = v > 0;
        if (cur && !prev) { st = c; run = 0; }
        if (!cur && prev) {
            if (c - st >= 10) { if (cnt < HV_NC) out[cnt++] = run / (double)(c - st); else atomicOr(&info[2], 1); }
        }
        if (cur) run += v;
        prev = cur;
    }
    for (int j = cnt; j < HV_NC; ++j) out[j] = 0;
}

// ---------------------------------------------------------------------------------------------- 7: overlap + refinement
__global__ __launch_bounds__(256) void hv_refine_kernel(const double* __restrict__ ypad, long ylen, const double* __restrict__ cand0, long nfr,
                                                       double fs, double f0_floor, double f0_ceil, double* __restrict__ cand,
                                                       double* __restrict__ score) {
    __shared__ double2 tw[1024];                 // (cos, sin)(2 pi q / 1024)
    __shared__ double win_s[4][384];
    for (int q = threadIdx.x; q < 1024; q += 256) { double s, c; sincospi((double)q / 512.0, &s, &c); tw[q] = make_double2(c, s); }
    __syncthreads();
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long unit = (long)blockIdx.x * 4 + wv;             // (frame, group)
    if (unit >= nfr * 7) return;
    const long fr = unit / 7; const int g = (int)(unit - fr * 7);
    const long src = g == 0 ? fr : (g <= 3 ? fr - g : fr + (g - 3));
    double* win = win_s[wv];
    const double pos = (double)fr / 1000.0;
    for (int j = 0; j < HV_NC; ++j) {
        const double f = (src >= 0 && src < nfr) ? cand0[src * HV_NC + j] : 0.0;
        double rf = 0, sc = 0;
        if (f > 0) {                                          // wave-uniform
            const int half = (int)(1.5 * fs / f + 1.0), n = 2 * half + 1;
            const double wl = (double)n / fs;
            const int lg = 31 - __builtin_clz((unsigned)n), fft = 1 << (2 + lg), tstep = 1024 / fft;
            const long base0 = hv_round((pos - (double)half / fs) * fs + 0.001);
            for (int m = lane; m < n; m += 64) {
                const double tt = ((double)(base0 + m) - 1.0) / fs - pos;
                win[m] = 0.42 + 0.5 * cos(2.0 * HV_PI * tt / wl) + 0.08 * cos(4.0 * HV_PI * tt / wl);
            }
            __builtin_amdgcn_wave_barrier();
            int nh = (int)(fs / 2.0 / f); nh = nh < 6 ? nh : 6;
            int idx[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) { long ix = hv_round(f * (double)fft / fs * (double)(k + 1)); idx[k] = (int)(ix < fft / 2 ? ix : fft / 2); }
            double sr[6] = {0, 0, 0, 0, 0, 0}, si[6] = {0, 0, 0, 0, 0, 0}, dr[6] = {0, 0, 0, 0, 0, 0}, di[6] = {0, 0, 0, 0, 0, 0};
            for (int m = lane; m < n; m += 64) {
                long p = base0 + m - 1; p = p < 0 ? 0 : (p > ylen - 1 ? ylen - 1 : p);
                const double xv = ypad[p + HV_YPAD];
                const double w = win[m];
                const double dw = m == 0 ? -win[1] / 2.0 : (m == n - 1 ? win[n - 2] / 2.0 : -(win[m + 1] - win[m - 1]) / 2.0);
                const double am = xv * w, ad = xv * dw;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const double2 t = tw[((idx[k] * m) & (fft - 1)) * tstep];
                    sr[k] = fma(am, t.x, sr[k]); si[k] = fma(-am, t.y, si[k]);
                    dr[k] = fma(ad, t.x, dr[k]); di[k] = fma(-ad, t.y, di[k]);
                }
            }
            double num = 0, den = 0, dev = 0;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double a = wave_sum_d(sr[k]), b = wave_sum_d(si[k]), c = wave_sum_d(dr[k]), d = wave_sum_d(di[k]);
                if (k < nh) {
                    const double pw = a * a + b * b, nm = a * d - b * c;
                    const double inst = pw == 0.0 ? 0.0 : (double)idx[k] * fs / (double)fft + nm / pw * fs / 2.0 / HV_PI;
                    const double amp = sqrt(pw);
                    num += amp * inst; den += amp * (double)(k + 1);
                    dev += fabs((inst / (double)(k + 1) - f) / f);
                }
            }
            rf = num / (den + 1e-12);
            sc = 1.0 / (1e-12 + dev / (double)nh);
            if (rf < f0_floor || rf > f0_ceil || sc < 2.5) { rf = 0; sc = 0; }
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) { cand[fr * HV_NS + g * HV_NC + j] = rf; score[fr * HV_NS + g * HV_NC + j] = sc; }
    }
}

// ---------------------------------------------------------------------------------------------- 8: unreliable candidates
__global__ __launch_bounds__(128) void hv_reliable_kernel(const double* __restrict__ cand, const double* __restrict__ score, long nfr,
                                                         double* __restrict__ cand2, double* __restrict__ score2) {
    const long fr = blockIdx.x; const int s = threadIdx.x;
    if (s >= HV_NS) return;
    const double ref = cand[fr * HV_NS + s];
    double c = ref, sc = score[fr * HV_NS + s];
    if (ref != 0 && fr >= 1 && fr <= nfr - 2) {
        double e1 = 1.0, e2 = 1.0;
        const double* nx = cand + (fr + 1) * HV_NS; const double* pv = cand + (fr - 1) * HV_NS;
        for (int j = 0; j < HV_NS; ++j) {
            const double a = fabs(ref - nx[j]) / ref, b = fabs(ref - pv[j]) / ref;
            e1 = a < e1 ? a : e1; e2 = b < e2 ? b : e2;
        }
        if ((e1 < e2 ? e1 : e2) > 0.05) { c = 0; sc = 0; }
    }
    cand2[fr * HV_NS + s] = c; score2[fr * HV_NS + s] = sc;
}

// ---------------------------------------------------------------------------------------------- 9: contour
__global__ __launch_bounds__(256) void hv_base_kernel(const double* __restrict__ cand, const double* __restrict__ score, long nfr,
                                                     double* __restrict__ base) {
    const long fr = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (fr >= nfr) return;
    double best = 0, bs = 0;
    for (int j = 0; j < HV_NS; ++j) { const double s = score[fr * HV_NS + j]; if (s > bs) { bs = s; best = cand[fr * HV_NS + j]; } }
    base[fr] = best;
}

__global__ __launch_bounds__(256) void hv_step1_kernel(const double* __restrict__ base, long nfr, double allowed, double* __restrict__ s1,
                                                      double* __restrict__ s2) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nfr) return;
    double v = 0;
    if (i >= 2 && base[i] != 0) {
        const double b0 = base[i], b1 = base[i - 1], b2 = base[i - 2];
        const double ref = b1 * 2 - b2;
        const bool a = fabs((b0 - ref) / ref) > allowed, b = fabs(b0 - b1) / b1 > allowed;     // NaN (0/0) compares false, as in C
        v = (a && b) ? 0.0 : b0;
    }
    s1[i] = v; s2[i] = v;
}

// voiced sections [st, ed] (inclusive) of f in time order; force: the first and the last frame never count as voiced
__device__ int hv_sections(const double* f, long n, bool force, int* st, int* ed, int cap, int* sh) {
    int nr = 0, nf = 0;
    for (long c0 = 0; c0 < n; c0 += blockDim.x) {
        const long i = c0 + threadIdx.x;
        bool rise = false, fall = false;
        if (i < n) {
            auto V = [&](long j) { return j >= 0 && j < n && f[j] > 0 && !(force && (j == 0 || j == n - 1)); };
            const bool vi = V(i);
            rise = vi && !V(i - 1); fall = vi && !V(i + 1);
        }
        int tr, tf;
        const int r = hv_block_scan(rise ? 1 : 0, &tr, sh);
        if (rise && nr + r < cap) st[nr + r] = (int)i;
        const int q = hv_block_scan(fall ? 1 : 0, &tf, sh);
        if (fall && nf + q < cap) ed[nf + q] = (int)i;
        nr += tr; nf += tf;
    }
    __syncthreads();
    return nr < cap ? nr : cap;
}

// info: [0] sections after step 2, [1] kept after extension, [2] overflow flags, [3] sections for the smoother
__global__ __launch_bounds__(1024) void hv_step2_kernel(const double* __restrict__ s1, double* __restrict__ s2, long nfr, int vmin,
                                                       int* __restrict__ st, int* __restrict__ ed, int cap, long* __restrict__ woff,
                                                       long chan_cap, int* __restrict__ info) {
    __shared__ int sh[16];
    const int n1 = hv_sections(s1, nfr, true, st, ed, cap, sh);
    for (int k = threadIdx.x; k < n1; k += blockDim.x)
        if (ed[k] - st[k] < vmin) for (int j = st[k]; j <= ed[k]; ++j) s2[j] = 0;
    __syncthreads();
    int n2 = hv_sections(s2, nfr, true, st, ed, cap, sh);
    if (threadIdx.x == 0) {
        long run = 0;
        for (int k = 0; k < n2; ++k) {
            const long ws = st[k] - HV_MARG > 0 ? st[k] - HV_MARG : 0, we = ed[k] + HV_MARG < nfr - 1 ? ed[k] + HV_MARG : nfr - 1;
            if (run + (we - ws + 1) > chan_cap) { info[2] |= 4; n2 = k; break; }
            woff[k] = run; run += we - ws + 1;
        }
        woff[n2] = run; info[0] = n2;
    }
}

// SelectBestF0 over the HV_NS slots of one frame by one wave: smallest relative distance <= allowed, the later slot on ties
__device__ __forceinline__ double hv_select(double ref, const double* __restrict__ c, double allowed, int lane) {
    double be = allowed, bv = 0; int bi = -1;
    for (int s = lane; s < HV_NS; s += 64) {
        const double v = c[s], t = fabs(ref - v) / ref;
        if (t > be) continue;
        be = t; bv = v; bi = s;
    }
    if (bi < 0) be = 2.0;                       // nothing within range on this lane
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double oe = __shfl_xor(be, o, 64), ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (oe < be || (oe == be && oi > bi)) { be = oe; bv = ov; bi = oi; }
    }
    return bi < 0 ? 0.0 : bv;
}

// one wave per section: copy the section into its window of `chan`, extend forward then backward, decide whether to keep it
__global__ __launch_bounds__(256) void hv_extend_kernel(const double* __restrict__ s2, const double* __restrict__ cand, long nfr,
                                                       const int* __restrict__ st, const int* __restrict__ ed, const long* __restrict__ woff,
                                                       const double* __restrict__ score, double* __restrict__ chan,
                                                       double* __restrict__ chs, int* __restrict__ xst, int* __restrict__ xed,
                                                       int* __restrict__ keep, const int* __restrict__ info, double allowed) {
    const int lane = threadIdx.x & 63;
    const int nsec = info[0];
    for (int k = blockIdx.x * 4 + (threadIdx.x >> 6); k < nsec; k += gridDim.x * 4) {
        const long ws = st[k] - HV_MARG > 0 ? st[k] - HV_MARG : 0, we = ed[k] + HV_MARG < nfr - 1 ? ed[k] + HV_MARG : nfr - 1;
        double* ch = chan + woff[k] - ws;                        // ch[frame]
        for (long j = ws + lane; j <= we; j += 64) ch[j] = (j >= st[k] && j <= ed[k]) ? s2[j] : 0.0;
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        int so_f = ed[k], so_b = st[k];
        {   // forward
            const long origin = ed[k], last = origin + 100 < nfr - 2 ? origin + 100 : nfr - 2;
            double tmp = s2[origin]; int count = 0;
            for (long i = 0; i <= (last > origin ? last - origin : origin - last); ++i) {
                const long p = origin + i + 1;
                if (p > we) break;
                const double b = hv_select(tmp, cand + p * HV_NS, allowed, lane);
                if (lane == 0) ch[p] = b;
                if (b == 0) ++count; else { tmp = b; count = 0; so_f = (int)p; }
                if (count == 4) break;
            }
        }
        {   // backward
            const long origin = st[k], last = origin - 100 > 1 ? origin - 100 : 1;
            double tmp = s2[origin]; int count = 0;
            for (long i = 0; i <= (origin > last ? origin - last : last - origin); ++i) {
                const long p = origin - i - 1;
                if (p < ws) break;
                const double b = hv_select(tmp, cand + p * HV_NS, allowed, lane);
                if (lane == 0) ch[p] = b;
                if (b == 0) ++count; else { tmp = b; count = 0; so_b = (int)p; }
                if (count == 4) break;
            }
        }
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        double sum = 0;
        for (long j = so_b + lane; j < so_f; j += 64) sum += ch[j];
        sum = wave_sum_d(sum);
        const double m = so_f > so_b ? sum / (double)(so_f - so_b) : 0.0;
        if (lane == 0) { xst[k] = so_b; xed[k] = so_f; keep[k] = (m > 0 && 2200.0 / m < (double)(so_f - so_b)) ? 1 : 0; }
        // SearchScore of every value of the channel (best score among the frame's candidates equal to it): the merge sums these
        double* cs = chs + woff[k] - ws;
        for (long j = ws + lane; j <= we; j += 64) {
            const double f = ch[j]; double b = 0;
            if (f != 0) { const double* c = cand + j * HV_NS; const double* sc = score + j * HV_NS;
                          for (int q = 0; q < HV_NS; ++q) if (c[q] == f && sc[q] > b) b = sc[q]; }
            cs[j] = b;
        }
    }
}

__global__ __launch_bounds__(1024) void hv_merge_kernel(const double* __restrict__ s2, const double* __restrict__ chs,
                                                       double* __restrict__ ms, long nfr, const int* __restrict__ st,
                                                       const int* __restrict__ xst, const int* __restrict__ xed, int* __restrict__ keep_kk,
                                                       const long* __restrict__ woff, const double* __restrict__ chan,
                                                       int* __restrict__ order, double* __restrict__ s3, double* __restrict__ s4,
                                                       int* __restrict__ gst, int* __restrict__ ged, int cap, long* __restrict__ soff,
                                                       long scratch_cap, int gap, int* __restrict__ info) {
    __shared__ int sh[16];
    __shared__ double p1[1024], p2[1024];
    __shared__ int nk_s, mode_s;
    const int tid = threadIdx.x;
    const int nsec = info[0];
    // MergeF0 as the library does it, quirks included (the reference's shipped tracks depend on them): the order comes from an
    // insertion pass that moves a new element at most ONE place forward, the merge starts from the first kept channel
    // whatever the order says, and the running bounds live in that channel's slots of the (mutable) boundary list.
    int* kk = keep_kk;                                // kept section ids, in place over the keep flags
    int* bst = gst; int* bed = ged;                   // mutable bounds of the kept sections (the lists are rebuilt below)
    if (tid == 0) {
        int nk = 0;
        for (int k = 0; k < nsec; ++k) if (keep_kk[k]) { bst[nk] = xst[k]; bed[nk] = xed[k]; kk[nk] = k; ++nk; }
        for (int i = 0; i < nk; ++i) order[i] = i;
        for (int i = 1; i < nk; ++i)
            for (int j = i - 1; j >= 0; --j) {
                if (bst[order[j]] > bst[order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; } else break;
            }
        nk_s = nk; info[1] = nk;
    }
    __syncthreads();
    const int nk = nk_s;
    auto CH = [&](int k) { const long ws = st[k] - HV_MARG > 0 ? st[k] - HV_MARG : 0; return chan + woff[k] - ws; };
    auto CS = [&](int k) { const long ws = st[k] - HV_MARG > 0 ? st[k] - HV_MARG : 0; return chs + woff[k] - ws; };
    if (nk == 0) {
        for (long i = tid; i < nfr; i += blockDim.x) s3[i] = s2[i];
    } else {
        for (long i = tid; i < nfr; i += blockDim.x) { s3[i] = 0; ms[i] = 0; }
        __syncthreads();
        {
            const int k = kk[0]; const double* ch = CH(k); const double* cs = CS(k);
            const long ws = st[k] - HV_MARG > 0 ? st[k] - HV_MARG : 0;
            const long we = ws + (woff[k + 1] - woff[k]) - 1;
            for (long i = ws + tid; i <= we; i += blockDim.x) { s3[i] = ch[i]; ms[i] = cs[i]; }
        }
        __syncthreads();
        for (int q = 1; q < nk; ++q) {
            const int o = order[q], k = kk[o]; const double* ch = CH(k); const double* cs = CS(k);
            const int a = bst[o], e = bed[o], b0 = bst[0], b1 = bed[0];
            __syncthreads();
            if (a - b1 > 0) {
                for (long i = a + tid; i <= e; i += blockDim.x) { s3[i] = ch[i]; ms[i] = cs[i]; }
                if (tid == 0) { bst[0] = a; bed[0] = e; }
            } else if (b0 <= a && b1 >= e) {
                // inside what is already merged
            } else {
                // score sums over the overlap: the same (fixed) association for both, so equal terms give equal sums
                double a1 = 0, a2 = 0;
                for (long i = a + tid; i <= b1; i += blockDim.x) { a1 += ms[i]; a2 += cs[i]; }
                p1[tid] = a1; p2[tid] = a2;
                __syncthreads();
                for (int w = 512; w > 0; w >>= 1) { if (tid < w) { p1[tid] += p1[tid + w]; p2[tid] += p2[tid + w]; } __syncthreads(); }
                if (tid == 0) mode_s = p1[0] > p2[0] ? 1 : 0;
                __syncthreads();
                const long from = mode_s ? b1 : a;
                for (long i = from + tid; i <= e; i += blockDim.x) { s3[i] = ch[i]; ms[i] = cs[i]; }
                if (tid == 0) bed[0] = e;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    // step 4: bridge gaps shorter than `gap` frames
    for (long i = tid; i < nfr; i += blockDim.x) s4[i] = s3[i];
    __syncthreads();
    const int n3 = hv_sections(s3, nfr, true, gst, ged, cap, sh);
    for (int k = tid; k + 1 < n3; k += blockDim.x) {
        const int dist = gst[k + 1] - ged[k] - 1;
        if (dist >= gap) continue;
        const double t0 = s3[ged[k]] + 1, t1 = s3[gst[k + 1]] - 1, co = (t1 - t0) / ((double)dist + 1.0);
        for (int j = 1; j <= dist; ++j) s4[ged[k] + j] = t0 + co * (double)j;
    }
    __syncthreads();
    int n4 = hv_sections(s4, nfr, false, gst, ged, cap, sh);
    if (tid == 0) {
        long run = 0;
        for (int k = 0; k < n4; ++k) {
            const long lo = gst[k] - HV_SMOOTH_MARG > -300 ? gst[k] - HV_SMOOTH_MARG : -300;
            const long hi = ged[k] + HV_SMOOTH_MARG < nfr + 299 ? ged[k] + HV_SMOOTH_MARG : nfr + 299;
            if (run + (hi - lo + 1) > scratch_cap) { info[2] |= 2; n4 = k; break; }
            soff[k] = run; run += hi - lo + 1;
        }
        info[3] = n4;
    }
}

// ---------------------------------------------------------------------------------------------- 10: smoothing, sampling
// one thread per section: the section, held constant beyond its ends (the library pads the contour by 300 frames either side
// and filters the whole padded array; here the run is cut HV_SMOOTH_MARG frames from the section and started from the
// constant input's steady state, which the full run has reached to 1e-35 by then)
__global__ __launch_bounds__(64) void hv_smooth_kernel(const double* __restrict__ s4, long nfr, const int* __restrict__ gst,
                                                      const int* __restrict__ ged, const long* __restrict__ soff, double* __restrict__ scratch,
                                                      double* __restrict__ sm, const int* __restrict__ info) {
    const double b0 = 0.0078202080334971724, b1 = 0.015640416066994345;
    const double a0 = 1.7347257688092754, a1 = -0.76600660094326412;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= info[3]) return;
    const long st = gst[k], ed = ged[k];
    const bool cut_lo = st - HV_SMOOTH_MARG > -300, cut_hi = ed + HV_SMOOTH_MARG < nfr + 299;
    const long lo = cut_lo ? st - HV_SMOOTH_MARG : -300, hi = cut_hi ? ed + HV_SMOOTH_MARG : nfr + 299;
    double* buf = scratch + soff[k] - lo;
    const double xl = s4[st], xr = s4[ed];
    double w0 = 0, w1 = 0;
    if (cut_lo) w0 = w1 = xl / (1.0 - a0 - a1);
    for (long i = lo; i <= hi; ++i) {
        const double x = i < st ? xl : (i > ed ? xr : s4[i]);
        const double wt = x + a0 * w0 + a1 * w1;
        buf[i] = b0 * wt + b1 * w0 + b0 * w1;
        w1 = w0; w0 = wt;
    }
    w0 = w1 = 0;
    if (cut_hi) w0 = w1 = buf[hi] / (1.0 - a0 - a1);
    for (long i = hi; i >= st; --i) {
        const double wt = buf[i] + a0 * w0 + a1 * w1;
        const double y = b0 * wt + b1 * w0 + b0 * w1;
        w1 = w0; w0 = wt;
        if (i <= ed) sm[i] = y;
    }
}

__global__ void hv_sample_kernel(const double* __restrict__ sm, long nfr, double frame_period, float zero_below, float* __restrict__ out,
                                 long nout, const int* __restrict__ info, int* __restrict__ status) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && status) *status = info[2];
    if (i >= nout) return;
    long j = hv_round((double)i * frame_period / 1000.0 * 1000.0);
    j = j < nfr - 1 ? j : nfr - 1;
    const double v = sm[j];
    out[i] = v < (double)zero_below ? 0.f : (float)v;
}

struct HvPlan {
    long L, text, ylen, ypad_len, ld, nfr, nout, ecap, chan_cap, scratch_cap; int nch, scap;
    size_t o_t1, o_t2, o_mean, o_ypad, o_bf0, o_hlen, o_filt, o_events, o_ecount, o_raw, o_cand0, o_cand, o_score, o_cand2, o_score2,
           o_base, o_s1, o_s2, o_s3, o_s4, o_sm, o_st, o_ed, o_xst, o_xed, o_keep, o_order, o_gst, o_ged, o_woff, o_soff, o_chan, o_chs, o_ms,
           o_scratch, o_info, total;
};

static HvPlan hv_plan(long L, double fs, double f0_floor, double f0_ceil, double frame_period) {
    HvPlan p{};
    p.L = L; p.text = L + 18; p.ylen = (L - 1) / 2 + 1; p.ypad_len = p.ylen + 2 * HV_YPAD; p.ld = (p.ylen + 2 + 15) / 16 * 16;
    p.nfr = (long)(1000.0 * (double)L / fs / 1.0) + 1;
    p.nout = (long)(1000.0 * (double)L / fs / frame_period) + 1;
    p.nch = 1 + (int)(log2((f0_ceil * 1.1) / (f0_floor * 0.9)) * 40.0);
    p.ecap = p.ylen / 2 + 2;
    p.scap = (int)(p.nfr / 2 + 2);
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) / 256 * 256; return at; };
    p.o_t1 = take(p.text * 8); p.o_t2 = take(p.text * 8); p.o_mean = take(8); p.o_ypad = take(p.ypad_len * 8);
    p.o_bf0 = take(p.nch * 8); p.o_hlen = take(p.nch * 4);
    p.o_filt = take((size_t)p.nch * p.ld * 8);
    p.o_events = take((size_t)p.nch * 4 * p.ecap * 8); p.o_ecount = take((size_t)p.nch * 4 * 4);
    p.o_raw = take((size_t)p.nch * p.nfr * 8);
    p.o_cand0 = take((size_t)p.nfr * HV_NC * 8);
    p.o_cand = take((size_t)p.nfr * HV_NS * 8); p.o_score = take((size_t)p.nfr * HV_NS * 8);
    p.o_cand2 = take((size_t)p.nfr * HV_NS * 8); p.o_score2 = take((size_t)p.nfr * HV_NS * 8);
    p.o_base = take(p.nfr * 8); p.o_s1 = take(p.nfr * 8); p.o_s2 = take(p.nfr * 8); p.o_s3 = take(p.nfr * 8); p.o_s4 = take(p.nfr * 8);
    p.o_sm = take(p.nfr * 8);
    p.o_st = take(p.scap * 4); p.o_ed = take(p.scap * 4); p.o_xst = take(p.scap * 4); p.o_xed = take(p.scap * 4);
    p.o_keep = take(p.scap * 4); p.o_order = take(p.scap * 4); p.o_gst = take(p.scap * 4); p.o_ged = take(p.scap * 4);
    p.o_woff = take(((size_t)p.scap + 1) * 8); p.o_soff = take(((size_t)p.scap + 1) * 8);
    // sections after step 2 are >= 7 frames long and >= 1 frame apart; after step 4 >= 7 long and >= 9 apart
    p.chan_cap = p.nfr + (long)(2 * HV_MARG + 1) * (p.nfr / 8 + 1);
    p.scratch_cap = p.nfr + 600 + (long)(2 * HV_SMOOTH_MARG + 1) * (p.nfr / 16 + 2);
    p.o_chan = take((size_t)p.chan_cap * 8); p.o_chs = take((size_t)p.chan_cap * 8); p.o_ms = take(p.nfr * 8);
    p.o_scratch = take((size_t)p.scratch_cap * 8);
    p.o_info = take(64);
    p.total = o;
    return p;
}

static int hv_check(int32_t sample_rate, int64_t L, float f0_floor, float f0_ceil, float frame_period) {
    KN_REQUIRE(sample_rate == 16000, "f0_harvest: the path runs at 16 kHz (decimation by 2 to 8 kHz); got %d", sample_rate);
    KN_REQUIRE(L >= 1600 && L < (1L << 30), "f0_harvest: need 0.1 s .. 18 h of audio (got %ld samples)", (long)L);
    KN_REQUIRE(f0_floor >= 40.f && f0_ceil > f0_floor && f0_ceil <= 1600.f && frame_period >= 1.f, "f0_harvest: bad range / frame period");
    return KNNSVC_OK;
}

}  // namespace

extern "C" int knnsvc_f0_harvest_workspace(int64_t L, int32_t sample_rate, float f0_floor, float f0_ceil, float frame_period,
                                           int64_t* n_frames, int64_t* bytes) {
    KN_REQUIRE(n_frames && bytes, "f0_harvest_workspace: null pointer");
    const int rc = hv_check(sample_rate, L, f0_floor, f0_ceil, frame_period);
    if (rc) return rc;
    const HvPlan p = hv_plan((long)L, (double)sample_rate, (double)f0_floor, (double)f0_ceil, (double)frame_period);
    *n_frames = p.nout; *bytes = (int64_t)p.total;
    return KNNSVC_OK;
}

extern "C" int knnsvc_f0_harvest(const float* x, int64_t L, int32_t sample_rate, float f0_floor, float f0_ceil, float frame_period,
                                 float zero_below, float* f0, int64_t n_frames, void* workspace, int64_t workspace_bytes,
                                 int32_t* status, void* stream) {
    KN_REQUIRE(x && f0 && workspace, "f0_harvest: null pointer");
    const int rc0 = hv_check(sample_rate, L, f0_floor, f0_ceil, frame_period);
    if (rc0) return rc0;
    const double fs = (double)sample_rate, afs = fs / 2.0;
    const HvPlan p = hv_plan((long)L, fs, (double)f0_floor, (double)f0_ceil, (double)frame_period);
    KN_REQUIRE(n_frames == p.nout, "f0_harvest: n_frames %ld, expected %ld", (long)n_frames, p.nout);
    KN_REQUIRE(workspace_bytes >= (int64_t)p.total, "f0_harvest: workspace %ld bytes, need %ld", (long)workspace_bytes, (long)p.total);
    KN_REQUIRE(((uintptr_t)workspace & 255) == 0, "f0_harvest: workspace must be 256-byte aligned");
    {   // the longest filter has to fit the bank kernel's LDS tile
        const double b0 = (double)f0_floor * 0.9 * pow(2.0, 1.0 / 40.0);
        KN_REQUIRE((long)(afs / b0 * 2.0 + 0.5) <= HV_HMAX - 2 && (long)(afs / b0 * 2.0 + 0.5) + 1 <= HV_YPAD, "f0_harvest: f0_floor too low");
        KN_REQUIRE(2 * (int)(1.5 * afs / (double)f0_floor + 1.0) + 1 <= 384, "f0_harvest: f0_floor too low for the refinement window");
    }
    char* ws = (char*)workspace;
    auto D = [&](size_t off) { return (double*)(ws + off); };
    auto I = [&](size_t off) { return (int*)(ws + off); };
    auto LL = [&](size_t off) { return (long*)(ws + off); };
    hipStream_t st = (hipStream_t)stream;
    int rc;
#define HV_LAUNCH(name, grid, block, ...)                                              \
    hipLaunchKernelGGL(name, dim3 grid, dim3 block, 0, st, __VA_ARGS__);               \
    if ((rc = knnsvc_check_launch("f0_harvest/" #name))) return rc;

    if ((rc = kn_zero_async(ws + p.o_info, 64, st))) return rc;
    const long nchunk = cdiv64(p.text, 64);
    HV_LAUNCH(hv_iir_kernel<true>, ((unsigned)cdiv64(nchunk, 256)), (256), x, (const double*)nullptr, D(p.o_t1), p.text, p.L);
    HV_LAUNCH(hv_iir_kernel<false>, ((unsigned)cdiv64(nchunk, 256)), (256), x, (const double*)D(p.o_t1), D(p.o_t2), p.text, p.L);
    const long nbeg = 2 - 2 * p.ylen + p.L;
    HV_LAUNCH(hv_mean_kernel, (1), (1024), (const double*)D(p.o_t2), nbeg, p.ylen, D(p.o_mean));
    HV_LAUNCH(hv_center_kernel, ((unsigned)cdiv64(p.ypad_len, 256)), (256), (const double*)D(p.o_t2), nbeg, p.ylen,
              (const double*)D(p.o_mean), D(p.o_ypad), p.ypad_len);
    HV_LAUNCH(hv_channels_kernel, ((unsigned)cdiv64(p.nch, 64)), (64), (double)f0_floor * 0.9, p.nch, afs, D(p.o_bf0), I(p.o_hlen));
    HV_LAUNCH(hv_bank_kernel, ((unsigned)cdiv64(p.ylen, HV_TILE), (unsigned)p.nch), (256), (const double*)D(p.o_ypad), p.ylen,
              (const double*)D(p.o_bf0), (const int*)I(p.o_hlen), afs, D(p.o_filt), p.ld);
    HV_LAUNCH(hv_events_kernel, ((unsigned)(p.nch * 4)), (256), (const double*)D(p.o_filt), p.ld, p.ylen, D(p.o_events), p.ecap,
              I(p.o_ecount));
    HV_LAUNCH(hv_raw_kernel, ((unsigned)cdiv64(p.nfr, 256), (unsigned)p.nch), (256), (const double*)D(p.o_events), p.ecap,
              (const int*)I(p.o_ecount), (const double*)D(p.o_bf0), afs, (double)f0_floor, (double)f0_ceil, p.nfr, D(p.o_raw));
    HV_LAUNCH(hv_detect_kernel, ((unsigned)cdiv64(p.nfr, 256)), (256), (const double*)D(p.o_raw), p.nch, p.nfr, D(p.o_cand0), I(p.o_info));
    HV_LAUNCH(hv_refine_kernel, ((unsigned)cdiv64(p.nfr * 7, 4)), (256), (const double*)D(p.o_ypad), p.ylen, (const double*)D(p.o_cand0),
              p.nfr, afs, (double)f0_floor, (double)f0_ceil, D(p.o_cand), D(p.o_score));
    HV_LAUNCH(hv_reliable_kernel, ((unsigned)p.nfr), (128), (const double*)D(p.o_cand), (const double*)D(p.o_score), p.nfr, D(p.o_cand2),
              D(p.o_score2));
    HV_LAUNCH(hv_base_kernel, ((unsigned)cdiv64(p.nfr, 256)), (256), (const double*)D(p.o_cand2), (const double*)D(p.o_score2), p.nfr,
              D(p.o_base));
    HV_LAUNCH(hv_step1_kernel, ((unsigned)cdiv64(p.nfr, 256)), (256), (const double*)D(p.o_base), p.nfr, 0.008, D(p.o_s1), D(p.o_s2));
    HV_LAUNCH(hv_step2_kernel, (1), (1024), (const double*)D(p.o_s1), D(p.o_s2), p.nfr, 6, I(p.o_st), I(p.o_ed), p.scap, LL(p.o_woff),
              p.chan_cap, I(p.o_info));
    HV_LAUNCH(hv_extend_kernel, (256), (256), (const double*)D(p.o_s2), (const double*)D(p.o_cand2), p.nfr, (const int*)I(p.o_st),
              (const int*)I(p.o_ed), (const long*)LL(p.o_woff), (const double*)D(p.o_score2), D(p.o_chan), D(p.o_chs), I(p.o_xst), I(p.o_xed), I(p.o_keep),
              (const int*)I(p.o_info), 0.18);
    HV_LAUNCH(hv_merge_kernel, (1), (1024), (const double*)D(p.o_s2), (const double*)D(p.o_chs), D(p.o_ms), p.nfr,
              (const int*)I(p.o_st), (const int*)I(p.o_xst), (const int*)I(p.o_xed), I(p.o_keep), (const long*)LL(p.o_woff),
              (const double*)D(p.o_chan), I(p.o_order), D(p.o_s3), D(p.o_s4), I(p.o_gst), I(p.o_ged), p.scap, LL(p.o_soff), p.scratch_cap, 9, I(p.o_info));
    if ((rc = kn_zero_async(ws + p.o_sm, (size_t)p.nfr * 8, st))) return rc;
    HV_LAUNCH(hv_smooth_kernel, ((unsigned)cdiv64(p.scap, 64)), (64), (const double*)D(p.o_s4), p.nfr, (const int*)I(p.o_gst),
              (const int*)I(p.o_ged), (const long*)LL(p.o_soff), D(p.o_scratch), D(p.o_sm), (const int*)I(p.o_info));
    HV_LAUNCH(hv_sample_kernel, ((unsigned)cdiv64(p.nout, 256)), (256), (const double*)D(p.o_sm), p.nfr, (double)frame_period, zero_below, f0,
              p.nout, (const int*)I(p.o_info), status);
#undef HV_LAUNCH
    return KNNSVC_OK;
}
