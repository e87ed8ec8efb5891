"""Oracle: Harvest f0 estimation as the reference calls it (ddsp_prematch_dataset.py:121-128:
``pyworld.harvest(x.double(), fs=16000, f0_floor=65, f0_ceil=1047, frame_period=20)``, values below 80 Hz zeroed).

pyworld (WORLD, M. Morise; pinned ``^0.3.5`` in the reference's pyproject.toml) is a third-party dependency that is ABSENT
offline, so this is a restatement of the published algorithm (M. Morise, "Harvest: A high-performance fundamental frequency
estimator from speech signals", Interspeech 2017, and the structure of WORLD's harvest.cpp) — NOT a port validated against the
library itself.  PINNED on the only pyworld outputs available: the two harvest tracks the reference ships next to its sample
clips (``sample_content/*_f0.npy``, 3002 frames each, committed under tests/golden/sample_content_full/).  On the full
minute of both clips this restatement gives the same voicing decision and the same pitch (to the fixtures' fp32 rounding,
1.5e-5 Hz) on all 6004 frames; tests/test_oracle_golden.py checks 12 s heads on the CPU, tests/test_gpu_f0.py the full clips
through the GPU front end, which in turn is compared with this file.

Pipeline: decimate to 8 kHz (zero-phase IIR) and remove DC -> bank of Nuttall-windowed band-pass filters, 40 per octave ->
per channel, f0 from four kinds of zero-crossing intervals -> per frame (1 ms), candidates = channel runs of >= 10 agreeing
channels -> overlap with +-3 neighbouring frames -> refinement by instantaneous frequency of up to six harmonics ->
removal of candidates without a neighbour within 5 % -> contour selection (jump removal, short-section removal, extension,
merging, gap filling) -> zero-phase Butterworth smoothing -> sampling at the requested frame period.
Test infrastructure only."""
from __future__ import annotations

import numpy as np

_EPS = 1e-12


def _round(x):
    """matlab_round: half away from zero."""
    return np.where(x > 0, np.floor(x + 0.5), np.ceil(x - 0.5)).astype(np.int64)


def _iir(x, a, b):
    """y[n] = b0 w[n] + b1 w[n-1] + b1 w[n-2] + b0 w[n-3],  w[n] = x[n] + a0 w[n-1] + a1 w[n-2] + a2 w[n-3]."""
    from scipy.signal import lfilter
    return lfilter([b[0], b[1], b[1], b[0]], [1.0, -a[0], -a[1], -a[2]], x)


def decimate2(x: np.ndarray) -> np.ndarray:
    """WORLD's decimate(x, r = 2): reflect-extend by 9 samples, third-order IIR forward and backward, every second sample."""
    a = (0.041156734567757189, -0.42599112459189636, 0.041037215479961225)
    b = (0.16797464681802227, 0.50392394045406674)
    nf, n, r = 9, len(x), 2
    t = np.concatenate([2 * x[0] - x[nf:0:-1], x, 2 * x[-1] - x[n - 2:n - 2 - nf:-1]])
    t = _iir(t, a, b)[::-1]
    t = _iir(t, a, b)[::-1]
    nout = (n - 1) // r + 1
    nbeg = r - r * nout + n
    idx = np.arange(nbeg, n + nf, r) + nf - 1
    return t[idx[idx < len(t)]][:nout]


def _nuttall(n):
    i = np.arange(n) / (n - 1.0)
    return 0.355768 - 0.487396 * np.cos(2 * np.pi * i) + 0.144232 * np.cos(4 * np.pi * i) - 0.012604 * np.cos(6 * np.pi * i)


def _interp1_extrap(x, y, xi):
    """WORLD's interp1: linear, first / last segment extended outside the data."""
    k = np.clip(np.searchsorted(x, xi, side="right"), 1, len(x) - 1)
    s = (xi - x[k - 1]) / (x[k] - x[k - 1])
    return y[k - 1] + s * (y[k] - y[k - 1])


def _zero_cross(sig, fs):
    """Negative-going zero crossings -> (interval centre times, interval f0)."""
    e = np.nonzero((sig[:-1] > 0) & (sig[1:] <= 0))[0] + 1                    # 1-based edge positions
    if len(e) < 2:
        return None
    fine = e - sig[e - 1] / (sig[e] - sig[e - 1])
    return (fine[:-1] + fine[1:]) / 2.0 / fs, fs / (fine[1:] - fine[:-1])


def _raw_candidates(y, y_spec, fft_size, fs, boundary_f0, f0_floor, f0_ceil, tpos):
    half = int(_round(np.float64(fs / boundary_f0 * 2.0)))
    bpf = np.zeros(fft_size)
    n = 2 * half + 1
    bpf[:n] = _nuttall(n) * np.cos(2 * np.pi * boundary_f0 * np.arange(-half, half + 1) / fs)
    filt = np.fft.irfft(np.fft.rfft(bpf) * y_spec, fft_size)[half + 1:half + 1 + len(y)].copy()
    sets = []
    z = _zero_cross(filt, fs); sets.append(z)
    filt = -filt
    z = _zero_cross(filt, fs); sets.append(z)
    d = filt[:-1] - filt[1:]
    z = _zero_cross(d, fs); sets.append(z)
    z = _zero_cross(-d, fs); sets.append(z)
    if any(s is None or len(s[0]) < 3 for s in sets):
        return np.zeros(len(tpos))
    f = np.mean([_interp1_extrap(s[0], s[1], tpos) for s in sets], 0)
    bad = (f > boundary_f0 * 1.1) | (f < boundary_f0 * 0.9) | (f > f0_ceil) | (f < f0_floor)
    f[bad] = 0.0
    return f


def _detect_candidates(raw):
    """raw [channels, frames] -> candidates [frames, max_count]: mean over every run of >= 10 voiced channels."""
    nch, nfr = raw.shape
    v = (raw > 0).astype(np.int8)
    v[0] = 0; v[-1] = 0
    d = np.diff(v, axis=0)
    out = []
    mx = 0
    for i in range(nfr):
        st = np.nonzero(d[:, i] == 1)[0] + 1
        ed = np.nonzero(d[:, i] == -1)[0] + 1
        c = [raw[s:e, i].mean() for s, e in zip(st, ed) if e - s >= 10]
        out.append(c); mx = max(mx, len(c))
    cand = np.zeros((nfr, max(mx, 1)))
    for i, c in enumerate(out):
        cand[i, :len(c)] = c
    return cand


def _overlap(cand, n=3):
    nfr, nc = cand.shape
    out = np.zeros((nfr, nc * (2 * n + 1)))
    out[:, :nc] = cand
    for i in range(1, n + 1):
        out[i:, nc * i:nc * (i + 1)] = cand[:nfr - i]
        out[:nfr - i, nc * (i + n):nc * (i + n + 1)] = cand[i:]
    return out


def _refine(y, fs, tpos, cand, f0_floor, f0_ceil):
    """Instantaneous-frequency refinement of every candidate (GetRefinedF0), batched by window length."""
    nfr, nc = cand.shape
    f0 = np.zeros_like(cand); score = np.zeros_like(cand)
    fi, ci = np.nonzero(cand > 0)
    cf = cand[fi, ci]
    half = (1.5 * fs / cf + 1.0).astype(np.int64)
    ylen = len(y)
    for h in np.unique(half):
        sel = np.nonzero(half == h)[0]
        n = 2 * h + 1
        wl = n / fs
        fft_size = int(2 ** (2 + int(np.log(n) / np.log(2.0))))
        bt = np.arange(-h, h + 1) / fs
        for c0 in range(0, len(sel), 20000):
            s = sel[c0:c0 + 20000]
            pos = tpos[fi[s]]; f = cf[s]
            base = _round((pos + bt[0]) * fs + 0.001)[:, None] + np.arange(n)[None]
            t = (base - 1.0) / fs - pos[:, None]
            win = 0.42 + 0.5 * np.cos(2 * np.pi * t / wl) + 0.08 * np.cos(4 * np.pi * t / wl)
            dw = np.empty_like(win)
            dw[:, 0] = -win[:, 1] / 2; dw[:, 1:-1] = -(win[:, 2:] - win[:, :-2]) / 2; dw[:, -1] = win[:, -2] / 2
            seg = y[np.clip(base - 1, 0, ylen - 1)]
            ms = np.fft.rfft(seg * win, fft_size); ds = np.fft.rfft(seg * dw, fft_size)
            num = ms.real * ds.imag - ms.imag * ds.real
            pw = ms.real ** 2 + ms.imag ** 2
            nh = np.minimum((fs / 2.0 / f).astype(np.int64), 6)
            k = np.arange(1, 7)[None]
            idx = _round(f[:, None] * fft_size / fs * k)
            valid = k <= nh[:, None]
            idx = np.where(valid, np.minimum(idx, fft_size // 2), 0)
            p = np.take_along_axis(pw, idx, 1); q = np.take_along_axis(num, idx, 1)
            inst = np.where(p == 0, 0.0, idx * fs / fft_size + q / np.where(p == 0, 1, p) * fs / 2.0 / np.pi)
            amp = np.sqrt(p) * valid
            rf = (amp * inst).sum(1) / ((amp * k).sum(1) + _EPS)
            dev = (np.abs((inst / k - f[:, None]) / f[:, None]) * valid).sum(1) / nh
            sc = 1.0 / (_EPS + dev)
            bad = (rf < f0_floor) | (rf > f0_ceil) | (sc < 2.5)
            rf[bad] = 0; sc[bad] = 0
            f0[fi[s], ci[s]] = rf; score[fi[s], ci[s]] = sc
    return f0, score


def _select_best(ref, cands, allowed):
    """SelectBestF0, vectorised over frames: candidate with the smallest relative distance <= allowed (last wins ties)."""
    err = np.abs(ref[:, None] - cands) / np.where(ref[:, None] == 0, 1, ref[:, None])
    err = np.where(err > allowed, np.inf, err)
    # the C loop keeps the LAST candidate among equal errors (tmp > best continues; equal replaces)
    j = err.shape[1] - 1 - np.argmin(err[:, ::-1], 1)
    best = np.take_along_axis(err, j[:, None], 1)[:, 0]
    return np.where(np.isfinite(best), np.take_along_axis(cands, j[:, None], 1)[:, 0], 0.0), np.where(np.isfinite(best), best, allowed)


def _remove_unreliable(f0, score, thr=0.05):
    tmp = f0.copy()
    for j in range(f0.shape[1]):
        ref = tmp[1:-1, j]
        _b1, e1 = _select_best(ref, tmp[2:], 1.0)
        _b2, e2 = _select_best(ref, tmp[:-2], 1.0)
        bad = (ref != 0) & (np.minimum(e1, e2) > thr)
        f0[1:-1, j][bad] = 0; score[1:-1, j][bad] = 0
    return f0, score


def _boundaries(f0):
    v = (f0 > 0).astype(np.int8)
    v[0] = 0; v[-1] = 0
    d = np.diff(v)
    return list(zip(np.nonzero(d == 1)[0] + 1, np.nonzero(d == -1)[0]))      # inclusive [start, end]


def _sel1(ref, cands, allowed):
    best, be = 0.0, allowed
    for c in cands:
        t = abs(ref - c) / ref
        if t > be:
            continue
        best, be = c, t
    return best


def _extend(f0, origin, last, shift, cand, allowed):
    tmp = f0[origin]; so = origin; count = 0
    for i in range(abs(last - origin) + 1):
        p = origin + shift * i + shift
        f0[p] = _sel1(tmp, cand[p], allowed)
        if f0[p] == 0:
            count += 1
        else:
            tmp = f0[p]; count = 0; so = p
        if count == 4:
            break
    return so


def _search_score(f, cands, scores):
    m = (cands == f)
    return scores[m].max() if m.any() else 0.0


def _fix_contour(cand, score, allowed1=0.008, vmin=6, allowed=0.18, gap=9):
    """FixF0Contour with WORLD's tuned constants: 0.8 % per-ms jump, 6-frame minimum section, 18 % while extending,
    gaps shorter than 9 frames bridged."""
    nfr = len(cand)
    base = np.where(score.max(1) > 0, np.take_along_axis(cand, score.argmax(1)[:, None], 1)[:, 0], 0.0)
    # step 1: jumps
    s1 = np.zeros(nfr)
    with np.errstate(divide="ignore", invalid="ignore"):
        ref = base[1:-1] * 2 - base[:-2]
        a = np.abs((base[2:] - ref) / ref) > allowed1
        b = np.abs(base[2:] - base[1:-1]) / base[1:-1] > allowed1
    s1[2:] = np.where((base[2:] != 0) & ~(a & b), base[2:], 0.0)
    # step 2: short sections
    s2 = s1.copy()
    for st, ed in _boundaries(s1):
        if ed - st < vmin:
            s2[st:ed + 1] = 0
    # step 3: extend / merge
    secs = _boundaries(s2)
    out = s2.copy()
    chans = []
    for st, ed in secs:
        ch = np.zeros(nfr); ch[st:ed + 1] = s2[st:ed + 1]
        ed2 = _extend(ch, ed, min(nfr - 2, ed + 100), 1, cand, allowed)
        st2 = _extend(ch, st, max(1, st - 100), -1, cand, allowed)
        m = ch[st2:ed2].mean() if ed2 > st2 else 0.0
        if m > 0 and 2200.0 / m < ed2 - st2:
            chans.append((st2, ed2, ch))
    if chans:
        # MergeF0 as the library does it, quirks included (the shipped tracks depend on them): the order comes from an
        # insertion pass that moves a new element at most ONE place forward; the merge starts from channel 0 whatever the
        # order says; and the running bounds live in channel 0's slots of the boundary list.
        bl = [[c[0], c[1]] for c in chans]
        order = list(range(len(chans)))
        for i in range(1, len(chans)):
            for j in range(i - 1, -1, -1):
                if bl[order[j]][0] > bl[order[i]][0]:
                    order[i], order[j] = order[j], order[i]
                else:
                    break
        merged = chans[0][2].copy()
        for i in range(1, len(chans)):
            st, ed = bl[order[i]]
            ch = chans[order[i]][2]
            b0, b1 = bl[0]
            if st - b1 > 0:
                merged[st:ed + 1] = ch[st:ed + 1]; bl[0] = [st, ed]
            elif b0 <= st and b1 >= ed:
                pass
            else:
                sc1 = sum(_search_score(merged[i2], cand[i2], score[i2]) for i2 in range(st, b1 + 1))
                sc2 = sum(_search_score(ch[i2], cand[i2], score[i2]) for i2 in range(st, b1 + 1))
                if sc1 > sc2:
                    merged[b1:ed + 1] = ch[b1:ed + 1]
                else:
                    merged[st:ed + 1] = ch[st:ed + 1]
                bl[0][1] = ed
        out = merged
    # step 4: short gaps
    s4 = out.copy()
    bl = _boundaries(out)
    for (s_a, e_a), (s_b, e_b) in zip(bl[:-1], bl[1:]):
        dist = s_b - e_a - 1
        if dist >= gap:
            continue
        t0 = out[e_a] + 1; t1 = out[s_b] - 1
        co = (t1 - t0) / (dist + 1.0)
        s4[e_a + 1:s_b] = t0 + co * np.arange(1, dist + 1)
    return s4


def _smooth(f0):
    from scipy.signal import lfilter
    b = [0.0078202080334971724, 0.015640416066994345, 0.0078202080334971724]
    a = [1.0, -1.7347257688092754, 0.76600660094326412]
    lag = 300
    c = np.concatenate([np.zeros(lag), f0, np.zeros(lag)])
    out = np.zeros(len(f0))
    for st, ed in _boundaries(c):
        x = np.zeros(len(c)); x[st:ed + 1] = c[st:ed + 1]
        x[:st] = x[st]; x[ed + 1:] = x[ed]
        y = lfilter(b, a, x)[::-1]
        y = lfilter(b, a, y)[::-1]
        out[st - lag:ed + 1 - lag] = y[st:ed + 1]
    return out


def harvest(x: np.ndarray, fs: int = 16000, f0_floor: float = 65.0, f0_ceil: float = 1047.0, frame_period: float = 20.0,
            zero_below: float = 80.0) -> np.ndarray:
    """-> f0 [int(1000 len(x) / fs / frame_period) + 1] float64, 0 = unvoiced."""
    x = np.asarray(x, np.float64)
    adj_floor, adj_ceil = f0_floor * 0.9, f0_ceil * 1.1
    nch = 1 + int(np.log2(adj_ceil / adj_floor) * 40)
    bf0 = adj_floor * 2.0 ** ((np.arange(nch) + 1) / 40.0)
    ratio = max(min(int(_round(np.float64(fs / 8000.0))), 12), 1)
    assert ratio == 2, "only the reference's 16 kHz case is restated"
    y = decimate2(x)
    afs = fs / ratio
    ylen = len(y)
    fft_size = int(2 ** (int(np.log(ylen + 5 + 2 * int(2.0 * afs / bf0[0])) / np.log(2.0)) + 1))
    y = y - y.mean()
    y_spec = np.fft.rfft(y, fft_size)
    nfr = int(1000.0 * len(x) / fs / 1.0) + 1
    tpos = np.arange(nfr) / 1000.0
    raw = np.stack([_raw_candidates(y, y_spec, fft_size, afs, b, f0_floor, f0_ceil, tpos) for b in bf0])
    cand = _overlap(_detect_candidates(raw))
    cand, score = _refine(y, afs, tpos, cand, f0_floor, f0_ceil)
    cand, score = _remove_unreliable(cand, score)
    f0 = _smooth(_fix_contour(cand, score))
    n_out = int(1000.0 * len(x) / fs / frame_period) + 1
    idx = np.minimum(nfr - 1, _round(np.arange(n_out) * frame_period / 1000.0 * 1000.0))
    out = f0[idx]
    out[out < zero_below] = 0.0
    return out
