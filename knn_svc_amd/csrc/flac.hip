// FLAC container: decoder and a plain encoder, HOST code only (no kernel here; it lives in the library so that the host side has
// one native dependency).  The reference reads .flac through torchaudio.load (ddsp_prematch_dataset.py:332; the prematch
// builder globs *.wav and *.flac, :1469-1473 — LibriSpeech is FLAC) and writes .flac through pydub / ffmpeg
// (lib_ongaku_test.py:122-143).  Neither library is available offline; FLAC is lossless, so a conforming decoder returns
// exactly the samples any other decoder returns.  Format: RFC 9639.  Every frame's CRC-8 / CRC-16 is verified here and the
// caller verifies the stream's MD5 (audio_io.read_flac), so a decoding error cannot pass silently.
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace {

struct BitReader {
    const uint8_t* p; int64_t n; int64_t pos = 0;        // pos in bits
    bool fail = false;
    BitReader(const uint8_t* d, int64_t size) : p(d), n(size) {}
    inline uint32_t bit() {
        if ((pos >> 3) >= n) { fail = true; return 0; }
        const uint32_t b = (p[pos >> 3] >> (7 - (pos & 7))) & 1u; ++pos; return b;
    }
    inline uint64_t bits(int k) {                        // k <= 57
        uint64_t v = 0;
        while (k > 0) {
            if ((pos >> 3) >= n) { fail = true; return 0; }
            const int avail = 8 - (int)(pos & 7), take = k < avail ? k : avail;
            v = (v << take) | ((p[pos >> 3] >> (avail - take)) & ((1u << take) - 1u));
            pos += take; k -= take;
        }
        return v;
    }
    inline int64_t sbits(int k) {
        if (k == 0) return 0;
        const uint64_t v = bits(k);
        return (int64_t)(v << (64 - k)) >> (64 - k);
    }
    inline uint32_t unary() {                            // zeros before the next one
        uint32_t q = 0;
        for (;;) {
            if ((pos >> 3) >= n) { fail = true; return q; }
            const uint32_t cur = (uint32_t)(p[pos >> 3] & (0xFFu >> (pos & 7)));
            if (cur) { const int lead = __builtin_clz(cur) - 24 - (int)(pos & 7); q += lead; pos += lead + 1; return q; }
            q += 8 - (int)(pos & 7); pos += 8 - (pos & 7);
        }
    }
    inline void align() { pos = (pos + 7) & ~7LL; }
};

uint8_t crc8(const uint8_t* d, int64_t n) {
    uint8_t c = 0;
    for (int64_t i = 0; i < n; ++i) { c ^= d[i]; for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : (c << 1)); }
    return c;
}
uint16_t crc16(const uint8_t* d, int64_t n) {
    static uint16_t tab[256]; static bool init = false;
    if (!init) { for (int i = 0; i < 256; ++i) { uint16_t c = (uint16_t)(i << 8); for (int b = 0; b < 8; ++b) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : (c << 1)); tab[i] = c; } init = true; }
    uint16_t c = 0;
    for (int64_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ tab[(c >> 8) ^ d[i]]);
    return c;
}

struct StreamInfo { int32_t sr, ch, bps; int64_t total; uint8_t md5[16]; int64_t audio_off; int32_t max_block; };

int parse_header(const uint8_t* d, int64_t n, StreamInfo* si) {
    int64_t o = 0;
    if (n >= 10 && !memcmp(d, "ID3", 3)) o = 10 + (((int64_t)(d[6] & 0x7F) << 21) | ((d[7] & 0x7F) << 14) | ((d[8] & 0x7F) << 7) | (d[9] & 0x7F));
    KN_REQUIRE(o + 4 <= n && !memcmp(d + o, "fLaC", 4), "flac: no fLaC marker");
    o += 4;
    bool have = false;
    for (;;) {
        KN_REQUIRE(o + 4 <= n, "flac: truncated metadata");
        const int last = d[o] >> 7, type = d[o] & 0x7F;
        const int64_t len = ((int64_t)d[o + 1] << 16) | (d[o + 2] << 8) | d[o + 3];
        o += 4;
        KN_REQUIRE(o + len <= n, "flac: truncated metadata block");
        if (type == 0) {
            KN_REQUIRE(len >= 34, "flac: short STREAMINFO");
            const uint8_t* s = d + o;
            si->max_block = (s[2] << 8) | s[3];
            si->sr = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
            si->ch = ((s[12] >> 1) & 7) + 1;
            si->bps = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
            si->total = ((int64_t)(s[13] & 0xF) << 32) | ((int64_t)s[14] << 24) | (s[15] << 16) | (s[16] << 8) | s[17];
            memcpy(si->md5, s + 18, 16);
            have = true;
        }
        o += len;
        if (last) break;
    }
    KN_REQUIRE(have, "flac: no STREAMINFO block");
    KN_REQUIRE(si->sr > 0 && si->bps >= 4 && si->bps <= 32, "flac: bad STREAMINFO (rate %d, %d bits)", si->sr, si->bps);
    si->audio_off = o;
    return KNNSVC_OK;
}

int decode_residual(BitReader& br, int blocksize, int order, int64_t* s) {
    const int method = (int)br.bits(2);
    KN_REQUIRE(method < 2, "flac: reserved residual coding method");
    const int pbits = method ? 5 : 4, esc = method ? 31 : 15;
    const int porder = (int)br.bits(4), parts = 1 << porder;
    KN_REQUIRE((blocksize >> porder) << porder == blocksize || porder == 0, "flac: partition order does not divide the block");
    int i = order;
    for (int pt = 0; pt < parts; ++pt) {
        int cnt = (blocksize >> porder) - (pt == 0 ? order : 0);
        KN_REQUIRE(cnt >= 0 && i + cnt <= blocksize, "flac: bad residual partition");
        const int k = (int)br.bits(pbits);
        if (k == esc) {
            const int nb = (int)br.bits(5);
            for (int e = 0; e < cnt; ++e) s[i++] = br.sbits(nb);
        } else {
            for (int e = 0; e < cnt; ++e) {
                const uint64_t q = br.unary();
                const uint64_t u = (q << k) | (k ? br.bits(k) : 0);
                s[i++] = (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
            }
        }
        if (br.fail) return knnsvc_fail(KNNSVC_EINVAL, "flac: truncated residual");
    }
    return KNNSVC_OK;
}

int decode_subframe(BitReader& br, int blocksize, int bps, int64_t* s) {
    KN_REQUIRE(br.bit() == 0, "flac: subframe padding bit set");
    const int type = (int)br.bits(6);
    int wasted = 0;
    if (br.bit()) wasted = (int)br.unary() + 1;
    bps -= wasted;
    KN_REQUIRE(bps >= 1, "flac: wasted bits exceed the sample size");
    if (type == 0) {
        const int64_t v = br.sbits(bps);
        for (int i = 0; i < blocksize; ++i) s[i] = v;
    } else if (type == 1) {
        for (int i = 0; i < blocksize; ++i) s[i] = br.sbits(bps);
    } else if (type >= 8 && type <= 12) {
        const int order = type - 8;
        KN_REQUIRE(order <= blocksize, "flac: predictor order exceeds the block");
        for (int i = 0; i < order; ++i) s[i] = br.sbits(bps);
        const int rc = decode_residual(br, blocksize, order, s);
        if (rc) return rc;
        for (int i = order; i < blocksize; ++i) {
            int64_t p = 0;
            if (order == 1) p = s[i - 1];
            else if (order == 2) p = 2 * s[i - 1] - s[i - 2];
            else if (order == 3) p = 3 * s[i - 1] - 3 * s[i - 2] + s[i - 3];
            else if (order == 4) p = 4 * s[i - 1] - 6 * s[i - 2] + 4 * s[i - 3] - s[i - 4];
            s[i] += p;
        }
    } else if (type >= 32) {
        const int order = type - 31;
        KN_REQUIRE(order <= blocksize, "flac: predictor order exceeds the block");
        for (int i = 0; i < order; ++i) s[i] = br.sbits(bps);
        const int prec = (int)br.bits(4) + 1;
        KN_REQUIRE(prec != 16, "flac: reserved LPC precision");
        const int shift = (int)br.sbits(5);
        KN_REQUIRE(shift >= 0, "flac: negative LPC shift");
        int64_t coef[32];
        for (int j = 0; j < order; ++j) coef[j] = br.sbits(prec);
        const int rc = decode_residual(br, blocksize, order, s);
        if (rc) return rc;
        for (int i = order; i < blocksize; ++i) {
            int64_t acc = 0;
            for (int j = 0; j < order; ++j) acc += coef[j] * s[i - 1 - j];
            s[i] += acc >> shift;
        }
    } else {
        return knnsvc_fail(KNNSVC_EINVAL, "flac: reserved subframe type %d", type);
    }
    if (wasted) for (int i = 0; i < blocksize; ++i) s[i] = (int64_t)((uint64_t)s[i] << wasted);
    if (br.fail) return knnsvc_fail(KNNSVC_EINVAL, "flac: truncated subframe");
    return KNNSVC_OK;
}

// ---- encoder helpers ------------------------------------------------------------------------------------
struct BitWriter {
    uint8_t* p; int64_t cap; int64_t pos = 0; bool fail = false;
    BitWriter(uint8_t* d, int64_t c) : p(d), cap(c) {}
    inline void put(uint64_t v, int k) {
        for (int b = k - 1; b >= 0; --b) {
            if ((pos >> 3) >= cap) { fail = true; return; }
            if ((pos & 7) == 0) p[pos >> 3] = 0;
            p[pos >> 3] |= (uint8_t)(((v >> b) & 1u) << (7 - (pos & 7)));
            ++pos;
        }
    }
    inline void unary(uint32_t q) { for (uint32_t i = 0; i < q; ++i) put(0, 1); put(1, 1); }
    inline void align() { while (pos & 7) put(0, 1); }
};

void put_utf8(BitWriter& bw, uint64_t v) {
    if (v < 0x80) { bw.put(v, 8); return; }
    int nb = 2; uint64_t lim = 0x800;
    while (v >= lim && nb < 7) { ++nb; lim <<= 5; }
    bw.put(((0xFFu << (8 - nb)) & 0xFF) | (v >> (6 * (nb - 1))), 8);
    for (int i = nb - 2; i >= 0; --i) bw.put(0x80 | ((v >> (6 * i)) & 0x3F), 8);
}

}  // namespace

extern "C" int knnsvc_flac_info(const uint8_t* data, int64_t size, int32_t* sample_rate, int32_t* channels, int32_t* bits,
                                int64_t* total_samples, uint8_t* md5) {
    KN_REQUIRE(data && sample_rate && channels && bits && total_samples, "flac_info: null pointer");
    StreamInfo si{};
    const int rc = parse_header(data, size, &si);
    if (rc) return rc;
    *sample_rate = si.sr; *channels = si.ch; *bits = si.bps; *total_samples = si.total;
    if (md5) memcpy(md5, si.md5, 16);
    return KNNSVC_OK;
}

extern "C" int knnsvc_flac_decode(const uint8_t* data, int64_t size, int32_t* out, int64_t capacity, int64_t* decoded) {
    KN_REQUIRE(data && out && decoded, "flac_decode: null pointer");
    StreamInfo si{};
    int rc = parse_header(data, size, &si);
    if (rc) return rc;
    int64_t o = si.audio_off, done = 0;
    int64_t* buf = (int64_t*)malloc(sizeof(int64_t) * 65536 * 8);
    KN_REQUIRE(buf, "flac_decode: out of memory");
    auto bail = [&](int code) { free(buf); return code; };
    while (o + 2 <= size) {
        if (!(data[o] == 0xFF && (data[o + 1] & 0xFE) == 0xF8)) {          // padding / trailing tags after the last frame
            if (si.total && done >= si.total) break;
            ++o; continue;
        }
        BitReader br(data + o, size - o);
        br.bits(14); br.bit(); br.bit();
        const int bs_code = (int)br.bits(4), sr_code = (int)br.bits(4), ch_code = (int)br.bits(4), ss_code = (int)br.bits(3);
        br.bit();
        {   // UTF-8 coded frame / sample number
            const uint32_t first = (uint32_t)br.bits(8);
            int extra = 0;
            if (first >= 0xFE) extra = 6; else if (first >= 0xFC) extra = 5; else if (first >= 0xF8) extra = 4; else if (first >= 0xF0) extra = 3;
            else if (first >= 0xE0) extra = 2; else if (first >= 0xC0) extra = 1;
            for (int i = 0; i < extra; ++i) br.bits(8);
        }
        int blocksize;
        if (bs_code == 1) blocksize = 192; else if (bs_code >= 2 && bs_code <= 5) blocksize = 576 << (bs_code - 2);
        else if (bs_code == 6) blocksize = (int)br.bits(8) + 1; else if (bs_code == 7) blocksize = (int)br.bits(16) + 1;
        else if (bs_code >= 8) blocksize = 256 << (bs_code - 8); else return bail(knnsvc_fail(KNNSVC_EINVAL, "flac: reserved block size code"));
        if (sr_code == 12) br.bits(8); else if (sr_code == 13 || sr_code == 14) br.bits(16);
        else if (sr_code == 15) return bail(knnsvc_fail(KNNSVC_EINVAL, "flac: invalid sample rate code"));
        const int hdr_bytes = (int)(br.pos >> 3);
        const uint8_t want8 = (uint8_t)br.bits(8);
        if (br.fail || crc8(data + o, hdr_bytes) != want8) {               // a false sync inside another frame's payload cannot happen
            if (si.total && done >= si.total) break;                       // between frames; after the last one it is trailing data
            return bail(knnsvc_fail(KNNSVC_EINVAL, "flac: frame header CRC mismatch at byte %ld", (long)o));
        }
        int bps = si.bps;
        const int ss_tab[8] = {0, 8, 12, -1, 16, 20, 24, 32};
        if (ss_code) { if (ss_tab[ss_code] < 0) return bail(knnsvc_fail(KNNSVC_EINVAL, "flac: reserved sample size code")); bps = ss_tab[ss_code]; }
        int nch; int mode = 0;                                             // 0 independent, 1 left/side, 2 side/right, 3 mid/side
        if (ch_code < 8) nch = ch_code + 1; else if (ch_code <= 10) { nch = 2; mode = ch_code - 7; }
        else return bail(knnsvc_fail(KNNSVC_EINVAL, "flac: reserved channel assignment"));
        if (nch != si.ch || blocksize > 65536) return bail(knnsvc_fail(KNNSVC_EINVAL, "flac: frame disagrees with STREAMINFO"));
        for (int c = 0; c < nch; ++c) {
            const int side = (mode == 1 && c == 1) || (mode == 2 && c == 0) || (mode == 3 && c == 1);
            rc = decode_subframe(br, blocksize, bps + side, buf + (int64_t)c * 65536);
            if (rc) return bail(rc);
        }
        br.align();
        const int frame_bytes = (int)(br.pos >> 3);
        const uint16_t want16 = (uint16_t)br.bits(16);
        if (br.fail || crc16(data + o, frame_bytes) != want16) return bail(knnsvc_fail(KNNSVC_EINVAL, "flac: frame CRC mismatch at byte %ld", (long)o));
        int64_t* a = buf; int64_t* b = buf + 65536;
        if (mode == 1) for (int i = 0; i < blocksize; ++i) b[i] = a[i] - b[i];
        else if (mode == 2) for (int i = 0; i < blocksize; ++i) a[i] = a[i] + b[i];
        else if (mode == 3) for (int i = 0; i < blocksize; ++i) { const int64_t m = (a[i] << 1) | (b[i] & 1), s = b[i]; a[i] = (m + s) >> 1; b[i] = (m - s) >> 1; }
        int64_t take = blocksize;
        if (si.total && done + take > si.total) take = si.total - done;
        if (done + take > capacity) return bail(knnsvc_fail(KNNSVC_EINVAL, "flac_decode: output capacity %ld too small", (long)capacity));
        for (int c = 0; c < nch; ++c) for (int64_t i = 0; i < take; ++i) out[(int64_t)c * capacity + done + i] = (int32_t)buf[(int64_t)c * 65536 + i];
        done += take;
        o += frame_bytes + 2;
    }
    free(buf);
    KN_REQUIRE(!si.total || done == si.total, "flac: decoded %ld of %ld samples", (long)done, (long)si.total);
    *decoded = done;
    return KNNSVC_OK;
}

// Plain encoder: fixed block size 4096, independent channels, per channel the best of the fixed predictors 0-4 (or CONSTANT /
// VERBATIM), one Rice partition (5-bit parameter form, so that 24-bit residuals fit) with the best parameter.  pcm: [channels][n] int32 holding `bits`-bit signed samples.
extern "C" int knnsvc_flac_encode(const int32_t* pcm, int32_t channels, int64_t n, int32_t bits, int32_t sample_rate, const uint8_t* md5,
                                  uint8_t* out, int64_t capacity, int64_t* size) {
    KN_REQUIRE(pcm && out && size, "flac_encode: null pointer");
    KN_REQUIRE(channels >= 1 && channels <= 8 && n >= 0 && n < (1LL << 36), "flac_encode: bad shape");
    KN_REQUIRE(bits == 16 || bits == 24 || bits == 8 || bits == 20 || bits == 12, "flac_encode: sample size must be 8, 12, 16, 20 or 24");
    KN_REQUIRE(sample_rate > 0 && sample_rate < (1 << 20), "flac_encode: bad sample rate");
    const int BS = 4096;
    BitWriter bw(out, capacity);
    bw.put('f', 8); bw.put('L', 8); bw.put('a', 8); bw.put('C', 8);
    bw.put(0x80, 8); bw.put(34, 24);                                          // last metadata block, STREAMINFO
    bw.put(BS, 16); bw.put(BS, 16); bw.put(0, 24); bw.put(0, 24);
    bw.put((uint64_t)sample_rate, 20); bw.put((uint64_t)(channels - 1), 3); bw.put((uint64_t)(bits - 1), 5); bw.put((uint64_t)n, 36);
    for (int i = 0; i < 16; ++i) bw.put(md5 ? md5[i] : 0, 8);
    const int ss_code = bits == 8 ? 1 : bits == 12 ? 2 : bits == 16 ? 4 : bits == 20 ? 5 : 6;
    int64_t frame_no = 0;
    for (int64_t f0 = 0; f0 < n; f0 += BS, ++frame_no) {
        const int bsz = (int)(n - f0 < BS ? n - f0 : BS);
        const int64_t start = bw.pos >> 3;
        bw.put(0x3FFE, 14); bw.put(0, 1); bw.put(0, 1);
        bw.put(bsz == BS ? 12 : 7, 4); bw.put(0, 4); bw.put((uint64_t)(channels - 1), 4); bw.put((uint64_t)ss_code, 3); bw.put(0, 1);
        put_utf8(bw, (uint64_t)frame_no);
        if (bsz != BS) bw.put((uint64_t)(bsz - 1), 16);
        if (bw.fail) break;
        bw.put(crc8(out + start, (bw.pos >> 3) - start), 8);
        for (int c = 0; c < channels; ++c) {
            const int32_t* s = pcm + (int64_t)c * n + f0;
            bool constant = true;
            for (int i = 1; i < bsz; ++i) if (s[i] != s[0]) { constant = false; break; }
            if (constant) { bw.put(0, 1); bw.put(0, 6); bw.put(0, 1); bw.put((uint64_t)(uint32_t)s[0] & ((1ULL << bits) - 1), bits); continue; }
            int best_o = -1, best_k = 0; uint64_t best_bits = (uint64_t)bsz * bits;            // VERBATIM cost
            for (int o = 0; o <= 4 && o < bsz; ++o) {
                uint64_t sum = 0;
                for (int i = o; i < bsz; ++i) {
                    int64_t p = 0;
                    if (o == 1) p = s[i - 1]; else if (o == 2) p = 2LL * s[i - 1] - s[i - 2];
                    else if (o == 3) p = 3LL * s[i - 1] - 3LL * s[i - 2] + s[i - 3];
                    else if (o == 4) p = 4LL * s[i - 1] - 6LL * s[i - 2] + 4LL * s[i - 3] - s[i - 4];
                    const int64_t r = s[i] - p;
                    sum += (uint64_t)(r < 0 ? -r : r);
                }
                const uint64_t cnt = (uint64_t)(bsz - o);
                int k = 0;
                while (k < 30 && (cnt << (k + 1)) < 2 * sum + cnt) ++k;                       // mean |r| ~ 2^k
                for (int kk = (k > 0 ? k - 1 : 0); kk <= k + 1 && kk <= 30; ++kk) {
                    const uint64_t est = (uint64_t)o * bits + 11 + cnt * (kk + 1) + ((2 * sum + cnt) >> (kk + 1));
                    if (est < best_bits) { best_bits = est; best_o = o; best_k = kk; }
                }
            }
            if (best_o < 0) {
                bw.put(0, 1); bw.put(1, 6); bw.put(0, 1);
                for (int i = 0; i < bsz; ++i) bw.put((uint64_t)(uint32_t)s[i] & ((1ULL << bits) - 1), bits);
                continue;
            }
            const int o = best_o, k = best_k;
            bw.put(0, 1); bw.put((uint64_t)(8 + o), 6); bw.put(0, 1);
            for (int i = 0; i < o; ++i) bw.put((uint64_t)(uint32_t)s[i] & ((1ULL << bits) - 1), bits);
            bw.put(1, 2); bw.put(0, 4); bw.put((uint64_t)k, 5);                               // Rice2 (5-bit parameter), partition order 0
            for (int i = o; i < bsz; ++i) {
                int64_t p = 0;
                if (o == 1) p = s[i - 1]; else if (o == 2) p = 2LL * s[i - 1] - s[i - 2];
                else if (o == 3) p = 3LL * s[i - 1] - 3LL * s[i - 2] + s[i - 3];
                else if (o == 4) p = 4LL * s[i - 1] - 6LL * s[i - 2] + 4LL * s[i - 3] - s[i - 4];
                const int64_t r = s[i] - p;
                const uint64_t u = r >= 0 ? (uint64_t)r << 1 : (((uint64_t)(-r)) << 1) - 1;
                bw.unary((uint32_t)(u >> k));
                if (k) bw.put(u & ((1ULL << k) - 1), k);
            }
        }
        bw.align();
        if (bw.fail) break;
        bw.put(crc16(out + start, (bw.pos >> 3) - start), 16);
    }
    KN_REQUIRE(!bw.fail, "flac_encode: output capacity %ld too small", (long)capacity);
    *size = bw.pos >> 3;
    return KNNSVC_OK;
}
