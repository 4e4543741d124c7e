// Audio ingest for the scoring path: WAV and FLAC files -> mono float32 -> crop or tile to a
// fixed length, many files at a time on host threads, written straight into the caller's
// (pinned) batch buffer.
//
// Replaces, for files that are already at the target sample rate, what the reference does per
// utterance in Python: `librosa.load(path, sr=16000)` followed by `pad(X, 64600)`
// (data_utils_SSL.py:58-65, :109-115, :130-135).  librosa decodes through libsndfile to float32:
// integer PCM of b bits is divided by 2^(b-1) (8-bit WAV is unsigned, offset 128), channels are
// averaged.  Resampling is not done here: a file at another rate is reported, not converted.
//
// The FLAC decoder follows the published format (RFC 9639): STREAMINFO, frame header with
// CRC-8, CONSTANT / VERBATIM / FIXED / LPC subframes, Rice-coded residuals with 4- and 5-bit
// parameters and escape partitions, wasted bits, left-side / right-side / mid-side stereo,
// frame CRC-16.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/slsmi.h"

namespace {

struct FileBytes {
    std::vector<uint8_t> data;
    bool read(const char* path) {
        FILE* f = fopen(path, "rb");
        if (!f) return false;
        if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return false; }
        const long n = ftell(f);
        if (n < 0 || fseek(f, 0, SEEK_SET) != 0) { fclose(f); return false; }
        data.resize((size_t)n);
        const size_t got = n ? fread(data.data(), 1, (size_t)n, f) : 0;
        fclose(f);
        return got == (size_t)n;
    }
};

struct Audio {
    int sample_rate = 0;
    int channels = 0;
    std::vector<float> mono;  // channel average, scaled to [-1, 1)
    bool header_only = false;  // in: stop once the frame count is known from the header
    int64_t frames = -1;       // out (header_only): frames the header announces; -1: decode to find out
};

// No clip this library is meant for comes near this (4.6 hours at 16 kHz); a header or a stream of
// constant subframes that claims more is refused instead of being allocated.
constexpr uint64_t MAX_SAMPLES = 1ull << 28;

// ------------------------------------------------------------------ WAV (RIFF)
inline uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

int decode_wav(const std::vector<uint8_t>& d, Audio* a) {
    if (d.size() < 12 || memcmp(d.data(), "RIFF", 4) != 0 || memcmp(d.data() + 8, "WAVE", 4) != 0)
        return SLSMI_ERR_AUDIO_FORMAT;
    size_t pos = 12;
    int fmt_tag = 0, bits = 0, block_align = 0;
    bool have_fmt = false;
    while (pos + 8 <= d.size()) {
        const uint8_t* ck = d.data() + pos;
        size_t len = le32(ck + 4);
        const size_t body = pos + 8;
        if (memcmp(ck, "fmt ", 4) == 0) {
            if (len < 16 || body + 16 > d.size()) return SLSMI_ERR_AUDIO_FORMAT;
            fmt_tag = le16(d.data() + body);
            a->channels = le16(d.data() + body + 2);
            a->sample_rate = (int)le32(d.data() + body + 4);
            block_align = le16(d.data() + body + 12);
            bits = le16(d.data() + body + 14);
            if (fmt_tag == 0xFFFE) {  // WAVE_FORMAT_EXTENSIBLE: the sub-format's first two bytes are the tag
                if (len < 40 || body + 40 > d.size()) return SLSMI_ERR_AUDIO_FORMAT;
                fmt_tag = le16(d.data() + body + 24);
            }
            have_fmt = true;
        } else if (memcmp(ck, "data", 4) == 0) {
            if (!have_fmt) return SLSMI_ERR_AUDIO_FORMAT;
            if (body + len > d.size()) len = d.size() - body;  // truncated file: take what is there
            const int ch = a->channels;
            const int bytes = bits / 8;
            if (ch < 1 || ch > 8 || bytes < 1 || block_align != ch * bytes) return SLSMI_ERR_AUDIO_FORMAT;
            const bool is_float = fmt_tag == 3;
            if (!(fmt_tag == 1 && (bits == 8 || bits == 16 || bits == 24 || bits == 32)) &&
                !(is_float && (bits == 32 || bits == 64)))
                return SLSMI_ERR_AUDIO_FORMAT;
            const size_t frames = len / (size_t)block_align;
            if (frames > MAX_SAMPLES) return SLSMI_ERR_AUDIO_DATA;
            if (a->header_only) {
                a->frames = (int64_t)frames;
                return SLSMI_OK;
            }
            a->mono.resize(frames);
            const uint8_t* p = d.data() + body;
            const float inv_ch = 1.0f / (float)ch;
            for (size_t i = 0; i < frames; ++i) {
                float s = 0.f;
                for (int c = 0; c < ch; ++c, p += bytes) {
                    float v;
                    if (is_float) {
                        if (bits == 32) { memcpy(&v, p, 4); }
                        else { double dv; memcpy(&dv, p, 8); v = (float)dv; }
                    } else if (bits == 8) {
                        v = (float)((int)p[0] - 128) * (1.0f / 128.0f);
                    } else if (bits == 16) {
                        v = (float)(int16_t)le16(p) * (1.0f / 32768.0f);
                    } else if (bits == 24) {
                        const int32_t x = (int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24) >> 8;
                        v = (float)x * (1.0f / 8388608.0f);
                    } else {
                        v = (float)((double)(int32_t)le32(p) * (1.0 / 2147483648.0));
                    }
                    s += v;
                }
                a->mono[i] = ch == 1 ? s : s * inv_ch;
            }
            return SLSMI_OK;
        }
        pos = body + len + (len & 1);  // chunks are word aligned
    }
    return SLSMI_ERR_AUDIO_FORMAT;
}

// ------------------------------------------------------------------ FLAC
struct BitReader {
    const uint8_t* base;
    size_t len;
    size_t ip = 0;     // next byte to load into the window
    uint64_t acc = 0;  // bits are consumed from the top; bits below the n valid ones are zero
    int n = 0;         // valid bits in acc
    BitReader(const uint8_t* b, const uint8_t* e) : base(b), len((size_t)(e - b)) {}
    inline void fill() {
        while (n <= 56) {  // bytes past the end read as zero; overrun() reports it
            const uint64_t byte = ip < len ? base[ip] : 0;
            ++ip;
            acc |= byte << (56 - n);
            n += 8;
        }
    }
    inline uint32_t bits(int k) {  // k in 0..32
        if (k == 0) return 0;
        if (n < k) fill();
        const uint32_t v = (uint32_t)(acc >> (64 - k));
        acc <<= k;
        n -= k;
        return v;
    }
    inline int32_t sbits(int k) {
        if (k == 0) return 0;
        const uint32_t v = bits(k);
        return (int32_t)(v << (32 - k)) >> (32 - k);
    }
    inline uint32_t unary() {  // number of zero bits before the next one bit
        uint32_t z = 0;
        for (;;) {
            if (n == 0) fill();
            if (acc == 0) {  // the whole window is zeros
                z += (uint32_t)n;
                n = 0;
                if (ip > len + 16) return z;  // ran off the end: overrun() is true
                continue;
            }
            const int lz = __builtin_clzll(acc);
            z += (uint32_t)lz;
            acc = lz == 63 ? 0 : acc << (lz + 1);
            n -= lz + 1;
            return z;
        }
    }
    inline void align() {
        const int r = n & 7;
        acc <<= r;
        n -= r;
    }
    // byte position of the next unread bit (call when byte aligned)
    inline const uint8_t* position() const { return base + ip - (size_t)(n / 8); }
    inline bool overrun() const { return ip - (size_t)(n / 8) > len; }
};

uint8_t crc8(const uint8_t* p, size_t n) {
    uint8_t c = 0;
    for (size_t i = 0; i < n; ++i) {
        c ^= p[i];
        for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : (c << 1));
    }
    return c;
}

struct Crc16Table {
    uint16_t t[256];
    Crc16Table() {
        for (int i = 0; i < 256; ++i) {
            uint16_t c = (uint16_t)(i << 8);
            for (int b = 0; b < 8; ++b) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : (c << 1));
            t[i] = c;
        }
    }
};

uint16_t crc16(const uint8_t* p, size_t n) {
    static const Crc16Table table;  // thread-safe one-time initialisation
    uint16_t c = 0;
    for (size_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ table.t[(c >> 8) ^ p[i]]);
    return c;
}

inline uint64_t load_be64(const uint8_t* p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return __builtin_bswap64(v);
}

bool read_residual(BitReader& br, int32_t* out, int blocksize, int order) {
    const int method = (int)br.bits(2);
    if (method > 1) return false;
    const int pbits = method == 0 ? 4 : 5;
    const uint32_t esc = method == 0 ? 15u : 31u;
    const int porder = (int)br.bits(4);
    const int parts = 1 << porder;
    if ((blocksize >> porder) << porder != blocksize && porder > 0) return false;
    int idx = order;
    for (int pt = 0; pt < parts; ++pt) {
        int count = (blocksize >> porder) - (pt == 0 ? order : 0);
        if (count < 0) return false;
        const uint32_t param = br.bits(pbits);
        if (param == esc) {
            const int raw = (int)br.bits(5);
            for (int i = 0; i < count; ++i) out[idx++] = br.sbits(raw);
        } else {
            // hot loop: the reader's window lives in locals; one unaligned 8-byte load refills it
            uint64_t acc = br.acc;
            int n = br.n;
            size_t ip = br.ip;
            const uint8_t* base = br.base;
            const size_t len = br.len;
            const int k = (int)param;
            int i = 0;
            for (; i < count; ++i) {
                if (ip + 8 <= len) {
                    if (n < 56) {
                        acc |= load_be64(base + ip) >> n;
                        ip += (size_t)((63 - n) >> 3);
                        n |= 56;
                    }
                } else {
                    break;  // near the end of the file: finish with the checked reader
                }
                if (acc == 0) break;  // a run of 56+ zero bits: rare, checked reader
                const int lz = __builtin_clzll(acc);
                if (lz + 1 + k > n) break;
                acc <<= lz;        // lz <= 55 here
                acc <<= 1;
                n -= lz + 1;
                const uint32_t low = k ? (uint32_t)(acc >> (64 - k)) : 0u;
                acc <<= k;
                n -= k;
                const uint32_t u = ((uint32_t)lz << k) | low;
                out[idx++] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
            }
            // bits below the n valid ones must be zero for the checked reader
            acc = n ? acc & (~0ull << (64 - n)) : 0;
            br.acc = acc;
            br.n = n;
            br.ip = ip;
            for (; i < count; ++i) {
                const uint32_t q = br.unary();
                const uint32_t u = (q << param) | br.bits((int)param);
                out[idx++] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
            }
        }
        if (br.overrun()) return false;
    }
    return idx == blocksize;
}

template <int ORDER>
void lpc_restore(int32_t* out, int blocksize, const int32_t* coef, int shift) {
    for (int i = ORDER; i < blocksize; ++i) {
        int64_t s = 0;
#pragma GCC unroll 32
        for (int j = 0; j < ORDER; ++j) s += (int64_t)coef[j] * out[i - 1 - j];
        out[i] += (int32_t)(s >> shift);
    }
}

void lpc_restore_any(int32_t* out, int blocksize, const int32_t* coef, int shift, int order) {
    switch (order) {
        case 1: lpc_restore<1>(out, blocksize, coef, shift); break;
        case 2: lpc_restore<2>(out, blocksize, coef, shift); break;
        case 3: lpc_restore<3>(out, blocksize, coef, shift); break;
        case 4: lpc_restore<4>(out, blocksize, coef, shift); break;
        case 5: lpc_restore<5>(out, blocksize, coef, shift); break;
        case 6: lpc_restore<6>(out, blocksize, coef, shift); break;
        case 7: lpc_restore<7>(out, blocksize, coef, shift); break;
        case 8: lpc_restore<8>(out, blocksize, coef, shift); break;
        case 9: lpc_restore<9>(out, blocksize, coef, shift); break;
        case 10: lpc_restore<10>(out, blocksize, coef, shift); break;
        case 11: lpc_restore<11>(out, blocksize, coef, shift); break;
        case 12: lpc_restore<12>(out, blocksize, coef, shift); break;
        default:
            for (int i = order; i < blocksize; ++i) {
                int64_t s = 0;
                for (int j = 0; j < order; ++j) s += (int64_t)coef[j] * out[i - 1 - j];
                out[i] += (int32_t)(s >> shift);
            }
    }
}

bool read_subframe(BitReader& br, int32_t* out, int blocksize, int bps) {
    if (br.bits(1) != 0) return false;
    const int type = (int)br.bits(6);
    int wasted = 0;
    if (br.bits(1)) wasted = (int)br.unary() + 1;
    bps -= wasted;
    if (bps < 1 || bps > 33) return false;
    auto sample = [&](int nb) -> int64_t {  // up to 33 bits (side channel of 32-bit audio)
        if (nb <= 32) return br.sbits(nb);
        const int64_t hi = br.sbits(nb - 32);
        return (hi << 32) | br.bits(32);
    };
    if (type == 0) {  // CONSTANT
        const int32_t v = (int32_t)sample(bps);
        for (int i = 0; i < blocksize; ++i) out[i] = v;
    } else if (type == 1) {  // VERBATIM
        for (int i = 0; i < blocksize; ++i) out[i] = (int32_t)sample(bps);
    } else if (type >= 8 && type <= 12) {  // FIXED, order type - 8
        const int order = type - 8;
        if (order > blocksize) return false;
        for (int i = 0; i < order; ++i) out[i] = (int32_t)sample(bps);
        if (!read_residual(br, out, blocksize, order)) return false;
        // residual -> signal, in place.  Unsigned arithmetic: a valid stream never leaves 32 bits, and a corrupt
        // one wraps around instead of overflowing a signed integer (its frame then fails the CRC-16 anyway)
        const auto u = [&](int i) { return (uint32_t)out[i]; };
        switch (order) {
            case 0: break;
            case 1: for (int i = 1; i < blocksize; ++i) out[i] = (int32_t)(u(i) + u(i - 1)); break;
            case 2: for (int i = 2; i < blocksize; ++i) out[i] = (int32_t)(u(i) + 2u * u(i - 1) - u(i - 2)); break;
            case 3: for (int i = 3; i < blocksize; ++i) out[i] = (int32_t)(u(i) + 3u * u(i - 1) - 3u * u(i - 2) + u(i - 3)); break;
            case 4: for (int i = 4; i < blocksize; ++i) out[i] = (int32_t)(u(i) + 4u * u(i - 1) - 6u * u(i - 2) + 4u * u(i - 3) - u(i - 4)); break;
        }
    } else if (type >= 32) {  // LPC, order type - 31
        const int order = type - 31;
        if (order > blocksize) return false;
        for (int i = 0; i < order; ++i) out[i] = (int32_t)sample(bps);
        const int prec = (int)br.bits(4) + 1;
        if (prec == 16) return false;
        const int shift = br.sbits(5);
        if (shift < 0) return false;
        int32_t coef[32];
        for (int i = 0; i < order; ++i) coef[i] = br.sbits(prec);
        if (!read_residual(br, out, blocksize, order)) return false;
        lpc_restore_any(out, blocksize, coef, shift, order);
    } else {
        return false;  // reserved
    }
    if (wasted)
        for (int i = 0; i < blocksize; ++i) out[i] = (int32_t)((uint32_t)out[i] << wasted);
    return !br.overrun();
}

int decode_flac(const std::vector<uint8_t>& d, Audio* a) {
    if (d.size() < 42 || memcmp(d.data(), "fLaC", 4) != 0) return SLSMI_ERR_AUDIO_FORMAT;
    size_t pos = 4;
    int bps = 0, max_block = 0;
    uint64_t total = 0;
    bool last = false, have_info = false;
    while (!last) {
        if (pos + 4 > d.size()) return SLSMI_ERR_AUDIO_FORMAT;
        last = (d[pos] & 0x80) != 0;
        const int type = d[pos] & 0x7F;
        const size_t len = ((size_t)d[pos + 1] << 16) | ((size_t)d[pos + 2] << 8) | d[pos + 3];
        pos += 4;
        if (pos + len > d.size()) return SLSMI_ERR_AUDIO_FORMAT;
        if (type == 0) {
            if (len < 34) return SLSMI_ERR_AUDIO_FORMAT;
            const uint8_t* s = d.data() + pos;
            max_block = (s[2] << 8) | s[3];
            a->sample_rate = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
            a->channels = ((s[12] >> 1) & 7) + 1;
            bps = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
            total = ((uint64_t)(s[13] & 0x0F) << 32) | ((uint64_t)s[14] << 24) | (s[15] << 16) | (s[16] << 8) | s[17];
            have_info = true;
        }
        pos += len;
    }
    if (!have_info || a->channels < 1 || a->channels > 8 || bps < 4 || bps > 32 || max_block < 16)
        return SLSMI_ERR_AUDIO_FORMAT;
    if (total > MAX_SAMPLES) return SLSMI_ERR_AUDIO_DATA;
    if (a->header_only && total) {  // STREAMINFO knows the length (0 there means "unknown": decode)
        a->frames = (int64_t)total;
        return SLSMI_OK;
    }
    const int ch = a->channels;
    const float scale = 1.0f / (float)(1ull << (bps - 1));
    a->mono.clear();
    // the header's count is untrusted: reserve no more than the file could plausibly hold (FLAC rarely goes
    // below one byte per sample; a stream of constant subframes may, and then the vector just grows)
    if (total) a->mono.reserve((size_t)std::min<uint64_t>(total, (uint64_t)d.size() * 4 + 65536));
    std::vector<int32_t> buf((size_t)ch * (size_t)(max_block > 0 ? max_block : 65535));
    const uint8_t* end = d.data() + d.size();
    while (pos + 6 <= d.size()) {
        if (total && a->mono.size() >= total) break;  // anything after the last frame is not audio
        const uint8_t* fr = d.data() + pos;
        if (fr[0] != 0xFF || (fr[1] & 0xFE) != 0xF8) return SLSMI_ERR_AUDIO_DATA;  // frame sync
        BitReader br(fr + 2, end);
        const int bs_code = (int)br.bits(4);
        const int sr_code = (int)br.bits(4);
        const int ch_code = (int)br.bits(4);
        const int ss_code = (int)br.bits(3);
        if (br.bits(1) != 0) return SLSMI_ERR_AUDIO_DATA;
        // UTF-8-like coded frame or sample number: its value is not needed, only its length
        uint32_t first = br.bits(8);
        int extra = 0;
        if (first & 0x80) {
            while (first & (0x80u >> (extra + 1))) ++extra;
            if (extra == 0 || extra > 6) return SLSMI_ERR_AUDIO_DATA;
        }
        for (int i = 0; i < extra; ++i) (void)br.bits(8);
        int blocksize;
        if (bs_code == 0) return SLSMI_ERR_AUDIO_DATA;
        else if (bs_code == 1) blocksize = 192;
        else if (bs_code <= 5) blocksize = 576 << (bs_code - 2);
        else if (bs_code == 6) blocksize = (int)br.bits(8) + 1;
        else if (bs_code == 7) blocksize = (int)br.bits(16) + 1;
        else blocksize = 256 << (bs_code - 8);
        if (sr_code == 12) (void)br.bits(8);
        else if (sr_code == 13 || sr_code == 14) (void)br.bits(16);
        else if (sr_code == 15) return SLSMI_ERR_AUDIO_DATA;
        const uint8_t* hdr_end = br.position();
        if (hdr_end + 1 > end) return SLSMI_ERR_AUDIO_DATA;
        if (crc8(fr, (size_t)(hdr_end - fr)) != *hdr_end) return SLSMI_ERR_AUDIO_DATA;
        (void)br.bits(8);
        int fbps = bps;
        switch (ss_code) {
            case 0: break;
            case 1: fbps = 8; break;
            case 2: fbps = 12; break;
            case 4: fbps = 16; break;
            case 5: fbps = 20; break;
            case 6: fbps = 24; break;
            case 7: fbps = 32; break;
            default: return SLSMI_ERR_AUDIO_DATA;
        }
        int nch;
        if (ch_code < 8) nch = ch_code + 1;
        else if (ch_code <= 10) nch = 2;
        else return SLSMI_ERR_AUDIO_DATA;
        if (nch != ch || fbps != bps || (size_t)blocksize * ch > buf.size()) return SLSMI_ERR_AUDIO_DATA;
        for (int c = 0; c < nch; ++c) {
            // the difference channel of a stereo pair carries one more bit
            const bool side = (ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1);
            if (!read_subframe(br, buf.data() + (size_t)c * blocksize, blocksize, fbps + (side ? 1 : 0)))
                return SLSMI_ERR_AUDIO_DATA;
        }
        br.align();
        const uint8_t* body_end = br.position();
        if (body_end + 2 > end) return SLSMI_ERR_AUDIO_DATA;
        if (crc16(fr, (size_t)(body_end - fr)) != (uint16_t)((body_end[0] << 8) | body_end[1])) return SLSMI_ERR_AUDIO_DATA;
        int32_t* c0 = buf.data();
        int32_t* c1 = buf.data() + blocksize;
        if (ch_code == 8) {  // left, side
            for (int i = 0; i < blocksize; ++i) c1[i] = c0[i] - c1[i];
        } else if (ch_code == 9) {  // side, right
            for (int i = 0; i < blocksize; ++i) c0[i] += c1[i];
        } else if (ch_code == 10) {  // mid, side
            for (int i = 0; i < blocksize; ++i) {
                const int32_t side = c1[i];
                const int32_t mid = (int32_t)(((uint32_t)c0[i] << 1) | ((uint32_t)side & 1));
                c0[i] = (mid + side) >> 1;
                c1[i] = (mid - side) >> 1;
            }
        }
        const size_t base = a->mono.size();
        if (base + (size_t)blocksize > MAX_SAMPLES) return SLSMI_ERR_AUDIO_DATA;
        a->mono.resize(base + (size_t)blocksize);
        if (ch == 1) {
            for (int i = 0; i < blocksize; ++i) a->mono[base + i] = (float)c0[i] * scale;
        } else {
            const float inv_ch = 1.0f / (float)ch;
            for (int i = 0; i < blocksize; ++i) {
                float s = 0.f;
                for (int c = 0; c < ch; ++c) s += (float)buf[(size_t)c * blocksize + i] * scale;
                a->mono[base + i] = s * inv_ch;
            }
        }
        pos = (size_t)(body_end + 2 - d.data());
    }
    if (total && a->mono.size() > total) a->mono.resize((size_t)total);
    if (a->header_only) a->frames = (int64_t)a->mono.size();
    return SLSMI_OK;
}

int decode_any(const char* path, Audio* a) {
    try {  // nothing may be thrown across the C boundary or out of a worker thread (std::bad_alloc on a hostile file)
        FileBytes f;
        if (!path || !f.read(path)) return SLSMI_ERR_AUDIO_OPEN;
        if (f.data.size() >= 4 && memcmp(f.data.data(), "fLaC", 4) == 0) return decode_flac(f.data, a);
        if (f.data.size() >= 4 && memcmp(f.data.data(), "RIFF", 4) == 0) return decode_wav(f.data, a);
        return SLSMI_ERR_AUDIO_FORMAT;
    } catch (...) {
        return SLSMI_ERR_AUDIO_DATA;
    }
}

// data_utils_SSL.py:58-65: crop to `cut`, or repeat the clip until it fills `cut`
void pad_into(const std::vector<float>& x, float* out, int64_t cut) {
    const int64_t n = (int64_t)x.size();
    if (n >= cut) {
        memcpy(out, x.data(), (size_t)cut * sizeof(float));
        return;
    }
    int64_t done = 0;
    while (done < cut) {
        const int64_t k = n < cut - done ? n : cut - done;
        memcpy(out + done, x.data(), (size_t)k * sizeof(float));
        done += k;
    }
}

// ---------------------------------------------------------------- sample-rate conversion
// Band-limited interpolation (J. O. Smith's windowed-sinc scheme): y(t) = sum_k x[k] h(t - k) with
// h = rolloff * sinc(rolloff * tau) * Kaiser window over ZEROS zero crossings per side, stretched by the rate ratio when
// going down so that the cut-off follows the lower Nyquist frequency.  The window's right half is tabulated at
// 2^PRECISION points per zero crossing and read with linear interpolation, so any pair of rates takes the same path.
constexpr int RS_ZEROS = 64;
constexpr int RS_PRECISION = 9;
constexpr double RS_ROLLOFF = 0.9475937167399596;
constexpr double RS_BETA = 14.769656459379492;

double bessel_i0(double x) {
    double sum = 1.0, term = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 200; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-17 * sum) break;
    }
    return sum;
}

const std::vector<float>& resample_window() {
    static const std::vector<float> win = [] {
        const int per = 1 << RS_PRECISION, n = RS_ZEROS * per;
        std::vector<float> w((size_t)n + 2, 0.f);
        const double i0b = bessel_i0(RS_BETA);
        for (int i = 0; i <= n; ++i) {
            const double tau = (double)i / per;                 // in zero crossings of the unit-rate sinc
            const double a = RS_ROLLOFF * tau * M_PI;
            const double sinc = i == 0 ? 1.0 : sin(a) / a;
            const double r = tau / RS_ZEROS;
            const double kaiser = r >= 1.0 ? 0.0 : bessel_i0(RS_BETA * sqrt(1.0 - r * r)) / i0b;
            w[(size_t)i] = (float)(RS_ROLLOFF * sinc * kaiser);
        }
        return w;
    }();
    return win;
}

// x at src_rate -> ceil(n * dst_rate / src_rate) samples at dst_rate
void resample(const std::vector<float>& x, int src_rate, int dst_rate, std::vector<float>* out) {
    const int64_t n = (int64_t)x.size();
    const double ratio = (double)dst_rate / (double)src_rate;
    const int64_t m = (int64_t)ceil((double)n * ratio);
    out->assign((size_t)m, 0.f);
    const std::vector<float>& win = resample_window();
    const int per = 1 << RS_PRECISION;
    const int64_t wlen = (int64_t)RS_ZEROS * per;      // last valid table index
    const double scale = ratio < 1.0 ? ratio : 1.0;
    const double step = scale * per;                    // table positions per input sample
    auto tap = [&](double pos) -> double {
        const int64_t i = (int64_t)pos;
        const double f = pos - (double)i;
        return (double)win[(size_t)i] + f * ((double)win[(size_t)i + 1] - (double)win[(size_t)i]);
    };
    for (int64_t t = 0; t < m; ++t) {
        const double time = (double)t / ratio;
        const int64_t k0 = (int64_t)time;
        const double frac = time - (double)k0;
        double acc = 0.0;
        double pos = frac * step;                       // left wing: x[k0], x[k0 - 1], ...
        for (int64_t k = k0; k >= 0 && pos <= (double)wlen; --k, pos += step)
            if (k < n) acc += tap(pos) * (double)x[(size_t)k];
        pos = (1.0 - frac) * step;                      // right wing: x[k0 + 1], ...
        for (int64_t k = k0 + 1; k < n && pos <= (double)wlen; ++k, pos += step) acc += tap(pos) * (double)x[(size_t)k];
        (*out)[(size_t)t] = (float)(acc * scale);
    }
}

}  // namespace

extern "C" {

int slsmi_audio_resample(const float* in, int64_t n, int src_rate, int dst_rate, float* out, int64_t capacity,
                         int64_t* n_out) {
    if (!in || !out || n < 0 || capacity < 0 || src_rate < 1 || dst_rate < 1 || (uint64_t)n > MAX_SAMPLES) return SLSMI_ERR_INVALID;
    if ((double)n * dst_rate / src_rate > (double)MAX_SAMPLES) return SLSMI_ERR_INVALID;
    try {
        std::vector<float> x(in, in + n), y;
        if (src_rate == dst_rate)
            y = x;
        else
            resample(x, src_rate, dst_rate, &y);
        const int64_t k = (int64_t)y.size() < capacity ? (int64_t)y.size() : capacity;
        memcpy(out, y.data(), (size_t)k * sizeof(float));
        if (n_out) *n_out = (int64_t)y.size();
        return SLSMI_OK;
    } catch (...) {
        return SLSMI_ERR_AUDIO_DATA;
    }
}

int slsmi_audio_info(const char* path, int* sample_rate, int* channels, int64_t* frames) {
    Audio a;
    a.header_only = true;  // WAV: the data chunk's length; FLAC: STREAMINFO's sample count (decodes only if that is 0)
    const int rc = decode_any(path, &a);
    if (rc != SLSMI_OK) return rc;
    if (sample_rate) *sample_rate = a.sample_rate;
    if (channels) *channels = a.channels;
    if (frames) *frames = a.frames;
    return SLSMI_OK;
}

int slsmi_audio_read(const char* path, float* out, int64_t capacity, int64_t* frames_out, int* sample_rate) {
    if (!out || capacity < 0) return SLSMI_ERR_INVALID;
    Audio a;
    const int rc = decode_any(path, &a);
    if (rc != SLSMI_OK) return rc;
    const int64_t n = (int64_t)a.mono.size() < capacity ? (int64_t)a.mono.size() : capacity;
    memcpy(out, a.mono.data(), (size_t)n * sizeof(float));
    if (frames_out) *frames_out = (int64_t)a.mono.size();
    if (sample_rate) *sample_rate = a.sample_rate;
    return SLSMI_OK;
}

int slsmi_audio_load_batch(const char* const* paths, int n, int64_t cut, int expect_rate, float* out, int threads,
                           int* status) {
    if (!paths || n < 0 || cut < 1 || !out) return SLSMI_ERR_INVALID;
    if (threads < 1) threads = 1;
    if (threads > n) threads = n > 0 ? n : 1;
    std::atomic<int> next{0};
    std::atomic<int> failures{0};
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) return;
            Audio a;
            int rc = decode_any(paths[i], &a);  // catches everything itself
            if (rc == SLSMI_OK && expect_rate > 0 && a.sample_rate != expect_rate) rc = SLSMI_ERR_AUDIO_RATE;
            if (rc == SLSMI_OK && expect_rate < 0 && a.sample_rate != -expect_rate) {  // convert to -expect_rate
                try {
                    if (a.sample_rate < 1 || (double)a.mono.size() * -expect_rate / a.sample_rate > (double)MAX_SAMPLES) {
                        rc = SLSMI_ERR_AUDIO_DATA;
                    } else {
                        std::vector<float> y;
                        resample(a.mono, a.sample_rate, -expect_rate, &y);
                        a.mono.swap(y);
                    }
                } catch (...) {
                    rc = SLSMI_ERR_AUDIO_DATA;
                }
            }
            if (rc == SLSMI_OK && a.mono.empty()) rc = SLSMI_ERR_AUDIO_DATA;
            float* dst = out + (size_t)i * (size_t)cut;
            if (rc == SLSMI_OK) {
                pad_into(a.mono, dst, cut);
            } else {
                memset(dst, 0, (size_t)cut * sizeof(float));
                failures.fetch_add(1);
            }
            if (status) status[i] = rc;
        }
    };
    std::vector<std::thread> pool;
    try {
        for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    } catch (...) {  // fewer threads than asked for: the ones that exist still share all the files
    }
    work();
    for (auto& t : pool) t.join();
    return failures.load() == 0 ? SLSMI_OK : SLSMI_ERR_AUDIO_DATA;
}

}  // extern "C"
