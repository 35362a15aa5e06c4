// Progressive JPEG files (SOF2) for the device decoder: the entropy decoding of every scan on the host, into the block-major
// coefficient array jpeg_didct_kernel reads (jpeg_dec.hip) -- the image is then made on the device like a baseline file's.
//
// Why the host: an AC refinement scan (T.81 G.1.2.3) reads one correction bit per coefficient that is ALREADY non-zero, so where
// a symbol starts depends on the block's history over all earlier scans, not only on the bit string -- the self-synchronising
// decoder of jpeg_dec.hip has no form for that.  What the device keeps: dequantisation, the IDCT, the colour conversion, and
// everything behind them (the quality search reads the planes where they lie).  A progressive file therefore crosses PCIe as
// 2 bytes per coefficient instead of 4 bytes per decoded pixel after a host decode, and the caller needs no host codec.
//
// Behaviour follows image/jpeg (reader.go, scan.go: processSOS, refine, refineNonZeroes, reconstructProgressiveImage) as
// published -- restated from ITU T.81 Annex G, not from Go's source; bit-exact against the CPU restatement the tests hold
// (tests/test_jpeg_progressive.py), which libjpeg's progressive files pin coefficient by coefficient:
//   * coefficients are collected over all scans and dequantised once, with the tables in force at EOI;
//   * a one-component frame is h = v = 1; an interleaved scan walks the frame's MCUs, a one-component scan its component's
//     blocks in raster order, without data for blocks wholly outside the image;
//   * image/jpeg counts FRAME MCUs between restart markers in every scan where T.81 counts the scan's own: the two agree only
//     for components of one block per MCU, anything else with a restart interval is FNX_ERR_UNSUPPORTED (the host codec's call).
// r5, later: the SEQUENTIAL files the device's scan decoder has no form for are read here as well -- SOF1 (extended sequential),
// components in scans of their own or out of frame order, a Huffman table that assigns the all-ones code.
// Four-component files (Adobe CMYK / YCbCrK, all components 1 x 1) are read here whatever their frame type.
// Not handled (FNX_ERR_UNSUPPORTED, as for baseline files): 12-bit samples, arithmetic coding, chroma factors
// other than 1 x 1, a component no scan mentions, coefficients beyond 16 bits.
// Plain C++ with no device code: this file reads untrusted bytes and is part of the sanitizer builds (make asan / tsan).
#include <cstring>
#include <new>
#include <vector>
#include "common.hpp"

namespace fnx {

namespace {

const uint8_t UNZIG_P[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// MSB-first bits of an entropy-coded segment: 0xff 0x00 is one 0xff byte, any other 0xff ends the segment (the reader
// stays in front of it and hands out zeros, counted: a scan that consumed one of them ran past its data)
struct BitReader {
    const uint8_t *p, *end;
    uint64_t acc = 0;
    int cnt = 0, fake = 0;
    BitReader(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
    inline void fill()
    {
        if (cnt > 32) return;                    // a 16-bit code and a 16-bit value, or 32 correction bits, are there
        if (p + 4 <= end) {                      // four bytes at once when none of them is 0xff
            uint32_t w;
            std::memcpy(&w, p, 4);
            const uint32_t v = ~w;
            if (!((v - 0x01010101u) & ~v & 0x80808080u)) {
                acc = (acc << 32) | __builtin_bswap32(w);
                cnt += 32;
                p += 4;
                return;
            }
        }
        while (cnt <= 56) {
            uint32_t b = 0;
            if (p < end && *p != 0xff) b = *p++;
            else if (p + 1 < end && p[1] == 0x00) { b = 0xff; p += 2; }
            else fake += 8;
            acc = (acc << 8) | b;
            cnt += 8;
        }
    }
    inline uint32_t peek16() { fill(); return static_cast<uint32_t>(acc >> (cnt - 16)) & 0xffffu; }
    inline void skip(int nb) { cnt -= nb; }
    inline uint32_t bits(int nb)                 // nb <= 16
    {
        if (nb == 0) return 0;
        fill();
        cnt -= nb;
        return static_cast<uint32_t>(acc >> cnt) & ((1u << nb) - 1u);
    }
    inline uint32_t bit() { return bits(1); }
    inline uint32_t bits32(int nb)               // 1 <= nb <= 32
    {
        fill();
        cnt -= nb;
        return static_cast<uint32_t>((acc >> cnt) & ((1ull << nb) - 1ull));
    }
    bool overran() const { return cnt < fake; }
    void restart() { acc = 0; cnt = 0; fake = 0; }
};

struct HTab {
    uint16_t fast[512];                          // 9 bits of look-ahead -> length << 8 | symbol; 0: a longer code
    int32_t maxcode[17], valoff[17];             // per length: largest code (-1: none), value index of code 0
    uint8_t val[256];
    bool have = false;
};

inline int huff(BitReader &br, const HTab &t)
{
    const uint32_t look = br.peek16();
    const uint16_t e = t.fast[look >> 7];
    if (e) { br.skip(e >> 8); return e & 0xff; }
    for (int L = 10; L <= 16; L++) {
        const int32_t code = static_cast<int32_t>(look >> (16 - L));
        if (code <= t.maxcode[L]) { br.skip(L); return t.val[t.valoff[L] + code]; }
    }
    return -1;
}

inline int32_t extend(BitReader &br, int s)          // T.81 F.2.2.1: s bits as a signed value
{
    if (s == 0) return 0;
    const int32_t v = static_cast<int32_t>(br.bits(s));
    return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
}

struct Frame {
    int w = 0, h = 0, ncomp = 0;
    int id[4] = {0, 0, 0, 0}, ch[4] = {1, 1, 1, 1}, cv[4] = {1, 1, 1, 1}, cq[4] = {0, 0, 0, 0};
    int hy = 1, vy = 1, mx = 0, my = 0, per = 1;
    int lhy = 0, lvy = 0;                        // log2 of hy (1, 2, 4) and vy (1, 2): block_at runs once per block and scan
};

// where block (bx, by) of component c lies in the scan-ordered array jpeg_didct_kernel reads
inline size_t block_at(const Frame &fr, int c, int bx, int by)
{
    if (c == 0) {
        const size_t m = static_cast<size_t>(by >> fr.lvy) * fr.mx + (bx >> fr.lhy);
        return m * fr.per + ((by & (fr.vy - 1)) << fr.lhy) + (bx & (fr.hy - 1));
    }
    return (static_cast<size_t>(by) * fr.mx + bx) * fr.per + fr.hy * fr.vy + c - 1;
}

inline bool put(int16_t *dst, int32_t v)
{
    *dst = static_cast<int16_t>(v);
    return v == static_cast<int16_t>(v);
}

// refineNonZeroes: over coefficients zig .. ze of b (zig >= 1); a non-zero one reads a correction bit; stops in front of the
// (nz + 1)-th zero one (nz < 0: never).  `mask` holds the block's non-zero coefficients by zig-zag position, so the walk costs
// what the block holds, not the band's width: an image's refinement scans are mostly blocks inside end-of-band runs, all of
// whose 63 positions the plain loop would visit (4K: 37 M positions for 2 M corrections).
inline int refine_nonzeroes(BitReader &br, int16_t *b, uint64_t mask, int zig, int ze, int nz, int32_t delta, bool *ok)
{
    const uint64_t band = (ze >= 63 ? ~0ull : ((1ull << (ze + 1)) - 1ull)) & ~((1ull << zig) - 1ull);
    int stop = ze + 1;
    if (nz >= 0) {
        uint64_t z = ~mask & band;
        for (int i = 0; i < nz && z; i++) z &= z - 1;
        if (z) stop = __builtin_ctzll(z);
    }
    uint64_t todo = mask & band & (stop >= 64 ? ~0ull : ((1ull << stop) - 1ull));
    while (todo) {                                   // the correction bits of up to 32 coefficients in one read
        const int have = __builtin_popcountll(todo), take = have < 32 ? have : 32;
        const uint32_t cb = br.bits32(take);
        for (int i = take - 1; i >= 0; i--) {
            const int k = __builtin_ctzll(todo);
            todo &= todo - 1;
            if (!((cb >> i) & 1u)) continue;
            int16_t *p = b + UNZIG_P[k];
            *ok = put(p, *p >= 0 ? *p + delta : *p - delta) && *ok;
        }
    }
    return stop;
}

}  // namespace

int jpeg_progressive_coefficients(const uint8_t *data, size_t n, JpegFile *f, int16_t *coef)
{
    // The frame is jpeg_parse's (f->comp_*, f->hy ...): this function never reads a SOF segment.  Up to the first SOS (f->scan)
    // it walks the file by jpeg_parse's marker rules exactly -- every segment starts with a marker, anything that is not a
    // stand-alone marker is length-prefixed, 0xff 0x00 included -- so what the sizes came from and what is decoded here are one
    // reading of the file (with two sets of rules an 0xff 0x00 pseudo-segment hid a second frame header whose chroma
    // factors indexed coef and nzmask at twice their size).  Behind the first scan it reads as reader.go does between segments.
    Frame fr;
    fr.w = f->w; fr.h = f->h; fr.ncomp = f->ncomp;
    if (fr.ncomp != 1 && fr.ncomp != 3 && fr.ncomp != 4) return jpeg_corrupt("bad component count");
    for (int c = 0; c < fr.ncomp; c++) {
        fr.id[c] = f->comp_id[c]; fr.ch[c] = f->comp_h[c]; fr.cv[c] = f->comp_v[c]; fr.cq[c] = f->comp_q[c];
        if (fr.cq[c] < 0 || fr.cq[c] > 3) return jpeg_corrupt("bad quantisation table selector");
        if (c > 0 && (fr.ch[c] != 1 || fr.cv[c] != 1)) return jpeg_unsupported("chroma factors other than 1 x 1");
    }
    if (fr.ncomp == 1) fr.ch[0] = fr.cv[0] = 1;
    if (fr.ncomp == 4 && (fr.ch[0] != 1 || fr.cv[0] != 1)) return jpeg_unsupported("a four-component file with subsampled components");
    fr.hy = fr.ch[0]; fr.vy = fr.cv[0];
    if (!(fr.hy == 1 || fr.hy == 2 || fr.hy == 4) || !(fr.vy == 1 || fr.vy == 2)) return jpeg_unsupported("luminance factors other than 1, 2, 4 across and 1, 2 down");
    fr.lhy = fr.hy == 4 ? 2 : fr.hy - 1; fr.lvy = fr.vy - 1;
    fr.mx = f->mx; fr.my = f->my;
    fr.per = fr.hy * fr.vy + fr.ncomp - 1;
    if (fr.w <= 0 || fr.h <= 0 || fr.hy != f->hy || fr.vy != f->vy || fr.per != f->nslots ||
        fr.mx != (fr.w + 8 * fr.hy - 1) / (8 * fr.hy) || fr.my != (fr.h + 8 * fr.vy - 1) / (8 * fr.vy))
        return jpeg_corrupt("the frame does not describe the block array");
    const size_t nblk_total = static_cast<size_t>(fr.mx) * fr.my * fr.per;      // what coef and nzmask hold
    // sequential frames (SOF0 / SOF1) whose scans the device's decoder has no form for come here too: processSOS is one function,
    // with Ss, Se, Ah, Al fixed at 0, 63, 0, 0 whatever the scan header says (Table B.3), and a block is dequantised when its scan
    // decodes it -- with the table in force then (qsnap), not at EOI
    const bool sequential = f->sof_sequential;
    uint8_t qsnap[4][64];
    uint8_t q[4][64];
    bool have_q[4] = {false, false, false, false}, seen[4] = {false, false, false, false};
    std::vector<HTab> tabs(8);                   // [tc * 4 + th]
    std::vector<uint64_t> nzmask;                // per block: its non-zero AC coefficients by zig-zag position
    try {
        nzmask.assign(nblk_total, 0);
    } catch (const std::bad_alloc &) {
        set_error("jpeg decode: no host memory for a progressive image of %d x %d", f->w, f->h);
        return FNX_ERR_OOM;
    }
    int ri = 0;
    bool range_ok = true, seen_sof = false;
    size_t pos = 2;
    if (f->scan < 2 || f->scan + 4 > n) return jpeg_corrupt("no scan where the first reading found one");
    for (;;) {
        if (pos + 2 > n) return jpeg_corrupt("the file ends before EOI");
        const bool header = pos < f->scan;                                  // jpeg_parse's rules up to the first SOS
        if (data[pos] != 0xff) {
            if (header) return jpeg_corrupt("a segment does not start with a marker");
            pos++;                                                          // reader.go skips what lies between segments
            continue;
        }
        const uint8_t m = data[pos + 1];
        if (m == 0xff) { pos++; continue; }
        if (m == 0x01 || (m >= 0xd0 && m <= 0xd7) || (m == 0x00 && !header)) { pos += 2; continue; }
        if (m == 0xd9) {
            if (header) return jpeg_corrupt("EOI before any scan");
            break;
        }
        if (pos + 4 > n) return jpeg_corrupt("the file ends inside a segment");
        const size_t len = (static_cast<size_t>(data[pos + 2]) << 8) | data[pos + 3];
        if (len < 2 || pos + 2 + len > n) return jpeg_corrupt("a segment runs past the end of the file");
        if (header && (pos + 2 + len > f->scan || m == 0xda)) return jpeg_corrupt("the two readings of the file's segments disagree");
        const uint8_t *seg = data + pos + 4;
        const size_t sl = len - 2;
        if (m == 0xdb) {
            size_t o = 0;
            while (o < sl) {
                const int pq = seg[o] >> 4, tq = seg[o] & 15;
                if (pq != 0) return jpeg_unsupported("a 16-bit quantisation table");
                if (tq > 3 || o + 65 > sl) return jpeg_corrupt("bad DQT segment");
                for (int zig = 0; zig < 64; zig++) q[tq][UNZIG_P[zig]] = seg[o + 1 + zig];
                have_q[tq] = true;
                o += 65;
            }
        } else if (m == 0xc2 || m == 0xc0 || m == 0xc1) {
            // (the one frame header jpeg_parse read; its contents are in *f)
            if (!header || seen_sof) return jpeg_corrupt("two SOF segments");
            seen_sof = true;
        } else if (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8) {
            return jpeg_corrupt("a second frame header of another kind");
        } else if (m == 0xc4) {
            size_t o = 0;
            while (o < sl) {
                const int tc = seg[o] >> 4, th = seg[o] & 15;
                if (tc > 1 || th > 3 || o + 17 > sl) return jpeg_corrupt("bad DHT segment");
                HTab &t = tabs[4 * tc + th];
                int total = 0;
                for (int L = 1; L <= 16; L++) total += seg[o + L];
                if (total > 256 || o + 17 + static_cast<size_t>(total) > sl) return jpeg_corrupt("bad DHT segment");
                std::memset(t.fast, 0, sizeof(t.fast));
                int32_t code = 0;
                int k = 0;
                for (int L = 1; L <= 16; L++) {
                    const int cnt = seg[o + L];
                    t.valoff[L] = k - code;
                    if (code + cnt > (1 << L)) return jpeg_corrupt("a Huffman table with more codes than its lengths allow");
                    for (int j = 0; j < cnt; j++, k++, code++) {
                        t.val[k] = seg[o + 17 + k];
                        if (L <= 9)
                            for (int32_t x = code << (9 - L); x < ((code + 1) << (9 - L)); x++)
                                t.fast[x] = static_cast<uint16_t>((L << 8) | t.val[k]);
                    }
                    t.maxcode[L] = cnt ? code - 1 : -1;
                    code <<= 1;
                }
                t.have = true;
                o += 17 + total;
            }
        } else if (m == 0xdd) {
            if (sl < 2) return jpeg_corrupt("bad DRI segment");
            ri = (seg[0] << 8) | seg[1];
        } else if (m == 0xee) {
            // (jpeg_parse has judged the Adobe segment against the component count)
        } else if (m == 0xda) {
            if (fr.ncomp == 0) return jpeg_corrupt("SOS before SOF");
            const int ns = sl >= 1 ? seg[0] : 0;
            if (ns < 1 || ns > fr.ncomp || sl != 4 + 2 * static_cast<size_t>(ns)) return jpeg_corrupt("bad SOS segment");
            int sc[4] = {0, 0, 0, 0}, td[4] = {0, 0, 0, 0}, ta[4] = {0, 0, 0, 0};
            for (int i = 0; i < ns; i++) {
                int c = -1;
                for (int j = 0; j < fr.ncomp; j++) if (fr.id[j] == seg[1 + 2 * i]) c = j;
                if (c < 0) return jpeg_corrupt("a scan names a component the frame does not hold");
                for (int j = 0; j < i; j++) if (sc[j] == c) return jpeg_corrupt("a scan names a component twice");
                sc[i] = c; td[i] = seg[2 + 2 * i] >> 4; ta[i] = seg[2 + 2 * i] & 15;
                if (td[i] > 3 || ta[i] > 3) return jpeg_corrupt("bad Huffman table selector");
            }
            int zs = seg[1 + 2 * ns], ze = seg[2 + 2 * ns], ah = seg[3 + 2 * ns] >> 4, al = seg[3 + 2 * ns] & 15;
            if (sequential) { zs = 0; ze = 63; ah = 0; al = 0; }
            if ((zs == 0 && ze != 0 && !sequential) || zs > ze || ze > 63) return jpeg_corrupt("bad spectral selection bounds");
            if (zs != 0 && ns != 1) return jpeg_corrupt("progressive AC coefficients for more than one component");
            if ((ah != 0 && ah != al + 1) || al > 13) return jpeg_corrupt("bad successive approximation values");
            for (int i = 0; i < ns; i++) {
                if (zs == 0 && ah == 0 && !tabs[td[i]].have) return jpeg_corrupt("the scan uses a Huffman table the file does not define");
                if ((zs != 0 || sequential) && !tabs[4 + ta[i]].have) return jpeg_corrupt("the scan uses a Huffman table the file does not define");
                if (sequential) {
                    if (seen[sc[i]]) return jpeg_corrupt("a sequential file codes a component twice");
                    if (!have_q[fr.cq[sc[i]]]) return jpeg_corrupt("the frame uses a quantisation table the file does not define");
                    std::memcpy(qsnap[sc[i]], q[fr.cq[sc[i]]], 64);
                }
                seen[sc[i]] = true;
            }
            if (ri > 0 && ns == 1 && fr.ch[sc[0]] * fr.cv[sc[0]] > 1)
                return jpeg_unsupported("a restart interval in a one-component scan of a component with several blocks per MCU");
            BitReader br(data + pos + 2 + len, data + n);
            const int32_t delta = 1 << al;
            int32_t pred[4] = {0, 0, 0, 0};
            uint32_t eob_run = 0;
            long long mcu = 0;
            int rbx = 0, rby = 0;                                                // a one-component scan's raster position
            const long long nmcu = static_cast<long long>(fr.mx) * fr.my;
            int expected_rst = 0;
            for (int my0 = 0; my0 < fr.my; my0++)
                for (int mx0 = 0; mx0 < fr.mx; mx0++) {
                    for (int i = 0; i < ns; i++) {
                        const int c = sc[i], hi = fr.ch[c], vi = fr.cv[c];
                        for (int j = 0; j < hi * vi; j++) {
                            int bx, by;
                            if (ns != 1) { bx = hi * mx0 + j % hi; by = vi * my0 + j / hi; }
                            else {
                                bx = rbx; by = rby;
                                if (++rbx == fr.mx * hi) { rbx = 0; rby++; }
                                // the component's own extent: ceil(w hi / hy) x ceil(h vi / vy) samples
                                if (8ll * bx * fr.hy >= static_cast<long long>(fr.w) * hi || 8ll * by * fr.vy >= static_cast<long long>(fr.h) * vi) continue;
                            }
                            const size_t blk = block_at(fr, c, bx, by);
                            if (blk >= nblk_total) return jpeg_corrupt("a scan addresses a block outside the frame");
                            int16_t *b = coef + 64 * blk;
                            uint64_t &mask = nzmask[blk];
                            if (ah != 0) {                                       // refinement (G.1.2.1, G.1.2.3)
                                if (zs == 0) {
                                    if (br.bit()) range_ok = put(b, b[0] | delta) && range_ok;
                                    continue;
                                }
                                const HTab &t = tabs[4 + ta[i]];
                                int zig = zs;
                                if (eob_run == 0) {
                                    for (; zig <= ze; zig++) {
                                        int32_t z = 0;
                                        const int rs = huff(br, t);
                                        if (rs < 0) return jpeg_corrupt("a scan holds a code outside its Huffman table");
                                        const int v0 = rs >> 4, v1 = rs & 15;
                                        if (v1 == 0) {
                                            if (v0 != 15) {
                                                eob_run = (1u << v0) | br.bits(v0);
                                                break;
                                            }
                                        } else if (v1 == 1) {
                                            z = br.bit() ? delta : -delta;
                                        } else {
                                            return jpeg_corrupt("a refinement scan holds a coefficient of more than one bit");
                                        }
                                        zig = refine_nonzeroes(br, b, mask, zig, ze, v0, delta, &range_ok);
                                        if (zig > ze) return jpeg_corrupt("a refinement scan runs past the end of its band");
                                        if (z != 0) { b[UNZIG_P[zig]] = static_cast<int16_t>(z); mask |= 1ull << zig; }
                                    }
                                }
                                if (eob_run > 0) {
                                    eob_run--;
                                    if (zig <= ze) refine_nonzeroes(br, b, mask, zig, ze, -1, delta, &range_ok);
                                }
                                continue;
                            }
                            int zig = zs;
                            if (zig == 0) {                                      // DC, first pass (G.1.2.1)
                                zig++;
                                const int s = huff(br, tabs[td[i]]);
                                if (s < 0 || s > 16) return jpeg_corrupt("a scan holds a code outside its Huffman table");
                                pred[c] += extend(br, s);
                                // the prediction is range-checked BEFORE it is scaled, the product taken in 64 bits (2^20 * 2^13
                                // does not fit 32: ADVICE r5 -- a wrapped product could pass for an in-range coefficient)
                                if (pred[c] < -(1 << 20) || pred[c] > (1 << 20)) { range_ok = false; pred[c] = 0; }
                                const long long dc = static_cast<long long>(pred[c]) * delta;
                                if (dc < -32768 || dc > 32767) range_ok = false;
                                b[0] = static_cast<int16_t>(dc);
                            }
                            if (zig <= ze && eob_run > 0) eob_run--;
                            else {                                               // AC, first pass (G.1.2.2)
                                const HTab &t = tabs[4 + ta[i]];
                                for (; zig <= ze; zig++) {
                                    const int rs = huff(br, t);
                                    if (rs < 0) return jpeg_corrupt("a scan holds a code outside its Huffman table");
                                    const int v0 = rs >> 4, v1 = rs & 15;
                                    if (v1 != 0) {
                                        zig += v0;
                                        if (zig > ze) break;
                                        range_ok = put(b + UNZIG_P[zig], extend(br, v1) * delta) && range_ok;
                                        mask |= 1ull << zig;
                                    } else {
                                        if (v0 != 15) {
                                            eob_run = ((1u << v0) | br.bits(v0)) - 1u;
                                            break;
                                        }
                                        zig += 15;
                                    }
                                }
                            }
                        }
                    }
                    if (br.overran()) return jpeg_corrupt("a scan ends before its last block");
                    mcu++;
                    if (ri > 0 && mcu % ri == 0 && mcu < nmcu) {
                        br.restart();
                        if (br.p + 2 > br.end || br.p[0] != 0xff || br.p[1] != 0xd0 + expected_rst) return jpeg_corrupt("restart markers out of sequence");
                        br.p += 2;
                        expected_rst = (expected_rst + 1) & 7;
                        pred[0] = pred[1] = pred[2] = pred[3] = 0;
                        eob_run = 0;
                    }
                }
            pos = static_cast<size_t>(br.p - data);                             // in front of the marker that ended the scan
            continue;
        }
        pos += 2 + len;
    }
    if (!range_ok) return jpeg_unsupported("coefficients beyond 16 bits");
    for (int c = 0; c < fr.ncomp; c++) {
        if (!seen[c]) return jpeg_unsupported("a component no scan mentions");
        if (!sequential && !have_q[fr.cq[c]]) return jpeg_corrupt("the frame uses a quantisation table the file does not define");
    }
    for (int c = 0; c < 4; c++)
        for (int k = 0; k < 64; k++) f->q[c][k] = c < fr.ncomp ? (sequential ? qsnap[c][k] : q[fr.cq[c]][k]) : 1;
    return FNX_OK;
}

}  // namespace fnx
