// The host side of the device JPEG decoder (jpeg_dec.hip): the file's segments, the Huffman decoding tables, and the
// entropy-coded segment without its stuffing.  Plain C++ with no device code in it -- this is the part that reads
// untrusted bytes, and the sanitizer builds (make asan / tsan) instrument it with the rest of the host layer.
#include <cstring>
#include <vector>
#include "common.hpp"

namespace fnx {

static const uint8_t UNZIG_H[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

int jpeg_unsupported(const char *what)
{
    set_error("jpeg decode: %s is not handled on the device (sequential or progressive Huffman coding, 8 bit, 1 or 3 components with luminance factors 1, 2 or 4 across and 1 or 2 down, or 4 at 1 x 1)", what);
    return FNX_ERR_UNSUPPORTED;
}

int jpeg_corrupt(const char *what)
{
    set_error("jpeg decode: %s", what);
    return FNX_ERR_INVALID;
}

int jpeg_parse(const uint8_t *data, size_t n, JpegFile *f)
{
    if (n < 4 || data[0] != 0xff || data[1] != 0xd8) return jpeg_corrupt("no SOI marker");
    bool have_q[4] = {false, false, false, false}, have_t[4] = {false, false, false, false}, have_sof = false;
    // what only a baseline scan minds (a progressive file's scans are decoded on the host, jpeg_prog.cpp, with tables of their own)
    const char *baseline_only = nullptr;
    bool sof1_or_2 = false, allones = false;
    f->progressive = false;
    uint8_t q[4][64];
    int ncomp = 0, comp_id[4] = {0, 0, 0, 0}, comp_h[4] = {0, 0, 0, 0}, comp_v[4] = {0, 0, 0, 0}, comp_q[4] = {0, 0, 0, 0};
    f->adobe = -1;
    std::memset(&f->tab, 0, sizeof(f->tab));
    f->ri = 0;
    size_t pos = 2;
    for (;;) {
        if (pos + 4 > n) return jpeg_corrupt("the file ends before its scan");
        if (data[pos] != 0xff) return jpeg_corrupt("a segment does not start with a marker");
        const uint8_t m = data[pos + 1];
        if (m == 0xff) { pos++; continue; }                              // fill bytes
        if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) { pos += 2; continue; }
        if (m == 0xd9) return jpeg_corrupt("EOI before any scan");
        const size_t len = (static_cast<size_t>(data[pos + 2]) << 8) | data[pos + 3];
        if (len < 2 || pos + 2 + len > n) return jpeg_corrupt("a segment runs past the end of the file");
        const uint8_t *seg = data + pos + 4;
        const size_t sl = len - 2;
        if (m == 0xdb) {
            size_t o = 0;
            while (o < sl) {
                const int pq = seg[o] >> 4, tq = seg[o] & 15;
                if (pq != 0) return jpeg_unsupported("a 16-bit quantisation table");
                if (tq > 3 || o + 65 > sl) return jpeg_corrupt("bad DQT segment");
                for (int zig = 0; zig < 64; zig++) q[tq][UNZIG_H[zig]] = seg[o + 1 + zig];
                have_q[tq] = true;
                o += 65;
            }
        } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) {
            if (have_sof) return jpeg_corrupt("two SOF segments");
            f->sof_sequential = m != 0xc2;
            f->progressive = m != 0xc0;      // SOF2, and SOF1 (extended sequential: up to four tables per class): the host reads the scans
            sof1_or_2 = m != 0xc0;
            if (sl < 6) return jpeg_corrupt("bad SOF segment");
            if (seg[0] != 8) return jpeg_unsupported("a sample precision other than 8 bits");
            if (seg[5] != 3 && seg[5] != 1 && seg[5] != 4) return jpeg_unsupported("a component count other than 1, 3 and 4");
            ncomp = seg[5];
            if (sl < 6 + 3 * static_cast<size_t>(ncomp)) return jpeg_corrupt("bad SOF segment");
            f->h = (seg[1] << 8) | seg[2];
            f->w = (seg[3] << 8) | seg[4];
            if (f->w <= 0 || f->h <= 0) return jpeg_unsupported("a zero dimension (DNL)");
            for (int c = 0; c < ncomp; c++) {
                comp_id[c] = seg[6 + 3 * c];
                comp_h[c] = seg[7 + 3 * c] >> 4;
                comp_v[c] = seg[7 + 3 * c] & 15;
                comp_q[c] = seg[8 + 3 * c];
                if (comp_q[c] > 3) return jpeg_corrupt("bad quantisation table selector");
            }
            have_sof = true;
        } else if (m == 0xc3 || (m >= 0xc5 && m <= 0xcf && m != 0xc8 && m != 0xcc)) {
            return jpeg_unsupported("a frame type other than sequential and progressive Huffman coding");
        } else if (m == 0xcc) {
            return jpeg_unsupported("arithmetic coding");
        } else if (m == 0xc4) {
            size_t o = 0;
            while (o < sl) {
                const int tc = seg[o] >> 4, th = seg[o] & 15;
                if (tc > 1 || th > 3 || o + 17 > sl) return jpeg_corrupt("bad DHT segment");
                int total = 0;
                for (int L = 1; L <= 16; L++) total += seg[o + L];
                if (total > 256 || o + 17 + total > sl) return jpeg_corrupt("bad DHT segment");
                if (th > 1) {                                             // (progressive files use up to four)
                    if (!baseline_only) baseline_only = "a Huffman table selector above 1";
                    o += 17 + total;
                    continue;
                }
                const int t = tc * 2 + th;
                std::memset(f->tab.fast[t], 0, sizeof(f->tab.fast[t]));
                uint32_t code = 0;
                int k = 0;
                for (int L = 1; L <= 16; L++) {
                    const int cnt = seg[o + L];
                    f->tab.delta[t][L] = k - static_cast<int32_t>(code);
                    for (int j = 0; j < cnt; j++, k++, code++) {
                        if (code >= (1u << L)) return jpeg_corrupt("a Huffman table with more codes than its lengths allow");
                        // T.81 C.2 keeps the all-ones code of every length unassigned, and the kernels lean on it: the 1-bits
                        // that pad the byte before a restart marker (and the string's last byte) can then never be a symbol.
                        // A table that assigns it is one the host codec gets to read.
                        if (code + 1 == (1u << L)) allones = true;                 // (image/jpeg reads such a table: the host decodes the scan)
                        f->tab.value[t][k] = seg[o + 17 + k];
                        if (L <= DEC_FAST_BITS)
                            for (uint32_t x = code << (DEC_FAST_BITS - L); x < ((code + 1) << (DEC_FAST_BITS - L)); x++)
                                f->tab.fast[t][x] = static_cast<uint16_t>((L << 8) | seg[o + 17 + k]);
                    }
                    f->tab.limit[t][L] = code << (16 - L);
                    code <<= 1;
                }
                have_t[t] = true;
                o += 17 + total;
            }
        } else if (m == 0xdd) {
            if (sl < 2) return jpeg_corrupt("bad DRI segment");
            f->ri = (seg[0] << 8) | seg[1];                               // MCUs per restart interval; 0: none
        } else if (m == 0xee) {
            if (sl >= 12 && std::memcmp(seg, "Adobe", 5) == 0) f->adobe = seg[11];     // (judged at the scan: what it means depends on the component count)
        } else if (m == 0xda) {
            if (!have_sof) return jpeg_corrupt("SOS before SOF");
            if (ncomp == 3 && comp_id[0] == 'R' && comp_id[1] == 'G' && comp_id[2] == 'B') return jpeg_unsupported("an RGB file");
            if (ncomp == 3 && f->adobe >= 0 && f->adobe != 1) return jpeg_unsupported("an Adobe colour transform other than YCbCr");
            if (ncomp == 4) {
                // image.CMYK (reader.go applyBlack): Adobe CMYK or YCbCrK.  Of image/jpeg's two layouts the one every encoder writes
                // -- all four components 1 x 1; without an APP14 segment image/jpeg refuses the file.  Always the host's scans.
                for (int c = 0; c < 4; c++)
                    if (comp_h[c] != 1 || comp_v[c] != 1) return jpeg_unsupported("a four-component file with subsampled components");
                if (f->adobe < 0) return jpeg_unsupported("a four-component file without an Adobe segment (image/jpeg refuses it)");
                if (baseline_only && !sof1_or_2) return jpeg_unsupported(baseline_only);
                f->ratio = -2; f->hy = 1; f->vy = 1;
                f->progressive = true;
            } else if (ncomp == 1) {
                // a one-component scan is not interleaved (T.81 A.2.2): one block per MCU whatever the factors say; image.Gray
                f->ratio = -1; f->hy = 1; f->vy = 1;
            } else {
                // image/jpeg takes luminance factors 1, 2, 4 across and 1, 2 down (reader.go: v == 4 is unsupported there too);
                // the chroma planes at 1 x 1 are what every encoder writes
                if (!(comp_h[1] == 1 && comp_v[1] == 1 && comp_h[2] == 1 && comp_v[2] == 1) ||
                    !(comp_h[0] == 1 || comp_h[0] == 2 || comp_h[0] == 4) || comp_v[0] < 1 || comp_v[0] > 2)
                    return jpeg_unsupported("a subsampling other than 4:4:4, 4:2:2, 4:2:0, 4:4:0, 4:1:1 and 4:1:0");
                f->hy = comp_h[0]; f->vy = comp_v[0];
                // image.YCbCrSubsampleRatio: 444, 422, 420, 440, 411, 410
                f->ratio = f->hy == 4 ? (f->vy == 2 ? 5 : 4) : f->hy == 2 ? (f->vy == 2 ? 2 : 1) : (f->vy == 2 ? 3 : 0);
            }
            f->ncomp = ncomp;
            for (int c = 0; c < 4; c++) {
                f->comp_id[c] = comp_id[c]; f->comp_q[c] = comp_q[c];
                // (one component: not interleaved, one block per MCU whatever the factors say)
                f->comp_h[c] = c < ncomp && ncomp != 1 ? comp_h[c] : 1;
                f->comp_v[c] = c < ncomp && ncomp != 1 ? comp_v[c] : 1;
            }
            f->nslots = f->hy * f->vy + ncomp - 1;
            f->mx = (f->w + 8 * f->hy - 1) / (8 * f->hy);
            f->my = (f->h + 8 * f->vy - 1) / (8 * f->vy);
            // A baseline frame whose scan the device's decoder has no form for -- the components in scans of their own or out of
            // frame order, a table that assigns the all-ones code -- is the host decoder's too (r5: jpeg_prog.cpp reads sequential
            // scans as well).  A baseline file with table selectors above 1 stays refused: image/jpeg refuses it ("bad Th value").
            if (!f->progressive && !baseline_only && sl >= 1 && seg[0] >= 1) {
                bool host = allones || seg[0] < ncomp;
                if (seg[0] == ncomp && sl >= 1 + 2 * static_cast<size_t>(ncomp))
                    for (int c = 0; c < ncomp; c++) host = host || seg[1 + 2 * c] != comp_id[c];
                f->progressive = host;
            }
            if (f->progressive) {                                         // the scans are jpeg_progressive_coefficients' to read
                if (!sof1_or_2 && baseline_only) return jpeg_unsupported(baseline_only);
                f->scan = pos;
                return FNX_OK;
            }
            if (baseline_only) return jpeg_unsupported(baseline_only);
            if (sl < 1 || seg[0] != ncomp) return jpeg_unsupported("a scan that does not hold all the frame's components");
            if (sl < 1 + 2 * static_cast<size_t>(ncomp) + 3) return jpeg_corrupt("bad SOS segment");
            int td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
            for (int c = 0; c < ncomp; c++) {
                if (seg[1 + 2 * c] != comp_id[c]) return jpeg_unsupported("scan components out of frame order");
                td[c] = seg[2 + 2 * c] >> 4;
                ta[c] = seg[2 + 2 * c] & 15;
                if (td[c] > 1 || ta[c] > 1) return jpeg_unsupported("a Huffman table selector above 1");
                if (!have_t[td[c]] || !have_t[2 + ta[c]]) return jpeg_corrupt("the scan uses a Huffman table the file does not define");
                if (!have_q[comp_q[c]]) return jpeg_corrupt("the frame uses a quantisation table the file does not define");
            }
            const int ny = f->hy * f->vy;
            f->dcpack = f->acpack = 0;
            for (int s = 0; s < f->nslots; s++) {
                const int c = s < ny ? 0 : s - ny + 1;
                f->dcpack |= static_cast<uint64_t>(td[c]) << (4 * s);
                f->acpack |= static_cast<uint64_t>(2 + ta[c]) << (4 * s);
            }
            for (int c = 0; c < 3; c++)
                for (int k = 0; k < 64; k++) f->q[c][k] = c < ncomp ? q[comp_q[c]][k] : 1;
            f->scan = pos + 2 + len;
            return FNX_OK;
        }
        pos += 2 + len;
    }
}

// The scan's bytes without the stuffing into `dst` (capacity: n - f.scan); *nbytes = what was written.
int jpeg_unstuff(const uint8_t *data, size_t n, const JpegFile &f, uint8_t *dst, size_t *nbytes, std::vector<uint32_t> *rst)
{
    const uint8_t *s = data + f.scan, *end = data + n;
    uint8_t *d = dst;
    const long long nmcu = static_cast<long long>(f.mx) * f.my;
    const long long want = f.ri > 0 ? (nmcu + f.ri - 1) / f.ri - 1 : 0;      // restart markers a complete scan holds
    rst->clear();
    for (;;) {
        const uint8_t *ff = static_cast<const uint8_t *>(std::memchr(s, 0xff, static_cast<size_t>(end - s)));
        if (!ff || ff + 1 >= end) return jpeg_corrupt("the scan runs to the end of the file (no EOI)");
        std::memcpy(d, s, static_cast<size_t>(ff - s));
        d += ff - s;
        const uint8_t m = ff[1];
        if (m == 0x00) {
            *d++ = 0xff;
            s = ff + 2;
        } else if (m == 0xff) {
            s = ff + 1;                                                   // fill byte before a marker
        } else if (m == 0xd9) {
            break;
        } else if (m >= 0xd0 && m <= 0xd7) {
            // RSTn: the next interval starts here, on a byte boundary, with every prediction and the MCU position reset
            if (f.ri <= 0) return jpeg_corrupt("a restart marker in a scan without restart intervals");
            if (m != 0xd0 + (rst->size() & 7u)) return jpeg_corrupt("restart markers out of sequence");
            if (static_cast<long long>(rst->size()) >= want) return jpeg_corrupt("more restart markers than the image has intervals");
            if (static_cast<size_t>(d - dst) > 0xfffffff0u) return jpeg_unsupported("a scan this large with restart intervals");
            rst->push_back(static_cast<uint32_t>(d - dst));
            s = ff + 2;
        } else {
            return jpeg_unsupported("a second scan (or another segment) behind the first");
        }
    }
    if (static_cast<long long>(rst->size()) != want) return jpeg_corrupt("fewer restart markers than the image has intervals");
    *nbytes = static_cast<size_t>(d - dst);
    return FNX_OK;
}


}  // namespace fnx
