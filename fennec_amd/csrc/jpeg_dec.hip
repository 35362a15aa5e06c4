// Baseline JPEG decoding on the device -- SURVEY 8(f)2, third slice: image.Decode of CompressBatch's SOURCE
// (batch.go:88-101 -> io.go:60-95) without the host codec, so that a JPEG crosses PCIe as its file bytes and the
// quality search (jpeg.hip) starts from pixels that never left the GPU.
//
// A scan is one long Huffman-coded bit string: where a symbol starts depends on every symbol before it.  What makes it
// parallel is that Huffman codes SELF-SYNCHRONISE: a decoder started at a wrong bit falls into step with the right one
// after a few symbols, and from there on both produce the same thing.  (The decoder state here is more than the bit
// position -- also the position inside the block and the block's place in the MCU, which picks the tables -- so "in
// step" means all three agree.)  The string is cut into spans of 1024 bits, one per lane:
//   1. jpeg_dsync_kernel<false>  every lane decodes its span from a guessed state (block start, first slot), notes
//                                the state it ends in and the blocks it finished; then a lane whose predecessor ended
//                                somewhere else than the lane assumed decodes again from there -- until nothing in the
//                                workgroup changes.  No values are read, only code lengths.
//   2. jpeg_dsync_kernel<true>   the same across workgroups: a workgroup whose predecessor's last lane ended somewhere
//                                else than its first lane assumed repairs its chain (which stops as soon as a lane
//                                ends where it ended before).  Launched until a round changes nothing.  When that
//                                holds every lane's start state follows from the file's first bit: exact, whatever
//                                the guesses were; content only decides how many rounds it takes.
//   3. prefix sum of the finished-block counts = the block every lane starts in
//   4. jpeg_dwrite_kernel        every lane decodes its span once more, now with the values, into the block-major
//                                coefficient array (natural order; DC still as the difference)
//   5. DC differences -> DC: one prefix sum over the blocks laid out component by component
//   6. jpeg_didct_kernel         lane = block: dequantise, idct.go's IDCT, level shift, clamp, into the planes an
//                                *image.YCbCr holds; convert.hip makes toNRGBARef's image of them
// The host parses the segments, builds the decoding tables and removes the 0x00 stuffed behind every 0xff while it
// copies the scan into pinned memory (one memchr pass).
//
// Handled: what jpeg.Encode, libjpeg and most cameras write -- baseline (SOF0), 8 bit, three components (4:4:4, 4:2:2,
// 4:2:0, 4:4:0, 4:1:1, 4:1:0) or one (image.Gray), one scan, with or without restart intervals.  Anything else is FNX_ERR_UNSUPPORTED (the caller decodes on the
// host); a scan that ends early or holds a code outside its table is FNX_ERR_INVALID.  Restated from ITU T.81 and
// reader.go / scan.go / huffman.go's published behaviour, not from Go's source: bit-exact against the CPU restatement
// the tests hold (which libjpeg-turbo's files exercise), parity with Go unpinned.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "common.hpp"
#include "devutil.hpp"
#include "jpeg_idct.hpp"

namespace fnx {

constexpr int DEC_WPT = 32;                          // words of the bit string per lane
constexpr int DEC_WARM = 16;                         // first pass: spans a workgroup decodes ahead of the ones it owns, so that its
constexpr int DEC_OWN = 256 - DEC_WARM;              // first own span almost always starts from a synchronised state
constexpr int DEC_SPAN = 32 * DEC_WPT;               // ... in bits
constexpr int DEC_WG_WORDS = 256 * DEC_WPT;
constexpr int DEC_SEGW = 256 * (DEC_WPT + 1) + 4;    // a workgroup's words in LDS: one pad word per span (a span's words sit in
                                                     // different banks than its neighbours'), 3 words of look-ahead

struct DecArgs {
    const uint32_t *ecs;                             // the scan without stuffing, bytes as in the file, zero-padded
    const DecTables *tab;                            // write pass: fast[] = length << 8 | symbol
    const DecSyncTables *tab_sync;                   // sync passes: what one symbol -- or two -- does to the state
    unsigned long long *s_in, *s_out;                // per lane: the state it starts in / ends in
    uint32_t *cnt;                                   // per lane: blocks finished inside its span
    uint32_t *flag;                                  // <true>: set by a workgroup that decoded again
    const unsigned long long *first_blk;             // write: exclusive prefix sum of cnt
    int16_t *coef;                                   // write: [nblk][64], natural order, zeroed
    uint32_t *err;                                   // write: != 0: a code outside its table or a run past the block
    uint32_t *dbg;                                   // sync: [0] max passes of a workgroup, [1] sum of passes, [2] lane-decodes (FNX_JPEG_TRACE)
    long long nwords;                                // words of ecs that may be read
    unsigned long long nbits;                        // length of the string
    int nlanes;                                      // spans in the string
    int nblk, nslots;
    uint64_t dcpack, acpack;                         // table of slot s: (pack >> 4 s) & 15
    const uint32_t *rst;                             // byte offsets at which restart intervals 1, 2, ... start (ascending)
    int nrst;
    int ri_blocks;                                   // blocks per restart interval
    // per boundary k: (the lane that stood on it) << 32 | the blocks that lane had counted before it -- written by the sync
    // passes, read by the write pass (see jpeg_dwrite_kernel: block numbers are counted from the interval's start)
    unsigned long long *rst_rec;
};

// Restart intervals (DRI): interval k starts on a byte boundary, in the state (block start, slot 0), with every DC
// prediction at 0; what is left of the byte before it is padding (1 bits, never a complete code: T.81 forbids the
// all-ones code).  For a decoder that means: standing on a boundary, start over; a symbol that would reach across
// the next boundary is not one -- go to the boundary.  A decoder in a WRONG state obeys the same two rules, so every
// boundary puts it right: with restart intervals no desynchronised run is longer than an interval.
__device__ __forceinline__ uint32_t rst_rel(const DecArgs &a, int k, long long wg_bit)
{
    if (k >= a.nrst) return 0xffffffffu;
    const long long b = 8ll * a.rst[k] - wg_bit;
    return b > 0xfffffff0ll ? 0xffffffffu : (b < 0 ? 0u : static_cast<uint32_t>(b));
}

// state: bit position | position in the block (0..63) << 40 | slot in the MCU << 48
__device__ __forceinline__ unsigned long long dec_state(unsigned long long p, int z, int slot)
{
    return p | (static_cast<unsigned long long>(z) << 40) | (static_cast<unsigned long long>(slot) << 48);
}

__device__ __constant__ uint8_t c_unzig[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                               41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                               30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct DecShared {
    uint32_t seg[DEC_SEGW];
    unsigned long long out[256];
    // the first sync kernel's later passes: the few lanes that decode again, packed into the workgroup's first wave
    unsigned long long tin[64], rout[64];
    uint32_t rcnt[64];
    uint16_t todo[64];
    uint32_t wcnt[4];
    union {
        DecTables tab;                               // write pass
        DecSyncTables stab;                          // sync passes
    };
    uint8_t unzig[64];
};

__device__ __forceinline__ uint32_t dec_word(const uint32_t *seg, uint32_t wi)
{
    return seg[wi + wi / DEC_WPT];
}

// Decodes from (rel, z, slot) to the first symbol that starts at or after `end`.  WRITE: with the values.
// The loop is one dependent chain per symbol -- window, table look-up, new position -- and a wave has at most one other
// wave on its SIMD to hide it behind, so the chain is kept short: three words of the string live in registers (the
// third is fetched a symbol ahead), the symbol's effect on (position, z) is two selects instead of branches, and a code
// longer than the fast table's 11 bits finds its length by counting, not by a loop.
template <bool WRITE, bool RST>
__device__ __forceinline__ void dec_span(const DecShared &sh, const DecArgs &a, uint32_t &rel, int &z, int &slot, uint32_t end, uint32_t &cnt,
                                         long long blk, uint32_t &bad, long long wg_bit, uint32_t lim = 0xffffffffu)
{
    uint32_t wi = rel >> 5;
    uint32_t hi = dec_word(sh.seg, wi), lo = dec_word(sh.seg, wi + 1), nxt = dec_word(sh.seg, wi + 2);
    int rk = 0;
    uint32_t bnext = 0xffffffffu;
    if (RST) {                                                     // the first boundary at or after this lane's start
        const unsigned long long at = static_cast<unsigned long long>(wg_bit + rel);
        int lo_k = 0, hi_k = a.nrst;
        while (lo_k < hi_k) {
            const int mid = (lo_k + hi_k) >> 1;
            if (8ull * a.rst[mid] < at) lo_k = mid + 1; else hi_k = mid;
        }
        rk = lo_k;
        bnext = rst_rel(a, rk, wg_bit);
    }
    while (rel < end) {
        if (WRITE && blk + cnt >= a.nblk) break;                   // what follows the last block is padding
        if (WRITE && rel >= lim) {                                 // a block the image needs starts past the end of the string
            bad |= 8u;
            break;
        }
        // write pass: every block of the intervals before the next boundary is complete -- what is left before it is not data.
        // (In a sound file that is the padding, which the symbol rule below skips as well; in a DAMAGED interval whose blocks
        // end early the leftover bits can read as the start of one more block: a decoder that counts MCUs never looks at
        // them, and written here they would stay in the coefficients of the next interval's first block -- the fuzzer's
        // find, seed 41: tests/golden/damaged_interval_ends_early.jpg.)
        if (WRITE && RST && rel < bnext && rk < a.nrst && blk + cnt >= static_cast<long long>(rk + 1) * a.ri_blocks) {
            rel = bnext;
            if (rel >= end) break;                                 // the boundary lies in a later span: that lane starts over there
        }
        if (RST && rel >= bnext) {                                 // on a boundary: the interval's first block, first slot
            // (write pass: exactly the intervals before it must be complete -- a damaged interval that yields a block too
            // many or too few would shift every block behind it)
            if (WRITE && blk + cnt != static_cast<long long>(rk + 1) * a.ri_blocks) bad |= 16u;
            if (!WRITE && blk >= 0) a.rst_rec[rk] = (static_cast<unsigned long long>(blk) << 32) | cnt;   // (sync passes: blk = the lane, or -1)
            z = 0; slot = 0;
            bnext = rst_rel(a, ++rk, wg_bit);
        }
        uint32_t off = rel - 32u * wi;                             // < 64: a symbol is at most 16 + 15 bits
        const bool adv = off >= 32u;
        hi = adv ? lo : hi;
        lo = adv ? nxt : lo;
        wi += adv ? 1u : 0u;
        off -= adv ? 32u : 0u;
        nxt = dec_word(sh.seg, wi + 2);
        const unsigned long long pair = (static_cast<unsigned long long>(hi) << 32) | lo;
        const uint32_t c16 = static_cast<uint32_t>((pair << off) >> 48);
        const int t = static_cast<int>(((z == 0 ? a.dcpack : a.acpack) >> (4 * slot)) & 15ull);
        uint32_t e;
        if constexpr (!WRITE) {
            // the sync passes' tables hold what a symbol does to the state, ready made -- and, for the AC tables, what the
            // next one does as well when its code lies inside the same 11 bits (about every second look-up on a photograph)
            const uint32_t px = c16 >> (16 - DEC_FAST_BITS);
            e = sh.stab.st[t][px];
            const uint32_t a1 = e & 0xffu, s1 = (e >> 8) & 0xffu, a2 = (e >> 16) & 0xffu, s2 = e >> 24;
            // both symbols only if the second one belongs to this span and to this block (and, with restart intervals,
            // never: the boundary tests are per symbol); DC entries hold one symbol (a2 == 0)
            const bool two = !RST && a2 != 0 && rel + a1 < end && z + static_cast<int>(s1) < 64;
            const uint32_t adv = two ? a2 : a1, step = two ? s2 : s1;
            if (e) {
                if (RST && rel + adv > bnext) {                    // padding before a boundary
                    rel = bnext;
                    continue;
                }
                rel += adv;
                z += static_cast<int>(step);
                const bool fin = z >= 64;
                z = fin ? 0 : z;
                cnt += fin ? 1u : 0u;
                const int s1n = slot + 1 == a.nslots ? 0 : slot + 1;
                slot = fin ? s1n : slot;
                continue;
            }
        } else {
            e = sh.tab.fast[t][c16 >> (16 - DEC_FAST_BITS)];
        }
        int len = static_cast<int>(e >> 8), sym = static_cast<int>(e & 0xffu);
        bool nocode = false;
        if (e == 0) {
            // limit[] rises with the length: the code's length is the first L with c16 < limit[L]
            int L = DEC_FAST_BITS + 1;
            const uint32_t (*limit)[18] = WRITE ? sh.tab.limit : sh.stab.limit;
            const int32_t (*delta)[18] = WRITE ? sh.tab.delta : sh.stab.delta;
            const uint8_t (*value)[256] = WRITE ? sh.tab.value : sh.stab.value;
#pragma unroll
            for (int k = DEC_FAST_BITS + 1; k < 16; k++) L += c16 >= limit[t][k] ? 1 : 0;
            const bool hit = c16 < limit[t][16];
            len = L;
            sym = hit ? value[t][(static_cast<int>(c16 >> (16 - L)) + delta[t][L]) & 255] : 0;
            nocode = !hit;
        }
        const int s = sym & 15, r = sym >> 4;
        const bool dc = z == 0;
        if (RST && rel + static_cast<uint32_t>(len + s) > bnext) {  // padding before a boundary (the boundary resets the state)
            rel = bnext;
            continue;
        }
        if (WRITE && nocode) bad |= 1u;                            // (after the boundary test: padding is not a code either)
        if (WRITE && rel + static_cast<uint32_t>(len + s) > lim) { // a code or a value that straddles the string's end would be completed
            bad |= 8u;                                             // from the zero padding behind it: Go reports unexpected EOF here
            break;
        }
        if (WRITE) {
            int32_t v = 0;
            if (s) {                                               // receive + extend (T.81 F.2.2.1)
                const uint32_t o2 = off + static_cast<uint32_t>(len);
                const unsigned long long w2 = o2 < 32u ? pair << o2 : ((static_cast<unsigned long long>(lo) << 32) | nxt) << (o2 - 32u);
                const uint32_t raw = static_cast<uint32_t>(w2 >> (64 - s));
                v = raw < (1u << (s - 1)) ? static_cast<int32_t>(raw) - (1 << s) + 1 : static_cast<int32_t>(raw);
            }
            if (dc) {
                if (sym > 11) bad |= 2u;
                a.coef[(blk + cnt) * 64] = static_cast<int16_t>(v);
            } else if (s) {
                if (z + r > 63) bad |= 4u;
                else a.coef[(blk + cnt) * 64 + sh.unzig[z + r]] = static_cast<int16_t>(v);
            }
        }
        rel += static_cast<uint32_t>(len + s);
        // DC: on to the first AC.  AC: a value after r zeros, or sixteen zeros (r = 15, s = 0), both z + r + 1; any other
        // s = 0 ends the block
        z = dc ? 1 : ((s == 0 && r != 15) ? 64 : z + r + 1);
        if (z >= 64) {
            z = 0;
            slot = slot + 1 == a.nslots ? 0 : slot + 1;
            cnt++;
        }
    }
}

// the 256 spans from span0 on (+ 3 words of look-ahead) into LDS; spans before the string's start or past its end read as zeros
__device__ __forceinline__ void dec_stage(DecShared &sh, const DecArgs &a, long long span0, const void *tab, int tab_bytes)
{
    const long long base = span0 * DEC_WPT;
    for (int i = threadIdx.x; i < DEC_WG_WORDS + 3; i += 256) {
        const long long w = base + i;
        sh.seg[i + i / DEC_WPT] = (w >= 0 && w < a.nwords) ? __builtin_bswap32(a.ecs[w]) : 0u;
    }
    const uint32_t *tw = reinterpret_cast<const uint32_t *>(tab);
    uint32_t *sw = reinterpret_cast<uint32_t *>(&sh.tab);
    for (int i = threadIdx.x; i < tab_bytes / 4; i += 256) sw[i] = tw[i];
    if (threadIdx.x < 64) sh.unzig[threadIdx.x] = c_unzig[threadIdx.x];
}

// <false>: workgroup g owns spans [g OWN, (g + 1) OWN) and decodes the WARM spans before them as well (their results
// are dropped).  <true>: the same ownership, no warm-up -- only a workgroup whose predecessor ended elsewhere runs.
template <bool FIX, bool RST>
__global__ __launch_bounds__(256) void jpeg_dsync_kernel(DecArgs a)
{
    __shared__ DecShared sh;
    const int g = blockIdx.x, t = threadIdx.x;
    const long long span0 = static_cast<long long>(g) * DEC_OWN - (FIX ? 0 : DEC_WARM);
    const long long gs = span0 + t;
    const bool live = gs >= 0 && gs < a.nlanes && (!FIX || t < DEC_OWN);
    const long long wg_bit = span0 * DEC_SPAN;
    unsigned long long my_in, my_out = 0;
    uint32_t my_cnt = 0;
    bool need;
    if (!FIX) {
        my_in = dec_state(static_cast<unsigned long long>(live ? gs : 0) * DEC_SPAN, 0, 0);
        my_out = my_in;
        need = live;
    } else {
        if (g == 0) return;
        // every lane reads the same two words: the branch is uniform
        const unsigned long long prev = __hip_atomic_load(&a.s_out[span0 - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == a.s_in[span0]) return;
        my_in = live ? a.s_in[gs] : 0;
        my_out = live ? a.s_out[gs] : 0;
        my_cnt = live ? a.cnt[gs] : 0;
        need = t == 0;
        if (t == 0) my_in = prev;
    }
    dec_stage(sh, a, span0, a.tab_sync, static_cast<int>(sizeof(DecSyncTables)));
    sh.out[t] = my_out;
    __syncthreads();
    const uint32_t end = static_cast<uint32_t>(t + 1) * DEC_SPAN;
    uint32_t passes = 0, decodes = 0;
    for (;;) {
        passes++;
        decodes += need ? 1u : 0u;
        // After the first pass only a few lanes decode again -- one per chain of corrections, anywhere in the workgroup --
        // and a wave with ONE such lane issues the whole span's instructions all the same: PMC counted 45 M VALU
        // wave-instructions per 4K file against 13 M for one full decode (the write pass).  When at most 64 lanes need
        // it, they hand their (span, start state) to the workgroup's first wave, which decodes them side by side; the
        // other three waves wait at the barrier.  Same decodes, same order of passes: the fixed point is untouched.
        bool packed = false;
        int slot_c = 0;
        if (!FIX && passes > 1) {
            const unsigned long long bal = __ballot(need);
            if ((t & 63) == 0) sh.wcnt[t >> 6] = static_cast<uint32_t>(__popcll(bal));
            __syncthreads();
            const uint32_t k0 = sh.wcnt[0], k1 = sh.wcnt[1], k2 = sh.wcnt[2], k3 = sh.wcnt[3];
            const uint32_t total = k0 + k1 + k2 + k3;
            packed = total <= 64u;                                  // workgroup-uniform
            if (packed) {
                const int wv = t >> 6;
                const uint32_t base = (wv > 0 ? k0 : 0u) + (wv > 1 ? k1 : 0u) + (wv > 2 ? k2 : 0u);
                slot_c = static_cast<int>(base) + __popcll(bal & ((1ull << (t & 63)) - 1ull));
                if (need) {
                    sh.todo[slot_c] = static_cast<uint16_t>(t);
                    sh.tin[slot_c] = my_in;
                }
                __syncthreads();
                if (t < static_cast<int>(total)) {
                    const int src = sh.todo[t];
                    const unsigned long long in = sh.tin[t];
                    uint32_t rel = static_cast<uint32_t>(static_cast<long long>(in & 0xffffffffffull) - wg_bit);
                    int z = static_cast<int>((in >> 40) & 0xffu), slot = static_cast<int>(in >> 48);
                    uint32_t bad = 0, cnt = 0;
                    dec_span<false, RST>(sh, a, rel, z, slot, static_cast<uint32_t>(src + 1) * DEC_SPAN, cnt,
                                         src >= DEC_WARM ? span0 + src : -1ll, bad, wg_bit);
                    sh.rout[t] = dec_state(static_cast<unsigned long long>(wg_bit + rel), z, slot);
                    sh.rcnt[t] = cnt;
                }
                __syncthreads();
                if (need) {
                    my_out = sh.rout[slot_c];
                    my_cnt = sh.rcnt[slot_c];
                    sh.out[t] = my_out;
                }
            }
        }
        if (!packed && need) {
            uint32_t rel = static_cast<uint32_t>(static_cast<long long>(my_in & 0xffffffffffull) - wg_bit);
            int z = static_cast<int>((my_in >> 40) & 0xffu), slot = static_cast<int>(my_in >> 48);
            uint32_t bad = 0;
            my_cnt = 0;
            dec_span<false, RST>(sh, a, rel, z, slot, end, my_cnt, (FIX || t >= DEC_WARM) ? gs : -1ll, bad, wg_bit);
            my_out = dec_state(static_cast<unsigned long long>(wg_bit + rel), z, slot);
            sh.out[t] = my_out;
        }
        __syncthreads();
        need = false;
        if (t > 0 && live && gs > 0) {                     // (span 0 starts where the string starts: never corrected)
            const unsigned long long pv = sh.out[t - 1];
            if (pv != my_in) {
                my_in = pv;
                need = true;
            }
        }
        if (!__syncthreads_or(need ? 1 : 0)) break;
    }
    if (live && (FIX || t >= DEC_WARM)) {
        a.s_in[gs] = my_in;
        __hip_atomic_store(&a.s_out[gs], my_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.cnt[gs] = my_cnt;
    }
    if (FIX && t == 0) a.flag[0] = 1u;
    if (a.dbg) {
        if (t == 0) {
            atomicMax(&a.dbg[0], passes);
            atomicAdd(&a.dbg[1], passes);
        }
        atomicAdd(&a.dbg[2], decodes);
    }
}

template <bool RST>
__global__ __launch_bounds__(256) void jpeg_dwrite_kernel(DecArgs a)
{
    __shared__ DecShared sh;
    const int g = blockIdx.x, t = threadIdx.x, gt = g * 256 + t;
    dec_stage(sh, a, static_cast<long long>(g) * 256, a.tab, static_cast<int>(sizeof(DecTables)));
    __syncthreads();
    if (gt >= a.nlanes) return;
    const unsigned long long st = a.s_in[gt];
    long long blk = static_cast<long long>(a.first_blk[gt]);
    if (RST && a.nrst > 0) {
        // Block numbers are counted from the restart interval's start, not from the file's: the sync passes go by position and
        // cannot know when an interval's blocks are complete, so the <= 7 bits a DAMAGED interval leaves before its boundary
        // can read as one more complete block there (DC category 0 + EOB is 4 bits) -- a block no decoder that counts MCUs
        // ever sees, and one that would shift every block number behind it (the fuzzer's find, seed 77:
        // tests/golden/damaged_interval_phantom_block.jpg).  With j the last boundary at or before this lane's start:
        // blocks before it = (j + 1) ri_blocks by definition, blocks since = the prefix sums' difference.
        const unsigned long long at = st & 0xffffffffffull;
        int lo_k = 0, hi_k = a.nrst;                               // first boundary AFTER the start
        while (lo_k < hi_k) {
            const int mid = (lo_k + hi_k) >> 1;
            if (8ull * a.rst[mid] <= at) lo_k = mid + 1; else hi_k = mid;
        }
        const int j = lo_k - 1;
        if (j >= 0) {
            const unsigned long long rec = a.rst_rec[j];
            const long long before = static_cast<long long>(a.first_blk[rec >> 32]) + static_cast<long long>(rec & 0xffffffffull);
            blk = static_cast<long long>(j + 1) * a.ri_blocks + (blk - before);
        }
    }
    if (blk >= a.nblk) return;                                     // padding behind the last block
    const unsigned long long wg_bit = static_cast<unsigned long long>(g) * 256u * DEC_SPAN;
    uint32_t rel = static_cast<uint32_t>((st & 0xffffffffffull) - wg_bit);
    int z = static_cast<int>((st >> 40) & 0xffu), slot = static_cast<int>(st >> 48);
    uint32_t cnt = 0, bad = 0;
    const unsigned long long left = a.nbits - wg_bit;              // (this lane exists: its span starts inside the string)
    dec_span<true, RST>(sh, a, rel, z, slot, static_cast<uint32_t>(t + 1) * DEC_SPAN, cnt, blk, bad, static_cast<long long>(wg_bit),
                        left > 0xfffffff0ull ? 0xfffffff0u : static_cast<uint32_t>(left));
    if (bad) atomicOr(a.err, bad);
}

// ---- DC prediction and the blocks ----
struct DcArgs {
    const int16_t *coef;
    uint32_t *dcb;            // [nblk] DC difference + 2048, component by component (Y in scan order, then Cb, then Cr)
    int nblk, nmcu, ny, nc;   // ny: Y blocks per MCU (1, 2 or 4); nc: chroma components (2 or 0)
};

// scan-order block -> its place in the component-major array
__device__ __forceinline__ int dc_place(int b, int nmcu, int ny, int nc)
{
    const int per = ny + nc, m = b / per, j = b - m * per;
    return j < ny ? m * ny + j : (ny + (j - ny)) * nmcu + m;
}

__global__ __launch_bounds__(256) void jpeg_dc_gather_kernel(DcArgs a)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= a.nblk) return;
    a.dcb[dc_place(b, a.nmcu, a.ny, a.nc)] = static_cast<uint32_t>(static_cast<int32_t>(a.coef[static_cast<size_t>(b) * 64]) + 2048);
}

struct IdctArgs {
    const int16_t *coef;
    const uint32_t *dcb;
    const unsigned long long *dcsum;     // exclusive prefix sum of dcb
    uint8_t *out[4];
    int stride[4], nbx[4], nblocks[4];
    int mx, nmcu, hy, vy, nc;            // MCUs per row, MCUs, Y blocks per MCU across / down, chroma components
    int ri;                              // MCUs per restart interval (the DC prediction starts over in each); 0: one interval
    int abs_dc;                          // progressive files (jpeg_prog.cpp): coef[0] IS the DC, no prediction to undo
    uint16_t q[4][64];                   // natural order (a fourth plane -- the black of a CMYK file -- only with abs_dc)
};

__global__ __launch_bounds__(256) void jpeg_didct_kernel(IdctArgs a)
{
    const int plane = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.nblocks[plane]) return;
    const int by = i / a.nbx[plane], bx = i - by * a.nbx[plane];
    const int ny = a.hy * a.vy, per = ny + a.nc;
    int sb, place, start;
    if (plane == 0) {
        const int m = (by / a.vy) * a.mx + bx / a.hy, j = (by % a.vy) * a.hy + bx % a.hy;
        sb = m * per + j;
        place = m * ny + j;
        start = a.ri > 0 ? (m / a.ri) * a.ri * ny : 0;
    } else {
        const int m = by * a.mx + bx;
        sb = m * per + ny + plane - 1;
        const int first = (ny + plane - 1) * a.nmcu;
        place = first + m;
        start = first + (a.ri > 0 ? (m / a.ri) * a.ri : 0);
    }
    const int16_t *cp = a.coef + static_cast<size_t>(sb) * 64;
    int32_t b[64];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(cp + 8 * r);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int32_t cv = static_cast<int16_t>((w[c >> 1] >> (16 * (c & 1))) & 0xffffu);
            b[8 * r + c] = cv * static_cast<int32_t>(a.q[plane][8 * r + c]);
        }
    }
    // the block's DC: the sum of its component's differences from the start of its restart interval up to it
    if (!a.abs_dc) {
        const long long dsum = static_cast<long long>(a.dcsum[place] - a.dcsum[start]) + a.dcb[place] - 2048ll * (place - start + 1);
        b[0] = static_cast<int32_t>(dsum) * static_cast<int32_t>(a.q[plane][0]);
    }
#pragma unroll
    for (int r = 0; r < 8; r++) idct8_row(b[8 * r], b[8 * r + 1], b[8 * r + 2], b[8 * r + 3], b[8 * r + 4], b[8 * r + 5], b[8 * r + 6], b[8 * r + 7]);
#pragma unroll
    for (int c = 0; c < 8; c++) idct8_col(b[c], b[8 + c], b[16 + c], b[24 + c], b[32 + c], b[40 + c], b[48 + c], b[56 + c]);
    const int stride = a.stride[plane];
    uint8_t *op = a.out[plane] + static_cast<size_t>(8 * by) * stride + 8 * bx;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        u32x2 v = {0, 0};
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int32_t s = b[8 * r + c];
            const uint32_t u = s < -128 ? 0u : (s > 127 ? 255u : static_cast<uint32_t>(s + 128));      // reader.go: level shift, clip
            if (c < 4) v.x |= u << (8 * c); else v.y |= u << (8 * (c - 4));
        }
        *reinterpret_cast<u32x2 *>(op + static_cast<size_t>(r) * stride) = v;
    }
}

// ---- the host side (segments, tables, the scan without its stuffing: jpeg_parse.cpp) and the launches ----
constexpr int SCAN_PER_WG_D = 2048;

// data (host): the file.  On return the planes (SLOT_JPEG_DEC_PLANES: Y, Cb, Cr back to back, MCU-padded) are
// enqueued and *f describes them; the scan has been validated (one small read-back).
// SOF2: every scan's entropy decoding on the host (jpeg_prog.cpp: why), the coefficients across PCIe at 2 bytes each, then the
// same dequantisation + IDCT launch as a baseline file's
static int jpeg_decode_planes_progressive(fnx_ctx *ctx, const uint8_t *data, size_t n, JpegFile *f, uint8_t *planes[4], int *ystride, int *cstride)
{
    const long long nmcu = static_cast<long long>(f->mx) * f->my;
    const long long nblk_ll = nmcu * f->nslots;
    if (nblk_ll > JPEG_HOST_MAX_BLOCKS) return jpeg_unsupported("a host-decoded file of more than 4 M blocks (FNX_JPEG_HOST_MAX_BLOCKS)");
    // a first DC scan costs every block at least one bit: a header that promises more blocks than the file has bits is refused
    // before anything is sized by it
    if (8ull * n < static_cast<unsigned long long>(nblk_ll)) return jpeg_corrupt("the file is too short for the image's blocks");
    const int nblk = static_cast<int>(nblk_ll);
    const int ys = 8 * f->hy * f->mx, yh = 8 * f->vy * f->my, cs = 8 * f->mx, chh = 8 * f->my;
    const size_t b_coef = sizeof(int16_t) * 64 * static_cast<size_t>(nblk);
    void *pin = nullptr;
    FNX_TRY(pinned_alloc(ctx, b_coef, &pin));
    std::memset(pin, 0, b_coef);
    FNX_TRY(jpeg_progressive_coefficients(data, n, f, static_cast<int16_t *>(pin)));
    auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
    void *sc = nullptr, *pl = nullptr;
    FNX_TRY(scratch(ctx, SLOT_JPEG_DEC, al(b_coef), &sc));
    const size_t b_y = al(static_cast<size_t>(ys) * yh), b_c = al(static_cast<size_t>(cs) * chh);
    FNX_TRY(scratch(ctx, SLOT_JPEG_DEC_PLANES, b_y + 3 * b_c, &pl));
    planes[0] = static_cast<uint8_t *>(pl);
    planes[1] = planes[0] + b_y;
    planes[2] = planes[1] + b_c;
    planes[3] = planes[2] + b_c;                  // (four components: all of one geometry, b_c == b_y)
    *ystride = ys; *cstride = cs;
    FNX_HIP(hipMemcpyAsync(sc, pin, b_coef, hipMemcpyHostToDevice, ctx->stream));
    IdctArgs ia{};
    ia.coef = static_cast<const int16_t *>(sc); ia.dcb = nullptr; ia.dcsum = nullptr;
    for (int c = 0; c < 4; c++) {
        ia.out[c] = planes[c];
        ia.stride[c] = c ? cs : ys;
        ia.nbx[c] = (c ? cs : ys) / 8;
        ia.nblocks[c] = ia.nbx[c] * ((c ? chh : yh) / 8);
        for (int k = 0; k < 64; k++) ia.q[c][k] = f->q[c][k];
    }
    ia.mx = f->mx; ia.nmcu = static_cast<int>(nmcu); ia.hy = f->hy; ia.vy = f->vy; ia.nc = f->ncomp - 1; ia.ri = 0; ia.abs_dc = 1;
    FNX_TRY(prof_begin(ctx, FNX_PROF_JPEG));
    hipLaunchKernelGGL(jpeg_didct_kernel, dim3((ia.nblocks[0] + 255) / 256, f->ncomp), dim3(256), 0, ctx->stream, ia);
    FNX_HIP(hipGetLastError());
    FNX_TRY(prof_end(ctx));
    f->rounds = 0;
    return FNX_OK;
}

int jpeg_decode_planes(fnx_ctx *ctx, const uint8_t *data, size_t n, JpegFile *f, uint8_t *planes[4], int *ystride, int *cstride)
{
    FNX_TRY(jpeg_parse(data, n, f));
    if (f->progressive) return jpeg_decode_planes_progressive(ctx, data, n, f, planes, ystride, cstride);
    void *pin = nullptr, *tpin = nullptr;
    const size_t cap = (n - f->scan + 64 + 63) & ~size_t(63);           // >= the scan + 4 words of zeros
    const long long nmcu0 = static_cast<long long>(f->mx) * f->my;
    const size_t rst_max = f->ri > 0 ? static_cast<size_t>((nmcu0 + f->ri - 1) / f->ri) : 0;
    FNX_TRY(pinned_alloc(ctx, cap + sizeof(DecTables) + sizeof(DecSyncTables) + 4 * rst_max + 64, &pin));   // one slice: a second request could wrap the ring onto it
    tpin = static_cast<uint8_t *>(pin) + cap;
    size_t nb = 0;
    std::vector<uint32_t> rst;
    FNX_TRY(jpeg_unstuff(data, n, *f, static_cast<uint8_t *>(pin), &nb, &rst));
    uint32_t *rpin = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(tpin) + sizeof(DecTables) + sizeof(DecSyncTables));
    if (!rst.empty()) std::memcpy(rpin, rst.data(), 4 * rst.size());
    const size_t nwords = (nb + 3) / 4 + 4;
    std::memset(static_cast<uint8_t *>(pin) + nb, 0, nwords * 4 - nb);
    const unsigned long long nbits = 8ull * nb;
    const size_t nlanes_z = static_cast<size_t>((nbits + DEC_SPAN - 1) / DEC_SPAN);
    const long long nmcu = static_cast<long long>(f->mx) * f->my;
    const long long nblk_ll = nmcu * f->nslots;
    if (nlanes_z == 0) return jpeg_corrupt("an empty scan");
    if (nlanes_z >= (size_t(1) << 30) || nblk_ll >= (1ll << 30)) return jpeg_unsupported("a file this large");
    // every block costs at least two bits (a DC code and an end of block): a header that promises more blocks than the scan
    // can hold is refused BEFORE anything is sized by it (65 535 x 65 535 in the frame header of a 1 KB file)
    if (nbits < 2ull * static_cast<unsigned long long>(nblk_ll)) return jpeg_corrupt("the scan is too short for the image's blocks");
    const int nlanes = static_cast<int>(nlanes_z), nblk = static_cast<int>(nblk_ll);
    const int nwg = (nlanes + DEC_OWN - 1) / DEC_OWN, nwg_write = (nlanes + 255) / 256;
    const int ys = 8 * f->hy * f->mx, yh = 8 * f->vy * f->my, cs = 8 * f->mx, chh = 8 * f->my;

    auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t lanes_pad = static_cast<size_t>(nwg) * 256;
    const size_t b_ecs = al(nwords * 4 + 64), b_tab = al(sizeof(DecTables) + sizeof(DecSyncTables) + 4 * rst_max + 16), b_state = al(8 * lanes_pad), b_cnt = al(4 * lanes_pad),
                 b_first = al(8 * lanes_pad), b_tot = al(8 * (lanes_pad / SCAN_PER_WG_D + 2)), b_flag = al(4 * 64 + 16),
                 b_coef = al(sizeof(int16_t) * 64 * static_cast<size_t>(nblk)), b_dcb = al(4 * static_cast<size_t>(nblk)),
                 b_dcs = al(8 * static_cast<size_t>(nblk)), b_tot2 = al(8 * (static_cast<size_t>(nblk) / SCAN_PER_WG_D + 2)),
                 b_rec = al(8 * (rst.size() + 1));
    void *sc = nullptr;
    FNX_TRY(scratch(ctx, SLOT_JPEG_DEC, b_ecs + b_tab + 2 * b_state + b_cnt + b_first + b_tot + b_flag + b_coef + b_dcb + b_dcs + b_tot2 + b_rec, &sc));
    unsigned char *p = static_cast<unsigned char *>(sc);
    uint32_t *d_ecs = reinterpret_cast<uint32_t *>(p); p += b_ecs;
    DecTables *d_tab = reinterpret_cast<DecTables *>(p); p += b_tab;
    unsigned long long *d_in = reinterpret_cast<unsigned long long *>(p); p += b_state;
    unsigned long long *d_out = reinterpret_cast<unsigned long long *>(p); p += b_state;
    uint32_t *d_cnt = reinterpret_cast<uint32_t *>(p); p += b_cnt;
    unsigned long long *d_first = reinterpret_cast<unsigned long long *>(p); p += b_first;
    unsigned long long *d_tot = reinterpret_cast<unsigned long long *>(p); p += b_tot;
    uint32_t *d_flag = reinterpret_cast<uint32_t *>(p); p += b_flag;          // [0..63] rounds, then err, then 2 x u64 totals
    int16_t *d_coef = reinterpret_cast<int16_t *>(p); p += b_coef;
    uint32_t *d_dcb = reinterpret_cast<uint32_t *>(p); p += b_dcb;
    unsigned long long *d_dcs = reinterpret_cast<unsigned long long *>(p); p += b_dcs;
    unsigned long long *d_tot2 = reinterpret_cast<unsigned long long *>(p); p += b_tot2;
    unsigned long long *d_rec = reinterpret_cast<unsigned long long *>(p);
    void *pl = nullptr;
    const size_t b_y = al(static_cast<size_t>(ys) * yh), b_c = al(static_cast<size_t>(cs) * chh);
    FNX_TRY(scratch(ctx, SLOT_JPEG_DEC_PLANES, b_y + 2 * b_c, &pl));
    planes[0] = static_cast<uint8_t *>(pl);
    planes[1] = planes[0] + b_y;
    planes[2] = planes[1] + b_c;
    *ystride = ys; *cstride = cs;

    std::memcpy(tpin, &f->tab, sizeof(DecTables));
    {   // the sync passes' form of the tables (common.hpp: DecSyncTables)
        DecSyncTables *ts = reinterpret_cast<DecSyncTables *>(static_cast<uint8_t *>(tpin) + sizeof(DecTables));
        std::memcpy(ts->limit, f->tab.limit, sizeof(ts->limit));
        std::memcpy(ts->delta, f->tab.delta, sizeof(ts->delta));
        std::memcpy(ts->value, f->tab.value, sizeof(ts->value));
        auto effect = [](uint32_t e, bool ac, uint32_t *bits, uint32_t *step) {        // one symbol of the write pass's table
            const uint32_t len = e >> 8, sym = e & 0xffu, sz = sym & 15u, r = sym >> 4;
            *bits = len + sz;
            *step = !ac ? 1u : ((sz == 0 && r != 15) ? 64u : r + 1u);
        };
        for (int t = 0; t < 2; t++)
            for (int i = 0; i < (1 << DEC_FAST_BITS); i++) {
                const uint32_t e = f->tab.fast[t][i];
                uint32_t b1 = 0, s1 = 0;
                if (e) effect(e, false, &b1, &s1);
                ts->st[t][i] = e ? (b1 | (s1 << 8)) : 0u;
            }
        for (int t = 0; t < 2; t++)
            for (int i = 0; i < (1 << DEC_FAST_BITS); i++) {
                const uint32_t e = f->tab.fast[2 + t][i];
                uint32_t v = 0;
                if (e) {
                    uint32_t b1, s1;
                    effect(e, true, &b1, &s1);
                    v = b1 | (s1 << 8);
                    // a second symbol: the block goes on, and the code after symbol 1's bits lies inside the prefix
                    if (s1 < 64 && b1 < static_cast<uint32_t>(DEC_FAST_BITS)) {
                        const uint32_t rest = (static_cast<uint32_t>(i) << b1) & ((1u << DEC_FAST_BITS) - 1u);   // what is known of the bits behind it
                        const uint32_t e2 = f->tab.fast[2 + t][rest];
                        if (e2 && (e2 >> 8) <= static_cast<uint32_t>(DEC_FAST_BITS) - b1) {      // every prefix with these known bits holds this code
                            uint32_t b2, s2;
                            effect(e2, true, &b2, &s2);
                            v |= ((b1 + b2) << 16) | ((s1 + s2) << 24);
                        }
                    }
                }
                ts->st[2 + t][i] = v;
            }
    }
    FNX_HIP(hipMemcpyAsync(d_ecs, pin, nwords * 4, hipMemcpyHostToDevice, ctx->stream));
    FNX_HIP(hipMemcpyAsync(d_tab, tpin, sizeof(DecTables) + sizeof(DecSyncTables) + 4 * rst.size(), hipMemcpyHostToDevice, ctx->stream));
    FNX_HIP(hipMemsetAsync(d_flag, 0, b_flag, ctx->stream));
    FNX_HIP(hipMemsetAsync(d_coef, 0, sizeof(int16_t) * 64 * static_cast<size_t>(nblk), ctx->stream));

    DecArgs a{};
    a.ecs = d_ecs; a.tab = d_tab; a.tab_sync = reinterpret_cast<const DecSyncTables *>(d_tab + 1); a.s_in = d_in; a.s_out = d_out; a.cnt = d_cnt; a.flag = d_flag;
    a.first_blk = d_first; a.coef = d_coef; a.err = d_flag + 64;
    const char *trc = std::getenv("FNX_JPEG_TRACE");
    const bool trace = trc && trc[0] == '1';
    a.dbg = trace ? d_flag + 72 : nullptr;
    a.nwords = static_cast<long long>(nwords); a.nbits = nbits; a.nlanes = nlanes; a.nblk = nblk; a.nslots = f->nslots;
    a.dcpack = f->dcpack; a.acpack = f->acpack;
    a.rst = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(d_tab) + sizeof(DecTables) + sizeof(DecSyncTables)); a.nrst = static_cast<int>(rst.size());
    a.ri_blocks = f->ri * f->nslots;
    a.rst_rec = d_rec;
    const bool has_rst = !rst.empty();
    if (has_rst) FNX_HIP(hipMemsetAsync(d_rec, 0, 8 * rst.size(), ctx->stream));
    FNX_TRY(prof_begin(ctx, FNX_PROF_JPEG));
    if (has_rst) hipLaunchKernelGGL((jpeg_dsync_kernel<false, true>), dim3(nwg), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL((jpeg_dsync_kernel<false, false>), dim3(nwg), dim3(256), 0, ctx->stream, a);
    FNX_HIP(hipGetLastError());
    FNX_TRY(prof_end(ctx));
    // rounds across workgroups, two per read-back; a round that changes nothing ends it (at most nwg rounds can change something)
    f->rounds = 0;
    if (nwg > 1) {
        uint32_t flags[64];
        int r = 0;
        for (;;) {
            const int r0 = r;
            for (int k = 0; k < 2 && r < 64; k++, r++) {
                a.flag = d_flag + r;
                if (has_rst) hipLaunchKernelGGL((jpeg_dsync_kernel<true, true>), dim3(nwg), dim3(256), 0, ctx->stream, a);
                else hipLaunchKernelGGL((jpeg_dsync_kernel<true, false>), dim3(nwg), dim3(256), 0, ctx->stream, a);
            }
            FNX_HIP(hipGetLastError());
            FNX_TRY(fetch_bytes(ctx, d_flag, flags, sizeof(uint32_t) * static_cast<size_t>(r)));
            bool quiet = false;
            for (int k = r0; k < r; k++) quiet = quiet || flags[k] == 0;
            f->rounds = r;
            if (quiet) break;
            if (r >= 64) {                       // start the flags over: the pattern that needs this many rounds is contrived, not wrong
                FNX_HIP(hipMemsetAsync(d_flag, 0, 4 * 64, ctx->stream));
                r = 0;
            }
        }
    }
    FNX_TRY(launch_scan(ctx, d_cnt, d_first, d_tot, nlanes, reinterpret_cast<unsigned long long *>(d_flag + 66)));
    if (has_rst) hipLaunchKernelGGL(jpeg_dwrite_kernel<true>, dim3(nwg_write), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(jpeg_dwrite_kernel<false>, dim3(nwg_write), dim3(256), 0, ctx->stream, a);
    DcArgs da{d_coef, d_dcb, nblk, static_cast<int>(nmcu), f->hy * f->vy, f->ncomp - 1};
    hipLaunchKernelGGL(jpeg_dc_gather_kernel, dim3((nblk + 255) / 256), dim3(256), 0, ctx->stream, da);
    FNX_TRY(launch_scan(ctx, d_dcb, d_dcs, d_tot2, nblk, nullptr));
    IdctArgs ia{};
    ia.coef = d_coef; ia.dcb = d_dcb; ia.dcsum = d_dcs;
    for (int c = 0; c < 3; c++) {
        ia.out[c] = planes[c];
        ia.stride[c] = c ? cs : ys;
        ia.nbx[c] = (c ? cs : ys) / 8;
        ia.nblocks[c] = ia.nbx[c] * ((c ? chh : yh) / 8);
        for (int k = 0; k < 64; k++) ia.q[c][k] = f->q[c][k];
    }
    ia.mx = f->mx; ia.nmcu = static_cast<int>(nmcu); ia.hy = f->hy; ia.vy = f->vy; ia.nc = f->ncomp - 1; ia.ri = f->ri; ia.abs_dc = 0;
    hipLaunchKernelGGL(jpeg_didct_kernel, dim3((ia.nblocks[0] + 255) / 256, f->ncomp), dim3(256), 0, ctx->stream, ia);
    FNX_HIP(hipGetLastError());
    // what the scan held: blocks finished inside the string, and the write pass's complaints
    struct { uint32_t err, pad; unsigned long long blocks; } chk;
    FNX_TRY(fetch_bytes(ctx, d_flag + 64, &chk, sizeof(chk)));
    if (trace) {
        uint32_t dbg[3];
        FNX_TRY(fetch_bytes(ctx, d_flag + 72, dbg, sizeof(dbg)));
        std::fprintf(stderr, "[fennec jpeg] %d x %d, %zu scan bytes, %d lanes in %d workgroups: passes max %u avg %.1f, lane-decodes %u (%.2f per lane), "
                             "cross-workgroup rounds %d\n", f->w, f->h, nb, nlanes, nwg, dbg[0], double(dbg[1]) / nwg, dbg[2], double(dbg[2]) / nlanes, f->rounds);
    }
    if (chk.blocks < static_cast<unsigned long long>(nblk)) return jpeg_corrupt("the scan ends before the last block");
    if (chk.err & 8u) return jpeg_corrupt("the scan ends before the last block");
    if (chk.err & 16u) return jpeg_corrupt("a restart interval does not hold the blocks it should");
    if (chk.err) return jpeg_corrupt("the scan holds a code outside its Huffman table or a run past the end of a block");
    return FNX_OK;
}

}  // namespace fnx
