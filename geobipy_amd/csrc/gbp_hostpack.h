// gbp_hostpack.h -- host-side helper of the results containers: rows of a hit map held as RUNS -> one zlib stream per row.
//
// A conductivity-depth hit map is [value bins, depth cells] int32 per sounding (440 KB), a few thousand runs of equal counts (depth is
// the fast axis, a layer fills a run of depth cells of one value bin).  An HDF5 container stores it chunked, one sounding per chunk,
// with the deflate filter (geobipy_amd/h5lite.py); a chunk handed to H5Dwrite_chunk must be a zlib stream of the chunk's dense bytes.
// zlib itself has to read all 440 KB to find the runs again (0.5 s per flight line at level 1); here the stream is written FROM the
// runs -- O(runs), not O(cells): a run of R cells of value v is, in deflate's terms, the four bytes of v followed by a copy from four
// bytes back of length 4R - 4; when the cell before it shares v's upper three bytes (the usual case: both counts below 256) the copy
// can start one byte after v's low byte (length 4R - 1) and one literal is enough.  One block with the fixed Huffman code (RFC 1951
// 3.2.6); Adler-32 of the dense bytes in closed form per run (RFC 1950).  ~7 KB per row against zlib level 1's ~6 KB.
// No reference counterpart (the reference stores the maps dense and uncompressed).
#pragma once
#include <cstdint>
#include <cstddef>

namespace hostpack {

struct BitWriter {
    uint8_t* out;
    size_t cap, n = 0;
    uint64_t acc = 0;
    int bits = 0;
    bool overflow = false;
    inline void put(uint32_t v, int nb)          // nb <= 24 bits, least-significant first (the order of deflate's bit stream)
    {
        acc |= (uint64_t)v << bits;
        bits += nb;
        while (bits >= 8) {
            if (n < cap) out[n++] = (uint8_t)acc; else overflow = true;
            acc >>= 8;
            bits -= 8;
        }
    }
    inline void flush()
    {
        if (bits > 0) {
            if (n < cap) out[n++] = (uint8_t)acc; else overflow = true;
            acc = 0; bits = 0;
        }
    }
};

static inline uint32_t rev(uint32_t v, int nb)   // Huffman codes go into the stream most-significant bit first
{
    uint32_t r = 0;
    for (int i = 0; i < nb; ++i) r |= ((v >> i) & 1u) << (nb - 1 - i);
    return r;
}
static inline void literal(BitWriter& w, uint32_t b)
{
    if (b < 144) w.put(rev(0x30 + b, 8), 8);
    else w.put(rev(0x190 + (b - 144), 9), 9);
}
// one copy of `len` (3 .. 258) bytes from `dist` (1 or 4) bytes back
static inline void match(BitWriter& w, int len, int dist)
{
    static const int base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const int extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    int c = 28;
    while (base[c] > len) --c;
    if (c == 28 && len != 258) c = 27;
    const int sym = 257 + c;
    if (sym < 280) w.put(rev(sym - 256, 7), 7);
    else w.put(rev(0xC0 + (sym - 280), 8), 8);
    if (extra[c]) w.put((uint32_t)(len - base[c]), extra[c]);
    w.put(rev(dist == 1 ? 0 : 3, 5), 5);          // distance codes 0 (= 1) and 3 (= 4): no extra bits
}
static inline void copy_chain(BitWriter& w, int64_t total, int dist)
{
    while (total > 0) {
        int64_t c = total < 258 ? total : 258;
        if (total - c == 1 || total - c == 2) c -= 2;     // never leave a tail shorter than the minimum match
        match(w, (int)c, dist);
        total -= c;
    }
}

// One row.  start[0] must be 0, starts strictly increasing and < cells.  Returns the stream's size, or 0 when `cap` is too small.
static size_t row_to_zlib(int64_t cells, int64_t nruns, const int32_t* start, const int32_t* value, uint8_t* out, size_t cap)
{
    BitWriter w{out, cap};
    w.put(0x78, 8); w.put(0x01, 8);               // zlib header: deflate, 32 KB window, fastest
    w.put(1, 1); w.put(1, 2);                     // final block, fixed Huffman code
    const uint64_t MOD = 65521;
    uint64_t s1 = 1, s2 = 0;
    uint32_t prev = 0;
    bool first = true;
    for (int64_t r = 0; r < nruns; ++r) {
        const int64_t R = (r + 1 < nruns ? (int64_t)start[r + 1] : cells) - (int64_t)start[r];
        if (R <= 0) return 0;
        const uint32_t v = (uint32_t)value[r];
        const uint32_t b[4] = {v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF, v >> 24};
        if (!first && (v >> 8) == (prev >> 8)) {
            literal(w, b[0]);
            copy_chain(w, 4 * R - 1, 4);
        } else {
            for (int j = 0; j < 4; ++j) literal(w, b[j]);
            copy_chain(w, 4 * R - 4, 4);
        }
        // Adler-32 over the run's n = 4R bytes: s2 += n s1 + sum_j b_j (R (n - j) - 2 R (R - 1)),  s1 += R sum_j b_j
        const uint64_t n = 4 * (uint64_t)R, RR = (uint64_t)R;
        uint64_t add2 = (n % MOD) * s1 % MOD;
        for (int j = 0; j < 4; ++j) {
            const uint64_t coef = (RR * (n - (uint64_t)j) - 2 * RR * (RR - 1)) % MOD;       // R (n - j) >= 2 R (R - 1) + R: no wrap
            add2 = (add2 + coef * b[j]) % MOD;
        }
        s2 = (s2 + add2) % MOD;
        s1 = (s1 + (RR % MOD) * (b[0] + b[1] + b[2] + b[3])) % MOD;
        prev = v;
        first = false;
    }
    w.put(0, 7);                                  // end of block (symbol 256: seven zero bits)
    w.flush();
    const uint32_t adler = (uint32_t)((s2 << 16) | s1);
    for (int sh = 24; sh >= 0; sh -= 8) w.put((adler >> sh) & 0xFF, 8);
    w.flush();
    return w.overflow ? 0 : w.n;
}

}  // namespace hostpack
