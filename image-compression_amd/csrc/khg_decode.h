// khg_decode.h -- .comp header parsing, tree rebuild, and chunked self-synchronising
// parallel Huffman decode.
//
// The format has no synchronisation markers (Writer.h:25-32 just concatenates codes), so
// each plane's stream is cut into 512-bit subsequences, one thread each.  A thread that
// starts inside a codeword decodes garbage for a few symbols and then falls into step
// with the true codeword boundaries (Huffman codes self-synchronise), so:
//   sync pass : every thread decodes from its assumed start to the first codeword boundary
//               at or after its subsequence's end; a workgroup (256 subsequences) repeats
//               this, each thread restarting from its predecessor's arrival point, until
//               nothing changes.  The workgroup's own start comes from the previous
//               workgroup's arrival in the previous pass; pass p+1 also verifies that every
//               arrival equals pass p's -- then all starts are exact by induction from the
//               first workgroup, whose start (bit 0) is exact.
//   scan      : symbol counts -> output index of every subsequence
//   write     : every thread decodes its subsequence once more from its exact start and
//               stores the symbols (Huffman.h:33-55: symbol i lands at raster index i).
#pragma once
#include "khg_common.h"
#include "khg_huffman.h"

enum {
    KHG_SUB_BITS = 512,       // bits per subsequence
    KHG_DEC_BLOCK = 256,      // subsequences per workgroup
    KHG_LUT_BITS = 12,
    KHG_LUT_SIZE = 1 << KHG_LUT_BITS,
    KHG_NOT_CONVERGED = 1     // internal status: the host must enqueue another sync pass
};

// Parsed header plus derived layout, written by khg_dec_header_body.
struct KhgDecHeader {
    int32_t nbits[3];         // stream bits without the pad (Decompress.h:37-39)
    int32_t k[3];             // table sizes
    uint32_t stream_byte[3];  // byte offset of each plane's stream in the file
    uint32_t table_byte[3];   // byte offset of each plane's table
    int32_t lvl;
    int32_t rows, cols;
    int32_t ok;
};

struct KhgSubInfo {
    uint32_t start;  // exact bit position of the first codeword starting in this subsequence
    uint32_t count;  // symbols whose codeword starts in it
    uint32_t out;    // raster index of the first of them (filled by the scan)
    uint32_t pad_;
};

struct KhgDecWorkspace {
    double* plane[3];   // reconstructed Y / Cb / Cr (f64)
    uint32_t* lut;      // [3][KHG_LUT_SIZE] first-level decode table, then block arrival arrays
    KhgSubInfo* sub;    // per subsequence, planes concatenated
    KhgDecHeader* hdr;
};

struct KhgDecWorkspaceSizes {
    size_t plane[3], lut, sub;
};

static inline uint32_t khg_max_subs(uint64_t nsym)
{
    // a plane's stream has fewer than 2^31 bits and at most 56 bits per symbol
    uint64_t bits = nsym * KHG_MAX_CODE_BITS;
    if (bits > 0x7FFFFFFFull) bits = 0x7FFFFFFFull;
    return (uint32_t)((bits + KHG_SUB_BITS - 1) / KHG_SUB_BITS);
}

struct KhgDecArgs {
    const uint8_t* comp;   // file bytes (4-byte aligned)
    uint64_t len;
    int rows, cols;        // expected dimensions
    KhgDecHeader* hdr;
    KhgResult* res;
    int32_t* tab_cnt;      // [3][KHG_NBINS]
    int32_t* tab_sym;
    int32_t* tree_left;    // [3][2*KHG_NBINS]
    int32_t* tree_right;
    int32_t* tree_root;    // [3]
    uint32_t* big_scratch;
    uint32_t* lut;         // [3][KHG_LUT_SIZE]
    uint32_t* arrive[2];   // [nblocks] block arrival positions, double-buffered by pass parity
    uint32_t* mismatch;    // [1]
    KhgSubInfo* sub;
    uint32_t sub0[3];      // first subsequence index of each plane (capacity layout)
    uint32_t blk0[3];      // first workgroup index of each plane
    uint32_t nblocks;
    int16_t* sym[3];
    uint32_t n[3];         // symbols per plane
    int pass;
};

static inline void khg_dec_workspace_sizes(int rows, int cols, KhgDecWorkspaceSizes* s)
{
    size_t ny = (size_t)rows * cols, nc = ny / 4;
    s->plane[0] = ny * sizeof(double);
    s->plane[1] = s->plane[2] = nc * sizeof(double);
    size_t subs = (size_t)khg_max_subs(ny) + 2 * (size_t)khg_max_subs(nc) + 3 * KHG_DEC_BLOCK;
    size_t blocks = subs / KHG_DEC_BLOCK + 4;
    s->lut = (3 * (size_t)KHG_LUT_SIZE + 2 * blocks + 16) * sizeof(uint32_t);
    s->sub = subs * sizeof(KhgSubInfo);
}

KHG_DEV int32_t khg_load_le32(const uint8_t* p)
{
    return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
}

// One thread.  Reader.h:5-23 plus the consistency checks the reference omits.
KHG_DEV void khg_dec_header_body(const KhgDecArgs& a)
{
    KHG_FOR_THREADS(256)
    {
        if (tid == 0) {
            KhgDecHeader* h = a.hdr;
            const uint8_t* c = a.comp;
            int ok = a.len >= 36;
            int32_t len[3] = {0, 0, 0}, k[3] = {0, 0, 0}, pad[3] = {0, 0, 0};
            if (ok) {
                for (int p = 0; p < 3; ++p) {
                    len[p] = khg_load_le32(c + 4 * p);
                    k[p] = khg_load_le32(c + 12 + 4 * p);
                    pad[p] = c[32 + p];
                }
                int32_t hh = khg_load_le32(c + 24), ww = khg_load_le32(c + 28);
                h->lvl = c[35];
                h->rows = hh;
                h->cols = ww;
                if (hh != a.rows || ww != a.cols || h->lvl > 2) ok = 0;
                uint64_t off = 36;
                for (int p = 0; p < 3 && ok; ++p) {
                    if (len[p] < 0 || (len[p] & 7) || pad[p] > 7 || pad[p] > len[p]) ok = 0;
                    if (k[p] < 1 || k[p] > KHG_NBINS) ok = 0;
                    h->stream_byte[p] = (uint32_t)off;
                    off += (uint64_t)(len[p] >> 3);
                }
                for (int p = 0; p < 3 && ok; ++p) {
                    h->table_byte[p] = (uint32_t)off;
                    off += 8ull * (uint64_t)k[p];
                    h->nbits[p] = len[p] - pad[p];
                    h->k[p] = k[p];
                    // a single-entry table has the empty code: its stream must be empty (Huffman.h:57-68)
                    if (k[p] == 1 && h->nbits[p] != 0) ok = 0;
                    if (k[p] > 1 && (uint64_t)h->nbits[p] > (uint64_t)a.n[p] * KHG_MAX_CODE_BITS) ok = 0;
                }
                if (ok && off > a.len) ok = 0;
            }
            h->ok = ok;
            if (!ok) khg_set_status(&a.res->status, -3);
            a.mismatch[0] = 0;
        }
    }
}

struct KhgDecTreeSmem {
    alignas(16) uint32_t w[6 * KHG_TREE_LDS_K];
    int32_t root;
};

// bid = colour component.  Rebuilds the tree from the stored table with the encoder's own
// construction (Decompress.h:45-47) and fills the first-level decode table.
// LUT entry: bit 31 set = leaf reached within KHG_LUT_BITS bits: bits 16..23 code length,
// bits 0..15 symbol (int16); otherwise the entry is the node reached after KHG_LUT_BITS bits.
KHG_DEV void khg_dec_tree_body(const KhgDecArgs& a, int bid, KhgDecTreeSmem& sh)
{
    const KhgDecHeader* h = a.hdr;
    if (!h->ok) return;
    const int k = h->k[bid];
    int32_t* tcnt = a.tab_cnt + (size_t)bid * KHG_NBINS;
    int32_t* tsym = a.tab_sym + (size_t)bid * KHG_NBINS;
    int32_t* left = a.tree_left + (size_t)bid * 2 * KHG_NBINS;
    int32_t* right = a.tree_right + (size_t)bid * 2 * KHG_NBINS;
    uint32_t* lut = a.lut + (size_t)bid * KHG_LUT_SIZE;
    const uint8_t* tb = a.comp + h->table_byte[bid];

    KHG_FOR_THREADS(256)
    {
        for (int i = tid; i < k; i += 256) {  // Reader.h:39-54
            int32_t cnt = khg_load_le32(tb + 8 * (size_t)i), s = khg_load_le32(tb + 8 * (size_t)i + 4);
            tcnt[i] = cnt;
            tsym[i] = s;
            if (s < -32768 || s > 32767) khg_set_status(&a.res->status, -4);
        }
    }
    KHG_BARRIER();
    uint64_t* heap;
    uint32_t *nfreq, *parent;
    if (k <= KHG_TREE_LDS_K) {
        heap = (uint64_t*)sh.w;
        nfreq = sh.w + 2 * KHG_TREE_LDS_K;
        parent = sh.w + 4 * KHG_TREE_LDS_K;
    } else {
        uint32_t* g = a.big_scratch + (size_t)bid * 10 * KHG_NBINS;
        heap = (uint64_t*)g;
        nfreq = g + 2 * KHG_NBINS;
        parent = g + 4 * KHG_NBINS;
    }
    KHG_FOR_THREADS(256)
    {
        if (tid == 0) {
            sh.root = khg_tree_build(tcnt, k, heap, nfreq, parent, left, right);
            a.tree_root[bid] = sh.root;
        }
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        const int root = sh.root;
        for (int v = tid; v < KHG_LUT_SIZE; v += 256) {
            int node = root, d = 0;
            uint32_t e;
            if (k == 1) {
                e = 0x80000000u;  // never consulted: a single-entry plane has no bits
            } else {
                while (d < KHG_LUT_BITS && left[node] >= 0) {
                    node = ((v >> (KHG_LUT_BITS - 1 - d)) & 1) ? right[node] : left[node];
                    ++d;
                }
                if (left[node] < 0) e = 0x80000000u | ((uint32_t)d << 16) | (uint32_t)(uint16_t)(int16_t)tsym[node];
                else e = (uint32_t)node;
            }
            lut[v] = e;
        }
    }
}

// ---------------------------------------------------------------- bit reader

struct KhgBits {
    const uint32_t* words;  // file as aligned 32-bit words
    uint64_t base;          // bit offset of the plane's stream in the file
    uint32_t nbits;         // stream length in bits
    uint32_t nwords;        // readable words
};

// 32 stream bits starting at stream position pos, first bit in bit 31 (zeros past the end).
KHG_DEV uint32_t khg_peek32(const KhgBits& b, uint32_t pos)
{
    uint64_t g = b.base + pos;
    uint32_t w = (uint32_t)(g >> 5), s = (uint32_t)(g & 31);
    uint32_t w0 = w < b.nwords ? khg_bswap32(b.words[w]) : 0;
    uint32_t w1 = w + 1 < b.nwords ? khg_bswap32(b.words[w + 1]) : 0;
    return s ? (w0 << s) | (w1 >> (32 - s)) : w0;
}

struct KhgDecLut {
    uint32_t lut[KHG_LUT_SIZE];
    uint32_t arrive[KHG_DEC_BLOCK];
    uint32_t count[KHG_DEC_BLOCK];
    uint32_t used[KHG_DEC_BLOCK];
    uint32_t changed[2];
    uint32_t blk_start;
};

// Decode one codeword at pos.  Returns its length (0 = the stream ends inside it) and symbol.
KHG_DEV uint32_t khg_decode_one(const KhgBits& b, const uint32_t* lut, const int32_t* left, const int32_t* right,
                                const int32_t* tsym, uint32_t pos, int* sym)
{
    uint32_t window = khg_peek32(b, pos);
    uint32_t e = lut[window >> (32 - KHG_LUT_BITS)];
    uint32_t len;
    if (e & 0x80000000u) {
        len = (e >> 16) & 0xFF;
        *sym = (int)(int16_t)(e & 0xFFFF);
    } else {
        // code longer than KHG_LUT_BITS: continue the walk of Huffman.h:39-51 bit by bit
        int node = (int)e;
        len = KHG_LUT_BITS;
        while (left[node] >= 0) {
            if (pos + len >= b.nbits) return 0;
            uint32_t bit = khg_peek32(b, pos + len) >> 31;
            node = bit ? right[node] : left[node];
            ++len;
        }
        *sym = tsym[node];
    }
    if (pos + len > b.nbits) return 0;
    return len;
}

// Decode from `start` to the first codeword boundary at or after `end` (or until the stream
// cannot supply another whole codeword).  Returns the arrival position; *count = symbols decoded.
KHG_DEV uint32_t khg_decode_span(const KhgBits& b, const uint32_t* lut, const int32_t* left, const int32_t* right,
                                 const int32_t* tsym, uint32_t start, uint32_t end, uint32_t* count)
{
    uint32_t pos = start, n = 0;
    while (pos < end) {
        int s;
        uint32_t len = khg_decode_one(b, lut, left, right, tsym, pos, &s);
        if (len == 0) {
            pos = 0xFFFFFFFFu;  // stream exhausted inside a codeword: nothing starts after this point
            break;
        }
        pos += len;
        ++n;
    }
    *count = n;
    return pos;
}

KHG_DEV int khg_block_plane(const KhgDecArgs& a, uint32_t blk)
{
    return blk >= a.blk0[2] ? 2 : blk >= a.blk0[1] ? 1 : 0;
}

KHG_DEV KhgBits khg_plane_bits(const KhgDecArgs& a, int p)
{
    KhgBits b;
    b.words = (const uint32_t*)a.comp;
    b.base = 8ull * a.hdr->stream_byte[p];
    b.nbits = (uint32_t)a.hdr->nbits[p];
    b.nwords = (uint32_t)(a.len >> 2);
    return b;
}

// One workgroup = 256 consecutive subsequences of one plane.
KHG_DEV void khg_dec_sync_body(const KhgDecArgs& a, uint32_t blk, KhgDecLut& sh)
{
    if (!a.hdr->ok) return;
    const int p = khg_block_plane(a, blk);
    const KhgBits bits = khg_plane_bits(a, p);
    const uint32_t lb = blk - a.blk0[p];                       // block index inside the plane
    const uint32_t blk_begin = lb * KHG_DEC_BLOCK * KHG_SUB_BITS;
    if (blk_begin >= bits.nbits && !(lb == 0)) return;         // beyond the stream (block 0 still records zeros)
    const int32_t* left = a.tree_left + (size_t)p * 2 * KHG_NBINS;
    const int32_t* right = a.tree_right + (size_t)p * 2 * KHG_NBINS;
    const int32_t* tsym = a.tab_sym + (size_t)p * KHG_NBINS;
    const uint32_t* glut = a.lut + (size_t)p * KHG_LUT_SIZE;
    const uint32_t* prev = a.arrive[(a.pass + 1) & 1];
    uint32_t* cur = a.arrive[a.pass & 1];

    KHG_FOR_THREADS(256)
    {
        for (int i = tid; i < KHG_LUT_SIZE; i += 256) sh.lut[i] = glut[i];
        if (tid == 0) {
            uint32_t s = blk_begin;  // pass 0: nominal; later: where the previous workgroup arrived
            if (lb == 0) s = 0;
            else if (a.pass > 0) s = prev[blk - 1];
            sh.blk_start = s;
            sh.changed[0] = 0;
            sh.changed[1] = 0;
        }
        sh.used[tid] = 0xFFFFFFFEu;  // "no start decoded yet"
        sh.arrive[tid] = blk_begin + ((uint32_t)tid + 1) * KHG_SUB_BITS;  // nominal arrival for the first round
        sh.count[tid] = 0;
    }
    KHG_BARRIER();
    // Round r: every thread restarts from its predecessor's current arrival if that differs from
    // the start it last decoded from.  A correct start propagates at least one thread per round.
    for (int round = 0; round <= KHG_DEC_BLOCK; ++round) {
        uint32_t mystart[KHG_NSLOT(256)];
        KHG_FOR_THREADS(256)
        {
            uint32_t s = tid == 0 ? sh.blk_start : sh.arrive[tid - 1];
            mystart[KHG_SLOT] = s;
        }
        KHG_BARRIER();
        KHG_FOR_THREADS(256)
        {
            uint32_t s = mystart[KHG_SLOT];
            uint32_t end = blk_begin + ((uint32_t)tid + 1) * KHG_SUB_BITS;
            if (s != sh.used[tid]) {
                uint32_t cnt = 0, arr;
                if (s >= bits.nbits) {
                    arr = s;  // nothing starts in this subsequence
                } else if (s >= end) {
                    arr = s;  // cannot happen for SUB_BITS > max code length; kept for safety
                } else {
                    arr = khg_decode_span(bits, sh.lut, left, right, tsym, s, end < bits.nbits ? end : bits.nbits, &cnt);
                    if (arr != 0xFFFFFFFFu && arr < end && arr >= bits.nbits) arr = bits.nbits;
                }
                sh.used[tid] = s;
                sh.arrive[tid] = arr;
                sh.count[tid] = cnt;
                sh.changed[round & 1] = 1;
            }
        }
        KHG_BARRIER();
        const bool any = sh.changed[round & 1] != 0;
        KHG_FOR_THREADS(256)
        {
            if (tid == 0) sh.changed[(round + 1) & 1] = 0;
        }
        KHG_BARRIER();
        if (!any) break;
    }
    KHG_FOR_THREADS(256)
    {
        KhgSubInfo si;
        si.start = sh.used[tid];
        si.count = sh.count[tid];
        si.out = 0;
        si.pad_ = 0;
        a.sub[a.sub0[p] + lb * KHG_DEC_BLOCK + tid] = si;
        if (tid == KHG_DEC_BLOCK - 1) {
            uint32_t arr = sh.arrive[tid];
            if (a.pass > 0 && prev[blk] != arr) khg_atomic_add(a.mismatch, 1u);
            cur[blk] = arr;
        }
    }
}

struct KhgDecScanSmem {
    uint64_t psum[3][1024];
    uint32_t nsub[3];
};

// Single workgroup, 1024 threads: output index of every subsequence; zero-fills raster cells
// past the decoded symbols (Huffman.h:35: the matrix starts zero-initialised).
KHG_DEV void khg_dec_scan_body(const KhgDecArgs& a, KhgDecScanSmem& sh)
{
    if (!a.hdr->ok) return;
    KHG_FOR_THREADS(1024)
    {
        if (tid < 3) {
            uint32_t nb = (uint32_t)a.hdr->nbits[tid];
            sh.nsub[tid] = (nb + KHG_SUB_BITS - 1) / KHG_SUB_BITS;
        }
        if (tid == 0 && a.mismatch[0] != 0) khg_set_status(&a.res->status, KHG_NOT_CONVERGED);
    }
    KHG_BARRIER();
    for (int p = 0; p < 3; ++p) {
        const uint32_t ns = sh.nsub[p];
        const uint32_t per = (ns + 1023) / 1024;
        KhgSubInfo* sub = a.sub + a.sub0[p];
        KHG_FOR_THREADS(1024)
        {
            uint64_t s = 0;
            uint32_t c0 = (uint32_t)tid * per, c1 = c0 + per < ns ? c0 + per : ns;
            for (uint32_t c = c0; c < c1; ++c) s += sub[c].count;
            sh.psum[p][tid] = s;
        }
        KHG_BARRIER();
        KHG_FOR_THREADS(1024)
        {
            if (tid == 0) {
                uint64_t run = 0;
                for (int t = 0; t < 1024; ++t) {
                    uint64_t v = sh.psum[p][t];
                    sh.psum[p][t] = run;
                    run += v;
                }
                if (run > a.n[p]) khg_set_status(&a.res->status, -3);  // the reference would write out of bounds
                a.res->bits[p] = (int32_t)(run > a.n[p] ? a.n[p] : run);  // symbols decoded
            }
        }
        KHG_BARRIER();
        KHG_FOR_THREADS(1024)
        {
            uint32_t c0 = (uint32_t)tid * per, c1 = c0 + per < ns ? c0 + per : ns;
            uint64_t run = sh.psum[p][tid];
            for (uint32_t c = c0; c < c1; ++c) {
                sub[c].out = (uint32_t)(run > 0xFFFFFFFFull ? 0xFFFFFFFFull : run);
                run += sub[c].count;
            }
            uint32_t decoded = (uint32_t)a.res->bits[p];
            for (uint32_t i = decoded + (uint32_t)tid; i < a.n[p]; i += 1024) a.sym[p][i] = 0;
        }
        KHG_BARRIER();
    }
}

// One workgroup = 256 subsequences: decode each from its exact start and store the symbols.
KHG_DEV void khg_dec_write_body(const KhgDecArgs& a, uint32_t blk, KhgDecLut& sh)
{
    if (!a.hdr->ok || a.res->status != 0) return;
    const int p = khg_block_plane(a, blk);
    const KhgBits bits = khg_plane_bits(a, p);
    const uint32_t lb = blk - a.blk0[p];
    const uint32_t blk_begin = lb * KHG_DEC_BLOCK * KHG_SUB_BITS;
    if (blk_begin >= bits.nbits) return;
    const int32_t* left = a.tree_left + (size_t)p * 2 * KHG_NBINS;
    const int32_t* right = a.tree_right + (size_t)p * 2 * KHG_NBINS;
    const int32_t* tsym = a.tab_sym + (size_t)p * KHG_NBINS;
    const uint32_t* glut = a.lut + (size_t)p * KHG_LUT_SIZE;
    KHG_FOR_THREADS(256)
    {
        for (int i = tid; i < KHG_LUT_SIZE; i += 256) sh.lut[i] = glut[i];
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        KhgSubInfo si = a.sub[a.sub0[p] + lb * KHG_DEC_BLOCK + tid];
        uint32_t pos = si.start;
        int16_t* q = a.sym[p];
        for (uint32_t i = 0; i < si.count; ++i) {
            int s;
            uint32_t len = khg_decode_one(bits, sh.lut, left, right, tsym, pos, &s);
            if (len == 0) break;
            uint32_t o = si.out + i;
            if (o < a.n[p]) q[o] = (int16_t)s;
            pos += len;
        }
    }
}
