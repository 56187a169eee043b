// khg_decode.h -- .comp header parsing, tree rebuild, and chunked self-synchronising
// parallel Huffman decode.
//
// The format has no synchronisation markers (Writer.h:25-32 just concatenates codes), so
// each plane's stream is cut into 512-bit subsequences, one thread each, 256 per workgroup;
// a workgroup stages its 16 KiB of stream in LDS.  A thread that starts inside a codeword
// decodes garbage for a few symbols and then falls into step with the true codeword
// boundaries (Huffman codes self-synchronise), so:
//   sync : every thread decodes from its assumed start to the first codeword boundary at or
//          after its subsequence's end, recording the codeword starts it saw in a 512-bit
//          map.  Then, until nothing changes, each thread restarts from its predecessor's
//          arrival point; as soon as it lands on a start its earlier run had already seen,
//          the rest of that run is known to be right and it stops.
//   fix  : the first thread of a workgroup started blind; it is re-run from where the
//          previous workgroup arrived.  If every such re-run rejoins its earlier run, all
//          starts are exact by induction from the first workgroup (bit 0 is exact);
//          otherwise the host repeats full sync passes chained through the workgroup
//          arrivals until they stop changing (fixed-length codes need this).
//   scan : symbol counts -> output index of every workgroup
//   write: every thread decodes its subsequence from its exact start and stores the
//          symbols (Huffman.h:33-55: symbol i lands at raster index i).
#pragma once
#include "khg_common.h"
#include "khg_huffman.h"

enum {
    KHG_SUB_BITS = 512,       // bits per subsequence
    KHG_SUB_WORDS = KHG_SUB_BITS / 32,
    KHG_DEC_BLOCK = 256,      // subsequences per workgroup
    KHG_BLK_BITS = KHG_SUB_BITS * KHG_DEC_BLOCK,
    KHG_STAGE_WORDS = KHG_BLK_BITS / 32 + 4,  // a workgroup's span plus the longest overshoot
    KHG_STAGE_LDS = KHG_STAGE_WORDS + KHG_STAGE_WORDS / 32 + 1,  // one pad word per 32: threads read 16 words apart
    KHG_LUT_BITS = 12,
    KHG_LUT_SIZE = 1 << KHG_LUT_BITS,
    KHG_NOT_CONVERGED = 1     // internal status: the host must run chained sync passes
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
};

// Per-workgroup record kept between the decode kernels.
struct KhgBlkInfo {
    uint32_t arrive[2];   // where the workgroup's last thread arrived, by pass parity
    uint32_t total;       // symbols starting in the workgroup's span
    uint32_t out;         // raster index of the first of them (filled by the scan)
    uint32_t pad_[4];
};

struct KhgDecWorkspace {
    double* plane[3];   // reconstructed Y / Cb / Cr (f64)
    uint32_t* lut;      // [3][KHG_LUT_SIZE] first-level decode tables, then the mismatch word
    KhgBlkInfo* blk;    // per workgroup, planes concatenated
    KhgSubInfo* sub;    // per subsequence, planes concatenated
    uint32_t* masks;    // [subsequence][KHG_SUB_WORDS] codeword-start maps
    KhgDecHeader* hdr;
};

struct KhgDecWorkspaceSizes {
    size_t plane[3], lut, sub, blk, masks;
};

// Decoder limit on a plane's stream: 16 bits per symbol on average.  A Huffman code built from the
// stored counts averages below log2(K) + 1 <= 15 bits when the stream follows those counts, as every
// stream written by Writer.h / by this library does; longer streams are rejected (KHG_ERR_RANGE).
enum { KHG_DEC_MAX_AVG_BITS = 16 };

static inline uint32_t khg_max_subs(uint64_t nsym)
{
    uint64_t bits = nsym * KHG_DEC_MAX_AVG_BITS;
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
    KhgBlkInfo* blk;       // [nblocks]
    uint32_t* mismatch;    // [1]
    KhgSubInfo* sub;
    uint32_t* masks;       // [subsequence][KHG_SUB_WORDS]
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
    s->lut = (3 * (size_t)KHG_LUT_SIZE + 16) * sizeof(uint32_t);
    s->sub = subs * sizeof(KhgSubInfo);
    s->blk = blocks * sizeof(KhgBlkInfo);
    s->masks = subs * KHG_SUB_WORDS * sizeof(uint32_t);
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
                    if (k[p] > 1 && (uint64_t)h->nbits[p] > (uint64_t)a.n[p] * KHG_DEC_MAX_AVG_BITS) {
                        ok = 0;
                        khg_set_status(&a.res->status, -4);
                    }
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
    alignas(16) uint32_t w[6 * KHG_TREE_LDS_K + 8];
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
    if (k <= KHG_TREE_LDS_K) {  // separate copies: LDS vs global address space of the heap arrays
        KHG_FOR_THREADS(256)
        {
            if (tid == 0) {
                sh.root = khg_tree_build(tcnt, k, (uint64_t*)sh.w + 1, sh.w + 2 * KHG_TREE_LDS_K + 4,
                                         sh.w + 4 * KHG_TREE_LDS_K + 4, left, right);
                a.tree_root[bid] = sh.root;
            }
        }
    } else {
        uint32_t* g = a.big_scratch + (size_t)bid * 10 * KHG_NBINS;
        KHG_FOR_THREADS(256)
        {
            if (tid == 0) {
                sh.root = khg_tree_build(tcnt, k, (uint64_t*)g + 1, g + 2 * KHG_NBINS + 4, g + 4 * KHG_NBINS + 4, left,
                                         right);
                a.tree_root[bid] = sh.root;
            }
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

// --------------------------------------------------------------- bit readers

// Stream bits straight from the file in global memory (used where only a few codewords are read).
struct KhgBitsG {
    const uint32_t* words;  // file as aligned 32-bit words
    uint64_t base;          // bit offset of the plane's stream in the file
    uint32_t nbits;         // stream length in bits
    uint32_t nwords;        // readable words
    // 32 stream bits starting at stream position pos, first bit in bit 31 (zeros past the end)
    KHG_DEV uint32_t peek32(uint32_t pos) const
    {
        uint64_t g = base + pos;
        uint32_t w = (uint32_t)(g >> 5), s = (uint32_t)(g & 31);
        uint32_t w0 = w < nwords ? khg_bswap32(words[w]) : 0;
        uint32_t w1 = w + 1 < nwords ? khg_bswap32(words[w + 1]) : 0;
        return s ? (w0 << s) | (w1 >> (32 - s)) : w0;
    }
};

// The same view over a workgroup's span staged in LDS as byte-swapped words.
KHG_DEV uint32_t khg_stage_slot(uint32_t w) { return w + (w >> 5); }

struct KhgBitsL {
    const uint32_t* stage;  // KHG_STAGE_WORDS words at khg_stage_slot() positions
    uint32_t bias;          // stream position of bit 31 of stage[0] (may be "negative": wraps, used modulo 2^32)
    uint32_t nbits;
    KHG_DEV uint32_t peek32(uint32_t pos) const
    {
        uint32_t l = pos - bias;
        uint32_t w = l >> 5, s = l & 31;
        uint32_t w0 = w < KHG_STAGE_WORDS ? stage[khg_stage_slot(w)] : 0;
        uint32_t w1 = w + 1 < KHG_STAGE_WORDS ? stage[khg_stage_slot(w + 1)] : 0;
        return s ? (w0 << s) | (w1 >> (32 - s)) : w0;
    }
};

KHG_DEV int khg_block_plane(const KhgDecArgs& a, uint32_t blk)
{
    return blk >= a.blk0[2] ? 2 : blk >= a.blk0[1] ? 1 : 0;
}

KHG_DEV KhgBitsG khg_plane_bits(const KhgDecArgs& a, int p)
{
    KhgBitsG b;
    b.words = (const uint32_t*)a.comp;
    b.base = 8ull * a.hdr->stream_byte[p];
    b.nbits = (uint32_t)a.hdr->nbits[p];
    b.nwords = (uint32_t)(a.len >> 2);
    return b;
}

// Decode one codeword at pos.  Returns its length (0 = the stream ends inside it) and symbol.
template <class Bits>
KHG_DEV uint32_t khg_decode_one(const Bits& b, const uint32_t* lut, const int32_t* left, const int32_t* right,
                                const int32_t* tsym, uint32_t pos, int* sym)
{
    uint32_t window = b.peek32(pos);
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
            uint32_t bit = b.peek32(pos + len) >> 31;
            node = bit ? right[node] : left[node];
            ++len;
        }
        *sym = tsym[node];
    }
    if (pos + len > b.nbits) return 0;
    return len;
}

// Codeword-start bitmap of one subsequence: word w of thread t lives at mask[w * stride + t * tstride];
// position r (0..511) is bit 31 - (r & 31) of word r >> 5.
struct KhgMaskRef {
    uint32_t* m;
    uint32_t wstride;
    KHG_DEV uint32_t& word(uint32_t w) const { return m[w * wstride]; }
};

// Decode from `start` (inside the subsequence beginning at sub_begin) until the first codeword
// boundary at or after `end`.  Codeword starts are recorded in `mask`.  With has_old, `mask`
// holds the starts of an earlier run of the same subsequence: landing on one of them means the
// two runs coincide from there on, so the earlier arrival (*arrive on entry) stands and the
// earlier starts at and after that point are kept and counted.  Returns the arrival position
// (0xFFFFFFFF: the stream ends inside a codeword) and the number of codewords started.
template <class Bits>
KHG_DEV void khg_decode_span(const Bits& b, const uint32_t* lut, const int32_t* left, const int32_t* right,
                             const int32_t* tsym, uint32_t sub_begin, uint32_t start, uint32_t end, bool has_old,
                             const KhgMaskRef& mask, uint32_t* arrive, uint32_t* count)
{
    uint32_t pos = start, n = 0;
    uint32_t widx = (start - sub_begin) >> 5;
    for (uint32_t w = 0; w < widx; ++w) mask.word(w) = 0;
    uint32_t oldw = has_old ? mask.word(widx) : 0, cur = 0;
    bool exhausted = false;
    while (pos < end) {
        uint32_t rel = pos - sub_begin, w = rel >> 5;
        while (widx < w) {
            mask.word(widx) = cur;
            ++widx;
            cur = 0;
            oldw = has_old ? mask.word(widx) : 0;
        }
        uint32_t bit = 0x80000000u >> (rel & 31);
        if (oldw & bit) {
            // rejoined the earlier run: keep its starts from here on
            uint32_t ge = bit | (bit - 1);
            uint32_t rest = (uint32_t)__builtin_popcount(oldw & ge);
            mask.word(widx) = cur | (oldw & ge);
            for (uint32_t w2 = widx + 1; w2 < KHG_SUB_WORDS; ++w2) rest += (uint32_t)__builtin_popcount(mask.word(w2));
            *count = n + rest;
            return;  // *arrive unchanged
        }
        int s;
        uint32_t len = khg_decode_one(b, lut, left, right, tsym, pos, &s);
        if (len == 0) {
            exhausted = true;
            break;
        }
        cur |= bit;
        pos += len;
        ++n;
    }
    mask.word(widx) = cur;
    for (uint32_t w2 = widx + 1; w2 < KHG_SUB_WORDS; ++w2) mask.word(w2) = 0;
    *arrive = exhausted ? 0xFFFFFFFFu : pos;
    *count = n;
}

struct KhgDecSmem {
    uint32_t lut[KHG_LUT_SIZE];
    uint32_t stage[KHG_STAGE_LDS];
    uint32_t mask[KHG_SUB_WORDS * KHG_DEC_BLOCK];  // word-major: thread t's word w at [w * 256 + t]
    uint32_t arrive[KHG_DEC_BLOCK];
    uint32_t count[KHG_DEC_BLOCK];
    uint32_t used[KHG_DEC_BLOCK];
    uint32_t grp[16];
    uint32_t changed[2];
    uint32_t blk_start;
};

// Stage the LUT and the workgroup's span of the stream (byte-swapped words) in LDS.
KHG_DEV KhgBitsL khg_dec_stage(const KhgDecArgs& a, int p, uint32_t blk_begin, KhgDecSmem& sh)
{
    const KhgBitsG g = khg_plane_bits(a, p);
    const uint32_t* glut = a.lut + (size_t)p * KHG_LUT_SIZE;
    const uint64_t gbit = g.base + blk_begin;
    const uint32_t w0 = (uint32_t)(gbit >> 5);
    KHG_FOR_THREADS(256)
    {
        for (int i = tid; i < KHG_LUT_SIZE; i += 256) sh.lut[i] = glut[i];
        for (uint32_t i = (uint32_t)tid; i < KHG_STAGE_WORDS; i += 256) {
            uint32_t w = w0 + i;
            sh.stage[khg_stage_slot(i)] = w < g.nwords ? khg_bswap32(g.words[w]) : 0;
        }
    }
    KhgBitsL l;
    l.stage = sh.stage;
    l.bias = blk_begin - (uint32_t)(gbit & 31);
    l.nbits = g.nbits;
    return l;
}

// One workgroup = 256 consecutive subsequences of one plane.
// pass 0: the workgroup's first thread starts at the nominal position (exact only for the plane's
// first workgroup).  pass > 0 (chained passes, only when the fix kernel reported a mismatch): it
// starts where the previous workgroup arrived in the previous pass, and a changed arrival is counted.
KHG_DEV void khg_dec_sync_body(const KhgDecArgs& a, uint32_t blk, KhgDecSmem& sh)
{
    if (!a.hdr->ok) return;
    const int p = khg_block_plane(a, blk);
    const uint32_t nbits = (uint32_t)a.hdr->nbits[p];
    const uint32_t lb = blk - a.blk0[p];  // workgroup index inside the plane
    const uint32_t blk_begin = lb * KHG_BLK_BITS;
    if (blk_begin >= nbits && lb != 0) return;  // beyond the stream (workgroup 0 still records zeros)
    const int32_t* left = a.tree_left + (size_t)p * 2 * KHG_NBINS;
    const int32_t* right = a.tree_right + (size_t)p * 2 * KHG_NBINS;
    const int32_t* tsym = a.tab_sym + (size_t)p * KHG_NBINS;
    KhgBlkInfo* bi = a.blk + blk;
    const KhgBitsL bits = khg_dec_stage(a, p, blk_begin, sh);

    KHG_FOR_THREADS(256)
    {
        if (tid == 0) {
            uint32_t s = blk_begin;
            if (lb == 0) s = 0;
            else if (a.pass > 0) s = a.blk[blk - 1].arrive[(a.pass + 1) & 1];
            sh.blk_start = s;
            sh.changed[0] = 0;
            sh.changed[1] = 0;
        }
        sh.used[tid] = 0xFFFFFFFEu;  // "no start decoded yet"
        sh.arrive[tid] = blk_begin + ((uint32_t)tid + 1) * KHG_SUB_BITS;  // nominal arrival for the first round
        sh.count[tid] = 0;
    }
    KHG_BARRIER();
    // Every round, a thread whose predecessor's arrival differs from the start it last decoded
    // from runs again.  An exact start propagates at least one thread per round.
    for (int round = 0; round <= KHG_DEC_BLOCK; ++round) {
        uint32_t mystart[KHG_NSLOT(256)];
        KHG_FOR_THREADS(256)
        {
            mystart[KHG_SLOT] = tid == 0 ? sh.blk_start : sh.arrive[tid - 1];
        }
        KHG_BARRIER();
        KHG_FOR_THREADS(256)
        {
            const uint32_t s = mystart[KHG_SLOT];
            const uint32_t sub_begin = blk_begin + (uint32_t)tid * KHG_SUB_BITS;
            const uint32_t end = sub_begin + KHG_SUB_BITS;
            if (s != sh.used[tid]) {
                const bool has_old = sh.used[tid] != 0xFFFFFFFEu && sh.used[tid] < nbits && sh.used[tid] < end;
                uint32_t cnt = 0, arr = sh.arrive[tid];
                KhgMaskRef mr = {&sh.mask[tid], KHG_DEC_BLOCK};
                if (s >= nbits || s >= end) {
                    arr = s;  // nothing starts in this subsequence (s >= end cannot happen: 512 > longest code)
                    for (uint32_t w = 0; w < KHG_SUB_WORDS; ++w) mr.word(w) = 0;
                } else {
                    khg_decode_span(bits, sh.lut, left, right, tsym, sub_begin, s, end < nbits ? end : nbits, has_old,
                                    mr, &arr, &cnt);
                }
                sh.used[tid] = s;
                sh.arrive[tid] = arr;
                sh.count[tid] = cnt;
                sh.changed[round & 1] = 1;
            }
        }
        KHG_BARRIER();
        const bool any = sh.changed[round & 1] != 0;
#ifdef KHG_SIM_STATS
        if (any) bi->pad_[0] = (uint32_t)round + 1;
#endif
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
        a.sub[a.sub0[p] + lb * KHG_DEC_BLOCK + tid] = si;
        if (tid < 16) {
            uint32_t t = 0;
            for (int i = 0; i < 16; ++i) t += sh.count[tid * 16 + i];
            sh.grp[tid] = t;
        }
        uint32_t* gm = a.masks + ((size_t)a.sub0[p] + (size_t)lb * KHG_DEC_BLOCK + tid) * KHG_SUB_WORDS;
        for (int w = 0; w < KHG_SUB_WORDS; ++w) gm[w] = sh.mask[w * KHG_DEC_BLOCK + tid];
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        if (tid == 0) {
            uint32_t t = 0;
            for (int i = 0; i < 16; ++i) t += sh.grp[i];
            bi->total = t;
            uint32_t arr = sh.arrive[KHG_DEC_BLOCK - 1];
            if (a.pass > 0 && bi->arrive[(a.pass + 1) & 1] != arr) khg_atomic_add(a.mismatch, 1u);
            bi->arrive[a.pass & 1] = arr;
        }
    }
}

// After sync pass 0: one thread per workgroup re-runs the workgroup's leading subsequences from
// where the previous workgroup arrived, subsequence after subsequence, until the run rejoins
// the speculative one (normally within a few codewords of the first subsequence).  If it walks
// through the whole workgroup and arrives somewhere else, the speculation failed for the next
// workgroup too and a mismatch is recorded.
KHG_DEV void khg_dec_fix_body(const KhgDecArgs& a, uint32_t bid)
{
    if (!a.hdr->ok) return;
    KHG_FOR_THREADS(256)
    {
        const uint32_t blk = bid * 256 + (uint32_t)tid;
        if (blk < a.nblocks) {
            const int p = khg_block_plane(a, blk);
            const uint32_t lb = blk - a.blk0[p];
            const KhgBitsG bits = khg_plane_bits(a, p);
            const uint32_t blk_begin = lb * KHG_BLK_BITS;
            if (lb != 0 && blk_begin < bits.nbits) {
                KhgBlkInfo* bi = a.blk + blk;
                KhgSubInfo* sub = a.sub + a.sub0[p] + (size_t)lb * KHG_DEC_BLOCK;
                uint32_t* masks = a.masks + ((size_t)a.sub0[p] + (size_t)lb * KHG_DEC_BLOCK) * KHG_SUB_WORDS;
                uint32_t s = a.blk[blk - 1].arrive[0];
                uint32_t total = bi->total;
                bool joined = false;
                for (uint32_t t = 0; t < KHG_DEC_BLOCK; ++t) {
                    if (s == sub[t].start) {
                        joined = true;
                        break;
                    }
                    const uint32_t sub_begin = blk_begin + t * KHG_SUB_BITS, end = sub_begin + KHG_SUB_BITS;
                    const uint32_t old_start = sub[t].start, old_cnt = sub[t].count;
                    const uint32_t old_arr = t + 1 < KHG_DEC_BLOCK ? sub[t + 1].start : bi->arrive[0];
                    uint32_t arr = old_arr, cnt = 0;
                    KhgMaskRef mr = {masks + (size_t)t * KHG_SUB_WORDS, 1};
                    if (s >= bits.nbits || s >= end) {
                        arr = s;
                    } else {
                        const bool has_old = old_start < bits.nbits && old_start < end;
                        khg_decode_span(bits, a.lut + (size_t)p * KHG_LUT_SIZE, a.tree_left + (size_t)p * 2 * KHG_NBINS,
                                        a.tree_right + (size_t)p * 2 * KHG_NBINS, a.tab_sym + (size_t)p * KHG_NBINS,
                                        sub_begin, s, end < bits.nbits ? end : bits.nbits, has_old, mr, &arr, &cnt);
                    }
                    sub[t].start = s;
                    sub[t].count = cnt;
                    total = total - old_cnt + cnt;
                    if (arr == old_arr) {
                        joined = true;
                        break;
                    }
                    s = arr;
                }
                bi->total = total;
                if (!joined) khg_atomic_add(a.mismatch, 1u);
            }
        }
    }
}

struct KhgDecScanSmem {
    uint64_t psum[1024];
};

// Single workgroup, 1024 threads: raster index of every workgroup's first symbol; zero-fills raster
// cells past the decoded symbols (Huffman.h:35: the matrix starts zero-initialised).
KHG_DEV void khg_dec_scan_body(const KhgDecArgs& a, KhgDecScanSmem& sh)
{
    if (!a.hdr->ok) return;
    KHG_FOR_THREADS(1024)
    {
        if (tid == 0 && a.mismatch[0] != 0) khg_set_status(&a.res->status, KHG_NOT_CONVERGED);
    }
    for (int p = 0; p < 3; ++p) {
        const uint32_t nbits = (uint32_t)a.hdr->nbits[p];
        const uint32_t nb = (nbits + KHG_BLK_BITS - 1) / KHG_BLK_BITS;  // active workgroups of this plane
        const uint32_t per = (nb + 1023) / 1024;
        KhgBlkInfo* blk = a.blk + a.blk0[p];
        KHG_FOR_THREADS(1024)
        {
            uint64_t s = 0;
            uint32_t c0 = (uint32_t)tid * per, c1 = c0 + per < nb ? c0 + per : nb;
            for (uint32_t c = c0; c < c1; ++c) s += blk[c].total;
            sh.psum[tid] = s;
        }
        KHG_BARRIER();
        KHG_FOR_THREADS(1024)
        {
            if (tid == 0) {
                uint64_t run = 0;
                for (int t = 0; t < 1024; ++t) {
                    uint64_t v = sh.psum[t];
                    sh.psum[t] = run;
                    run += v;
                }
                if (run > a.n[p]) khg_set_status(&a.res->status, -3);  // the reference would write out of bounds
                a.res->bits[p] = (int32_t)(run > a.n[p] ? a.n[p] : run);  // symbols decoded
            }
        }
        KHG_BARRIER();
        KHG_FOR_THREADS(1024)
        {
            uint32_t c0 = (uint32_t)tid * per, c1 = c0 + per < nb ? c0 + per : nb;
            uint64_t run = sh.psum[tid];
            for (uint32_t c = c0; c < c1; ++c) {
                blk[c].out = (uint32_t)(run > 0xFFFFFFFFull ? 0xFFFFFFFFull : run);
                run += blk[c].total;
            }
            uint32_t decoded = (uint32_t)a.res->bits[p];
            for (uint32_t i = decoded + (uint32_t)tid; i < a.n[p]; i += 1024) a.sym[p][i] = 0;
        }
        KHG_BARRIER();
    }
}

// One workgroup = 256 subsequences: decode each from its exact start and store the symbols.
KHG_DEV void khg_dec_write_body(const KhgDecArgs& a, uint32_t blk, KhgDecSmem& sh)
{
    if (!a.hdr->ok || a.res->status != 0) return;
    const int p = khg_block_plane(a, blk);
    const uint32_t nbits = (uint32_t)a.hdr->nbits[p];
    const uint32_t lb = blk - a.blk0[p];
    const uint32_t blk_begin = lb * KHG_BLK_BITS;
    if (blk_begin >= nbits) return;
    const int32_t* left = a.tree_left + (size_t)p * 2 * KHG_NBINS;
    const int32_t* right = a.tree_right + (size_t)p * 2 * KHG_NBINS;
    const int32_t* tsym = a.tab_sym + (size_t)p * KHG_NBINS;
    const KhgSubInfo* sub = a.sub + a.sub0[p] + (size_t)lb * KHG_DEC_BLOCK;
    const KhgBitsL bits = khg_dec_stage(a, p, blk_begin, sh);
    KHG_FOR_THREADS(256)
    {
        sh.used[tid] = sub[tid].start;
        sh.count[tid] = sub[tid].count;
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        if (tid < 16) {  // exclusive scan of the counts inside each group of 16 threads
            uint32_t run = 0;
            for (int i = 0; i < 16; ++i) {
                sh.arrive[tid * 16 + i] = run;
                run += sh.count[tid * 16 + i];
            }
            sh.grp[tid] = run;
        }
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        uint32_t o = a.blk[blk].out + sh.arrive[tid];
        for (int g = 0; g < (tid >> 4); ++g) o += sh.grp[g];
        uint32_t pos = sh.used[tid];
        const uint32_t cnt = sh.count[tid];
        int16_t* q = a.sym[p];
        for (uint32_t i = 0; i < cnt; ++i) {
            int s;
            uint32_t len = khg_decode_one(bits, sh.lut, left, right, tsym, pos, &s);
            if (len == 0) break;
            if (o + i < a.n[p]) q[o + i] = (int16_t)s;
            pos += len;
        }
    }
}
