// khg_decode.h -- .comp header parsing, tree rebuild, and chunked parallel Huffman decode.
//
// The format has no synchronisation markers (Writer.h:25-32 just concatenates codes), so the
// first codeword of a chunk of the stream starts at an unknown offset.  Each plane's stream is
// cut into subsequences of KHG_SUB_BITS bits (one thread each, 256 per workgroup, staged in LDS).  The first
// codeword that starts inside a subsequence starts at one of its first M bits, M = the longest
// code of the plane.  Instead of guessing, a thread works out, for EVERY start offset o < M,
// where a decoder that starts there arrives in the next subsequence and how many codewords it
// decodes on the way:
//   map   : per subsequence, offset -> (arrival offset, count).  Offsets whose first codeword
//           lands on another offset < M reuse that offset's entry; the others run on.  The first
//           such run records the codeword starts it sees in a bit map; every later run
//           stops as soon as it lands on one of them (Huffman codes self-synchronise, so this
//           is normally within a few codewords; a fixed-length stretch costs one full run per
//           residue instead).  The workgroup then composes its 256 maps into one, one lane per
//           start offset, and notes on the way where each subsequence is entered and how many
//           codewords precede it (the walk table).
//   chain : one thread per plane walks the workgroup maps from bit 0: exact start offset and
//           output index of every workgroup.
//   write : every thread reads its exact start and output index from the walk table column of
//           its workgroup's start offset, decodes its subsequence and stores the symbols
//           (Huffman.h:33-55: symbol i lands at raster index i).
// Nothing is speculative: the result is exact for any code.
#pragma once
#include "khg_common.h"
#include "khg_huffman.h"

// Bits per subsequence, a build-time choice (256 or 512).  Measured on 8K frames against 512: 256 halves
// the serial chains per lane in the stretches that need one run per residue class (map 290 -> 157 us, write 73 ->
// 55 us for one frame alone) at the price of twice the first-codeword work, workgroup records and chain steps
// (35 -> 66 us, one lane); in the pipelined batch it decodes 4 % more frames per second.
#ifndef KHG_SUB_BITS_CFG
#define KHG_SUB_BITS_CFG 256
#endif

enum {
    KHG_SUB_BITS = KHG_SUB_BITS_CFG,  // bits per subsequence (build-time choice)
    KHG_SUB_WORDS = KHG_SUB_BITS / 32,
    KHG_DEC_BLOCK = 256,      // subsequences per workgroup
    KHG_BLK_BITS = KHG_SUB_BITS * KHG_DEC_BLOCK,
    KHG_STAGE_WORDS = KHG_BLK_BITS / 32 + 8,  // a workgroup's span plus the longest overshoot and the cursor's read-ahead
    KHG_STAGE_LDS = KHG_STAGE_WORDS + 2 * (KHG_STAGE_WORDS / 32) + 2,  // two pad words per 32 (khg_stage_slot)
    KHG_LUT_BITS = 12,
    KHG_LUT_SIZE = 1 << KHG_LUT_BITS,
    KHG_LUT2_BITS = 8,        // second-level tables: the next 8 bits after a first-level miss
    KHG_LUT2_SIZE = 1 << KHG_LUT2_BITS,
    KHG_MAXOFF = KHG_MAX_CODE_BITS,  // start offsets 0..55
    KHG_OFF_TEMP = 62,        // map entry not final yet: the count field holds the landing offset
    KHG_OFF_NONE = 63         // no codeword starts from here on (the stream has ended)
};

// (128-bit subsequences decode streams with codes of up to 32 bits and fail the test with 44-bit codes: a start
// offset plus the longest code must stay well inside a subsequence.)
static_assert(KHG_SUB_BITS == 256 || KHG_SUB_BITS == 512, "subsequences of 256 or 512 bits");

// Decoder limit on a plane's stream: 16 bits per symbol on average.  A Huffman code built from the
// stored counts averages below log2(K) + 1 <= 15 bits when the stream follows those counts, as every
// stream written by Writer.h / by this library does; longer streams are rejected (KHG_ERR_RANGE).
enum { KHG_DEC_MAX_AVG_BITS = 16 };

// Parsed header plus derived layout, written by khg_dec_header_body / khg_dec_tree_body.
struct KhgDecHeader {
    int32_t nbits[3];         // stream bits without the pad (Decompress.h:37-39)
    int32_t k[3];             // table sizes
    uint32_t stream_byte[3];  // byte offset of each plane's stream in the file
    uint32_t table_byte[3];   // byte offset of each plane's table
    int32_t maxlen[3];        // longest code of each plane
    int32_t one_bit[3];       // value of the plane's 1-bit code (0 or 1), -1 if it has none
    int32_t one_sym[3];       // its symbol
    int32_t lvl;
    int32_t rows, cols;
    int32_t ok;
};

// Subsequence map entry: (arrival offset << 10) | codeword count.
KHG_DEV uint16_t khg_map_entry(uint32_t arr, uint32_t cnt) { return (uint16_t)((arr << 10) | cnt); }
KHG_DEV uint32_t khg_map_arr(uint16_t e) { return (uint32_t)e >> 10; }
KHG_DEV uint32_t khg_map_cnt(uint16_t e) { return (uint32_t)e & 1023u; }

// Per-workgroup record kept between the decode kernels.
struct KhgBlkInfo {
    uint32_t map[KHG_MAXOFF];  // start offset -> (arrival offset << 24) | codewords in the workgroup's span
    uint32_t start_off;        // exact start offset (filled by the chain)
    uint32_t out;              // raster index of the workgroup's first symbol (filled by the chain)
    uint32_t pad_[6];
};

struct KhgDecWorkspace {
    uint32_t* lut;      // [3][KHG_LUT_SIZE] first-level decode tables
    uint32_t* lut2;     // [3][KHG_LUT_SIZE][KHG_LUT2_SIZE] second-level tables (only rows of missing prefixes are filled)
    KhgBlkInfo* blk;    // per workgroup, planes concatenated
    uint32_t* walk;     // [workgroup][KHG_DEC_BLOCK][maxlen]: see khg_dec_map_body
    KhgDecHeader* hdr;
};

struct KhgDecWorkspaceSizes {
    size_t lut, lut2, blk, walk;
};

static inline uint32_t khg_max_blocks(uint64_t nsym)
{
    uint64_t bits = nsym * KHG_DEC_MAX_AVG_BITS;
    if (bits > 0x7FFFFFFFull) bits = 0x7FFFFFFFull;
    uint32_t b = (uint32_t)((bits + KHG_BLK_BITS - 1) / KHG_BLK_BITS);
    return b ? b : 1;
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
    uint32_t* lut2;        // [3][KHG_LUT_SIZE][KHG_LUT2_SIZE]
    KhgBlkInfo* blk;       // [nblocks]
    uint32_t* walk;        // [nblocks][KHG_DEC_BLOCK * KHG_MAXOFF] words; a workgroup uses [t * maxlen + o]
    uint32_t blk0[3];      // first workgroup index of each plane (capacity layout)
    uint32_t nblocks;
    int16_t* sym[3];
    uint32_t n[3];         // symbols per plane
    // Row-tiled images: the map and write launches of rank part_rank of part_nranks take the workgroups
    // [nb * rank / nranks, nb * (rank + 1) / nranks) of each plane's nb active ones (0 ranks = 1: all of them).
    uint32_t part_rank, part_nranks;
};

// Active workgroups of plane p's stream.
KHG_HOST_DEV uint32_t khg_dec_active_blocks(uint32_t nbits) { return (nbits + KHG_BLK_BITS - 1) / KHG_BLK_BITS; }
// First workgroup of rank r's share of nb workgroups split over n ranks.
KHG_HOST_DEV uint32_t khg_part_begin(uint32_t nb, uint32_t r, uint32_t n) { return (uint32_t)((uint64_t)nb * r / n); }

static inline void khg_dec_workspace_sizes(int rows, int cols, KhgDecWorkspaceSizes* s)
{
    size_t ny = (size_t)rows * cols, nc = ny / 4;
    size_t blocks = (size_t)khg_max_blocks(ny) + 2 * (size_t)khg_max_blocks(nc);
    s->lut = 3 * (size_t)KHG_LUT_SIZE * sizeof(uint32_t);
    s->lut2 = 3 * (size_t)KHG_LUT_SIZE * KHG_LUT2_SIZE * sizeof(uint32_t);
    s->blk = blocks * sizeof(KhgBlkInfo);
    s->walk = blocks * KHG_MAXOFF * KHG_DEC_BLOCK * sizeof(uint32_t);
}

KHG_DEV int32_t khg_load_le32(const uint8_t* p)
{
    return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
}

// One thread.  Reader.h:5-23 plus the consistency checks the reference omits.  Starts the job's result block afresh
// (it is the first kernel of every decode).
KHG_DEV void khg_dec_header_body(const KhgDecArgs& a)
{
    KHG_FOR_THREADS(256)
    {
        if (tid == 0) {
            KhgResult zero;
            __builtin_memset(&zero, 0, sizeof zero);
            *a.res = zero;
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
                    h->maxlen[p] = 0;
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
        }
    }
}

enum { KHG_FILL_LIST = 160 };  // LUT spans of 64+ entries are filled by the whole workgroup; a tree has at most 126

struct KhgDecTreeSmem {
    alignas(16) uint32_t w[6 * KHG_TREE_LDS_K + 8];
    int32_t root;
    int maxlen;
    uint32_t nfill;
    uint32_t fill_at[KHG_FILL_LIST], fill_n[KHG_FILL_LIST], fill_e[KHG_FILL_LIST];
};

// Depth of a node and the bits of the path from the root to it, first edge most significant (Huffman.h:57-68).
KHG_DEV int khg_node_path(const uint32_t* parent, int node, int root, uint64_t* code_out)
{
    int len = 0;
    uint64_t code = 0;
    while (node != root) {
        uint32_t p = parent[node];
        if (len < 64) code |= (uint64_t)(p >> 31) << len;
        node = (int)(p & 0x7FFFFFFFu);
        ++len;
    }
    *code_out = code;
    return len;
}

// Decode-table entries of tree node `node` (leaf when node < k): every first-level slot whose
// KHG_LUT_BITS-bit prefix decides on this node, and likewise one level down.
//   leaf, depth d <= 12         : lut[code << (12-d) ...] x 2^(12-d)
//   internal node, depth 12     : lut[code] = node
//   leaf, 12 < d <= 20          : lut2[prefix][rest << (20-d) ...] x 2^(20-d)
//   internal node, depth 20     : lut2[prefix][rest] = node
// Spans of 64 entries or more are queued for the whole workgroup.
KHG_DEV void khg_lut_scatter(KhgDecTreeSmem& sh, uint32_t* lut, uint32_t* lut2, int node, int k, int d, uint64_t code,
                             int sym)
{
    const bool leaf = node < k;
    const uint32_t e = leaf ? 0x80000000u | ((uint32_t)d << 16) | (uint32_t)(uint16_t)(int16_t)sym : (uint32_t)node;
    if (d <= KHG_LUT_BITS) {
        if (!leaf && d != KHG_LUT_BITS) return;
        const uint32_t n = 1u << (KHG_LUT_BITS - d), at = (uint32_t)code << (KHG_LUT_BITS - d);
        if (n >= 64) {
            uint32_t i = khg_atomic_add(&sh.nfill, 1u);
            sh.fill_at[i] = at; sh.fill_n[i] = n; sh.fill_e[i] = e;
        } else {
            for (uint32_t j = 0; j < n; ++j) lut[at + j] = e;
        }
    } else if (d <= KHG_LUT_BITS + KHG_LUT2_BITS) {
        if (!leaf && d != KHG_LUT_BITS + KHG_LUT2_BITS) return;
        const int r = d - KHG_LUT_BITS;  // bits below the first-level prefix
        const uint32_t v = (uint32_t)(code >> r), rest = (uint32_t)code & ((1u << r) - 1);
        const uint32_t n = 1u << (KHG_LUT2_BITS - r);
        uint32_t* row = lut2 + (size_t)v * KHG_LUT2_SIZE + (rest << (KHG_LUT2_BITS - r));
        for (uint32_t j = 0; j < n; ++j) row[j] = e;
    }
}

// bid = colour component.  Rebuilds the tree from the stored table with the encoder's own
// construction (Decompress.h:45-47), finds the longest code and fills the decode tables.
// LUT entry: bit 31 set = leaf reached: bits 16..23 code length, bits 0..15 symbol (int16);
// otherwise the entry is the node reached after KHG_LUT_BITS bits (first level) or after
// KHG_LUT_BITS + KHG_LUT2_BITS bits (second level, one 256-entry table per 12-bit prefix that
// ends on an internal node; rows of other prefixes are never read and stay unwritten).
KHG_DEV void khg_dec_tree_body(const KhgDecArgs& a, int bid, KhgDecTreeSmem& sh)
{
    KhgDecHeader* h = a.hdr;
    if (!h->ok) return;
    const int k = h->k[bid];
    int32_t* tcnt = a.tab_cnt + (size_t)bid * KHG_NBINS;
    int32_t* tsym = a.tab_sym + (size_t)bid * KHG_NBINS;
    int32_t* left = a.tree_left + (size_t)bid * 2 * KHG_NBINS;
    int32_t* right = a.tree_right + (size_t)bid * 2 * KHG_NBINS;
    uint32_t* lut = a.lut + (size_t)bid * KHG_LUT_SIZE;
    uint32_t* lut2 = a.lut2 + (size_t)bid * KHG_LUT_SIZE * KHG_LUT2_SIZE;
    const uint8_t* tb = a.comp + h->table_byte[bid];
    const bool in_lds = k <= KHG_TREE_LDS_K;  // heap arrays in LDS, else in global scratch
    uint32_t* g = a.big_scratch + (size_t)bid * 10 * KHG_NBINS;
    uint32_t* nfreq_l = sh.w + 2 * KHG_TREE_LDS_K + 4;
    uint32_t* parent_l = sh.w + 4 * KHG_TREE_LDS_K + 4;
    uint32_t* nfreq_g = g + 2 * KHG_NBINS + 4;
    uint32_t* parent_g = g + 4 * KHG_NBINS + 4;

    KHG_FOR_THREADS(256)
    {
        for (int i = tid; i < k; i += 256) {  // Reader.h:39-54
            int32_t cnt = khg_load_le32(tb + 8 * (size_t)i), s = khg_load_le32(tb + 8 * (size_t)i + 4);
            tcnt[i] = cnt;
            tsym[i] = s;
            if (s < -32768 || s > 32767) khg_set_status(&a.res->status, -4);
            if (in_lds) nfreq_l[i] = (uint32_t)cnt;
            else nfreq_g[i] = (uint32_t)cnt;
            left[i] = -1;
            right[i] = -1;
        }
        if (tid == 0) {
            sh.maxlen = 0;
            sh.nfill = 0;
        }
    }
    KHG_BARRIER();
    // separate copies of the sequential part and of the node walk: LDS vs global address space
    if (in_lds) {
        KHG_FOR_THREADS(256)
        {
            if (tid == 0) {
                sh.root = khg_tree_build(k, (uint64_t*)sh.w + 1, nfreq_l, parent_l, left, right);
                a.tree_root[bid] = sh.root;
            }
        }
        KHG_BARRIER();
        KHG_FOR_THREADS(256)
        {
            int m = 0;
            for (int node = tid; node < 2 * k - 1; node += 256) {
                uint64_t code;
                int d = khg_node_path(parent_l, node, sh.root, &code);
                if (node < k && d > m) m = d;
                khg_lut_scatter(sh, lut, lut2, node, k, d, code, node < k ? tsym[node] : 0);
            }
            if (m) khg_atomic_max_i(&sh.maxlen, m);
        }
    } else {
        KHG_FOR_THREADS(256)
        {
            if (tid == 0) {
                sh.root = khg_tree_build(k, (uint64_t*)g + 1, nfreq_g, parent_g, left, right);
                a.tree_root[bid] = sh.root;
            }
        }
        KHG_BARRIER();
        KHG_FOR_THREADS(256)
        {
            int m = 0;
            for (int node = tid; node < 2 * k - 1; node += 256) {
                uint64_t code;
                int d = khg_node_path(parent_g, node, sh.root, &code);
                if (node < k && d > m) m = d;
                khg_lut_scatter(sh, lut, lut2, node, k, d, code, node < k ? tsym[node] : 0);
            }
            if (m) khg_atomic_max_i(&sh.maxlen, m);
        }
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        const int root = sh.root;
        if (tid == 0) {
            h->maxlen[bid] = sh.maxlen;
            int ob = -1, os = 0;
            if (k > 1) {
                if (left[left[root]] < 0) { ob = 0; os = tsym[left[root]]; }
                else if (left[right[root]] < 0) { ob = 1; os = tsym[right[root]]; }
            }
            h->one_bit[bid] = ob;
            h->one_sym[bid] = os;
            if (sh.maxlen > KHG_MAX_CODE_BITS) khg_set_status(&a.res->status, -4);
        }
        for (uint32_t f = 0; f < sh.nfill; ++f) {
            const uint32_t at = sh.fill_at[f], n = sh.fill_n[f], e = sh.fill_e[f];
            for (uint32_t j = (uint32_t)tid; j < n; j += 256) lut[at + j] = e;
        }
    }
}

// Which workgroup of the decoder's capacity layout (KhgDecArgs::blk0) block b of a map / write launch works on.
// Only the workgroups a stream really spans are launched (the host bounds their number by the file length), and
// they are dispatched Cr, Cb, Y: workgroups run in launch order and the slow ones sit at the start of each plane's
// stream (the near-fixed-length codes of the LL band need one full run per residue), so this starts every slow
// workgroup in the first wave of dispatches instead of leaving the chroma ones for last.  ~0u = nothing to do.
KHG_DEV uint32_t khg_dec_dispatch_order(const KhgDecArgs& a, uint32_t b)
{
    if (!a.hdr->ok) return ~0u;
    const uint32_t n = a.part_nranks ? a.part_nranks : 1u, r = a.part_nranks ? a.part_rank : 0u;
    for (int p = 2; p >= 0; --p) {
        const uint32_t nb = khg_dec_active_blocks((uint32_t)a.hdr->nbits[p]);
        const uint32_t lo = khg_part_begin(nb, r, n), hi = khg_part_begin(nb, r + 1, n);
        if (b < hi - lo) return a.blk0[p] + lo + b;
        b -= hi - lo;
    }
    return ~0u;
}

// --------------------------------------------------------------- bit reader

// Where word w of a staged span lives: two pad slots follow every 32 words, so that threads whose subsequences lie a
// power of two words apart start in different banks.  The pad slots hold copies of the two words after them: the
// three slots from khg_stage_slot(w) on are words w, w + 1, w + 2 for every w, which lets a reader take 96
// consecutive stream bits with one address.
KHG_DEV uint32_t khg_stage_slot(uint32_t w) { return w + 2 * (w >> 5); }
KHG_DEV void khg_stage_put(uint32_t* stage, uint32_t i, uint32_t v)
{
    stage[khg_stage_slot(i)] = v;
    if (i >= 32 && (i & 31) < 2) stage[khg_stage_slot(i) - 2] = v;  // the copy in the pad slots before its chunk
}

// A workgroup's span of the stream staged in LDS as byte-swapped words at khg_stage_slot() positions.
struct KhgBitsL {
    const uint32_t* stage;
    uint32_t bias;   // stream position of bit 31 of stage word 0 (modulo 2^32)
    uint32_t nbits;  // stream length in bits
    // 32 stream bits starting at stream position pos, first bit in bit 31
    KHG_DEV uint32_t peek32(uint32_t pos) const
    {
        uint32_t l = pos - bias;
        uint32_t w = l >> 5, s = l & 31;
        // callers stay inside the staged span: pos < span end + 56, and 8 spare words are staged
        uint64_t two = ((uint64_t)stage[khg_stage_slot(w)] << 32) | stage[khg_stage_slot(w + 1)];
        return (uint32_t)((two << s) >> 32);
    }
};

KHG_DEV int khg_block_plane(const KhgDecArgs& a, uint32_t blk)
{
    return blk >= a.blk0[2] ? 2 : blk >= a.blk0[1] ? 1 : 0;
}

// Decode one codeword at pos.  Returns its length (0 = the stream ends inside it) and symbol.
KHG_DEV uint32_t khg_decode_one(const KhgBitsL& b, const uint32_t* lut, const uint32_t* lut2, const int32_t* left,
                                const int32_t* right, const int32_t* tsym, uint32_t pos, int* sym)
{
    const uint32_t window = b.peek32(pos);
    const uint32_t v = window >> (32 - KHG_LUT_BITS);
    uint32_t e = lut[v];
    uint32_t len;
    if (e & 0x80000000u) {
        len = (e >> 16) & 0xFF;
        *sym = (int)(int16_t)(e & 0xFFFF);
    } else {
        e = lut2[(size_t)v * KHG_LUT2_SIZE + ((window >> (32 - KHG_LUT_BITS - KHG_LUT2_BITS)) & (KHG_LUT2_SIZE - 1))];
        if (e & 0x80000000u) {
            len = (e >> 16) & 0xFF;
            *sym = (int)(int16_t)(e & 0xFFFF);
        } else {
            // code longer than 20 bits: continue the walk of Huffman.h:39-51 bit by bit
            int node = (int)e;
            len = KHG_LUT_BITS + KHG_LUT2_BITS;
            while (left[node] >= 0) {
                if (pos + len >= b.nbits) return 0;
                uint32_t bit = b.peek32(pos + len) >> 31;
                node = bit ? right[node] : left[node];
                ++len;
            }
            *sym = tsym[node];
        }
    }
    if (pos + len > b.nbits) return 0;
    return len;
}

// LDS of the map kernel.  MO = start offsets it has room for: with 32 (codes up to 32 bits, i.e. nearly every
// stream) a workgroup needs 40 488 bytes and four fit on a CU (163 840 bytes); streams with longer codes are handled by the
// KHG_MAXOFF instantiation (three per CU).  A launch of either does nothing for planes that belong to the other.
enum { KHG_MAP_SMALL = 32, KHG_MAP_THREADS = 2 * KHG_DEC_BLOCK };  // two threads per subsequence (khg_dec_map_body)
template <int MO>
struct KhgDecSmem {
    uint8_t len8[KHG_LUT_SIZE];                    // see khg_dec_stage_lens
    uint32_t stage[KHG_STAGE_LDS];
    uint32_t mask[(KHG_SUB_WORDS + 1) * KHG_DEC_BLOCK];  // word-major: thread t's word w at [w * 256 + t]; a subsequence
                                                         // that starts inside a staged word reaches into one more
    uint16_t map[MO * KHG_DEC_BLOCK];              // offset-major: thread t's entry for offset o at [o * 256 + t]
    uint32_t main_arr[KHG_DEC_BLOCK];              // where subsequence t's recording run lands | its offset << 8 | 0x8000
    uint16_t runs[KHG_MAP_THREADS];                // step 2b's work list: subsequence | offset << 8
    uint32_t scan[KHG_MAP_THREADS / 64];
};

// Stage the LUT and the workgroup's span of the stream (byte-swapped words) in LDS.
template <class SMEM>
KHG_DEV KhgBitsL khg_dec_stage(const KhgDecArgs& a, int p, uint32_t blk_begin, SMEM& sh)
{
    const uint32_t* words = (const uint32_t*)a.comp;
    const uint32_t nwords = (uint32_t)(a.len >> 2);
    const uint32_t* glut = a.lut + (size_t)p * KHG_LUT_SIZE;
    const uint64_t gbit = 8ull * a.hdr->stream_byte[p] + blk_begin;
    const uint32_t w0 = (uint32_t)(gbit >> 5);
    KHG_FOR_THREADS(256)
    {
        for (int i = tid; i < KHG_LUT_SIZE; i += 256) sh.lut[i] = glut[i];
        for (uint32_t i = (uint32_t)tid; i < KHG_STAGE_WORDS; i += 256) {
            uint32_t w = w0 + i;
            khg_stage_put(sh.stage, i, w < nwords ? khg_bswap32(words[w]) : 0);
        }
    }
    KhgBitsL l;
    l.stage = sh.stage;
    l.bias = blk_begin - (uint32_t)(gbit & 31);
    l.nbits = (uint32_t)a.hdr->nbits[p];
    return l;
}

// The map kernel's staging: code lengths instead of the decode table (len8[v] = length of the code that the
// KHG_LUT_BITS-bit prefix v starts with, 0 if it is longer than that: then the general decoder, with the tables in
// global memory, finds it), and the workgroup's span of the stream.
template <class SMEM>
KHG_DEV KhgBitsL khg_dec_stage_lens(const KhgDecArgs& a, int p, uint32_t blk_begin, SMEM& sh)
{
    const uint32_t* words = (const uint32_t*)a.comp;
    const uint32_t nwords = (uint32_t)(a.len >> 2);
    const uint32_t* glut = a.lut + (size_t)p * KHG_LUT_SIZE;
    const uint64_t gbit = 8ull * a.hdr->stream_byte[p] + blk_begin;
    const uint32_t w0 = (uint32_t)(gbit >> 5);
    KHG_FOR_THREADS(KHG_MAP_THREADS)
    {
        for (int i = tid; i < KHG_LUT_SIZE; i += KHG_MAP_THREADS) {
            const uint32_t e = glut[i];
            sh.len8[i] = (e & 0x80000000u) ? (uint8_t)((e >> 16) & 0xFF) : (uint8_t)0;
        }
        for (uint32_t i = (uint32_t)tid; i < KHG_STAGE_WORDS; i += KHG_MAP_THREADS) {
            uint32_t w = w0 + i;
            khg_stage_put(sh.stage, i, w < nwords ? khg_bswap32(words[w]) : 0);
        }
    }
    KhgBitsL l;
    l.stage = sh.stage;
    l.bias = blk_begin - (uint32_t)(gbit & 31);
    l.nbits = (uint32_t)a.hdr->nbits[p];
    return l;
}

// One workgroup = 256 consecutive subsequences of one plane: subsequence maps and their composition.
template <int MO>
KHG_DEV void khg_dec_map_body(const KhgDecArgs& a, uint32_t blk, KhgDecSmem<MO>& sh)
{
    if (!a.hdr->ok) return;
    const int p = khg_block_plane(a, blk);
    const uint32_t nbits = (uint32_t)a.hdr->nbits[p];
    const uint32_t lb = blk - a.blk0[p];  // workgroup index inside the plane
    const uint32_t blk_begin = lb * KHG_BLK_BITS;
    if (blk_begin >= nbits) return;
    const uint32_t M = (uint32_t)a.hdr->maxlen[p];
    if (M == 0 || M > KHG_MAXOFF) return;  // flagged by the tree kernel
    if ((M <= KHG_MAP_SMALL) != (MO == KHG_MAP_SMALL)) return;  // the other instantiation's plane
    const uint32_t* glut = a.lut + (size_t)p * KHG_LUT_SIZE;
    const int32_t* left = a.tree_left + (size_t)p * 2 * KHG_NBINS;
    const int32_t* right = a.tree_right + (size_t)p * 2 * KHG_NBINS;
    const int32_t* tsym = a.tab_sym + (size_t)p * KHG_NBINS;
    const uint32_t* lut2 = a.lut2 + (size_t)p * KHG_LUT_SIZE * KHG_LUT2_SIZE;
    const int onebit = a.hdr->one_bit[p];
#ifdef KHG_TIMING
    const long long tmap0 = wall_clock64();
#endif
    const KhgBitsL bits = khg_dec_stage_lens(a, p, blk_begin, sh);
    KHG_BARRIER();

    // A subsequence is walked one 32-bit word of the staged stream at a time: the word, its successor and the
    // word's start marks sit in registers while the codewords that begin in it are stepped over, so a step costs one
    // LDS access (the code length under the bit window) instead of four (two stream words, the mark word, the
    // length).  Positions are kept relative to the
    // first staged word of the subsequence: lr = (position in the subsequence) + s0, the same s0 < 32 for every
    // thread (subsequences are 256 bits, a whole number of words, apart).
    const uint32_t s0 = (blk_begin - bits.bias) & 31u;
    // Two threads per subsequence: thread t and thread 256 + t share subsequence t's columns of `map` and `mask`
    // and split its start offsets by parity for the first codewords; the recording run is one thread's work (the
    // partner waits at the barrier), and the runs behind it are dealt out to all the workgroup's threads (step 2b).
    // Where runs never rejoin (stretches of near-equal code lengths: the LL band) a subsequence has one whole-length
    // run per offset, and the workgroups of those stretches last longest -- the launch lasts as long as they do;
    // spreading their runs over twice the threads is what shortens it.
#define KHG_MAP_SUB_SETUP /* subsequence t of the workgroup */                                                     \
        const uint32_t sub_begin = blk_begin + (uint32_t)t * KHG_SUB_BITS;                                         \
        const uint32_t end = sub_begin + KHG_SUB_BITS < nbits ? sub_begin + KHG_SUB_BITS : nbits;                  \
        const uint32_t lr_of = s0 - sub_begin;            /* lr = pos + lr_of */                                   \
        const uint32_t end_lr = sub_begin < end ? end + lr_of : 0;  /* first lr behind the subsequence (or the stream) */ \
        const uint32_t nb_lr = nbits + lr_of;             /* first lr behind the stream (sub_begin < nbits: no wrap) */ \
        const uint32_t wbase = (uint32_t)t * KHG_SUB_WORDS;  /* first staged word of the subsequence */           \
        uint16_t* map = &sh.map[t];                                                                                \
        uint32_t* mask = &sh.mask[t];                                                                              \
        (void)lr_of; (void)end_lr; (void)nb_lr; (void)wbase; (void)map; (void)mask;
#define KHG_MAP_THREAD_SETUP                                                                                       \
        const int t = tid & (KHG_DEC_BLOCK - 1);                                                                   \
        const uint32_t half = (uint32_t)tid >> 8;                                                                  \
        (void)half;                                                                                                \
        KHG_MAP_SUB_SETUP

    // 1. first codeword from every start offset: where does it land?  The offsets lie in the first three words.
    //    escs: the thread's offsets whose first codeword leaves the first M bits (bit o), kept for step 2b.
    uint64_t escs[KHG_NSLOT(KHG_MAP_THREADS)];
    KHG_FOR_THREADS(KHG_MAP_THREADS)
    {
        KHG_MAP_THREAD_SETUP
        uint64_t esc = 0;
        const uint32_t w0 = bits.stage[khg_stage_slot(wbase)], w1 = bits.stage[khg_stage_slot(wbase + 1)],
                       w2 = bits.stage[khg_stage_slot(wbase + 2)], w3 = bits.stage[khg_stage_slot(wbase + 3)];
        for (uint32_t o = half; o < M; o += 2) {
            const uint32_t lr = s0 + o, j = lr >> 5, sh_ = lr & 31u;
            uint16_t e = khg_map_entry(KHG_OFF_NONE, 0);
            if (lr < end_lr) {
                const uint32_t hi = j == 0 ? w0 : j == 1 ? w1 : w2, lo = j == 0 ? w1 : j == 1 ? w2 : w3;
                const uint32_t win = (uint32_t)(((((uint64_t)hi << 32) | lo) << sh_) >> 32);
                uint32_t len = sh.len8[win >> (32 - KHG_LUT_BITS)];
                if (len == 0) {  // longer than the first-level table: the general decoder
                    int sy;
                    len = khg_decode_one(bits, glut, lut2, left, right, tsym, sub_begin + o, &sy);
                } else if (lr + len > nb_lr) {
                    len = 0;  // the stream ends inside this codeword
                }
                if (len) e = khg_map_entry(KHG_OFF_TEMP, o + len);  // landing offset < 2 * 56
                if (len && o + len >= M) esc |= 1ull << o;
            }
            map[o * KHG_DEC_BLOCK] = e;
        }
        escs[KHG_SLOT] = esc;
        if (half == 0)
            for (uint32_t w = 0; w <= KHG_SUB_WORDS; ++w) mask[w * KHG_DEC_BLOCK] = 0;
    }
    KHG_BARRIER();
    // 2. offsets whose first codeword leaves the first M bits run on.  The first such run (from the highest
    //    offset) goes through the whole subsequence and records its codeword starts; every other run stops as
    //    soon as it lands on one of them (inside a run of the 1-bit code every position is one).  Most rejoin
    //    within a few codewords; a stretch of near-equal code lengths has one run per residue class that never
    //    does.
    //    The inner loops are straight-line code with one exit at the bottom (the lanes of a wave stay in one
    //    loop); whatever is rare -- a code beyond the first-level table, the end of the stream -- leaves them
    //    and is dealt with once per word.
    const uint32_t runx = onebit < 0 ? 0xFFFFFFFFu : 0u;       // planes without a 1-bit code: no runs
    const uint32_t runflip = onebit > 0 ? 0xFFFFFFFFu : 0u;    // the run's bits read as zeros
    // 2a. the recording run; a run of the 1-bit code is stepped over at once (as far as the word goes)
    KHG_FOR_THREADS(KHG_MAP_THREADS)
    {
        KHG_MAP_THREAD_SETUP
        if (half == 0) {
            int oo = (int)M;
            bool found = false;
            uint32_t ro = 0, lr = 0;
            while (oo > 0) {
                --oo;
                const uint16_t e0 = map[oo * KHG_DEC_BLOCK];
                if (khg_map_arr(e0) == KHG_OFF_TEMP && khg_map_cnt(e0) >= M) {
                    found = true;
                    ro = (uint32_t)oo;
                    lr = s0 + khg_map_cnt(e0);
                    break;
                }
            }
            uint32_t main_arr = KHG_OFF_NONE;
            if (found) {
                uint32_t n = 1;
                bool dead = false;
                if (lr < end_lr) {
                    uint32_t j = lr >> 5, adv = 0;
                    uint32_t wa = bits.stage[khg_stage_slot(wbase + j)];
                    for (;;) {  // one word of the stream per trip
                        const uint32_t wb = bits.stage[khg_stage_slot(wbase + j + 1)];
                        const uint32_t lim = end_lr < 32u * (j + 1) ? end_lr : 32u * (j + 1);  // this trip's positions end here
                        uint32_t acc = 0;
                        do {  // the codewords that begin in word j
                            const uint32_t sh_ = lr & 31u;
                            const uint32_t win = (uint32_t)(((((uint64_t)wa << 32) | wb) << sh_) >> 32);
                            const uint32_t x = (win ^ runflip) | runx;
                            uint32_t r = x ? (uint32_t)__builtin_clz(x) : 32u;  // the run of the 1-bit code here
                            r = r > lim - lr ? lim - lr : r;
                            const uint32_t len = sh.len8[win >> (32 - KHG_LUT_BITS)];  // 0: beyond the table
                            adv = r ? r : len;
                            const uint32_t starts = r ? r : (len ? 1u : 0u);
                            acc |= (uint32_t)(0xFFFFFFFF00000000ull >> starts) >> sh_;  // `starts` ones from here on
                            lr += adv;
                            n += starts;
                        } while (adv != 0 && lr < lim);
                        if (adv == 0) {  // a code beyond the first-level table (or the stream's ragged end)
                            int sy;
                            const uint32_t len = khg_decode_one(bits, glut, lut2, left, right, tsym, lr - lr_of, &sy);
                            if (len == 0) {
                                dead = true;
                            } else {
                                acc |= 0x80000000u >> (lr & 31u);
                                lr += len;
                                ++n;
                            }
                        }
                        mask[j * KHG_DEC_BLOCK] |= acc;  // (a word is entered twice only behind a long code that ends in it)
                        if (dead || lr >= end_lr) break;
                        const uint32_t jn = lr >> 5;
                        if (jn != j) wa = jn == j + 1 ? wb : bits.stage[khg_stage_slot(wbase + jn)];
                        j = jn;
                    }
                    if (!dead && lr > nb_lr) {  // the stream ends inside the last codeword: not a codeword, not a start
                        dead = true;
                        --n;
                        const uint32_t at = lr - adv;
                        mask[(at >> 5) * KHG_DEC_BLOCK] &= ~(0x80000000u >> (at & 31u));
                    }
                }
                if (dead || lr >= nb_lr) main_arr = KHG_OFF_NONE;
                else main_arr = lr - (s0 + KHG_SUB_BITS);  // behind the subsequence: an overshoot < M
                map[ro * KHG_DEC_BLOCK] = khg_map_entry(main_arr, n);
            }
            // where the recording run lands, and the offset it started from: offsets below that one may still run on
            sh.main_arr[t] = main_arr | (found ? 0x8000u | (ro << 8) : 0u);
#ifdef KHG_TIMING
            if (tid == 0) a.blk[blk].pad_[4] = (uint32_t)wall_clock64();
#endif
        }
    }
    KHG_BARRIER();
    // 2b. the other runs.  How many a subsequence has differs (none to one per residue class), and so does what a
    //     thread would find among its own offsets; a wave that lets every lane work through its own list lasts as
    //     long as its longest list and runs mostly empty.  So the workgroup's runs are numbered (a scan over the
    //     threads' counts), their (subsequence, offset) pairs are listed in LDS, KHG_MAP_THREADS at a time, and every
    //     thread takes one from the list: a run reads the start marks and writes only its own map entry, so any
    //     thread can do it.
    uint32_t rcnt[KHG_NSLOT(KHG_MAP_THREADS)], rbase[KHG_NSLOT(KHG_MAP_THREADS)], rtotal = 0;
    KHG_FOR_THREADS(KHG_MAP_THREADS)
    {
        const int t = tid & (KHG_DEC_BLOCK - 1);
        const uint32_t mw = sh.main_arr[t];
        uint64_t esc = escs[KHG_SLOT];
        esc = (mw & 0x8000u) ? esc & ((1ull << ((mw >> 8) & 63u)) - 1) : 0ull;  // below the recording run's offset
        escs[KHG_SLOT] = esc;
        rcnt[KHG_SLOT] = (uint32_t)__builtin_popcountll(esc);
    }
    KHG_WG_EXCL_SCAN_NW(KHG_MAP_THREADS / 64, rbase, rcnt, sh.scan, rtotal)
    for (uint32_t r0 = 0; r0 < rtotal; r0 += KHG_MAP_THREADS) {
        if (r0) KHG_BARRIER();  // the list of the pass before has been read
        KHG_FOR_THREADS(KHG_MAP_THREADS)
        {
            const uint32_t t = (uint32_t)tid & (KHG_DEC_BLOCK - 1);
            uint64_t esc = escs[KHG_SLOT];
            uint32_t idx = rbase[KHG_SLOT] - r0;  // of the thread's first run in this pass's list (may lie outside)
            while (esc) {
                const uint32_t o = 63u - (uint32_t)__builtin_clzll(esc);
                esc &= ~(1ull << o);
                if (idx < (uint32_t)KHG_MAP_THREADS) sh.runs[idx] = (uint16_t)(t | (o << 8));
                ++idx;
            }
        }
        KHG_BARRIER();
        KHG_FOR_THREADS(KHG_MAP_THREADS)
        {
            if (r0 + (uint32_t)tid < rtotal) {
                const uint32_t item = sh.runs[tid];
                const int t = (int)(item & 255u);
                const uint32_t ro = item >> 8;
                KHG_MAP_SUB_SETUP
                const uint32_t main_arr = sh.main_arr[t] & 63u;
                uint32_t lr = s0 + khg_map_cnt(map[ro * KHG_DEC_BLOCK]);  // where the offset's first codeword lands
                uint32_t n = 1, join = 0xFFFFFFFFu;
                bool dead = false;
                if (lr < end_lr) {
                    uint32_t j = lr >> 5;
                    uint32_t wa = bits.stage[khg_stage_slot(wbase + j)];
                    for (;;) {
                        const uint32_t wb = bits.stage[khg_stage_slot(wbase + j + 1)];
                        const uint32_t mw = mask[j * KHG_DEC_BLOCK];
                        const uint32_t lim = end_lr < 32u * (j + 1) ? end_lr : 32u * (j + 1);
                        uint32_t tt, len;
                        do {
                            const uint32_t sh_ = lr & 31u;
                            const uint32_t win = (uint32_t)(((((uint64_t)wa << 32) | wb) << sh_) >> 32);
                            tt = mw << sh_;  // top bit: a start of the recording run, identical from here on
                            len = sh.len8[win >> (32 - KHG_LUT_BITS)];
                            len = (int32_t)tt < 0 ? 0u : len;
                            lr += len;
                            n += len ? 1u : 0u;
                        } while (len != 0 && lr < lim);
                        if ((int32_t)tt < 0) {
                            join = lr;
                            break;
                        }
                        if (len == 0) {  // a code beyond the first-level table
                            int sy;
                            const uint32_t l2 = khg_decode_one(bits, glut, lut2, left, right, tsym, lr - lr_of, &sy);
                            if (l2 == 0) {
                                dead = true;
                                break;
                            }
                            lr += l2;
                            ++n;
                        }
                        if (lr >= end_lr) break;
                        const uint32_t jn = lr >> 5;
                        if (jn != j) wa = jn == j + 1 ? wb : bits.stage[khg_stage_slot(wbase + jn)];
                        j = jn;
                    }
                    if (!dead && join == 0xFFFFFFFFu && lr > nb_lr) {  // the stream ends inside the last codeword
                        dead = true;
                        --n;
                    }
                }
                uint32_t arr;
                if (join != 0xFFFFFFFFu) {
                    const uint32_t wq = join >> 5, bit = 0x80000000u >> (join & 31u);
                    uint32_t rest = (uint32_t)__builtin_popcount(mask[wq * KHG_DEC_BLOCK] & (bit | (bit - 1)));
                    for (uint32_t w2 = wq + 1; w2 <= KHG_SUB_WORDS; ++w2)
                        rest += (uint32_t)__builtin_popcount(mask[w2 * KHG_DEC_BLOCK]);
                    n += rest;
                    arr = main_arr;
                } else if (dead || lr >= nb_lr) {
                    arr = KHG_OFF_NONE;
                } else {
                    arr = lr - (s0 + KHG_SUB_BITS);  // behind the subsequence: an overshoot < M
                }
                map[ro * KHG_DEC_BLOCK] = khg_map_entry(arr, n);
            }
        }
    }
    KHG_BARRIER();
    // 3. the other offsets reuse the entry of the offset their first codeword lands on
    KHG_FOR_THREADS(KHG_MAP_THREADS)
    {
        KHG_MAP_THREAD_SETUP
        if (half == 0) {
            for (uint32_t o3 = M; o3-- > 0;) {
                const uint16_t e0 = map[o3 * KHG_DEC_BLOCK];
                if (khg_map_arr(e0) != KHG_OFF_TEMP) continue;
                const uint16_t r = map[khg_map_cnt(e0) * KHG_DEC_BLOCK];  // landing offset > o3: already final
                map[o3 * KHG_DEC_BLOCK] = khg_map_entry(khg_map_arr(r), khg_map_cnt(r) + 1);
            }
        }
#ifdef KHG_TIMING
        if (tid == 0) a.blk[blk].pad_[5] = (uint32_t)wall_clock64();
#endif
    }
#undef KHG_MAP_THREAD_SETUP
    KHG_BARRIER();
    // Compose the 256 subsequence maps.  For a workgroup entered at offset o the result is its own map entry
    // (landing offset, codewords) and, for every subsequence t, the offset it is entered at and the codewords
    // before it: walk[t * M + o] = (entry offset << 24) | count.  Once the chain has fixed the workgroup's true
    // entry offset, column o of this table is all the write kernel needs to start 256 independent decoders.
    // Composition is associative, so it goes in three short sweeps instead of one of 256 dependent steps:
    // 16 groups of 16 subsequences are composed for every entry offset (one lane per group and offset), one lane
    // per o chains the 16 group maps, and the first sweep is repeated from the true group entries to write the
    // table.  The start masks are dead by now and lend their memory to the two small tables.
    uint32_t* gmap = sh.mask;            // [16][MO]: group g entered at o -> (landing offset << 24) | codewords
    uint32_t* gent = sh.mask + 16 * MO;  // [16][MO]: workgroup entered at o -> state at the start of group g
    KHG_FOR_THREADS(KHG_MAP_THREADS)
    {
        for (uint32_t idx = (uint32_t)tid; idx < 16 * M; idx += KHG_MAP_THREADS) {
            const uint32_t g = idx / M, o = idx - g * M;
            uint32_t off = o, tot = 0;
            for (uint32_t t = 16 * g; t < 16 * g + 16; ++t) {
                if (off != KHG_OFF_NONE) {
                    const uint16_t e = sh.map[off * KHG_DEC_BLOCK + t];
                    tot += khg_map_cnt(e);
                    off = khg_map_arr(e);
                }
            }
            gmap[g * MO + o] = (off << 24) | tot;
        }
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(KHG_MAP_THREADS)
    {
        if ((uint32_t)tid < M) {
            uint32_t off = (uint32_t)tid, tot = 0;
            for (uint32_t g = 0; g < 16; ++g) {
                gent[g * MO + (uint32_t)tid] = (off << 24) | tot;
                if (off != KHG_OFF_NONE) {
                    const uint32_t v = gmap[g * MO + off];
                    tot += v & 0xFFFFFFu;
                    off = v >> 24;
                }
            }
            a.blk[blk].map[tid] = (off << 24) | tot;
        }
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(KHG_MAP_THREADS)
    {
        uint32_t* gw = a.walk + (size_t)blk * KHG_MAXOFF * KHG_DEC_BLOCK;
        for (uint32_t idx = (uint32_t)tid; idx < 16 * M; idx += KHG_MAP_THREADS) {
            const uint32_t g = idx / M, o = idx - g * M;
            const uint32_t v = gent[g * MO + o];
            uint32_t off = v >> 24, tot = v & 0xFFFFFFu;
            for (uint32_t t = 16 * g; t < 16 * g + 16; ++t) {
                gw[t * M + o] = (off << 24) | tot;
                if (off != KHG_OFF_NONE) {
                    const uint16_t e = sh.map[off * KHG_DEC_BLOCK + t];
                    tot += khg_map_cnt(e);
                    off = khg_map_arr(e);
                }
            }
        }
#ifdef KHG_TIMING
        if (tid == 0) {  // developer build: workgroup start / end stamps (100 MHz), read back by khg_debug_blocks
            a.blk[blk].pad_[0] = (uint32_t)tmap0;
            a.blk[blk].pad_[1] = (uint32_t)wall_clock64();
        }
#endif
    }
}

struct KhgDecChainSmem {
    uint32_t map[KHG_DEC_BLOCK * KHG_MAXOFF];
    uint32_t gmap[16 * KHG_MAXOFF];  // 16 workgroups in a row entered at offset o -> (landing offset << 24) | codewords
    uint32_t goff[16], gout[16];     // state at the start of each such group
    uint32_t off, out;               // state carried from one batch of 256 workgroups to the next
    uint32_t total;
};

// bid = colour component.  Follows the workgroup maps from bit 0 of the stream: where each workgroup is entered and
// how many codewords precede it; afterwards zero-fills raster cells past the decoded symbols (Huffman.h:35: the matrix
// starts zero-initialised).  The walk is a chain of dependent lookups, one per workgroup; composition is associative,
// so 256 workgroups at a time (their maps staged in LDS) it goes in three short sweeps instead: 16 groups of 16
// workgroups composed for every entry offset (one thread per group and offset), one thread chaining the 16 group
// maps, 16 threads walking their group from its true entry.
KHG_DEV void khg_dec_chain_body(const KhgDecArgs& a, int bid, KhgDecChainSmem& sh)
{
    if (!a.hdr->ok) return;
    const int p = bid;
    const uint32_t nbits = (uint32_t)a.hdr->nbits[p];
    const uint32_t nb = (nbits + KHG_BLK_BITS - 1) / KHG_BLK_BITS;  // active workgroups of this plane
    KhgBlkInfo* blk = a.blk + a.blk0[p];
    const uint32_t M = (uint32_t)a.hdr->maxlen[p] <= (uint32_t)KHG_MAXOFF ? (uint32_t)a.hdr->maxlen[p] : 0u;  // offsets in use
    KHG_FOR_THREADS(256)
    {
        if (tid == 0) {
            sh.off = M ? 0 : KHG_OFF_NONE;
            sh.out = 0;
        }
    }
    for (uint32_t t0 = 0; t0 < nb; t0 += KHG_DEC_BLOCK) {
        const uint32_t nt = nb - t0 < KHG_DEC_BLOCK ? nb - t0 : KHG_DEC_BLOCK;
        KHG_FOR_THREADS(256)
        {
            for (uint32_t i = (uint32_t)tid; i < nt * M; i += 256) {  // the M entries in use of every map
                const uint32_t b = i / M, o = i - b * M;
                sh.map[b * KHG_MAXOFF + o] = blk[t0 + b].map[o];
            }
        }
        KHG_BARRIER();
        KHG_FOR_THREADS(256)
        {
            for (uint32_t idx = (uint32_t)tid; idx < 16 * M; idx += 256) {
                const uint32_t g = idx / M, o = idx - g * M;
                uint32_t off = o, tot = 0;
                for (uint32_t b = 16 * g; b < 16 * g + 16 && b < nt; ++b) {
                    if (off < M) {  // else: the stream has ended (KHG_OFF_NONE)
                        const uint32_t e = sh.map[b * KHG_MAXOFF + off];
                        tot += e & 0xFFFFFFu;  // at most 16 x 2^16 codewords
                        off = e >> 24;
                    }
                }
                sh.gmap[g * KHG_MAXOFF + o] = ((off < M ? off : (uint32_t)KHG_OFF_NONE) << 24) | tot;
            }
        }
        KHG_BARRIER();
        KHG_FOR_THREADS(256)
        {
            if (tid == 0) {
                uint32_t off = sh.off, out = sh.out;
                for (uint32_t g = 0; g < 16; ++g) {
                    sh.goff[g] = off;
                    sh.gout[g] = out;
                    if (off < M) {
                        const uint32_t e = sh.gmap[g * KHG_MAXOFF + off];
                        const uint32_t tot = e & 0xFFFFFFu;
                        out = out + tot < out ? 0xFFFFFFFFu : out + tot;
                        off = e >> 24;
                    }
                }
                sh.off = off;
                sh.out = out;
            }
        }
        KHG_BARRIER();
        KHG_FOR_THREADS(256)
        {
            if (tid < 16) {
                uint32_t off = sh.goff[tid], out = sh.gout[tid];
                for (uint32_t b = 16 * (uint32_t)tid; b < 16 * (uint32_t)tid + 16 && b < nt; ++b) {
                    blk[t0 + b].start_off = off;
                    blk[t0 + b].out = out;
                    if (off < M) {
                        const uint32_t e = sh.map[b * KHG_MAXOFF + off];
                        const uint32_t tot = e & 0xFFFFFFu;
                        out = out + tot < out ? 0xFFFFFFFFu : out + tot;
                        off = e >> 24 < M ? e >> 24 : (uint32_t)KHG_OFF_NONE;
                    }
                }
            }
        }
        KHG_BARRIER();
    }
    KHG_FOR_THREADS(256)
    {
        if (tid == 0) {
            uint32_t out = sh.out;
            if (out > a.n[p]) {
                khg_set_status(&a.res->status, -3);  // the reference would write out of bounds
                out = a.n[p];
            }
            a.res->bits[p] = (int32_t)out;  // symbols decoded
            sh.total = out;
        }
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        for (uint32_t i = sh.total + (uint32_t)tid; i < a.n[p]; i += 256) a.sym[p][i] = 0;
    }
}

// ---- The same chain for long streams, in two launches with one workgroup per batch of workgroup maps instead of one
// workgroup per plane walking through all batches (a 32768 x 32768 image has 69 batches of 256 in its luma stream).
// First launch: every batch's composed map, entry offset -> (landing offset, codewords), kept in spare words of the
// batch's own records.  Second launch: every workgroup chains the maps of the batches before its own (staged in LDS,
// one dependent lookup per batch) and then finishes its batch as khg_dec_chain_body does; the last one also writes
// the totals and clears the tail.
#ifndef KHG_CHAIN_GROUPS
#define KHG_CHAIN_GROUPS 16  // groups of 16 workgroup maps per batch (the simulator's build uses fewer)
#endif
enum { KHG_CHAIN_BATCH = 16 * KHG_CHAIN_GROUPS };

// Batch map entry o of the batch whose first record is `first`: two spare words of record o / 3.
KHG_DEV uint32_t* khg_batch_map_slot(KhgBlkInfo* first, uint32_t o) { return &first[o / 3].pad_[2 * (o % 3)]; }

struct KhgDecChainSplitSmem {
    uint32_t map[KHG_CHAIN_BATCH * KHG_MAXOFF];
    uint32_t gmap[KHG_CHAIN_GROUPS * KHG_MAXOFF];
    uint32_t goff[KHG_CHAIN_GROUPS], gout[KHG_CHAIN_GROUPS];
    uint32_t off, out;
    uint32_t total;
};

// Stages the maps of batch `batch` and composes its groups of 16 for every entry offset (the first two steps of
// khg_dec_chain_body).  Returns the number of maps staged (0: nothing to do).
KHG_DEV uint32_t khg_chain_stage_batch(const KhgDecArgs& a, int p, uint32_t batch, uint32_t nb, uint32_t M,
                                       KhgDecChainSplitSmem& sh)
{
    const KhgBlkInfo* blk = a.blk + a.blk0[p];
    const uint32_t t0 = batch * KHG_CHAIN_BATCH;
    const uint32_t nt = nb - t0 < (uint32_t)KHG_CHAIN_BATCH ? nb - t0 : (uint32_t)KHG_CHAIN_BATCH;
    KHG_FOR_THREADS(256)
    {
        for (uint32_t i = (uint32_t)tid; i < nt * M; i += 256) {
            const uint32_t b = i / M, o = i - b * M;
            sh.map[b * KHG_MAXOFF + o] = blk[t0 + b].map[o];
        }
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        for (uint32_t idx = (uint32_t)tid; idx < (uint32_t)KHG_CHAIN_GROUPS * M; idx += 256) {
            const uint32_t g = idx / M, o = idx - g * M;
            uint32_t off = o, tot = 0;
            for (uint32_t b = 16 * g; b < 16 * g + 16 && b < nt; ++b) {
                if (off < M) {
                    const uint32_t e = sh.map[b * KHG_MAXOFF + off];
                    tot += e & 0xFFFFFFu;
                    off = e >> 24;
                }
            }
            sh.gmap[g * KHG_MAXOFF + o] = ((off < M ? off : (uint32_t)KHG_OFF_NONE) << 24) | tot;
        }
    }
    KHG_BARRIER();
    return nt;
}

KHG_DEV void khg_dec_chain_maps_body(const KhgDecArgs& a, int p, uint32_t batch, KhgDecChainSplitSmem& sh)
{
    if (!a.hdr->ok) return;
    const uint32_t nb = khg_dec_active_blocks((uint32_t)a.hdr->nbits[p]);
    const uint32_t M = (uint32_t)a.hdr->maxlen[p] <= (uint32_t)KHG_MAXOFF ? (uint32_t)a.hdr->maxlen[p] : 0u;
    // the last batch's map is nobody's predecessor
    if (M == 0 || (uint64_t)(batch + 1) * KHG_CHAIN_BATCH >= nb) return;
    khg_chain_stage_batch(a, p, batch, nb, M, sh);
    KHG_FOR_THREADS(256)
    {
        if ((uint32_t)tid < M) {
            uint32_t off = (uint32_t)tid, tot = 0;  // at most 256 x 2^16 codewords: a word of its own
            for (uint32_t g = 0; g < (uint32_t)KHG_CHAIN_GROUPS; ++g) {
                if (off < M) {
                    const uint32_t e = sh.gmap[g * KHG_MAXOFF + off];
                    tot += e & 0xFFFFFFu;
                    off = e >> 24;
                }
            }
            uint32_t* slot = khg_batch_map_slot(a.blk + a.blk0[p] + batch * KHG_CHAIN_BATCH, (uint32_t)tid);
            slot[0] = off < M ? off : (uint32_t)KHG_OFF_NONE;
            slot[1] = tot;
        }
    }
}

KHG_DEV void khg_dec_chain_finish_body(const KhgDecArgs& a, int p, uint32_t batch, KhgDecChainSplitSmem& sh)
{
    if (!a.hdr->ok) return;
    const uint32_t nb = khg_dec_active_blocks((uint32_t)a.hdr->nbits[p]);
    const uint32_t M = (uint32_t)a.hdr->maxlen[p] <= (uint32_t)KHG_MAXOFF ? (uint32_t)a.hdr->maxlen[p] : 0u;
    const uint32_t nbatch = (nb + KHG_CHAIN_BATCH - 1) / KHG_CHAIN_BATCH;
    const bool tail = batch + 1 == nbatch || (nbatch == 0 && batch == 0);  // this workgroup closes the plane
    if (batch >= nbatch && !tail) return;
    if ((uint64_t)2 * batch * M > (uint64_t)KHG_CHAIN_BATCH * KHG_MAXOFF) {  // more batches than the staging area holds
        KHG_FOR_THREADS(256) { if (tid == 0) khg_set_status(&a.res->status, -6); }
        return;
    }
    KhgBlkInfo* blk = a.blk + a.blk0[p];
    // the maps of the batches before this one, two words per entry offset, in the (still free) map area
    KHG_FOR_THREADS(256)
    {
        for (uint32_t i = (uint32_t)tid; i < batch * M; i += 256) {
            const uint32_t k = i / M, o = i - k * M;
            const uint32_t* slot = khg_batch_map_slot(blk + k * KHG_CHAIN_BATCH, o);
            sh.map[2 * i] = slot[0];
            sh.map[2 * i + 1] = slot[1];
        }
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        if (tid == 0) {
            uint32_t off = M ? 0 : (uint32_t)KHG_OFF_NONE, out = 0;
            for (uint32_t k = 0; k < batch; ++k) {
                if (off < M) {
                    const uint32_t tot = sh.map[2 * (k * M + off) + 1];
                    out = out + tot < out ? 0xFFFFFFFFu : out + tot;
                    off = sh.map[2 * (k * M + off)];
                }
            }
            sh.off = off;
            sh.out = out;
        }
    }
    KHG_BARRIER();
    const uint32_t nt = nbatch ? khg_chain_stage_batch(a, p, batch, nb, M, sh) : 0;
    const uint32_t t0 = batch * KHG_CHAIN_BATCH;
    KHG_FOR_THREADS(256)
    {
        if (tid == 0) {
            uint32_t off = sh.off, out = sh.out;
            for (uint32_t g = 0; g < (uint32_t)KHG_CHAIN_GROUPS; ++g) {
                sh.goff[g] = off;
                sh.gout[g] = out;
                if (off < M && 16 * g < nt) {
                    const uint32_t e = sh.gmap[g * KHG_MAXOFF + off];
                    const uint32_t tot = e & 0xFFFFFFu;
                    out = out + tot < out ? 0xFFFFFFFFu : out + tot;
                    off = e >> 24;
                }
            }
            sh.off = off;
            sh.out = out;
        }
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        if (tid < KHG_CHAIN_GROUPS) {
            uint32_t off = sh.goff[tid], out = sh.gout[tid];
            for (uint32_t b = 16 * (uint32_t)tid; b < 16 * (uint32_t)tid + 16 && b < nt; ++b) {
                blk[t0 + b].start_off = off;
                blk[t0 + b].out = out;
                if (off < M) {
                    const uint32_t e = sh.map[b * KHG_MAXOFF + off];
                    const uint32_t tot = e & 0xFFFFFFu;
                    out = out + tot < out ? 0xFFFFFFFFu : out + tot;
                    off = e >> 24 < M ? e >> 24 : (uint32_t)KHG_OFF_NONE;
                }
            }
        }
    }
    if (!tail) return;
    KHG_FOR_THREADS(256)
    {
        if (tid == 0) {
            uint32_t out = sh.out;
            if (out > a.n[p]) {
                khg_set_status(&a.res->status, -3);  // the reference would write out of bounds
                out = a.n[p];
            }
            a.res->bits[p] = (int32_t)out;  // symbols decoded
            sh.total = out;
        }
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        for (uint32_t i = sh.total + (uint32_t)tid; i < a.n[p]; i += 256) a.sym[p][i] = 0;
    }
}

enum { KHG_WRITE_BUCKETS = 32 };
struct KhgDecWriteSmem {
    uint32_t lut[KHG_LUT_SIZE];
    uint32_t stage[KHG_STAGE_LDS];
    uint32_t pre[KHG_DEC_BLOCK + 1];                 // (start offset << 24) | codewords before subsequence t; [256]: all
    uint32_t hist[KHG_WRITE_BUCKETS], hbase[KHG_WRITE_BUCKETS];  // subsequences per bucket of codeword counts; before it
    uint8_t perm[KHG_DEC_BLOCK];                     // the subsequence thread i decodes
};

// One workgroup = 256 subsequences: every thread looks up its exact start and output position, then
// decodes and stores its symbols.
KHG_DEV void khg_dec_write_body(const KhgDecArgs& a, uint32_t blk, KhgDecWriteSmem& sh)
{
    if (!a.hdr->ok || a.res->status != 0) return;
    const int p = khg_block_plane(a, blk);
    const uint32_t nbits = (uint32_t)a.hdr->nbits[p];
    const uint32_t lb = blk - a.blk0[p];
    const uint32_t blk_begin = lb * KHG_BLK_BITS;
    if (blk_begin >= nbits) return;
    const uint32_t M = (uint32_t)a.hdr->maxlen[p];
    if (M == 0 || M > KHG_MAXOFF) return;
    const uint32_t entry = a.blk[blk].start_off;
    if (entry == KHG_OFF_NONE) return;  // the stream broke off before this workgroup
    const int32_t* left = a.tree_left + (size_t)p * 2 * KHG_NBINS;
    const int32_t* right = a.tree_right + (size_t)p * 2 * KHG_NBINS;
    const int32_t* tsym = a.tab_sym + (size_t)p * KHG_NBINS;
    const uint32_t* lut2 = a.lut2 + (size_t)p * KHG_LUT_SIZE * KHG_LUT2_SIZE;
    const int onebit = a.hdr->one_bit[p], onesym = a.hdr->one_sym[p];
    const uint32_t* gw = a.walk + (size_t)blk * KHG_MAXOFF * KHG_DEC_BLOCK;
#ifdef KHG_TIMING
    const long long tm0 = wall_clock64();
#endif
    const KhgBitsL bits = khg_dec_stage(a, p, blk_begin, sh);
    KHG_FOR_THREADS(256)
    {
        sh.pre[tid] = gw[(uint32_t)tid * M + entry];  // (start offset << 24) | codewords before the subsequence
        if (tid == 0) sh.pre[KHG_DEC_BLOCK] = a.blk[blk].map[entry] & 0xFFFFFFu;  // codewords of the whole workgroup
        if (tid < KHG_WRITE_BUCKETS) sh.hist[tid] = 0;
    }
    KHG_BARRIER();
    // Who decodes what.  A subsequence costs one trip per codeword that is not the 1-bit one, and of those it holds
    // the more the fewer codewords it has (a stretch of LL symbols: 30 codewords of 8 bits; a stretch of quiet detail:
    // 200 of them, nearly all one bit).  A wave lasts as long as its costliest lane, so the workgroup's subsequences
    // are dealt out sorted by their codeword counts (a counting sort in LDS, eight counts to a bucket) instead of in
    // stream order: at the High preset a wave then takes 22 trips on average instead of 34.
    uint32_t ranks[KHG_NSLOT(256)];
    KHG_FOR_THREADS(256)
    {
        const uint32_t cnt = (sh.pre[tid + 1] & 0xFFFFFFu) - (sh.pre[tid] & 0xFFFFFFu);
        const uint32_t key = cnt >> 3 < (uint32_t)KHG_WRITE_BUCKETS - 1 ? cnt >> 3 : (uint32_t)KHG_WRITE_BUCKETS - 1;
        ranks[KHG_SLOT] = (key << 16) | khg_atomic_add(&sh.hist[key], 1u);
    }
    // The workgroup's symbols are one contiguous stretch of the plane, and most of them are the plane's 1-bit
    // symbol: the stretch is first filled with it in 16-byte stores by all threads (what a lane would otherwise
    // write run by run, every lane of a wave into a different cache line), then each thread stores only the
    // other symbols of its subsequence.
    if (onebit >= 0) {
        KHG_FOR_THREADS(256)
        {
            const uint32_t lim = a.n[p], o0 = a.blk[blk].out;
            uint32_t o1 = o0 + sh.pre[KHG_DEC_BLOCK];
            if (o1 > lim) o1 = lim;  // never store past the plane
            if (o0 < o1) {
                int16_t* base = a.sym[p];  // 16-byte aligned (workspace)
                uint32_t a0 = (o0 + 7) & ~7u, a1 = o1 & ~7u;
                if (a0 > o1) a0 = o1;
                if (a1 < a0) a1 = a0;
                const uint64_t z8[2] = {0x0001000100010001ull * (uint64_t)(uint16_t)(int16_t)onesym,
                                        0x0001000100010001ull * (uint64_t)(uint16_t)(int16_t)onesym};
                for (uint32_t i = o0 + (uint32_t)tid; i < a0; i += 256) base[i] = (int16_t)onesym;
                for (uint32_t i = a0 + 8 * (uint32_t)tid; i < a1; i += 8 * 256) __builtin_memcpy(base + i, z8, 16);
                for (uint32_t i = a1 + (uint32_t)tid; i < o1; i += 256) base[i] = (int16_t)onesym;
            }
        }
    }
    KHG_BARRIER();  // the fill is complete (and visible to this workgroup) before single symbols land in it
    KHG_FOR_THREADS(256)
    {
        if (tid < KHG_WRITE_BUCKETS) {
            uint32_t before = 0;
            for (int k = 0; k < tid; ++k) before += sh.hist[k];
            sh.hbase[tid] = before;
        }
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256) { sh.perm[sh.hbase[ranks[KHG_SLOT] >> 16] + (ranks[KHG_SLOT] & 0xFFFFu)] = (uint8_t)tid; }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        const uint32_t t = sh.perm[tid];  // this thread's subsequence
        // Start and count come from the walk table -- from the map kernel, or, for a file with an index, from the
        // file, which may say anything: a subsequence holds at most one codeword per bit from its start offset on,
        // and a codeword that does not begin inside the subsequence is not this thread's, whatever the count says
        // (so the cursor never leaves the staged span and no thread decodes more than its 256 bits).
        const uint32_t off = sh.pre[t] >> 24;
        uint32_t cnt = (sh.pre[t + 1] & 0xFFFFFFu) - (sh.pre[t] & 0xFFFFFFu);
        if (off < KHG_SUB_BITS && cnt > KHG_SUB_BITS - off) cnt = KHG_SUB_BITS - off;
        if (off < KHG_SUB_BITS && cnt) {
            const uint32_t o = a.blk[blk].out + (sh.pre[t] & 0xFFFFFFu);
            const uint32_t pos = blk_begin + t * KHG_SUB_BITS + off;
            const uint32_t sub_end = blk_begin + (t + 1) * KHG_SUB_BITS;
            const uint32_t lim = a.n[p];
            const uint32_t todo = o >= lim ? 0 : (cnt < lim - o ? cnt : lim - o);  // never store past the plane
            int16_t* const base = a.sym[p];  // 16-byte aligned (workspace)
            int16_t* q = base + o;
            const uint32_t g_lo = (o + 7) >> 3, g_hi = (o + todo) >> 3;  // the groups of eight that are all this thread's
            const uint32_t ng = g_hi > g_lo ? g_hi - g_lo : 0;
            const uint32_t fill1 = onebit >= 0 ? (uint32_t)(uint16_t)(int16_t)onesym : 0u;
            const uint64_t fill4 = 0x0001000100010001ull * fill1;
            uint64_t grp[2] = {fill4, fill4};
            uint32_t gcur = 0xFFFFFFFFu;  // the group in grp; none yet
            // Each trip steps over a run of the 1-bit code (possibly empty; already filled in) and then takes one
            // other codeword, so the lanes of a wave stay in step through "zeros, then a value".  A trip has no
            // branch but the one for codes beyond the first-level table: it reads the 96 stream bits at its
            // position anew (one address, see khg_stage_slot; enough for a run of 32 and a table lookup behind
            // it) instead of topping up a window under a condition.
            const uint32_t lend = sub_end - bits.bias, lnb = bits.nbits - bits.bias;
            const uint32_t runx = onebit > 0 ? 0xFFFFFFFFu : 0u;  // the run's bits read as zeros
            const uint32_t rmax = onebit >= 0 ? 32u : 0u;
            uint32_t lpos = pos - bits.bias;  // position in the staged span
            uint32_t i = 0;
            bool go = todo != 0;
            while (go) {  // one exit, at the bottom
                const uint32_t w = lpos >> 5, sh_ = lpos & 31, sl = khg_stage_slot(w);
                const uint32_t x0 = bits.stage[sl], x1 = bits.stage[sl + 1], x2 = bits.stage[sl + 2];
                const uint32_t t0 = (uint32_t)(((((uint64_t)x0 << 32) | x1) << sh_) >> 32);
                const uint32_t t1 = (uint32_t)(((((uint64_t)x1 << 32) | x2) << sh_) >> 32);
                uint32_t rl = todo - i < lend - lpos ? todo - i : lend - lpos;  // > 0 here
                rl = rl < rmax ? rl : rmax;
                const uint32_t zx = t0 ^ runx;
                uint32_t r = zx ? (uint32_t)__builtin_clz(zx) : 32u;
                r = r < rl ? r : rl;  // 0..32
                const uint32_t top = (uint32_t)(((((uint64_t)t0 << 32) | t1) << r) >> 32);
                const uint32_t e = sh.lut[top >> (32 - KHG_LUT_BITS)];
                i += r;
                lpos += r;
                const bool want = r < 32 && i < todo && lpos < lend;  // a run cut at 32 may simply go on
                uint32_t len = (e >> 16) & 0xFF;
                int sym = (int)(int16_t)(e & 0xFFFF);
                bool whole = true;
                if (want && !(e & 0x80000000u)) {  // longer than the first-level table
                    len = khg_decode_one(bits, sh.lut, lut2, left, right, tsym, lpos + bits.bias, &sym);
                    whole = len != 0;
                    // The symbol may come from memory: it is waited for here, on the rare path.  Left pending, it
                    // is waited for where the paths meet, and that wait also covers the previous trip's store.
                    KHG_OPAQUE(sym);
                }
                const bool put = want && whole && lpos + len <= lnb;  // else the stream broke off
                // Where the symbol goes: a lane's single store opens a cache line of its own, and at a mild
                // preset most symbols are single stores (they cost as much as everything else in this kernel).
                // The aligned groups of four symbols that lie wholly inside this thread's stretch are therefore
                // put together in a register -- starting from what the fill left there -- and stored as 8 bytes
                // when the thread moves on to another group; the symbols of the ragged groups at its two ends,
                // which it shares with its neighbours, go out one by one.
                const uint32_t idx = o + i, g = idx >> 3;
                const bool own = want && g - g_lo < ng;
                if (want && !own) q[i] = (int16_t)sym;
                const bool fresh = own && g != gcur;
                if (fresh && gcur != 0xFFFFFFFFu) __builtin_memcpy(base + 8 * (size_t)gcur, grp, 16);
                grp[0] = fresh ? fill4 : grp[0];
                grp[1] = fresh ? fill4 : grp[1];
                gcur = fresh ? g : gcur;
                const uint64_t ins = (uint64_t)(((uint32_t)sym ^ fill1) & 0xFFFFu) << (16 * (idx & 3));
                grp[0] ^= own && !(idx & 4) ? ins : 0ull;
                grp[1] ^= own && (idx & 4) ? ins : 0ull;
                i += put ? 1u : 0u;
                lpos += put ? len : 0u;
                go = (put || !want) && i < todo && lpos < lend;
            }
            if (gcur != 0xFFFFFFFFu) __builtin_memcpy(base + 8 * (size_t)gcur, grp, 16);
        }
#ifdef KHG_TIMING
        if ((tid & 63) == 0) {
            a.blk[blk].pad_[2] = (uint32_t)tm0;
            khg_atomic_max_i((int*)&a.blk[blk].pad_[3], (int)((uint32_t)wall_clock64() - (uint32_t)tm0));
        }
#endif
    }
}
