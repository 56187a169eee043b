// khg_index.h -- an optional seek index behind a .comp file ("format 2").  The reference's format (Writer.h:9-69) has
// no synchronisation points: a decoder that wants to start in the middle of a stream must first find out where
// codewords begin (khg_dec_map_body / khg_dec_chain_body do that, exactly, at about a third of the decoder's work).
// Format 2 keeps the file of format 1 byte for byte and appends what those two kernels would work out: for every
// KHG_SUB_BITS bits of each stream, where the first codeword that starts in them lies and which symbol it is.  A
// reader of format 1 (the reference's ReadAll) stops before the index and never sees it; the indexed decoder skips the
// two kernels.  The index also carries both dimensions as they are, and stream lengths in fields 64 bits wide -- room
// for streams beyond the 36-byte header's int lengths (Type.h:5), which this encoder and decoder do not produce or
// accept: the high words are written as zero and anything else is refused.  It has no counterpart in the reference.
//
// Layout, little-endian 32-bit words unless noted, from the first multiple of 8 at or behind the end of the format-1
// file (the gap holds zeros):
//   [0] magic "KHI2"   [1] version (1)   [2] bits per entry (KHG_SUB_BITS)   [3] rows   [4] cols   [5] preset
//   [6..11]  bits of the Y, Cb, Cr streams without their pad bits, as (low, high) pairs
//   [12..14] symbols each stream holds (0 for an empty stream: a plane of one value, which decodes to zeros)
//   [15..17] entries of each plane = ceil(bits / bits per entry)
//   first[]  one word per entry, Y's then Cb's then Cr's: raster index of the first symbol whose codeword starts in
//            the entry's bits; the plane's symbol count if none does (only possible for a plane's last entry)
//   off[]    one BYTE per entry, same order: that codeword's first bit, counted from the entry's first bit; 255 if
//            none; zero bytes up to a multiple of 4
//   [bytes of the whole index, these two words included]   [magic "2IHK"]
#pragma once
#include "khg_decode.h"
#include "khg_huffman.h"

enum {
    KHG_IDX_MAGIC = 0x3249484B,      // "KHI2"
    KHG_IDX_MAGIC_END = 0x4B484932,  // "2IHK"
    KHG_IDX_VERSION = 1,
    KHG_IDX_HEAD_WORDS = 18,
    KHG_IDX_NONE = 255
};

KHG_HOST_DEV uint64_t khg_idx_begin(uint64_t v1_len) { return (v1_len + 7) & ~7ull; }
KHG_HOST_DEV uint64_t khg_idx_bytes(uint64_t entries)
{
    return 4ull * KHG_IDX_HEAD_WORDS + 4 * entries + ((entries + 3) & ~3ull) + 8;
}
KHG_HOST_DEV uint32_t khg_idx_entries(uint64_t bits) { return (uint32_t)((bits + KHG_SUB_BITS - 1) / KHG_SUB_BITS); }

KHG_DEV void khg_idx_store32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }  // p is 4-byte aligned
KHG_DEV uint32_t khg_idx_load32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// ------------------------------------------------------------------------------------------------ encoder
// Where the index of the file that the single-pass packer and khg_meta_body have just written goes.
struct KhgIdxPlace {
    uint64_t begin;         // byte offset of the index in the output
    uint64_t bits[3];       // unpadded stream bits
    uint32_t entries[3];
    uint64_t bytes;         // of the whole index
    bool fits;
};
KHG_DEV KhgIdxPlace khg_idx_place(const KhgEncArgs& a, uint64_t v1_len)
{
    KhgIdxPlace pl;
    uint64_t total = 0;
    for (int p = 0; p < 3; ++p) {
        pl.bits[p] = (uint64_t)(uint32_t)a.res->bits[p] - (uint64_t)(uint32_t)a.res->pads[p];
        pl.entries[p] = khg_idx_entries(pl.bits[p]);
        total += pl.entries[p];
    }
    pl.begin = khg_idx_begin(v1_len);
    pl.bytes = khg_idx_bytes(total);
    pl.fits = pl.begin + pl.bytes <= a.cap;
    return pl;
}

struct KhgIdxSmem {
    uint64_t hot[KHG_HOT_SYMS];
    uint32_t wave[KHG_SUPER][4];
    uint16_t pairlen[256];  // the two code lengths of a pair of symbols in -8 .. 7 (index as in khg_pair_entry); 0: no entry
};

// One workgroup per entry of the packer's chain (khg_chain_entry: 8 chunks of 2048 symbols), after khg_packscan_body
// and khg_meta_body have finished: the bit position of every symbol is worked out once more (code lengths, the same
// scans), and the symbol whose codeword reaches or crosses a multiple of KHG_SUB_BITS announces its successor as the
// first of that entry.  v1_len: the length khg_meta_body stored (a.res->total_len is rewritten by khg_index_meta_body
// only after this launch).
KHG_DEV void khg_index_body(const KhgEncArgs& a, uint32_t t, KhgIdxSmem& sh)
{
    if (a.res->status != 0 || !a.res->ok_to_pack) return;
    const KhgChainEntry en = khg_chain_entry(a, t);
    if (en.pad) return;
    const int p = en.plane;
    const KhgIdxPlace pl = khg_idx_place(a, a.res->total_len);
    if (!pl.fits) return;  // khg_index_meta_body reports it
    const uint64_t* ctab = a.codetab + (size_t)p * KHG_NBINS;
    const uint32_t t0 = t - en.sym0 / (uint32_t)(KHG_SUPER * KHG_CHUNK);  // the plane's first chain entry
    const uint64_t plane_start = t0 ? khg_chain_load(khg_chain_pre(a, t0 - 1)) & ~KHG_CHAIN_VALID : 36 * 8;
    const uint64_t entry_start = t ? khg_chain_load(khg_chain_pre(a, t - 1)) & ~KHG_CHAIN_VALID : 36 * 8;
    const uint64_t rel0 = entry_start - plane_start;  // the entry's first bit, counted in the plane's stream
    uint32_t before = 0;                              // index entries of the planes before this one
    for (int q = 0; q < p; ++q) before += pl.entries[q];
    uint8_t* first = a.out + pl.begin + 4 * KHG_IDX_HEAD_WORDS + 4 * (size_t)before;
    uint8_t* off = a.out + pl.begin + 4 * KHG_IDX_HEAD_WORDS +
                   4 * ((size_t)pl.entries[0] + pl.entries[1] + pl.entries[2]) + before;
    const uint32_t nent = pl.entries[p], nsym = a.n[p];

    uint64_t lens[KHG_SUPER][KHG_NSLOT(256)];  // code lengths of a thread's eight symbols, a byte each
    uint32_t tbs[KHG_SUPER][KHG_NSLOT(256)], excl[KHG_SUPER][KHG_NSLOT(256)];
    int nvs[KHG_SUPER][KHG_NSLOT(256)];
    KHG_FOR_THREADS(256)
    {
        sh.hot[tid] = ctab[KHG_SYM_BIAS - KHG_HOT_SYMS / 2 + tid];
        const int q0 = (int)(((uint32_t)tid & 15u) ^ 8u) - 8, q1 = (int)((((uint32_t)tid >> 4) & 15u) ^ 8u) - 8;
        const uint32_t l0 = (uint32_t)(ctab[q0 + KHG_SYM_BIAS] >> 56), l1 = (uint32_t)(ctab[q1 + KHG_SYM_BIAS] >> 56);
        sh.pairlen[tid] = (l0 && l1) ? (uint16_t)(l0 | (l1 << 8)) : (uint16_t)0;
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        for (int j = 0; j < KHG_SUPER; ++j) {
            int16_t v[8];
            const uint32_t s0 = en.sym0 + (uint32_t)j * KHG_CHUNK + (uint32_t)tid * 8;
            const int nv = s0 >= en.sym1 ? 0 : khg_load8(a.sym[p], s0, en.sym1, v);
            uint32_t tb = 0;
            uint64_t lv = 0;
            // two symbols per lookup, as in the packer; `odd`: a symbol outside -8 .. 7, a pair without an entry, a
            // ragged end
            uint32_t w[4] = {0, 0, 0, 0};
            if (nv == 8) __builtin_memcpy(w, v, 16);
            uint32_t odd = nv == 8 ? 0u : 1u;
            KHG_UNROLL
            for (int q = 0; q < 4; ++q) {
                odd |= (w[q] ^ (w[q] << 1)) & 0xFFF0FFF0u;
                const uint32_t e = sh.pairlen[(w[q] & 15u) | ((w[q] >> 12) & 0xF0u)];
                odd |= e == 0;
                lv |= (uint64_t)e << (16 * q);
                tb += (e & 255u) + (e >> 8);
            }
            if (odd) {
                tb = 0;
                lv = 0;
                for (int i = 0; i < 8; ++i) {
                    if (i < nv) {
                        const uint32_t len = (uint32_t)(khg_pack_code_lds(sh.hot, ctab, v[i]) >> 56);
                        lv |= (uint64_t)len << (8 * i);
                        tb += len;
                    }
                }
            }
            lens[j][KHG_SLOT] = lv;
            tbs[j][KHG_SLOT] = tb;
            nvs[j][KHG_SLOT] = nv;
        }
    }
    uint32_t tot[KHG_SUPER], sub_base[KHG_SUPER], total = 0;
    KHG_WG_EXCL_SCAN_N(KHG_SUPER, excl, tbs, sh.wave, tot)
    for (int j = 0; j < KHG_SUPER; ++j) {
        sub_base[j] = total;
        total += tot[j];
    }
    KHG_FOR_THREADS(256)
    {
        if (en.sym0 == 0 && tid == 0 && nent) {  // the plane's first symbol opens entry 0
            khg_idx_store32(first, 0);
            off[0] = 0;
        }
        for (int j = 0; j < KHG_SUPER; ++j) {
            uint64_t b = rel0 + sub_base[j] + excl[j][KHG_SLOT];
            const uint32_t s0 = en.sym0 + (uint32_t)j * KHG_CHUNK + (uint32_t)tid * 8;
            const uint64_t lv = lens[j][KHG_SLOT];
            // nearly always none of the thread's codewords reaches a multiple of KHG_SUB_BITS
            if (b / KHG_SUB_BITS == (b + tbs[j][KHG_SLOT]) / KHG_SUB_BITS) continue;
            for (int i = 0; i < nvs[j][KHG_SLOT]; ++i) {
                const uint64_t e = b + ((lv >> (8 * i)) & 255u);
                const uint64_t s = e / KHG_SUB_BITS;
                if (s != b / KHG_SUB_BITS && s < nent) {  // this codeword reaches the first bit of entry s, or crosses it
                    const uint32_t next = s0 + (uint32_t)i + 1;
                    if (next < nsym) {
                        khg_idx_store32(first + 4 * s, next);
                        off[s] = (uint8_t)(e - s * KHG_SUB_BITS);
                    } else {  // the plane's last codeword ends inside entry s: nothing starts there
                        khg_idx_store32(first + 4 * s, nsym);
                        off[s] = KHG_IDX_NONE;
                    }
                }
                b = e;
            }
        }
    }
}

// The index's fixed words and the file's new length.  One thread (of a workgroup of 64), after khg_index_body.
KHG_DEV void khg_index_meta_body(const KhgEncArgs& a)
{
    KHG_FOR_THREADS(64)
    {
        if (tid == 0 && a.res->status == 0 && a.res->ok_to_pack) {
            const uint64_t v1_len = a.res->total_len;
            const KhgIdxPlace pl = khg_idx_place(a, v1_len);
            if (!pl.fits) {
                khg_set_status(&a.res->status, -2);
            } else {
                uint8_t* o = a.out + pl.begin;
                const uint32_t ne = pl.entries[0] + pl.entries[1] + pl.entries[2];
                for (uint64_t i = v1_len; i < pl.begin; ++i) a.out[i] = 0;
                const uint32_t w[KHG_IDX_HEAD_WORDS] = {
                    (uint32_t)KHG_IDX_MAGIC, KHG_IDX_VERSION, KHG_SUB_BITS, (uint32_t)a.rows, (uint32_t)a.cols, (uint32_t)a.lvl,
                    (uint32_t)pl.bits[0], (uint32_t)(pl.bits[0] >> 32), (uint32_t)pl.bits[1], (uint32_t)(pl.bits[1] >> 32),
                    (uint32_t)pl.bits[2], (uint32_t)(pl.bits[2] >> 32), pl.bits[0] ? a.n[0] : 0u, pl.bits[1] ? a.n[1] : 0u,
                    pl.bits[2] ? a.n[2] : 0u, pl.entries[0], pl.entries[1], pl.entries[2]};
                for (int i = 0; i < KHG_IDX_HEAD_WORDS; ++i) khg_idx_store32(o + 4 * i, w[i]);
                uint8_t* tail = o + 4 * KHG_IDX_HEAD_WORDS + 4 * (size_t)ne;
                for (uint32_t i = ne; i < ((ne + 3) & ~3u); ++i) tail[i] = 0;
                uint8_t* end = tail + ((ne + 3) & ~3u);
                khg_idx_store32(end, (uint32_t)pl.bytes);
                khg_idx_store32(end + 4, (uint32_t)KHG_IDX_MAGIC_END);
                a.res->total_len = pl.begin + pl.bytes;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ decoder
// The index of the file in a.comp, if it has a sound one: checked against the 36-byte header (parsed into a.hdr by
// khg_dec_header_body) and the file's length.
struct KhgIdxView {
    bool ok;
    const uint8_t* first[3];
    const uint8_t* off[3];
    uint32_t entries[3];
    uint32_t nsym[3];
};
KHG_DEV KhgIdxView khg_idx_view(const KhgDecArgs& a)
{
    KhgIdxView v;
    v.ok = false;
    const KhgDecHeader& h = *a.hdr;
    const uint64_t v1_len = (uint64_t)h.table_byte[2] + 8ull * (uint64_t)h.k[2];
    const uint64_t begin = khg_idx_begin(v1_len);
    if (begin + khg_idx_bytes(0) > a.len) return v;
    const uint8_t* o = a.comp + begin;
    uint32_t w[KHG_IDX_HEAD_WORDS];
    for (int i = 0; i < KHG_IDX_HEAD_WORDS; ++i) w[i] = khg_idx_load32(o + 4 * i);
    if (w[0] != (uint32_t)KHG_IDX_MAGIC || w[1] != KHG_IDX_VERSION || w[2] != KHG_SUB_BITS) return v;
    if (w[3] != (uint32_t)h.rows || w[4] != (uint32_t)h.cols || w[5] != (uint32_t)h.lvl) return v;
    uint64_t ne = 0;
    for (int p = 0; p < 3; ++p) {
        if (w[6 + 2 * p] != (uint32_t)h.nbits[p] || w[7 + 2 * p] != 0) return v;  // this decoder takes 31-bit streams
        v.nsym[p] = w[12 + p];
        v.entries[p] = w[15 + p];
        if (v.nsym[p] > a.n[p] || v.entries[p] != khg_idx_entries((uint32_t)h.nbits[p])) return v;
        ne += v.entries[p];
    }
    const uint64_t bytes = khg_idx_bytes(ne);
    if (begin + bytes > a.len) return v;
    const uint8_t* end = o + bytes - 8;
    if (khg_idx_load32(end) != (uint32_t)bytes || khg_idx_load32(end + 4) != (uint32_t)KHG_IDX_MAGIC_END) return v;
    const uint8_t* f = o + 4 * KHG_IDX_HEAD_WORDS;
    const uint8_t* b = f + 4 * ne;
    for (int p = 0; p < 3; ++p) {
        v.first[p] = f;
        v.off[p] = b;
        f += 4 * (size_t)v.entries[p];
        b += v.entries[p];
    }
    v.ok = true;
    return v;
}

// Stands in for khg_dec_map_body and khg_dec_chain_body when the file brings an index: fills the workgroup records and
// the walk table that khg_dec_write_body reads (column 0: every workgroup is "entered at offset 0" of its table).
// blk < a.nblocks: one workgroup of the write launch; blk = a.nblocks + p: what the chain does once per plane (the
// count of decoded symbols, zeros behind them, Huffman.h:35) and, for p = 0, the verdict on the index.
KHG_DEV void khg_dec_index_body(const KhgDecArgs& a, uint32_t blk)
{
    if (!a.hdr->ok) return;
    const KhgIdxView v = khg_idx_view(a);
    if (blk >= a.nblocks) {
        const int p = (int)(blk - a.nblocks);
        KHG_FOR_THREADS(256)
        {
            if (!v.ok) {
                if (tid == 0 && p == 0) khg_set_status(&a.res->status, -3);
            } else {
                if (tid == 0) a.res->bits[p] = (int32_t)v.nsym[p];
                for (uint32_t i = v.nsym[p] + (uint32_t)tid; i < a.n[p]; i += 256) a.sym[p][i] = 0;
            }
        }
        return;
    }
    if (!v.ok) return;
    const int p = khg_block_plane(a, blk);
    const uint32_t lb = blk - a.blk0[p];
    const uint32_t e0 = lb * KHG_DEC_BLOCK;
    if (e0 >= v.entries[p]) return;
    const uint32_t M = (uint32_t)a.hdr->maxlen[p];
    if (M == 0 || M > KHG_MAXOFF) return;
    const uint32_t nsym = v.nsym[p];
    uint32_t out = khg_idx_load32(v.first[p] + 4 * (size_t)e0);
    if (out > nsym) out = nsym;
    uint32_t* gw = a.walk + (size_t)blk * KHG_MAXOFF * KHG_DEC_BLOCK;
    KHG_FOR_THREADS(256)
    {
        const uint32_t e = e0 + (uint32_t)tid;
        uint32_t o = KHG_OFF_NONE, pre = 0;
        uint32_t f = nsym;
        if (e < v.entries[p]) {
            f = khg_idx_load32(v.first[p] + 4 * (size_t)e);
            const uint32_t ob = v.off[p][e];
            if (ob < M && ob < KHG_OFF_TEMP) o = ob;  // a first codeword starts within the longest code's length
        }
        if (f > nsym) f = nsym;
        pre = f >= out ? f - out : 0;
        if (pre > 0xFFFFFFu) pre = 0xFFFFFFu;
        gw[(uint32_t)tid * M] = (o << 24) | pre;
        if (tid == 0) {
            uint32_t nx = nsym;  // the first symbol of the next workgroup
            if (e0 + KHG_DEC_BLOCK < v.entries[p]) nx = khg_idx_load32(v.first[p] + 4 * (size_t)(e0 + KHG_DEC_BLOCK));
            if (nx > nsym) nx = nsym;
            uint32_t tot = nx >= out ? nx - out : 0;
            if (tot > 0xFFFFFFu) tot = 0xFFFFFFu;
            a.blk[blk].map[0] = ((uint32_t)KHG_OFF_NONE << 24) | tot;
            a.blk[blk].start_off = 0;
            a.blk[blk].out = out;
        }
    }
}
