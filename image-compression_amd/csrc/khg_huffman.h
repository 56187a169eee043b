// khg_huffman.h -- symbol table, Huffman tree/codes, and the bit-packing encoder.
//
//   khg_table_body : dense (count, first raster index) histogram -> (count, symbol) table in
//                    first-appearance order (Huffman.h:110-139), then the tree and codes
//   khg_tree_build : Huffman.h:87-108 with std::priority_queue's heap (libstdc++
//                    bits/stl_heap.h __push_heap / __adjust_heap) reproduced move for move
//   khg_lens_body  : bits per 2048-symbol chunk
//   khg_scan_body  : chunk bit offsets, pads, header (Writer.h:9-23) and tables (Writer.h:34-41)
//   khg_pack_body  : codes concatenated in raster order, MSB first (Huffman.h:76-82, Writer.h:25-32)
#pragma once
#include "khg_common.h"

enum { KHG_CHAIN_HEAD = 16 };  // words of KhgEncArgs::chain before the per-entry state ([0]: next ticket, alone in its 128-byte line)

enum {
    KHG_TREE_LDS_K = 4096,  // tables up to this size build their tree in LDS
    KHG_TABLE_LDS_WORDS = 2 * KHG_NBINS  // 128 KiB: entry list (first, bin), later heap + node arrays
};

struct KhgTreeOut {
    int32_t* left;     // [2*KHG_NBINS] children of internal nodes, -1 for leaves
    int32_t* right;
    int32_t* root;     // [1]
};

struct KhgTableArgs {
    const uint32_t* hist_cnt;    // [3][KHG_NBINS]
    const uint32_t* hist_first;  // [3][KHG_NBINS]
    int32_t* tab_cnt;            // [3][KHG_NBINS] table in first-appearance order
    int32_t* tab_sym;            // [3][KHG_NBINS]
    int32_t* tab_k;              // [3]
    uint64_t* codetab;           // [3][KHG_NBINS]: (len << 56) | code, indexed by symbol + 8192
    uint32_t* big_scratch;       // [3][5 * 2 * KHG_NBINS] words, used when K > KHG_TREE_LDS_K
    int32_t* tree_left;          // [3][2*KHG_NBINS]
    int32_t* tree_right;         // [3][2*KHG_NBINS]
    int32_t* tree_root;          // [3]
    int* status;
};

struct KhgTableSmem {
    alignas(16) uint32_t w[KHG_TABLE_LDS_WORDS];
    uint32_t k;
    int32_t root;
};

struct alignas(16) KhgPair {
    uint64_t a, b;
};

// Heap entry: freq in the high word, node id in the low word.  Compare (Huffman.h:25-31) looks at freq only.
KHG_DEV bool khg_heap_comp(uint64_t l, uint64_t r) { return (uint32_t)(l >> 32) > (uint32_t)(r >> 32); }

// bits/stl_heap.h __push_heap(first, holeIndex, topIndex, value, comp)
KHG_DEV void khg_heap_sift_up(uint64_t* heap, int hole, int top, uint64_t value)
{
    int parent = (hole - 1) / 2;
    while (hole > top && khg_heap_comp(heap[parent], value)) {
        heap[hole] = heap[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    heap[hole] = value;
}

// priority_queue::top() + pop(): pop_heap (__pop_heap + __adjust_heap) then pop_back
KHG_DEV uint64_t khg_heap_pop(uint64_t* heap, int& n)
{
    uint64_t top = heap[0];
    if (n > 1) {
        int len = n - 1;
        uint64_t value = heap[len];
        heap[len] = top;
        int hole = 0, child = 0;
        uint64_t up = 0;  // the entry moved last, i.e. what the hole's parent holds now
        // heap[] starts 8 bytes past a 16-byte boundary, so the sibling pair (2i+1, 2i+2) is one aligned 16-byte load
        const int lim = (len - 1) / 2;
        // two levels per memory round trip while both are complete: the children of `child` and the
        // children of both of them are fetched together, then the two picks are made from registers
        while (2 * child + 2 < lim) {
            const int c2 = 2 * (child + 1);
            const KhgPair pr = *(const KhgPair*)&heap[c2 - 1];
            const KhgPair ga = *(const KhgPair*)&heap[2 * c2 - 1];  // children of c2 - 1
            const KhgPair gb = *(const KhgPair*)&heap[2 * c2 + 1];  // children of c2
            const bool la = khg_heap_comp(pr.b, pr.a);              // step 1 takes the left sibling
            const uint64_t pick = la ? pr.a : pr.b;
            const KhgPair g = la ? ga : gb;
            int c4 = 2 * ((la ? c2 - 1 : c2) + 1);
            uint64_t pick2 = g.b;
            if (khg_heap_comp(g.b, g.a)) {
                c4--;
                pick2 = g.a;
            }
            heap[hole] = pick;
            heap[la ? c2 - 1 : c2] = pick2;
            up = pick2;
            hole = c4;
            child = c4;
        }
        while (child < lim) {
            child = 2 * (child + 1);
            const KhgPair pr = *(const KhgPair*)&heap[child - 1];
            uint64_t pick = pr.b;
            if (khg_heap_comp(pr.b, pr.a)) {
                child--;
                pick = pr.a;
            }
            heap[hole] = pick;
            up = pick;
            hole = child;
        }
        if ((len & 1) == 0 && child == (len - 2) / 2) {
            child = 2 * (child + 1);
            up = heap[child - 1];
            heap[hole] = up;
            hole = child - 1;
        }
        // __push_heap from the hole: its first comparison is with the entry just moved, still in a register; the
        // re-inserted entry nearly always stays where the hole is
        if (hole > 0 && khg_heap_comp(up, value)) khg_heap_sift_up(heap, hole, 0, value);
        else heap[hole] = value;
    }
    n--;
    return top;
}

// One thread.  Leaves are nodes 0..k-1 in table order; internal nodes k..2k-2 in creation order.
// On entry nfreq[0..k) holds the table counts and left/right[0..k) are already -1 (both filled by all
// threads beforehand, so this sequential part touches global memory with stores only).
// parent[] holds the parent id with bit 31 set when the node is the right ('1') child.
KHG_DEV int khg_tree_build(int k, uint64_t* heap, uint32_t* nfreq, uint32_t* parent, int32_t* left, int32_t* right)
{
    int n = 0;
#ifdef KHG_TIMING
    long long t0 = wall_clock64();
#endif
    uint32_t fnext = nfreq[0];
    for (int i = 0; i < k; ++i) {  // Huffman.h:91-94; the next count is fetched while this one sifts up
        const uint32_t f = fnext;
        fnext = nfreq[i + 1 < k ? i + 1 : i];
        khg_heap_sift_up(heap, n++, 0, ((uint64_t)f << 32) | (uint32_t)i);
    }
#ifdef KHG_TIMING
    long long t1 = wall_clock64();
#endif
    int next = k;
    while (n != 1) {  // Huffman.h:96-105
        const uint64_t er = khg_heap_pop(heap, n), el = khg_heap_pop(heap, n);  // entries carry their node's count
        const uint32_t r = (uint32_t)er, l = (uint32_t)el;
        uint32_t sum = (uint32_t)(int)((uint32_t)(el >> 32) + (uint32_t)(er >> 32));
        nfreq[next] = sum;
        parent[l] = (uint32_t)next;
        parent[r] = (uint32_t)next | 0x80000000u;
        left[next] = (int32_t)l;
        right[next] = (int32_t)r;
        khg_heap_sift_up(heap, n++, 0, ((uint64_t)sum << 32) | (uint32_t)next);
        next++;
    }
#ifdef KHG_TIMING
    long long t2 = wall_clock64();
    if (k > 100) printf("tree k=%d pushes %lld merges %lld (100MHz ticks)\n", k, t1 - t0, t2 - t1);
#endif
    return (int)(uint32_t)heap[0];
}

// All threads: what khg_tree_build expects to find.
KHG_DEV void khg_tree_prefill(int tid, int nthreads, const int32_t* tab_cnt, int k, uint32_t* nfreq, int32_t* left,
                              int32_t* right)
{
    for (int i = tid; i < k; i += nthreads) {
        nfreq[i] = (uint32_t)tab_cnt[i];
        left[i] = -1;
        right[i] = -1;
    }
}

// Huffman.h:57-68 from the leaf upwards: the edge nearest the root is the first (most significant) bit.
KHG_DEV uint64_t khg_leaf_code(const uint32_t* parent, int leaf, int root, int* status)
{
    uint64_t code = 0;
    int len = 0;
    int node = leaf;
    while (node != root) {
        uint32_t p = parent[node];
        if (len < 64) code |= (uint64_t)(p >> 31) << len;
        len++;
        node = (int)(p & 0x7FFFFFFFu);
    }
    if (len > KHG_MAX_CODE_BITS) {
        khg_set_status(status, -4);
        len = 0;
        code = 0;
    }
    return ((uint64_t)len << 56) | code;
}

// bid = colour component 0..2.  256 threads.
KHG_DEV void khg_table_body(const KhgTableArgs& a, int bid, KhgTableSmem& sh)
{
    const uint32_t* hc = a.hist_cnt + (size_t)bid * KHG_NBINS;
    const uint32_t* hf = a.hist_first + (size_t)bid * KHG_NBINS;
    int32_t* tcnt = a.tab_cnt + (size_t)bid * KHG_NBINS;
    int32_t* tsym = a.tab_sym + (size_t)bid * KHG_NBINS;
    uint64_t* ctab = a.codetab + (size_t)bid * KHG_NBINS;
    int32_t* left = a.tree_left + (size_t)bid * 2 * KHG_NBINS;
    int32_t* right = a.tree_right + (size_t)bid * 2 * KHG_NBINS;
    uint32_t* efirst = sh.w;             // [KHG_NBINS]
    uint32_t* ebin = sh.w + KHG_NBINS;   // [KHG_NBINS]

    KHG_FOR_THREADS(256)
    {
        if (tid == 0) sh.k = 0;
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        // 64 consecutive bins per thread, read 16 at a time so the loads overlap
        for (int g = 0; g < 4; ++g) {
            const int s0 = tid * 64 + g * 16;
            uint32_t c[16];
            KHG_UNROLL
            for (int j = 0; j < 16; ++j) c[j] = hc[s0 + j];
            KHG_UNROLL
            for (int j = 0; j < 16; ++j)
                if (c[j]) {
                    uint32_t i = khg_atomic_add(&sh.k, 1u);
                    efirst[i] = hf[s0 + j];
                    ebin[i] = (uint32_t)(s0 + j);
                }
        }
    }
    KHG_BARRIER();
    const int k = (int)sh.k;
    KHG_FOR_THREADS(256)
    {
        // rank by first raster index = position in the reference's first-appearance table
        for (int i = tid; i < k; i += 256) {
            uint32_t f = efirst[i];
            int rank = 0;
            for (int j = 0; j < k; ++j) rank += efirst[j] < f;
            uint32_t s = ebin[i];
            tcnt[rank] = (int32_t)hc[s];
            tsym[rank] = (int32_t)s - KHG_SYM_BIAS;
        }
        if (tid == 0) a.tab_k[bid] = k;
    }
    KHG_BARRIER();  // table complete (global), entry list dead: LDS is reused below

    // Two copies of the sequential part so the compiler sees which address space each one walks
    // (LDS for ordinary tables, global scratch for tables above KHG_TREE_LDS_K entries).
    if (k <= KHG_TREE_LDS_K) {
        uint64_t* heap = (uint64_t*)sh.w + 1;               // [4096] u64, element 0 at byte 8 (see khg_heap_pop)
        uint32_t* nfreq = sh.w + 2 * KHG_TREE_LDS_K + 4;    // [8192]
        uint32_t* parent = sh.w + 4 * KHG_TREE_LDS_K + 4;   // [8192]
        KHG_FOR_THREADS(256) { khg_tree_prefill(tid, 256, tcnt, k, nfreq, left, right); }
        KHG_BARRIER();
        KHG_FOR_THREADS(256)
        {
            if (tid == 0) {
                if (k == 0) {
                    khg_set_status(a.status, -3);
                    sh.root = 0;
                } else {
                    sh.root = khg_tree_build(k, heap, nfreq, parent, left, right);
                }
                a.tree_root[bid] = sh.root;
            }
        }
        KHG_BARRIER();
        KHG_FOR_THREADS(256)
        {
            const int root = sh.root;
            for (int i = tid; i < k; i += 256) ctab[tsym[i] + KHG_SYM_BIAS] = khg_leaf_code(parent, i, root, a.status);
        }
    } else {
        uint32_t* g = a.big_scratch + (size_t)bid * 10 * KHG_NBINS;
        uint64_t* heap = (uint64_t*)g + 1;                  // [16384] u64, element 0 at byte 8
        uint32_t* nfreq = g + 2 * KHG_NBINS + 4;            // [32768]
        uint32_t* parent = g + 4 * KHG_NBINS + 4;           // [32768]
        KHG_FOR_THREADS(256) { khg_tree_prefill(tid, 256, tcnt, k, nfreq, left, right); }
        KHG_BARRIER();
        KHG_FOR_THREADS(256)
        {
            if (tid == 0) {
                sh.root = khg_tree_build(k, heap, nfreq, parent, left, right);
                a.tree_root[bid] = sh.root;
            }
        }
        KHG_BARRIER();
        KHG_FOR_THREADS(256)
        {
            const int root = sh.root;
            for (int i = tid; i < k; i += 256) ctab[tsym[i] + KHG_SYM_BIAS] = khg_leaf_code(parent, i, root, a.status);
        }
    }
}

// ------------------------------------------------------------------ encoder

struct KhgEncArgs {
    const int16_t* sym[3];     // quantised planes, raster order
    uint32_t n[3];             // symbols per plane
    uint32_t chunk0[3];        // first chunk index of each plane
    // Row-tiled images (nranks > 1): a rank computed the symbols of its row pairs at each level, which in the
    // Mallat layout are, per plane of prow x pcol symbols (hh = prow/2, hw = pcol/2),
    //   left half of row r < hh   (level 2)           : rank of r within [0, hh/2) or [hh/2, hh), split evenly
    //   right half of row r < hh  (level 1, band 4)   : rank of row pair k = r
    //   whole row r >= hh         (level 1, bands 5|6): rank of row pair k = r - hh
    // so the raster is cut into these runs, each run into chunks of its own, and a rank lens/packs its runs only.
    int rank, nranks;
    int prow[3], pcol[3];
    int write_meta;            // this rank writes the header and the tables
    uint32_t nchunks;          // total
    const uint64_t* codetab;   // [3][KHG_NBINS]
    uint32_t* chunk_bits;      // [nchunks]
    uint64_t* chunk_off;       // [nchunks] bit offset in the output file
    uint64_t* chain;           // state of the single-pass packer (khg_packscan_body)
    uint32_t chain_entries;    // its entries: superchunks of the three planes + 3 pad entries
    const int32_t* tab_cnt;    // [3][KHG_NBINS]
    const int32_t* tab_sym;
    const int32_t* tab_k;      // [3]
    uint8_t* out;              // .comp bytes (4-byte aligned)
    uint64_t cap;              // bytes available at out (multiple of 4)
    int rows, cols, lvl;
    KhgResult* res;
};

KHG_DEV int khg_chunk_plane(const KhgEncArgs& a, uint32_t chunk)
{
    return chunk >= a.chunk0[2] ? 2 : chunk >= a.chunk0[1] ? 1 : 0;
}

// Chunks of one run of `len` symbols.
KHG_DEV uint32_t khg_run_chunks(uint32_t len) { return (len + KHG_CHUNK - 1) / KHG_CHUNK; }

// A chunk of the symbol stream: plane, raster range [sym0, sym0 + nsym), and whether this rank owns it.
struct KhgChunk {
    int plane;
    uint32_t sym0, nsym;
    bool mine;
};

KHG_DEV KhgChunk khg_chunk_info(const KhgEncArgs& a, uint32_t chunk)
{
    KhgChunk c;
    c.plane = khg_chunk_plane(a, chunk);
    const uint32_t lc = chunk - a.chunk0[c.plane];
    if (a.nranks <= 1) {  // the plane's raster in chunks of KHG_CHUNK
        c.sym0 = lc * KHG_CHUNK;
        c.nsym = a.n[c.plane] - c.sym0 < KHG_CHUNK ? a.n[c.plane] - c.sym0 : KHG_CHUNK;
        c.mine = true;
        return c;
    }
    const uint32_t prow = (uint32_t)a.prow[c.plane], pcol = (uint32_t)a.pcol[c.plane], hh = prow / 2, hw = pcol / 2;
    const uint32_t ch = khg_run_chunks(hw), cw = khg_run_chunks(pcol);
    uint32_t run_sym0, run_len, j, owner;
    if (lc < hh * 2 * ch) {
        const uint32_t r = lc / (2 * ch), rest = lc - r * 2 * ch, right = rest >= ch;
        j = right ? rest - ch : rest;
        run_sym0 = r * pcol + (right ? hw : 0);
        run_len = hw;
        if (right) owner = r / (hh / (uint32_t)a.nranks);
        else owner = (r < hh / 2 ? r : r - hh / 2) / (hh / 2 / (uint32_t)a.nranks);
    } else {
        const uint32_t l2 = lc - hh * 2 * ch, k = l2 / cw;
        j = l2 - k * cw;
        run_sym0 = (hh + k) * pcol;
        run_len = pcol;
        owner = k / (hh / (uint32_t)a.nranks);
    }
    c.sym0 = run_sym0 + j * KHG_CHUNK;
    c.nsym = run_len - j * KHG_CHUNK < KHG_CHUNK ? run_len - j * KHG_CHUNK : KHG_CHUNK;
    c.mine = owner == (uint32_t)a.rank;
    return c;
}

// Symbols [s0, s0 + 8) of a plane that lie below `end`: one 16-byte load when all eight do (the address is a
// multiple of 8 bytes at least), element by element at the end of a run.  Returns how many are valid.
KHG_DEV int khg_load8(const int16_t* sym, uint32_t s0, uint32_t end, int16_t v[8])
{
    if (s0 + 8 <= end) {
        __builtin_memcpy(v, sym + s0, 16);
        return 8;
    }
    int n = 0;
    for (int i = 0; i < 8; ++i) {
        v[i] = 0;
        if (s0 + i < end) {
            v[i] = sym[s0 + i];
            n = i + 1;
        }
    }
    return n;
}

// Code table entries of the symbols -128 .. 127 (nearly every symbol) are fetched once per workgroup, coalesced,
// into LDS; the others are gathered from the global table.
enum { KHG_HOT_SYMS = 256 };

struct KhgLensSmem {
    uint32_t wave[4];
    uint8_t len8[KHG_HOT_SYMS];
};

// One chunk per workgroup, 8 symbols per thread.
KHG_DEV void khg_lens_body(const KhgEncArgs& a, uint32_t chunk, KhgLensSmem& sh)
{
    const KhgChunk ck = khg_chunk_info(a, chunk);
    if (!ck.mine) return;  // another rank's symbols: its entry of chunk_bits stays zero here
    const int p = ck.plane;
    const uint64_t* ctab = a.codetab + (size_t)p * KHG_NBINS;
    const uint32_t base = ck.sym0, end = ck.sym0 + ck.nsym;
    int16_t vs[KHG_NSLOT(256)][8];
    int nvs[KHG_NSLOT(256)];
    uint32_t bitsv[KHG_NSLOT(256)], excl[KHG_NSLOT(256)];
    KHG_FOR_THREADS(256)
    {
        sh.len8[tid] = (uint8_t)(ctab[KHG_SYM_BIAS - KHG_HOT_SYMS / 2 + tid] >> 56);
        nvs[KHG_SLOT] = khg_load8(a.sym[p], base + (uint32_t)tid * 8, end, vs[KHG_SLOT]);
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        uint32_t bits = 0;
        const int16_t* v = vs[KHG_SLOT];
        const int nv = nvs[KHG_SLOT];
        KHG_UNROLL
        for (int i = 0; i < 8; ++i) {
            if (i < nv) {
                const uint32_t h = (uint32_t)(v[i] + KHG_HOT_SYMS / 2);
                uint32_t len = sh.len8[h & (KHG_HOT_SYMS - 1)];
                if (h >= (uint32_t)KHG_HOT_SYMS) {
                    KHG_OPAQUE(len);  // the LDS load stays a load of its own (not one flat load of a selected address)
                    len = (uint32_t)(ctab[v[i] + KHG_SYM_BIAS] >> 56);
                }
                bits += len;
            }
        }
        bitsv[KHG_SLOT] = bits;
    }
    uint32_t total;
    KHG_WG_EXCL_SCAN(excl, bitsv, sh.wave, total)
    KHG_FOR_THREADS(256)
    {
        (void)excl[KHG_SLOT];
        if (tid == 0) a.chunk_bits[chunk] = total;
    }
}

struct KhgScanSmem {
    uint32_t wave[2][16];  // wave totals of the workgroup scan, alternating between the tiles of chunks
    uint64_t table_off;    // byte offset of the tables
    int ok;
};

KHG_DEV void khg_store_le32(uint8_t* p, int32_t v)
{
    uint32_t u = (uint32_t)v;
    p[0] = (uint8_t)u;
    p[1] = (uint8_t)(u >> 8);
    p[2] = (uint8_t)(u >> 16);
    p[3] = (uint8_t)(u >> 24);
}

// Single workgroup, 1024 threads.
KHG_DEV void khg_scan_body(const KhgEncArgs& a, KhgScanSmem& sh)
{
    // One pass over the chunk bit counts, plane after plane, 8192 chunks (eight consecutive ones per thread) at a
    // time: a workgroup scan gives every chunk its offset behind the bits counted so far, and a plane's padded length
    // is known when its last chunk is (so is where the next plane begins).  The packer ORs the words that chunks
    // share into the output: the caller hands over a zeroed buffer (khg_tile_pack_dev fills it before this launch).
    uint64_t off = 36 * 8;  // Writer.h:9-23: the streams start behind the 36-byte header
    bool ok = true;
    uint32_t it = 0;
    for (int p = 0; p < 3; ++p) {
        const uint32_t c_begin = a.chunk0[p], c_end = p < 2 ? a.chunk0[p + 1] : a.nchunks;
        uint64_t run = 0;  // the plane's bits so far
        // tiles begin at multiples of eight chunks, so a thread's eight counts are one aligned 32-byte load and its
        // eight offsets one aligned 64-byte store (chunks of the plane before, in the first tile, count as empty)
        for (uint32_t tb = c_begin & ~7u; tb < c_end; tb += 8192, ++it) {
            uint32_t v[KHG_NSLOT(1024)][8], sum[KHG_NSLOT(1024)], excl[KHG_NSLOT(1024)], total;
            KHG_FOR_THREADS(1024)
            {
                const uint32_t c = tb + (uint32_t)tid * 8;
                uint32_t t = 0;
                if (c >= c_begin && c + 8 <= c_end) {
                    __builtin_memcpy(v[KHG_SLOT], (const uint32_t*)KHG_ASSUME_ALIGNED(a.chunk_bits + c, 32), 32);
                } else {
                    KHG_UNROLL
                    for (int i = 0; i < 8; ++i) v[KHG_SLOT][i] = (c + i >= c_begin && c + i < c_end) ? a.chunk_bits[c + i] : 0;
                }
                KHG_UNROLL
                for (int i = 0; i < 8; ++i) t += v[KHG_SLOT][i];
                sum[KHG_SLOT] = t;
            }
            KHG_WG_EXCL_SCAN_NW(16, excl, sum, sh.wave[it & 1], total)
            KHG_FOR_THREADS(1024)
            {
                const uint32_t c = tb + (uint32_t)tid * 8;
                uint64_t o[8];
                uint64_t at = off + run + excl[KHG_SLOT];
                KHG_UNROLL
                for (int i = 0; i < 8; ++i) {
                    o[i] = at;
                    at += v[KHG_SLOT][i];
                }
                if (c >= c_begin && c + 8 <= c_end) {
                    __builtin_memcpy((uint64_t*)KHG_ASSUME_ALIGNED(a.chunk_off + c, 64), o, 64);
                } else {
                    KHG_UNROLL
                    for (int i = 0; i < 8; ++i)
                        if (c + i >= c_begin && c + i < c_end) a.chunk_off[c + i] = o[i];
                }
            }
            run += total;
        }
        // Compress.h:310-328: pad with '0' to a multiple of 8 bits
        const uint32_t pad = (uint32_t)((8 - (run & 7)) & 7);
        const uint64_t padded = run + pad;
        if (padded >= ((uint64_t)1 << 31)) ok = false;  // HeaderInfo lengths are int (Type.h:5)
        KHG_FOR_THREADS(1024)
        {
            if (tid == 0) {
                if (padded >= ((uint64_t)1 << 31)) khg_set_status(&a.res->status, -4);
                a.res->bits[p] = (int32_t)padded;
                a.res->pads[p] = (int32_t)pad;
                a.res->k[p] = a.tab_k[p];
            }
        }
        off += padded;
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(1024)
    {
        if (tid == 0) {
            KhgResult* res = a.res;
            if (res->status != 0) ok = false;
            const uint64_t table_off = off >> 3;
            const uint64_t total = table_off + 8ull * ((uint64_t)a.tab_k[0] + a.tab_k[1] + a.tab_k[2]);
            res->total_len = total;
            if (((total + 3) & ~3ull) > a.cap) {
                khg_set_status(&res->status, -2);
                ok = false;
            }
            sh.table_off = table_off;
            sh.ok = ok;
            res->ok_to_pack = ok;
            if (ok && a.write_meta) {
                // Writer.h:9-23.  Offsets 24/28: rows, cols (equal for the square images the
                // reference supports; it passes (img_height, img_width) into Type.h:18's (w, h) slots).
                uint8_t* o = a.out;
                khg_store_le32(o + 0, res->bits[0]);
                khg_store_le32(o + 4, res->bits[1]);
                khg_store_le32(o + 8, res->bits[2]);
                khg_store_le32(o + 12, res->k[0]);
                khg_store_le32(o + 16, res->k[1]);
                khg_store_le32(o + 20, res->k[2]);
                khg_store_le32(o + 24, a.rows);
                khg_store_le32(o + 28, a.cols);
                o[32] = (uint8_t)res->pads[0];
                o[33] = (uint8_t)res->pads[1];
                o[34] = (uint8_t)res->pads[2];
                o[35] = (uint8_t)a.lvl;
            }
        }
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(1024)
    {
        if (sh.ok && a.write_meta) {
            // Writer.h:34-41: (int32 count, int32 symbol) per entry, Y then Cb then Cr
            uint8_t* t = a.out + sh.table_off;
            uint32_t kk[3] = {(uint32_t)a.tab_k[0], (uint32_t)a.tab_k[1], (uint32_t)a.tab_k[2]};
            uint32_t tot = kk[0] + kk[1] + kk[2];
            for (uint32_t e = (uint32_t)tid; e < tot; e += 1024) {
                int p = e >= kk[0] + kk[1] ? 2 : e >= kk[0] ? 1 : 0;
                uint32_t i = e - (p == 2 ? kk[0] + kk[1] : p == 1 ? kk[0] : 0);
                khg_store_le32(t + 8 * (size_t)e, a.tab_cnt[(size_t)p * KHG_NBINS + i]);
                khg_store_le32(t + 8 * (size_t)e + 4, a.tab_sym[(size_t)p * KHG_NBINS + i]);
            }
        }
    }
}

enum { KHG_PACK_WORDS = (31 + KHG_CHUNK * KHG_MAX_CODE_BITS + 31) / 32 + 2 };

struct KhgPackSmem {
    uint32_t words[KHG_PACK_WORDS];
    uint32_t wave[4];
    uint64_t hot[KHG_HOT_SYMS];  // see KHG_HOT_SYMS
};

// Table entry of symbol v: (length << 56) | code.  hot: the workgroup's LDS copy of the entries of -128 .. 127.
template <int NHOT>
KHG_DEV uint64_t khg_pack_code_lds_n(const uint64_t* hot, const uint64_t* ctab, int v)
{
    // the LDS entry is fetched unconditionally (index wrapped) and replaced on the rare path: a select between
    // an LDS and a global address would turn both into one flat load
    const uint32_t h = (uint32_t)(v + NHOT / 2);
    uint64_t c = hot[h & (NHOT - 1)];
    if (h >= (uint32_t)NHOT) {
        KHG_OPAQUE(c);  // the LDS load stays a load of its own
        c = ctab[v + KHG_SYM_BIAS];
    }
    return c;
}
KHG_DEV uint64_t khg_pack_code_lds(const uint64_t* hot, const uint64_t* ctab, int v)
{
    return khg_pack_code_lds_n<KHG_HOT_SYMS>(hot, ctab, v);
}

// ORs the n <= 64 bits of `bits` (right-aligned) into the workgroup's word buffer at bit position pos (first
// stream bit = most significant bit of a word).  Zero bits set nothing, so an all-zero string costs nothing.
KHG_DEV void khg_pack_put(uint32_t* words, uint32_t pos, uint64_t bits, uint32_t n)
{
    if (n == 0 || bits == 0) return;
    const uint64_t v = bits << (64 - n);  // left-aligned
    const uint32_t w = pos >> 5, s = pos & 31;
    const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    if (hi >> s) khg_atomic_or(&words[w], hi >> s);
    if (s + n > 32) {
        const uint32_t mid = s ? (hi << (32 - s)) | (lo >> s) : lo;
        if (mid) khg_atomic_or(&words[w + 1], mid);
        if (s + n > 64 && (lo << (32 - s))) khg_atomic_or(&words[w + 2], lo << (32 - s));
    }
}

// One chunk per workgroup, 8 symbols per thread.  A thread first strings its eight codes together in a register
// (nearly always 64 bits or fewer); the chunk's bits are assembled in LDS in big-endian 32-bit words aligned to the
// output word holding its first bit, then written out; the first and last word may be shared with neighbouring
// chunks and are OR-merged.
KHG_DEV void khg_pack_body(const KhgEncArgs& a, uint32_t chunk, KhgPackSmem& sh)
{
    if (!a.res->ok_to_pack) return;
    const KhgChunk ck = khg_chunk_info(a, chunk);
    if (!ck.mine) return;
    const int p = ck.plane;
    const uint64_t* ctab = a.codetab + (size_t)p * KHG_NBINS;
    const uint32_t base = ck.sym0, end = ck.sym0 + ck.nsym;
    const uint32_t bits = a.chunk_bits[chunk];
    if (bits == 0) return;
    const uint64_t off = a.chunk_off[chunk];
    const uint32_t lead = (uint32_t)(off & 31);
    const uint32_t nw = (lead + bits + 31) >> 5;

    uint64_t acc[KHG_NSLOT(256)];
    uint32_t tbs[KHG_NSLOT(256)], excl[KHG_NSLOT(256)];
    int16_t vs[KHG_NSLOT(256)][8];
    int nvs[KHG_NSLOT(256)];
    KHG_FOR_THREADS(256)
    {
        sh.hot[tid] = ctab[KHG_SYM_BIAS - KHG_HOT_SYMS / 2 + tid];
        for (uint32_t i = (uint32_t)tid; i < nw; i += 256) sh.words[i] = 0;
        nvs[KHG_SLOT] = khg_load8(a.sym[p], base + (uint32_t)tid * 8, end, vs[KHG_SLOT]);
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        uint32_t tb = 0;
        uint64_t ac = 0;
        const int16_t* v = vs[KHG_SLOT];
        const int nv = nvs[KHG_SLOT];
        KHG_UNROLL
        for (int i = 0; i < 8; ++i) {
            if (i < nv) {
                const uint64_t c = khg_pack_code_lds(sh.hot, ctab, v[i]);
                const uint32_t len = (uint32_t)(c >> 56);
                ac = (ac << len) | (c & 0x00FFFFFFFFFFFFFFull);  // meaningful while tb <= 64 (len <= 56)
                tb += len;
            }
        }
        acc[KHG_SLOT] = ac;
        tbs[KHG_SLOT] = tb;
    }
    uint32_t total;
    KHG_WG_EXCL_SCAN(excl, tbs, sh.wave, total)
    KHG_FOR_THREADS(256)
    {
        (void)total;
        uint32_t pos = lead + excl[KHG_SLOT];
        const uint32_t tb = tbs[KHG_SLOT];
        if (tb <= 64) {
            khg_pack_put(sh.words, pos, acc[KHG_SLOT], tb);
        } else {  // long codes: one at a time
            const int16_t* v = vs[KHG_SLOT];
            const int nv = nvs[KHG_SLOT];
            KHG_UNROLL
            for (int i = 0; i < 8; ++i) {  // constant indices: the symbols stay in registers
                if (i < nv) {
                    const uint64_t c = khg_pack_code_lds(sh.hot, ctab, v[i]);
                    const uint32_t len = (uint32_t)(c >> 56);
                    khg_pack_put(sh.words, pos, c & 0x00FFFFFFFFFFFFFFull, len);
                    pos += len;
                }
            }
        }
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        uint32_t* out32 = (uint32_t*)a.out + (off >> 5);
        for (uint32_t i = (uint32_t)tid; i < nw; i += 256) {
            uint32_t v = khg_bswap32(sh.words[i]);  // first stream bit = MSB of the first byte
            if (i == 0 || i == nw - 1) khg_atomic_or(&out32[i], v);
            else out32[i] = v;
        }
    }
}

// ---------------------------------------------------------------- single-pass packer
//
// khg_lens_body + khg_scan_body + khg_pack_body in one pass over the symbols (whole images on one GPU; row-tiled
// images keep the three steps, because the chunk bit counts travel between the ranks).  Where a chunk's bits go
// depends on the lengths of all chunks before it; instead of a first pass that only measures, the workgroups of
// the launch tell each other (decoupled look-back, two levels):
//   * the launch's entries are, in order: Y superchunks (KHG_SUPER chunks = 16384 symbols), pad, Cb superchunks,
//     pad, Cr superchunks, pad ("pad" = the zero bits that fill a plane's stream up to a byte, Compress.h:310-328).
//     A workgroup takes the next entry when it STARTS (a ticket), so every entry before its own belongs to a
//     workgroup that is already running;
//   * a workgroup first turns its symbols into per-thread code strings held in registers, which gives the entry's
//     bit count, and publishes AGG = (bit count, the entry's last <= 32 bits).  The last entry of every group of 64
//     also publishes the group's bit count (GAGG);
//   * wave 0 then finds the entry's start: the bit counts of the entries before it in its group (one load per
//     lane) plus the end of the previous group -- its published end (GPRE), or the nearest published group end plus
//     the group counts in between.  It publishes its own end (PRE; a group's last entry also GPRE).  A pad entry
//     can only publish once the plane before it is complete; that is the one real dependency of the format;
//   * the workgroup then assembles chunk after chunk in LDS and writes the 32-bit words each one COMPLETES: the
//     first word begins with the last (start mod 32) bits of the stream before it -- the previous chunk's, or for
//     the entry's first chunk the tail in its predecessors' AGG words; the entry's own unfinished last word is left
//     to its successor.  No word is written twice, so nothing has to be zeroed or OR-merged.
// Every wait is bounded: a workgroup that gives up flags the job (KHG_ERR_HIP) and ends, so the grid drains.

enum {
    KHG_SUPER = 8,               // chunks per entry
    KHG_CHAIN_GROUP = 64,        // entries per group
    KHG_CHAIN_SPINS = 1 << 17    // polls (about a microsecond each) before a workgroup gives up
};
#define KHG_CHAIN_VALID (1ull << 63)

KHG_HOST_DEV uint32_t khg_chain_plane_entries(uint32_t plane_chunks) { return (plane_chunks + KHG_SUPER - 1) / KHG_SUPER; }

KHG_DEV uint64_t* khg_chain_agg(const KhgEncArgs& a, uint32_t i) { return a.chain + KHG_CHAIN_HEAD + 2 * (size_t)i; }
KHG_DEV uint64_t* khg_chain_pre(const KhgEncArgs& a, uint32_t i) { return a.chain + KHG_CHAIN_HEAD + 2 * (size_t)i + 1; }
KHG_DEV uint64_t* khg_chain_gagg(const KhgEncArgs& a, uint32_t g) { return a.chain + KHG_CHAIN_HEAD + 2 * ((size_t)a.chain_entries + g); }
KHG_DEV uint64_t* khg_chain_gpre(const KhgEncArgs& a, uint32_t g) { return a.chain + KHG_CHAIN_HEAD + 2 * ((size_t)a.chain_entries + g) + 1; }

// Entry t of the launch: plane, whether it is the plane's pad entry, and otherwise its symbols.
struct KhgChainEntry {
    int plane;
    bool pad;
    uint32_t sym0, sym1;  // raster range of the entry's symbols
    uint32_t plane_entries;
};
KHG_DEV KhgChainEntry khg_chain_entry(const KhgEncArgs& a, uint32_t t)
{
    KhgChainEntry e = {2, true, 0, 0, 0};
    for (int p = 0; p < 3; ++p) {
        const uint32_t ne = khg_chain_plane_entries((p < 2 ? a.chunk0[p + 1] : a.nchunks) - a.chunk0[p]);
        e.plane = p;
        e.plane_entries = ne;
        if (t < ne) {
            e.pad = false;
            e.sym0 = t * (uint32_t)(KHG_SUPER * KHG_CHUNK);
            e.sym1 = a.n[p] - e.sym0 < (uint32_t)(KHG_SUPER * KHG_CHUNK) ? a.n[p] : e.sym0 + (uint32_t)(KHG_SUPER * KHG_CHUNK);
            return e;
        }
        t -= ne;
        if (t == 0) return e;
        t -= 1;
    }
    return e;
}

// Waits (bounded) for a published word.  Returns 0 if it never came.
KHG_DEV uint64_t khg_chain_wait(const uint64_t* p)
{
    uint64_t v = khg_chain_load(p);
#ifndef KHG_SIM
    for (int spin = 0; !(v & KHG_CHAIN_VALID) && spin < KHG_CHAIN_SPINS; ++spin) {
        __builtin_amdgcn_s_sleep(2);
        v = khg_chain_load(p);
    }
#endif
    return (v & KHG_CHAIN_VALID) ? v : 0;
}

KHG_DEV uint32_t khg_agg_bits(uint64_t v) { return (uint32_t)((v >> 32) & 0x7FFFFFFFu); }

// The last (up to 32) bits of the stream before entry c, right-aligned (fewer than 32 exist only at the very start
// of the file's streams, which is word-aligned, so they are never needed there).  One thread.
KHG_DEV uint32_t khg_chain_tail_before(const KhgEncArgs& a, uint32_t c)
{
    uint32_t bits = 0, have = 0;
    for (uint32_t i = c; i-- > 0 && have < 32;) {
        const uint64_t v = khg_chain_wait(khg_chain_agg(a, i));
        if (!v) {
            khg_set_status(&a.res->status, -6);
            break;
        }
        const uint32_t b = khg_agg_bits(v), tail = (uint32_t)v;
        const uint32_t take = b < 32 - have ? b : 32 - have;  // the entry's last `take` bits (its tail holds min(32, b))
        if (take) {
            const uint32_t part = take >= 32 ? tail : (tail & ((1u << take) - 1));
            bits |= part << have;
            have += take;
        }
    }
    return bits;
}

// ---- wave collectives of the look-back: `lane` 0..63 of wave 0 call them together (the simulator runs the lanes one
// after another, so there they are plain loops executed by lane 0 and no-ops for the others).
// Sum of the bit counts of entries [first, first + n) (n <= 64); 0 valid flag -> failure reported through *ok.
KHG_DEV uint64_t khg_chain_sum_window(const uint64_t* words, uint32_t n, int lane, bool* ok)
{
#ifdef KHG_SIM
    uint64_t s = 0;
    if (lane == 0)
        for (uint32_t l = 0; l < n; ++l) {
            const uint64_t v = khg_chain_wait(words + 2 * (size_t)l);
            if (!v) *ok = false;
            s += khg_agg_bits(v);
        }
    return s;
#else
    uint64_t v = KHG_CHAIN_VALID;
    if ((uint32_t)lane < n) v = khg_chain_wait(words + 2 * (size_t)lane);
    if (__any(v == 0)) *ok = false;
    unsigned long long part = (uint32_t)lane < n ? khg_agg_bits(v) : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
    return part;
#endif
}

// End position of the stream before group g (g >= 1): the nearest published group end plus the bit counts of the
// groups in between.
KHG_DEV uint64_t khg_chain_group_start(const KhgEncArgs& a, uint32_t g, int lane, bool* ok)
{
    const uint64_t origin = 36 * 8;  // the streams start behind the 36-byte header (Writer.h:9-23)
#ifdef KHG_SIM
    (void)lane;
    const uint64_t v = khg_chain_wait(khg_chain_gpre(a, g - 1));  // groups complete one after another
    if (!v) *ok = false;
    (void)origin;
    return v & ~KHG_CHAIN_VALID;
#else
    uint64_t sum = 0;
    int64_t hi = (int64_t)g - 1;
    for (;;) {
        if (hi < 0) return origin + sum;
        const int64_t i = hi - lane;
        uint64_t gagg = KHG_CHAIN_VALID, gpre = 0;
        if (i >= 0) {
            gpre = khg_chain_load(khg_chain_gpre(a, (uint32_t)i));
            if (!(gpre & KHG_CHAIN_VALID)) gagg = khg_chain_wait(khg_chain_gagg(a, (uint32_t)i));
        }
        if (__any(gagg == 0)) {
            *ok = false;
            return origin;
        }
        const unsigned long long known = __ballot(i >= 0 && (gpre & KHG_CHAIN_VALID));
        const int first = known ? __builtin_ctzll(known) : 64;  // nearest group whose end is known
        unsigned long long part = (i >= 0 && lane < first) ? (gagg & ~KHG_CHAIN_VALID) : 0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
        sum += part;
        if (known) return __shfl((unsigned long long)(gpre & ~KHG_CHAIN_VALID), first, 64) + sum;
        hi -= 64;
    }
#endif
}

// The single-pass packer's LDS tables: the codes of the symbols -64 .. 63 one by one, and the joint code of every PAIR
// of symbols in -8 .. 7 (nearly all symbols of a detail band): one lookup and one merge for two symbols.
enum { KHG_PS_HOT = 128 };
#define KHG_PAIR_NONE (1ull << 63)  // no joint entry: a symbol without a code, or more than 56 bits together

// Entry idx of the pair table: the first symbol's low four bits in bits 0-3 of idx, the second symbol's in bits 4-7.
KHG_DEV uint64_t khg_pair_entry(const uint64_t* ctab, uint32_t idx)
{
    const int s0 = (int)((idx & 15u) ^ 8u) - 8, s1 = (int)(((idx >> 4) & 15u) ^ 8u) - 8;
    const uint64_t c0 = ctab[s0 + KHG_SYM_BIAS], c1 = ctab[s1 + KHG_SYM_BIAS];
    const uint32_t l0 = (uint32_t)(c0 >> 56), l1 = (uint32_t)(c1 >> 56);
    if (l0 == 0 || l1 == 0 || l0 + l1 > (uint32_t)KHG_MAX_CODE_BITS) return KHG_PAIR_NONE;
    return ((uint64_t)(l0 + l1) << 56) | ((c0 & 0x00FFFFFFFFFFFFFFull) << l1) | (c1 & 0x00FFFFFFFFFFFFFFull);
}

// The entry's last (up to 32) bits, from its last (up to 64) symbols: every code has at least one bit, so 32
// symbols always suffice.  Wave 0.
KHG_DEV uint32_t khg_entry_tail32(const KhgEncArgs& a, int p, const uint64_t* hot, uint32_t sym0, uint32_t sym1, int lane)
{
    const uint64_t* ctab = a.codetab + (size_t)p * KHG_NBINS;
    const uint32_t n = sym1 - sym0 < 64u ? sym1 - sym0 : 64u;
#ifdef KHG_SIM
    uint32_t tail = 0, pos = 0;
    if (lane == 0)
        for (uint32_t i = 0; i < n && pos < 32; ++i) {  // from the last symbol backwards
            const uint64_t c = khg_pack_code_lds_n<KHG_PS_HOT>(hot, ctab, a.sym[p][sym1 - 1 - i]);
            const uint32_t len = (uint32_t)(c >> 56);
            tail |= (uint32_t)((c & 0x00FFFFFFFFFFFFFFull) << pos);
            pos += len;
        }
    return tail;
#else
    // lane l takes the symbol l places before the last one; pos = bits of the symbols behind it
    uint32_t len = 0;
    uint64_t code = 0;
    if ((uint32_t)lane < n) {
        const uint64_t c = khg_pack_code_lds_n<KHG_PS_HOT>(hot, ctab, a.sym[p][sym1 - 1 - (uint32_t)lane]);
        len = (uint32_t)(c >> 56);
        code = c & 0x00FFFFFFFFFFFFFFull;
    }
    const uint32_t pos = khg_wave_incl_scan(len) - len;
    uint32_t part = pos < 32 ? (uint32_t)(code << pos) : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part |= (uint32_t)__shfl_xor((int)part, d, 64);
    return part;
#endif
}

struct KhgPackScanSmem {
    uint32_t words[KHG_PACK_WORDS];
    uint32_t wave[KHG_SUPER][4];
    uint64_t pair[256];        // khg_pair_entry
    uint64_t hot[KHG_PS_HOT];  // (length << 56) | code of the symbols -64 .. 63
    uint64_t start;      // absolute bit position of the entry
    uint32_t carry[2];   // the 32 stream bits before the chunk being written (ping-pong between chunks)
    uint32_t nbits;      // a pad entry: its pad bits
};

// Bits a batch of chunks may have (the simulator's build lowers it, so that ordinary pictures take several batches).
#ifndef KHG_PS_BATCH_BITS
#define KHG_PS_BATCH_BITS ((KHG_PACK_WORDS - 3) * 32)
#endif

// The batch that begins with chunk j0 (whose first bit is bit base0 of the entry) ends before chunk j1 (returned):
// as many whole chunks as fit KHG_PS_BATCH_BITS, at least one.  *bits: the batch's bits.
KHG_DEV int khg_ps_batch_end(const uint32_t (&sub_base)[KHG_SUPER], const uint32_t (&tot)[KHG_SUPER], int nsub, int j0,
                             uint32_t base0, uint32_t* bits)
{
    int j1 = nsub;
    KHG_UNROLL
    for (int j = KHG_SUPER - 1; j >= 1; --j)
        if (j > j0 && j < nsub && sub_base[j] + tot[j] - base0 > (uint32_t)KHG_PS_BATCH_BITS) j1 = j;
    uint32_t end = base0;
    KHG_UNROLL
    for (int j = 0; j < KHG_SUPER; ++j)
        if (j == j1 - 1) end = sub_base[j] + tot[j];
    *bits = end - base0;
    return j1;
}

// Thread tid ORs its code strings of the chunks [j0, j1) into the word buffer; bit base0 of the entry is bit 0 of the
// buffer.
KHG_DEV void khg_ps_put_chunks(const KhgEncArgs& a, const KhgChainEntry& en, int p, KhgPackScanSmem& sh, int tid, int slot,
                               const uint64_t (&acc)[KHG_SUPER][KHG_NSLOT(256)], const uint32_t (&tbs)[KHG_SUPER][KHG_NSLOT(256)],
                               const uint32_t (&excl)[KHG_SUPER][KHG_NSLOT(256)], const uint32_t (&sub_base)[KHG_SUPER], int j0,
                               int j1, uint32_t base0)
{
    const uint64_t* ctab = a.codetab + (size_t)p * KHG_NBINS;
    KHG_UNROLL
    for (int j = 0; j < KHG_SUPER; ++j) {
        if (j < j0 || j >= j1) continue;
        uint32_t pos = sub_base[j] - base0 + excl[j][slot];
        const uint32_t tb = tbs[j][slot];
        if (tb <= 64) {
            khg_pack_put(sh.words, pos, acc[j][slot], tb);
        } else {  // long codes: one at a time, from the symbols again (rare: kept small, not unrolled)
            const uint32_t s0 = en.sym0 + (uint32_t)j * KHG_CHUNK + (uint32_t)tid * 8;
            KHG_NOUNROLL
            for (uint32_t i = s0; i < s0 + 8 && i < en.sym1; ++i) {
                const uint64_t c = ctab[a.sym[p][i] + KHG_SYM_BIAS];
                const uint32_t len = (uint32_t)(c >> 56);
                khg_pack_put(sh.words, pos, c & 0x00FFFFFFFFFFFFFFull, len);
                pos += len;
            }
        }
    }
}

// One entry of the chain per workgroup (`t` = its ticket; the GPU kernel draws it, the simulator passes 0, 1, 2 ...).
KHG_DEV void khg_packscan_body(const KhgEncArgs& a, uint32_t t, KhgPackScanSmem& sh)
{
    const KhgChainEntry en = khg_chain_entry(a, t);
    const int p = en.plane;
    const uint64_t* ctab = a.codetab + (size_t)p * KHG_NBINS;
    const uint32_t group = t / KHG_CHAIN_GROUP, in_group = t % KHG_CHAIN_GROUP;
    const bool closes_group = in_group == KHG_CHAIN_GROUP - 1 || t == a.chain_entries - 1;

#ifdef KHG_TIMING
    KHG_FOR_THREADS(256) { if (tid == 0) a.chunk_off[8 * (size_t)t + 0] = khg_clock(); }
#endif
    uint64_t acc[KHG_SUPER][KHG_NSLOT(256)];
    uint32_t tbs[KHG_SUPER][KHG_NSLOT(256)], excl[KHG_SUPER][KHG_NSLOT(256)];
    KHG_FOR_THREADS(256)
    {
        if (tid < KHG_PS_HOT) sh.hot[tid] = ctab[KHG_SYM_BIAS - KHG_PS_HOT / 2 + tid];
        sh.pair[tid] = khg_pair_entry(ctab, (uint32_t)tid);
        for (uint32_t i = (uint32_t)tid; i < (uint32_t)KHG_PACK_WORDS; i += 256) sh.words[i] = 0;  // for the whole-entry path
    }
    KHG_BARRIER();
    // every thread's eight symbols of every chunk as one code string (nearly always 64 bits or fewer)
    KHG_FOR_THREADS(256)
    {
        KHG_UNROLL
        for (int j = 0; j < KHG_SUPER; ++j) {
            int16_t v[8];
            const uint32_t s0 = en.sym0 + (uint32_t)j * KHG_CHUNK + (uint32_t)tid * 8;
            const int nv = (en.pad || s0 >= en.sym1) ? 0 : khg_load8(a.sym[p], s0, en.sym1, v);
            uint32_t tb = 0;
            uint64_t ac = 0;
            // two symbols per lookup; `odd` collects what the pair table does not cover: a symbol outside -8 .. 7
            // (bits 3 .. 15 of a symbol are not all alike), a pair without a joint entry, a ragged end
            uint32_t w[4] = {0, 0, 0, 0};
            if (nv == 8) __builtin_memcpy(w, v, 16);
            uint32_t odd = nv == 8 ? 0u : 0x10u;
            KHG_UNROLL
            for (int q = 0; q < 4; ++q) {
                odd |= (w[q] ^ (w[q] << 1)) & 0xFFF0FFF0u;
                const uint64_t c = sh.pair[(w[q] & 15u) | ((w[q] >> 12) & 0xF0u)];
                odd |= (uint32_t)(c >> 32) & 0x80000000u;
                const uint32_t len = (uint32_t)(c >> 56) & 0x7Fu;
                ac = (ac << len) | (c & 0x00FFFFFFFFFFFFFFull);  // meaningful while tb <= 64 (len <= 56)
                tb += len;
            }
            if (odd) {  // one symbol at a time: all eight from the LDS table, or all eight from the global one
                uint64_t c[8];
                uint32_t far = 0;
                KHG_UNROLL
                for (int i = 0; i < 8; ++i) {
                    const uint32_t h = (uint32_t)(v[i] + KHG_PS_HOT / 2);
                    far |= h;
                    c[i] = sh.hot[h & (KHG_PS_HOT - 1)];
                }
                if (far >= (uint32_t)KHG_PS_HOT) {
                    KHG_UNROLL
                    for (int i = 0; i < 8; ++i) c[i] = ctab[(i < nv ? v[i] : 0) + KHG_SYM_BIAS];
                }
                tb = 0;
                ac = 0;
                KHG_UNROLL
                for (int i = 0; i < 8; ++i) {
                    const uint32_t len = i < nv ? (uint32_t)(c[i] >> 56) : 0u;
                    ac = (ac << len) | (i < nv ? c[i] & 0x00FFFFFFFFFFFFFFull : 0ull);
                    tb += len;
                }
            }
            acc[j][KHG_SLOT] = ac;
            tbs[j][KHG_SLOT] = tb;
        }
    }
    uint32_t tot[KHG_SUPER], sub_base[KHG_SUPER], total = 0;  // bits of each chunk, their offsets in the entry
    KHG_WG_EXCL_SCAN_N(KHG_SUPER, excl, tbs, sh.wave, tot)
    for (int j = 0; j < KHG_SUPER; ++j) {
        sub_base[j] = total;
        total += tot[j];
    }
    // The entry is assembled in LDS, at bit 0 of words[0], in batches of whole chunks whose bits fit the word buffer:
    // nearly always one batch of all eight (streams below seven bits per symbol), put together while wave 0 finds
    // out where the entry goes -- three barriers per entry.  Further batches cost three barriers each.
#ifdef KHG_TIMING
    KHG_FOR_THREADS(256) { if (tid == 0) a.chunk_off[8 * (size_t)t + 1] = khg_clock(); }
#endif
    const int nsub = en.pad ? 1 : KHG_SUPER;
    int j0 = 0;
    uint32_t base0 = 0, batch_bits;
    int j1 = khg_ps_batch_end(sub_base, tot, nsub, j0, base0, &batch_bits);
    const bool last = en.pad && p == 2;  // the file's streams end here: the unfinished word is flushed too
    bool first = true;
    KHG_NOUNROLL
    for (int cy = 0;; cy ^= 1) {
        if (!first) {  // a further batch: the buffer is free once the batch before is written
            KHG_BARRIER();
            j0 = j1;
            base0 += batch_bits;
            j1 = khg_ps_batch_end(sub_base, tot, nsub, j0, base0, &batch_bits);
            KHG_FOR_THREADS(256)
            {
                // the words the batch's bits reach, and one more (the funnel shift below reads a word past the last bit)
                for (uint32_t i = (uint32_t)tid; i < ((batch_bits + 31) >> 5) + 2; i += 256) sh.words[i] = 0;
            }
            KHG_BARRIER();
        }
        KHG_FOR_THREADS(256)
        {
            // first batch: publish the entry's bit count and tail, find its place, publish its end
            if (first && tid < 64) {
                bool ok = true;
                if (!en.pad) {
                    const uint32_t tail = khg_entry_tail32(a, p, sh.hot, en.sym0, en.sym1, tid);
                    if (tid == 0) khg_chain_store(khg_chain_agg(a, t), KHG_CHAIN_VALID | ((uint64_t)total << 32) | tail);
                    if (closes_group) {  // the group's bit count, for look-backs that find no published group end nearby
                        const uint64_t gsum = khg_chain_sum_window(khg_chain_agg(a, group * KHG_CHAIN_GROUP), in_group + 1, tid, &ok);
                        if (tid == 0) khg_chain_store(khg_chain_gagg(a, group), KHG_CHAIN_VALID | gsum);
                    }
                }
                // entries before this one in its group, then everything before the group
                uint64_t start = khg_chain_sum_window(khg_chain_agg(a, group * KHG_CHAIN_GROUP), in_group, tid, &ok);
                start += group ? khg_chain_group_start(a, group, tid, &ok) : 36 * 8;
                uint32_t bits = total;
                if (en.pad) bits = (uint32_t)((8 - (start & 7)) & 7);  // Compress.h:310-328: '0' bits up to a byte
                if (tid == 0) {
                    if (!ok) khg_set_status(&a.res->status, -6);
                    if (en.pad) {  // the plane's lengths for the header
                        uint64_t plane_start = 36 * 8;
                        if (p > 0) plane_start = khg_chain_wait(khg_chain_pre(a, t - 1 - en.plane_entries)) & ~KHG_CHAIN_VALID;
                        const uint64_t padded = start + bits - plane_start;
                        a.res->bits[p] = padded >= ((uint64_t)1 << 31) ? -1 : (int32_t)padded;  // checked by khg_meta_body
                        a.res->pads[p] = (int32_t)bits;
                        khg_chain_store(khg_chain_agg(a, t), KHG_CHAIN_VALID | ((uint64_t)bits << 32));
                    }
                    khg_chain_store(khg_chain_pre(a, t), KHG_CHAIN_VALID | (start + bits));
                    if (closes_group) khg_chain_store(khg_chain_gpre(a, group), KHG_CHAIN_VALID | (start + bits));
                    sh.start = start;
                    sh.carry[0] = (start & 31) ? khg_chain_tail_before(a, t) : 0;
                    sh.nbits = bits;
                }
                if (closes_group && en.pad) {  // a pad entry's own count is only known now
                    const uint64_t gsum = khg_chain_sum_window(khg_chain_agg(a, group * KHG_CHAIN_GROUP), in_group + 1, tid, &ok);
                    if (tid == 0) khg_chain_store(khg_chain_gagg(a, group), KHG_CHAIN_VALID | gsum);
                }
            }
#ifdef KHG_TIMING
            if (first && tid == 0) a.chunk_off[8 * (size_t)t + 2] = khg_clock();
#endif
            khg_ps_put_chunks(a, en, p, sh, tid, KHG_SLOT, acc, tbs, excl, sub_base, j0, j1, base0);
        }
        KHG_BARRIER();
#ifdef KHG_TIMING
        KHG_FOR_THREADS(256) { if (first && tid == 0) a.chunk_off[8 * (size_t)t + 3] = khg_clock(); }
#endif
        first = false;
        const uint32_t bits = en.pad ? sh.nbits : batch_bits;
        const uint64_t start = sh.start + base0;
        const uint32_t r = (uint32_t)(start & 31);
        KHG_FOR_THREADS(256)
        {
            const uint32_t carry = sh.carry[cy];
            const uint32_t ncomplete = (r + bits) >> 5, rem = (r + bits) & 31;
            const uint32_t nwrite = ncomplete + ((last && rem) ? 1 : 0);
            const uint64_t w0 = start >> 5;
            const bool fits = (w0 + nwrite) * 4 <= a.cap;
            if (!fits && tid == 0 && nwrite) khg_set_status(&a.res->status, -2);
            const uint32_t carry_hi = r ? (carry & ((1u << r) - 1)) << (32 - r) : 0;  // the r bits before `start`
            uint32_t* out32 = (uint32_t*)a.out + w0;
            for (uint32_t k = (uint32_t)tid; k <= ncomplete; k += 256) {
                uint32_t v;
                if (r == 0) {
                    v = sh.words[k];
                } else {
                    const uint32_t hi = k == 0 ? carry_hi : sh.words[k - 1] << (32 - r);
                    v = hi | (sh.words[k] >> r);
                }
                if (k < nwrite && fits) out32[k] = khg_bswap32(v);  // first stream bit = MSB of the first byte
                if (k == ncomplete) sh.carry[cy ^ 1] = rem ? v >> (32 - rem) : 0;  // left to the next batch
            }
        }
        if (j1 >= nsub) break;
    }
#ifdef KHG_TIMING
    KHG_FOR_THREADS(256) { if (tid == 0) a.chunk_off[8 * (size_t)t + 4] = khg_clock(); }
#endif
}

// Header (Writer.h:9-23) and tables (Writer.h:34-41) behind the streams the single-pass packer wrote.  One
// workgroup of 1024 threads.
KHG_DEV void khg_meta_body(const KhgEncArgs& a, KhgScanSmem& sh)
{
    KHG_FOR_THREADS(1024)
    {
        if (tid == 0) {
            KhgResult* res = a.res;
            int ok = res->status == 0;
            uint64_t off = 36 * 8;
            for (int p = 0; p < 3; ++p) {
                if (res->bits[p] < 0) {  // HeaderInfo lengths are int (Type.h:5)
                    khg_set_status(&res->status, -4);
                    ok = 0;
                }
                off += (uint64_t)(uint32_t)res->bits[p];
                res->k[p] = a.tab_k[p];
            }
            const uint64_t table_off = off >> 3;
            const uint64_t total = table_off + 8ull * ((uint64_t)a.tab_k[0] + a.tab_k[1] + a.tab_k[2]);
            res->total_len = total;
            if (ok && ((total + 3) & ~3ull) > a.cap) {
                khg_set_status(&res->status, -2);
                ok = 0;
            }
            sh.table_off = table_off;
            sh.ok = ok;
            res->ok_to_pack = ok;
            if (ok) {
                uint8_t* o = a.out;
                khg_store_le32(o + 0, res->bits[0]);
                khg_store_le32(o + 4, res->bits[1]);
                khg_store_le32(o + 8, res->bits[2]);
                khg_store_le32(o + 12, res->k[0]);
                khg_store_le32(o + 16, res->k[1]);
                khg_store_le32(o + 20, res->k[2]);
                khg_store_le32(o + 24, a.rows);
                khg_store_le32(o + 28, a.cols);
                o[32] = (uint8_t)res->pads[0];
                o[33] = (uint8_t)res->pads[1];
                o[34] = (uint8_t)res->pads[2];
                o[35] = (uint8_t)a.lvl;
            }
        }
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(1024)
    {
        if (sh.ok) {
            uint8_t* tb = a.out + sh.table_off;
            uint32_t kk[3] = {(uint32_t)a.tab_k[0], (uint32_t)a.tab_k[1], (uint32_t)a.tab_k[2]};
            uint32_t tot = kk[0] + kk[1] + kk[2];
            for (uint32_t e = (uint32_t)tid; e < tot; e += 1024) {
                int p = e >= kk[0] + kk[1] ? 2 : e >= kk[0] ? 1 : 0;
                uint32_t i = e - (p == 2 ? kk[0] + kk[1] : p == 1 ? kk[0] : 0);
                khg_store_le32(tb + 8 * (size_t)e, a.tab_cnt[(size_t)p * KHG_NBINS + i]);
                khg_store_le32(tb + 8 * (size_t)e + 4, a.tab_sym[(size_t)p * KHG_NBINS + i]);
            }
        }
    }
}
