/*
 * khugle_oracle.c -- CPU restatement of the KhuGle codec path (see khugle_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP path, never the product.
 * Build with -O2 -ffp-contract=off (no FMA contraction: the reference is MSVC
 * /fp:precise x64, every * and + rounds separately).
 */
#include "khugle_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ colour */

void khgo_rgb_to_ycbcr(const uint8_t *r, const uint8_t *g, const uint8_t *b, int h, int w,
                       double *y, double *cb, double *cr)
{
    /* Colorspace.h:7-13 */
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            size_t p = (size_t)i * w + j;
            double R = r[p], G = g[p], B = b[p];
            y[p] = R * 0.299 + G * 0.587 + B * 0.114;
        }
    /* Colorspace.h:15-22: plain decimation at even (row, col) */
    int cw = w / 2;
    for (int i = 0; i < h; i += 2)
        for (int j = 0; j < w; j += 2) {
            size_t p = (size_t)i * w + j;
            size_t c = (size_t)(i / 2) * cw + j / 2;
            double R = r[p], G = g[p], B = b[p];
            cb[c] = R * -0.16874 + G * -0.33126 + B * 0.5;
            cr[c] = R * 0.5 + G * -0.41869 + B * -0.08131;
        }
}

/* the Win32 min/max macros the reference relies on (Colorspace.h:31-33) */
#define KHGO_MIN(a, b) (((a) < (b)) ? (a) : (b))
#define KHGO_MAX(a, b) (((a) > (b)) ? (a) : (b))

void khgo_ycbcr_to_rgb(const double *y, const double *cb, const double *cr, int h, int w,
                       uint8_t *r, uint8_t *g, uint8_t *b)
{
    int cw = w / 2;
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            size_t p = (size_t)i * w + j;
            size_t c = (size_t)(i / 2) * cw + j / 2;
            /* Colorspace.h:31-33 */
            double R = KHGO_MAX(0, KHGO_MIN(y[p] * 1 + cr[c] * 1.402, 255));
            double G = KHGO_MAX(0, KHGO_MIN(y[p] * 1 + cb[c] * -0.34414 + cr[c] * -0.71414, 255));
            double B = KHGO_MAX(0, KHGO_MIN(y[p] * 1 + cb[c] * 1.772, 255));
            /* Main.cpp:235-237: unsigned char = double (truncation toward zero) */
            r[p] = (unsigned char)R;
            g[p] = (unsigned char)G;
            b[p] = (unsigned char)B;
        }
}

/* --------------------------------------------------------------------- DWT */

void khgo_fwt97(double *x, int n, double *tmp)
{
    double a;
    int i;
    /* DWT.h:12-18 predict 1 */
    a = -1.586134342;
    for (i = 1; i < n - 2; i += 2) x[i] += a * (x[i - 1] + x[i + 1]);
    x[n - 1] += 2 * a * x[n - 2];
    /* DWT.h:20-26 update 1 */
    a = -0.05298011854;
    for (i = 2; i < n; i += 2) x[i] += a * (x[i - 1] + x[i + 1]);
    x[0] += 2 * a * x[1];
    /* DWT.h:28-34 predict 2 */
    a = 0.8829110762;
    for (i = 1; i < n - 2; i += 2) x[i] += a * (x[i - 1] + x[i + 1]);
    x[n - 1] += 2 * a * x[n - 2];
    /* DWT.h:36-42 update 2 */
    a = 0.4435068522;
    for (i = 2; i < n; i += 2) x[i] += a * (x[i - 1] + x[i + 1]);
    x[0] += 2 * a * x[1];
    /* DWT.h:44-50 scale: odd *= a, even /= a with a = 1/1.149604398 rounded once */
    a = 1 / 1.149604398;
    for (i = 0; i < n; i++) {
        if (i % 2) x[i] *= a;
        else x[i] /= a;
    }
    /* DWT.h:52-59 pack */
    for (i = 0; i < n; i++) {
        if (i % 2 == 0) tmp[i / 2] = x[i];
        else tmp[n / 2 + i / 2] = x[i];
    }
    for (i = 0; i < n; i++) x[i] = tmp[i];
}

void khgo_iwt97(double *x, int n, double *tmp)
{
    double a;
    int i;
    /* DWT.h:67-74 unpack */
    for (i = 0; i < n / 2; i++) {
        tmp[i * 2] = x[i];
        tmp[i * 2 + 1] = x[i + n / 2];
    }
    for (i = 0; i < n; i++) x[i] = tmp[i];
    /* DWT.h:76-82 undo scale */
    a = 1.149604398;
    for (i = 0; i < n; i++) {
        if (i % 2) x[i] *= a;
        else x[i] /= a;
    }
    /* DWT.h:84-90 */
    a = -0.4435068522;
    for (i = 2; i < n; i += 2) x[i] += a * (x[i - 1] + x[i + 1]);
    x[0] += 2 * a * x[1];
    /* DWT.h:92-98 */
    a = -0.8829110762;
    for (i = 1; i < n - 2; i += 2) x[i] += a * (x[i - 1] + x[i + 1]);
    x[n - 1] += 2 * a * x[n - 2];
    /* DWT.h:100-106 */
    a = 0.05298011854;
    for (i = 2; i < n; i += 2) x[i] += a * (x[i - 1] + x[i + 1]);
    x[0] += 2 * a * x[1];
    /* DWT.h:108-114 */
    a = 1.586134342;
    for (i = 1; i < n - 2; i += 2) x[i] += a * (x[i - 1] + x[i + 1]);
    x[n - 1] += 2 * a * x[n - 2];
}

int khgo_fwt2d(double *img, int stride, int rows, int cols)
{
    int m = rows > cols ? rows : cols;
    double *tmp = (double *)malloc(sizeof(double) * 2 * (size_t)m);
    if (!tmp) return KHGO_ERR_NOMEM;
    double *col = tmp + m;
    /* DWT.h:121-125 rows */
    for (int i = 0; i < rows; i++) khgo_fwt97(img + (size_t)i * stride, cols, tmp);
    /* DWT.h:126-140 columns */
    for (int j = 0; j < cols; j++) {
        for (int i = 0; i < rows; i++) col[i] = img[(size_t)i * stride + j];
        khgo_fwt97(col, rows, tmp);
        for (int i = 0; i < rows; i++) img[(size_t)i * stride + j] = col[i];
    }
    free(tmp);
    return KHGO_OK;
}

int khgo_iwt2d(double *img, int stride, int rows, int cols)
{
    int m = rows > cols ? rows : cols;
    double *tmp = (double *)malloc(sizeof(double) * 2 * (size_t)m);
    if (!tmp) return KHGO_ERR_NOMEM;
    double *col = tmp + m;
    /* DWT.h:145-159 columns first */
    for (int j = 0; j < cols; j++) {
        for (int i = 0; i < rows; i++) col[i] = img[(size_t)i * stride + j];
        khgo_iwt97(col, rows, tmp);
        for (int i = 0; i < rows; i++) img[(size_t)i * stride + j] = col[i];
    }
    /* DWT.h:160-164 rows */
    for (int i = 0; i < rows; i++) khgo_iwt97(img + (size_t)i * stride, cols, tmp);
    free(tmp);
    return KHGO_OK;
}

int khgo_forward_dwt(double *y, double *cb, double *cr, int h, int w)
{
    int rc;
    /* Compress.h:108-110 */
    if ((rc = khgo_fwt2d(y, w, h, w))) return rc;
    if ((rc = khgo_fwt2d(cb, w / 2, h / 2, w / 2))) return rc;
    if ((rc = khgo_fwt2d(cr, w / 2, h / 2, w / 2))) return rc;
    /* Compress.h:112-114 */
    if ((rc = khgo_fwt2d(y, w, h / 2, w / 2))) return rc;
    if ((rc = khgo_fwt2d(cb, w / 2, h / 4, w / 4))) return rc;
    if ((rc = khgo_fwt2d(cr, w / 2, h / 4, w / 4))) return rc;
    return KHGO_OK;
}

int khgo_inverse_dwt(double *y, double *cb, double *cr, int h, int w)
{
    int rc;
    /* Decompress.h:274-276 */
    if ((rc = khgo_iwt2d(y, w, h / 2, w / 2))) return rc;
    if ((rc = khgo_iwt2d(cb, w / 2, h / 4, w / 4))) return rc;
    if ((rc = khgo_iwt2d(cr, w / 2, h / 4, w / 4))) return rc;
    /* Decompress.h:278-280 */
    if ((rc = khgo_iwt2d(y, w, h, w))) return rc;
    if ((rc = khgo_iwt2d(cb, w / 2, h / 2, w / 2))) return rc;
    if ((rc = khgo_iwt2d(cr, w / 2, h / 2, w / 2))) return rc;
    return KHGO_OK;
}

/* --------------------------------------------------------------- quantiser */

void khgo_divisor_plane(int h, int w, int lvl, int chroma, double *div)
{
    int h_half = h / 2, w_half = w / 2, h_qut = h / 4, w_qut = w / 4;
    /* Compress.h:154-160: {y0,y1,x0,x1} of LL1, LH1, HL1, HH1, LH, HL, HH */
    int map[7][4] = {
        {0, h_qut, 0, w_qut},           {0, h_qut, w_qut, w_half},      {h_qut, h_half, 0, w_qut},
        {h_qut, h_half, w_qut, w_half}, {0, h_half, w_half, w},         {h_half, h, 0, w_half},
        {h_half, h, w_half, w}};
    int pw = chroma ? w / 2 : w;
    for (int index = 0; index < 7; index++) {
        /* Compress.h:172-199 */
        double r = lvl == 0 ? 8 : lvl == 1 ? 10 : 12;
        double i = 2;
        double c = 8;
        double f = lvl == 0 ? 23 : lvl == 1 ? 230 : 2300;
        double tau = pow(2, r - c + i) * (1 + f / pow(2, 11));
        double step_size;
        if (index == 0) step_size = tau / pow(2, i);
        else if (index == 1 || index == 2) step_size = tau / pow(2, 2 - 1);
        else if (index == 3) step_size = tau / pow(2, 2 - 2);
        else if (index == 4 || index == 5) step_size = tau / pow(2, 1 - 1);
        else step_size = tau / pow(2, 1 - 2);

        int y0 = map[index][0], y1 = map[index][1], x0 = map[index][2], x1 = map[index][3];
        if (chroma) { /* Compress.h:223-230: integer halves of the luma rectangle */
            y0 /= 2; y1 /= 2; x0 /= 2; x1 /= 2;
        }
        /* Compress.h:206-209 / 223-226 */
        int alpha = y0 + x0;
        int beta = y1 + x1;
        double a = 1.5 / (double)(beta - alpha);
        double b = 0.5 - (double)(a * alpha);
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                double mul = a * (x + y) + b;                       /* Compress.h:215 / 232 */
                div[(size_t)y * pw + x] = step_size * (lvl >= 1 ? mul : 1); /* Compress.h:216-218 */
            }
    }
}

void khgo_quantise(const double *coef, const double *div, size_t n, int32_t *q)
{
    for (size_t i = 0; i < n; i++) q[i] = (int)round(coef[i] / div[i]); /* Compress.h:218 */
}

void khgo_dequantise(const int32_t *q, const double *div, size_t n, double *out)
{
    for (size_t i = 0; i < n; i++) out[i] = q[i] * div[i]; /* Decompress.h:105 */
}

/* ------------------------------------------------------------ symbol table */

typedef struct { uint64_t first; int32_t sym; int32_t cnt; } khgo_ent;

static int khgo_ent_cmp(const void *a, const void *b)
{
    const khgo_ent *x = (const khgo_ent *)a, *y = (const khgo_ent *)b;
    return x->first < y->first ? -1 : x->first > y->first ? 1 : 0;
}

long khgo_symbol_table(const int32_t *q, size_t n, int32_t *counts, int32_t *symbols, int faithful)
{
    long k = 0;
    if (n == 0) return 0;
    if (faithful) {
        /* Huffman.h:113-136 verbatim search order */
        for (size_t i = 0; i < n; i++) {
            long index = -1;
            for (long j = 0; j < k; j++)
                if (symbols[j] == q[i]) { index = j; break; }
            if (index == -1) { counts[k] = 1; symbols[k] = q[i]; k++; }
            else counts[index]++;
        }
        return k;
    }
    int32_t lo = q[0], hi = q[0];
    for (size_t i = 1; i < n; i++) { if (q[i] < lo) lo = q[i]; if (q[i] > hi) hi = q[i]; }
    uint64_t range = (uint64_t)((int64_t)hi - (int64_t)lo) + 1;
    if (range > ((uint64_t)1 << 26)) return khgo_symbol_table(q, n, counts, symbols, 1);
    int32_t *cnt = (int32_t *)calloc(range, sizeof(int32_t));
    uint64_t *first = (uint64_t *)malloc(range * sizeof(uint64_t));
    if (!cnt || !first) { free(cnt); free(first); return KHGO_ERR_NOMEM; }
    for (size_t i = 0; i < n; i++) {
        size_t s = (size_t)((int64_t)q[i] - lo);
        if (cnt[s]++ == 0) first[s] = i;
    }
    for (uint64_t s = 0; s < range; s++) if (cnt[s]) k++;
    khgo_ent *e = (khgo_ent *)malloc(sizeof(khgo_ent) * (size_t)k);
    if (!e) { free(cnt); free(first); return KHGO_ERR_NOMEM; }
    long m = 0;
    for (uint64_t s = 0; s < range; s++)
        if (cnt[s]) { e[m].first = first[s]; e[m].sym = (int32_t)((int64_t)s + lo); e[m].cnt = cnt[s]; m++; }
    qsort(e, (size_t)k, sizeof(khgo_ent), khgo_ent_cmp);
    for (long i = 0; i < k; i++) { counts[i] = e[i].cnt; symbols[i] = e[i].sym; }
    free(e); free(cnt); free(first);
    return k;
}

/* ----------------------------------------------------- Huffman tree + codes */

typedef struct {
    uint32_t *freq;   /* HuffmanNode::freq_ (unsigned, Huffman.h:12) */
    int32_t *left, *right;
    int32_t *heap;    /* node indices, std::vector<HuffmanNode*> of the priority_queue */
    long len;
} khgo_pq;

/* Huffman.h:25-31: comp(l, r) = l->freq_ > r->freq_ */
static inline int khgo_comp(const khgo_pq *p, int32_t l, int32_t r) { return p->freq[l] > p->freq[r]; }

/* bits/stl_heap.h __push_heap */
static void khgo_sift_up(khgo_pq *p, long hole, long top, int32_t value)
{
    long parent = (hole - 1) / 2;
    while (hole > top && khgo_comp(p, p->heap[parent], value)) {
        p->heap[hole] = p->heap[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    p->heap[hole] = value;
}

/* priority_queue::push = push_back + push_heap */
static void khgo_push(khgo_pq *p, int32_t node)
{
    p->heap[p->len++] = node;
    khgo_sift_up(p, p->len - 1, 0, node);
}

/* priority_queue::top + pop = pop_heap (__pop_heap + __adjust_heap) + pop_back */
static int32_t khgo_pop(khgo_pq *p)
{
    int32_t top = p->heap[0];
    if (p->len > 1) {
        long len = p->len - 1;          /* __pop_heap: range [first, last-1) */
        int32_t value = p->heap[len];
        p->heap[len] = p->heap[0];
        long hole = 0, child = 0;       /* __adjust_heap(first, 0, len, value) */
        while (child < (len - 1) / 2) {
            child = 2 * (child + 1);
            if (khgo_comp(p, p->heap[child], p->heap[child - 1])) child--;
            p->heap[hole] = p->heap[child];
            hole = child;
        }
        if ((len & 1) == 0 && child == (len - 2) / 2) {
            child = 2 * (child + 1);
            p->heap[hole] = p->heap[child - 1];
            hole = child - 1;
        }
        khgo_sift_up(p, hole, 0, value);
    }
    p->len--;
    return top;
}

int khgo_build_codes(const int32_t *counts, const int32_t *symbols, long k, uint64_t *codes, uint8_t *lens)
{
    (void)symbols;
    if (k <= 0) return KHGO_ERR_FORMAT;
    long nn = 2 * k - 1;
    khgo_pq p;
    p.freq = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)nn);
    p.left = (int32_t *)malloc(sizeof(int32_t) * (size_t)nn);
    p.right = (int32_t *)malloc(sizeof(int32_t) * (size_t)nn);
    p.heap = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
    int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (size_t)nn);
    uint64_t *scode = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nn);
    int32_t *slen = (int32_t *)malloc(sizeof(int32_t) * (size_t)nn);
    int rc = KHGO_OK;
    if (!p.freq || !p.left || !p.right || !p.heap || !stack || !scode || !slen) { rc = KHGO_ERR_NOMEM; goto done; }
    p.len = 0;
    /* Huffman.h:91-94: leaves pushed in table order */
    for (long i = 0; i < k; i++) {
        p.freq[i] = (uint32_t)counts[i];
        p.left[i] = p.right[i] = -1;
        khgo_push(&p, (int32_t)i);
    }
    long next = k;
    /* Huffman.h:96-105 */
    while (p.len != 1) {
        int32_t right = khgo_pop(&p);
        int32_t left = khgo_pop(&p);
        int sum = (int)(p.freq[left] + p.freq[right]);
        p.freq[next] = (uint32_t)sum;
        p.left[next] = left;
        p.right[next] = right;
        khgo_push(&p, (int32_t)next);
        next++;
    }
    /* Huffman.h:57-68: depth-first, left appends '0', right appends '1' */
    {
        long sp = 0;
        stack[sp] = p.heap[0]; scode[sp] = 0; slen[sp] = 0; sp++;
        while (sp) {
            sp--;
            int32_t nd = stack[sp]; uint64_t cd = scode[sp]; int32_t ln = slen[sp];
            if (p.left[nd] < 0 && p.right[nd] < 0) {
                if (ln > 64) { rc = KHGO_ERR_RANGE; goto done; }
                codes[nd] = cd; lens[nd] = (uint8_t)ln;
                continue;
            }
            stack[sp] = p.right[nd]; scode[sp] = (cd << 1) | 1; slen[sp] = ln + 1; sp++;
            stack[sp] = p.left[nd]; scode[sp] = cd << 1; slen[sp] = ln + 1; sp++;
        }
    }
done:
    free(p.freq); free(p.left); free(p.right); free(p.heap); free(stack); free(scode); free(slen);
    return rc;
}

/* Rebuild the tree arrays for the decoder (same construction as above). */
static int khgo_build_tree(const int32_t *counts, long k, int32_t **left_out, int32_t **right_out, int32_t *root)
{
    long nn = 2 * k - 1;
    khgo_pq p;
    p.freq = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)nn);
    p.left = (int32_t *)malloc(sizeof(int32_t) * (size_t)nn);
    p.right = (int32_t *)malloc(sizeof(int32_t) * (size_t)nn);
    p.heap = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
    if (!p.freq || !p.left || !p.right || !p.heap) {
        free(p.freq); free(p.left); free(p.right); free(p.heap);
        return KHGO_ERR_NOMEM;
    }
    p.len = 0;
    for (long i = 0; i < k; i++) {
        p.freq[i] = (uint32_t)counts[i];
        p.left[i] = p.right[i] = -1;
        khgo_push(&p, (int32_t)i);
    }
    long next = k;
    while (p.len != 1) {
        int32_t right = khgo_pop(&p);
        int32_t left = khgo_pop(&p);
        int sum = (int)(p.freq[left] + p.freq[right]);
        p.freq[next] = (uint32_t)sum;
        p.left[next] = left;
        p.right[next] = right;
        khgo_push(&p, (int32_t)next);
        next++;
    }
    *root = p.heap[0];
    *left_out = p.left;
    *right_out = p.right;
    free(p.freq); free(p.heap);
    return KHGO_OK;
}

/* ------------------------------------------------------------- bit streams */

typedef struct { uint8_t *buf; size_t cap; uint64_t bits; int overflow; } khgo_bw;

/* Huffman.h:80 + Writer.h:25-32: codes concatenated, 8 chars -> one byte, MSB first */
static inline void khgo_put(khgo_bw *w, uint64_t code, int len)
{
    for (int i = len - 1; i >= 0; i--) {
        size_t byte = (size_t)(w->bits >> 3);
        if (byte >= w->cap) { w->overflow = 1; w->bits++; continue; }
        if ((w->bits & 7) == 0) w->buf[byte] = 0;
        if ((code >> i) & 1) w->buf[byte] |= (uint8_t)(0x80u >> (w->bits & 7));
        w->bits++;
    }
}

static void khgo_wr32(uint8_t *p, int32_t v)
{
    uint32_t u = (uint32_t)v;
    p[0] = (uint8_t)u; p[1] = (uint8_t)(u >> 8); p[2] = (uint8_t)(u >> 16); p[3] = (uint8_t)(u >> 24);
}
static int32_t khgo_rd32(const uint8_t *p)
{
    return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
}

static int khgo_check_dims(int h, int w, int lvl)
{
    if (h <= 0 || w <= 0 || (h % 8) || (w % 8) || lvl < 0 || lvl > 2) return KHGO_ERR_DIMS;
    return KHGO_OK;
}

/* ------------------------------------------------------------------ encode */

int khgo_forward_planes(const uint8_t *r, const uint8_t *g, const uint8_t *b, int h, int w,
                        double *y, double *cb, double *cr)
{
    int rc = khgo_check_dims(h, w, 0);
    if (rc) return rc;
    khgo_rgb_to_ycbcr(r, g, b, h, w, y, cb, cr);
    return khgo_forward_dwt(y, cb, cr, h, w);
}

int khgo_symbols(const uint8_t *r, const uint8_t *g, const uint8_t *b, int h, int w, int lvl,
                 int32_t *qy, int32_t *qcb, int32_t *qcr)
{
    int rc = khgo_check_dims(h, w, lvl);
    if (rc) return rc;
    size_t ny = (size_t)h * w, nc = ny / 4;
    double *y = (double *)malloc(sizeof(double) * (ny + 2 * nc));
    double *div = (double *)malloc(sizeof(double) * ny);
    if (!y || !div) { free(y); free(div); return KHGO_ERR_NOMEM; }
    double *cb = y + ny, *cr = cb + nc;
    rc = khgo_forward_planes(r, g, b, h, w, y, cb, cr);
    if (!rc) {
        khgo_divisor_plane(h, w, lvl, 0, div);
        khgo_quantise(y, div, ny, qy);
        khgo_divisor_plane(h, w, lvl, 1, div);
        khgo_quantise(cb, div, nc, qcb);
        khgo_quantise(cr, div, nc, qcr);
    }
    free(y); free(div);
    return rc;
}

typedef struct {
    long k; int32_t *counts, *symbols; uint64_t *codes; uint8_t *lens;
    uint64_t bits; int pad;
} khgo_plane_enc;

static int khgo_encode_plane(const int32_t *q, size_t n, khgo_plane_enc *pe, khgo_bw *bw, int faithful)
{
    pe->counts = (int32_t *)malloc(sizeof(int32_t) * n);
    pe->symbols = (int32_t *)malloc(sizeof(int32_t) * n);
    if (!pe->counts || !pe->symbols) return KHGO_ERR_NOMEM;
    pe->k = khgo_symbol_table(q, n, pe->counts, pe->symbols, faithful); /* Compress.h:290-292 */
    if (pe->k < 0) return (int)pe->k;
    pe->codes = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)pe->k);
    pe->lens = (uint8_t *)malloc((size_t)pe->k);
    if (!pe->codes || !pe->lens) return KHGO_ERR_NOMEM;
    int rc = khgo_build_codes(pe->counts, pe->symbols, pe->k, pe->codes, pe->lens); /* Compress.h:302-304 */
    if (rc) return rc;
    /* symbol -> table slot (std::map<int,std::string> of Huffman.h:72) */
    int32_t lo = pe->symbols[0], hi = pe->symbols[0];
    for (long i = 1; i < pe->k; i++) {
        if (pe->symbols[i] < lo) lo = pe->symbols[i];
        if (pe->symbols[i] > hi) hi = pe->symbols[i];
    }
    size_t range = (size_t)((int64_t)hi - lo) + 1;
    int32_t *slot = (int32_t *)malloc(sizeof(int32_t) * range);
    if (!slot) return KHGO_ERR_NOMEM;
    for (long i = 0; i < pe->k; i++) slot[(size_t)((int64_t)pe->symbols[i] - lo)] = (int32_t)i;
    uint64_t start = bw->bits;
    for (size_t i = 0; i < n; i++) { /* Huffman.h:76-82 raster order */
        int32_t s = slot[(size_t)((int64_t)q[i] - lo)];
        khgo_put(bw, pe->codes[s], pe->lens[s]);
    }
    free(slot);
    pe->bits = bw->bits - start;
    pe->pad = 0;
    while ((bw->bits - start) % 8 != 0) { khgo_put(bw, 0, 1); pe->pad++; } /* Compress.h:314-328 */
    pe->bits = bw->bits - start;
    return KHGO_OK;
}

int khgo_encode(const uint8_t *r, const uint8_t *g, const uint8_t *b, int h, int w, int lvl,
                uint8_t *out, size_t cap, size_t *out_len, int faithful)
{
    int rc = khgo_check_dims(h, w, lvl);
    if (rc) return rc;
    if (cap < 36) return KHGO_ERR_CAP;
    size_t ny = (size_t)h * w, nc = ny / 4;
    int32_t *qy = (int32_t *)malloc(sizeof(int32_t) * (ny + 2 * nc));
    if (!qy) return KHGO_ERR_NOMEM;
    int32_t *qcb = qy + ny, *qcr = qcb + nc;
    khgo_plane_enc pe[3];
    memset(pe, 0, sizeof pe);
    rc = khgo_symbols(r, g, b, h, w, lvl, qy, qcb, qcr);
    khgo_bw bw = {out + 36, cap - 36, 0, 0};
    if (!rc) rc = khgo_encode_plane(qy, ny, &pe[0], &bw, faithful);
    if (!rc) rc = khgo_encode_plane(qcb, nc, &pe[1], &bw, faithful);
    if (!rc) rc = khgo_encode_plane(qcr, nc, &pe[2], &bw, faithful);
    if (!rc) {
        for (int p = 0; p < 3; p++) if (pe[p].bits >= ((uint64_t)1 << 31)) rc = KHGO_ERR_RANGE;
    }
    if (!rc) {
        size_t stream = (size_t)(bw.bits >> 3);
        size_t total = 36 + stream + 8 * (size_t)(pe[0].k + pe[1].k + pe[2].k);
        if (out_len) *out_len = total;
        if (bw.overflow || total > cap) rc = KHGO_ERR_CAP;
        else {
            /* Writer.h:9-23; Compress.h:393-403 passes (img_height, img_width) into the
             * (w, h) constructor slots of Type.h:18, so field h (offset 24) receives the
             * column count and field w (offset 28) the row count -- equal for the square
             * images the reference supports.  Rectangular: offset 24 = rows, 28 = cols. */
            khgo_wr32(out + 0, (int32_t)pe[0].bits);
            khgo_wr32(out + 4, (int32_t)pe[1].bits);
            khgo_wr32(out + 8, (int32_t)pe[2].bits);
            khgo_wr32(out + 12, (int32_t)pe[0].k);
            khgo_wr32(out + 16, (int32_t)pe[1].k);
            khgo_wr32(out + 20, (int32_t)pe[2].k);
            khgo_wr32(out + 24, h);
            khgo_wr32(out + 28, w);
            out[32] = (uint8_t)pe[0].pad; out[33] = (uint8_t)pe[1].pad; out[34] = (uint8_t)pe[2].pad;
            out[35] = (uint8_t)lvl;
            /* Writer.h:34-41: (int32 count, int32 symbol) per entry */
            uint8_t *t = out + 36 + stream;
            for (int p = 0; p < 3; p++)
                for (long i = 0; i < pe[p].k; i++) {
                    khgo_wr32(t, pe[p].counts[i]); khgo_wr32(t + 4, pe[p].symbols[i]); t += 8;
                }
        }
    }
    for (int p = 0; p < 3; p++) { free(pe[p].counts); free(pe[p].symbols); free(pe[p].codes); free(pe[p].lens); }
    free(qy);
    return rc;
}

/* ------------------------------------------------------------------ decode */

int khgo_peek(const uint8_t *comp, size_t len, khgo_header *hdr)
{
    if (len < 36) return KHGO_ERR_FORMAT;
    /* Reader.h:5-23 */
    hdr->y_len = khgo_rd32(comp); hdr->cb_len = khgo_rd32(comp + 4); hdr->cr_len = khgo_rd32(comp + 8);
    hdr->rle_y_len = khgo_rd32(comp + 12); hdr->rle_cb_len = khgo_rd32(comp + 16); hdr->rle_cr_len = khgo_rd32(comp + 20);
    hdr->h = khgo_rd32(comp + 24); hdr->w = khgo_rd32(comp + 28);
    hdr->y_pad = comp[32]; hdr->cb_pad = comp[33]; hdr->cr_pad = comp[34]; hdr->lvl = comp[35];
    if (khgo_check_dims(hdr->h, hdr->w, hdr->lvl)) return KHGO_ERR_FORMAT;
    if (hdr->y_len < 0 || hdr->cb_len < 0 || hdr->cr_len < 0 || (hdr->y_len % 8) || (hdr->cb_len % 8) || (hdr->cr_len % 8))
        return KHGO_ERR_FORMAT;
    if (hdr->y_pad > 7 || hdr->cb_pad > 7 || hdr->cr_pad > 7) return KHGO_ERR_FORMAT;
    if (hdr->y_pad > hdr->y_len || hdr->cb_pad > hdr->cb_len || hdr->cr_pad > hdr->cr_len) return KHGO_ERR_FORMAT;
    if (hdr->rle_y_len < 1 || hdr->rle_cb_len < 1 || hdr->rle_cr_len < 1) return KHGO_ERR_FORMAT;
    uint64_t need = 36 + ((uint64_t)hdr->y_len + hdr->cb_len + hdr->cr_len) / 8 +
                    8 * ((uint64_t)hdr->rle_y_len + hdr->rle_cb_len + hdr->rle_cr_len);
    if (need > len) return KHGO_ERR_FORMAT;
    return KHGO_OK;
}

/* Huffman.h:33-55: bit-serial tree walk; symbol i lands at [i / n][i % n] (flat index i). */
static int khgo_decode_plane(const uint8_t *bits, uint64_t nbits, const uint8_t *table, long k,
                             size_t n, int32_t *q)
{
    memset(q, 0, sizeof(int32_t) * n); /* Huffman.h:35 zero-initialised */
    int32_t *counts = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
    int32_t *symbols = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
    if (!counts || !symbols) { free(counts); free(symbols); return KHGO_ERR_NOMEM; }
    for (long i = 0; i < k; i++) { counts[i] = khgo_rd32(table + 8 * i); symbols[i] = khgo_rd32(table + 8 * i + 4); }
    int32_t *left, *right, root;
    int rc = khgo_build_tree(counts, k, &left, &right, &root); /* Decompress.h:45-47 */
    if (rc) { free(counts); free(symbols); return rc; }
    size_t index = 0;
    int32_t curr = root;
    if (k == 1 && nbits) rc = KHGO_ERR_FORMAT; /* the reference would dereference null here */
    for (uint64_t i = 0; i < nbits && !rc; i++) {
        int bit = (bits[i >> 3] >> (7 - (i & 7))) & 1;
        curr = bit ? right[curr] : left[curr];
        if (left[curr] < 0 && right[curr] < 0) {
            if (index >= n) { rc = KHGO_ERR_FORMAT; break; } /* the reference would write out of bounds */
            q[index++] = symbols[curr];
            curr = root;
        }
    }
    free(left); free(right); free(counts); free(symbols);
    return rc;
}

int khgo_decode_planes(const uint8_t *comp, size_t len, int32_t *qy, int32_t *qcb, int32_t *qcr,
                       double *y, double *cb, double *cr)
{
    khgo_header hd;
    int rc = khgo_peek(comp, len, &hd);
    if (rc) return rc;
    int h = hd.h, w = hd.w;
    size_t ny = (size_t)h * w, nc = ny / 4;
    const uint8_t *sy = comp + 36, *scb = sy + hd.y_len / 8, *scr = scb + hd.cb_len / 8;
    const uint8_t *ty = scr + hd.cr_len / 8, *tcb = ty + 8 * (size_t)hd.rle_y_len, *tcr = tcb + 8 * (size_t)hd.rle_cb_len;
    /* Decompress.h:37-39 strip the pad bits, 49-51 decode */
    rc = khgo_decode_plane(sy, (uint64_t)(hd.y_len - hd.y_pad), ty, hd.rle_y_len, ny, qy);
    if (!rc) rc = khgo_decode_plane(scb, (uint64_t)(hd.cb_len - hd.cb_pad), tcb, hd.rle_cb_len, nc, qcb);
    if (!rc) rc = khgo_decode_plane(scr, (uint64_t)(hd.cr_len - hd.cr_pad), tcr, hd.rle_cr_len, nc, qcr);
    if (rc) return rc;
    double *div = (double *)malloc(sizeof(double) * ny);
    if (!div) return KHGO_ERR_NOMEM;
    khgo_divisor_plane(h, w, hd.lvl, 0, div);
    khgo_dequantise(qy, div, ny, y);
    khgo_divisor_plane(h, w, hd.lvl, 1, div);
    khgo_dequantise(qcb, div, nc, cb);
    khgo_dequantise(qcr, div, nc, cr);
    free(div);
    return khgo_inverse_dwt(y, cb, cr, h, w);
}

int khgo_decode(const uint8_t *comp, size_t len, uint8_t *r, uint8_t *g, uint8_t *b)
{
    khgo_header hd;
    int rc = khgo_peek(comp, len, &hd);
    if (rc) return rc;
    size_t ny = (size_t)hd.h * hd.w, nc = ny / 4;
    int32_t *q = (int32_t *)malloc(sizeof(int32_t) * (ny + 2 * nc));
    double *y = (double *)malloc(sizeof(double) * (ny + 2 * nc));
    if (!q || !y) { free(q); free(y); return KHGO_ERR_NOMEM; }
    rc = khgo_decode_planes(comp, len, q, q + ny, q + ny + nc, y, y + ny, y + ny + nc);
    if (!rc) khgo_ycbcr_to_rgb(y, y + ny, y + ny + nc, hd.h, hd.w, r, g, b);
    free(q); free(y);
    return rc;
}

/* ----------------------------------------------------------------- metrics */

/* SSIM.h:42-55 on one 8x8 block */
static double khgo_ssim_block(const uint8_t *a, const uint8_t *b, int stride)
{
    double L = 255.0;
    double K1 = 0.01, K2 = 0.03;
    double C1 = (K1 * L) * (K1 * L), C2 = (K2 * L) * (K2 * L);
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) { s1 += a[i * stride + j]; s2 += b[i * stride + j]; }
    double avg1 = s1 / 64.0, avg2 = s2 / 64.0;
    double v1 = 0.0, v2 = 0.0, cv = 0.0;
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) {
        v1 += (a[i * stride + j] - avg1) * (a[i * stride + j] - avg1);
        v2 += (b[i * stride + j] - avg2) * (b[i * stride + j] - avg2);
        cv += (a[i * stride + j] - avg1) * (b[i * stride + j] - avg2);
    }
    double var1 = v1 / 64.0, var2 = v2 / 64.0, cov = cv / 64.0;
    return (2.0 * avg1 * avg2 + C1) * (2.0 * cov + C2) / ((avg1 * avg1 + avg2 * avg2 + C1) * (var1 + var2 + C2));
}

long khgo_ssim_blocks(const uint8_t *a, const uint8_t *b, int width, int height, double *out)
{
    long n = 0;
    for (int i = 0; i <= height - 8; i += 8)      /* SSIM.h:60 */
        for (int j = 0; j <= width - 8; j += 8) { /* SSIM.h:61 */
            double v = khgo_ssim_block(a + (size_t)i * width + j, b + (size_t)i * width + j, width);
            if (out) out[n] = v;
            n++;
        }
    return n;
}

double khgo_ssim(const uint8_t *a, const uint8_t *b, int width, int height)
{
    double sum = 0.0; /* SSIM.h:68-71: sequential sum in raster order */
    long n = 0;
    for (int i = 0; i <= height - 8; i += 8)
        for (int j = 0; j <= width - 8; j += 8) {
            sum += khgo_ssim_block(a + (size_t)i * width + j, b + (size_t)i * width + j, width);
            n++;
        }
    return sum / n;
}

static double khgo_mse(const uint8_t *I, const uint8_t *O, int nW, int nH)
{
    double Mse = 0; /* KhuGleBase.cpp:604-610 */
    for (int y = 0; y < nH; ++y)
        for (int x = 0; x < nW; ++x) {
            size_t p = (size_t)y * nW + x;
            Mse += (I[p] - O[p]) * (I[p] - O[p]);
        }
    Mse /= nW * nH;
    return Mse;
}

double khgo_psnr(const uint8_t *ir, const uint8_t *ig, const uint8_t *ib,
                 const uint8_t *or_, const uint8_t *og, const uint8_t *ob, int w, int h)
{
    /* KhuGleBase.cpp:613-627 */
    double MseR = khgo_mse(ir, or_, w, h), MseG = khgo_mse(ig, og, w, h), MseB = khgo_mse(ib, ob, w, h);
    double Mse = (MseR + MseG + MseB) / 3.;
    if (Mse == 0) return 100.;
    return 10 * log10(255 * 255 / Mse);
}

/* --------------------------------------- the application's by-product pictures */

/* Compress.h:245-300 (Decompress.h:190-232, 265-271 are the same statements): the step sizes in red, their
 * extremes in the first status line, the table sizes in the second.  str / str2: at least 100 bytes each (the
 * reference's buffer), or NULL. */
int khgo_vis_steps(int h, int w, int lvl, long k_y, long k_cb, long k_cr, uint8_t *r, uint8_t *g, uint8_t *b,
                   char *str, char *str2)
{
    int rc = khgo_check_dims(h, w, lvl);
    if (rc) return rc;
    size_t n = (size_t)h * w;
    double *step_size_2d = (double *)malloc(sizeof(double) * n);
    if (!step_size_2d) return KHGO_ERR_NOMEM;
    khgo_divisor_plane(h, w, lvl, 0, step_size_2d); /* Compress.h:216: step_size * (lvl >= 1 ? mul : 1) */
    /* step_size_vec: Compress.h:172-199 */
    double step_size_vec[7];
    int h_half = h / 2, w_half = w / 2, h_qut = h / 4, w_qut = w / 4;
    int map[7][4] = {
        {0, h_qut, 0, w_qut},           {0, h_qut, w_qut, w_half},      {h_qut, h_half, 0, w_qut},
        {h_qut, h_half, w_qut, w_half}, {0, h_half, w_half, w},         {h_half, h, 0, w_half},
        {h_half, h, w_half, w}};
    for (int index = 0; index < 7; index++) {
        double rr = lvl == 0 ? 8 : lvl == 1 ? 10 : 12;
        double i = 2;
        double c = 8;
        double f = lvl == 0 ? 23 : lvl == 1 ? 230 : 2300;
        double tau = pow(2, rr - c + i) * (1 + f / pow(2, 11));
        double step_size;
        if (index == 0) step_size = tau / pow(2, i);
        else if (index == 1 || index == 2) step_size = tau / pow(2, 2 - 1);
        else if (index == 3) step_size = tau / pow(2, 2 - 2);
        else if (index == 4 || index == 5) step_size = tau / pow(2, 1 - 1);
        else step_size = tau / pow(2, 1 - 2);
        step_size_vec[index] = step_size;
    }
    /* Compress.h:259-263 */
    double max_step_size = step_size_vec[0], min_step_size = step_size_vec[0];
    for (int i = 1; i < 7; i++) {
        if (step_size_vec[i] > max_step_size) max_step_size = step_size_vec[i];
        if (step_size_vec[i] < min_step_size) min_step_size = step_size_vec[i];
    }
    double max_step_size_2d = step_size_2d[0], min_step_size_2d = step_size_2d[0];
    for (size_t i = 1; i < n; i++) {
        if (step_size_2d[i] > max_step_size_2d) max_step_size_2d = step_size_2d[i];
        if (step_size_2d[i] < min_step_size_2d) min_step_size_2d = step_size_2d[i];
    }
    if (r && g && b) {
        for (int index = 0; index < 7; index++)
            for (int y = map[index][0]; y < map[index][1]; y++)
                for (int x = map[index][2]; x < map[index][3]; x++) {
                    double nn = 0.0; /* Compress.h:274-288 */
                    if (lvl >= 1) nn = (step_size_2d[(size_t)y * w + x] - min_step_size_2d) / max_step_size_2d;
                    else nn = (step_size_vec[index] - min_step_size) / max_step_size;
                    nn *= 255;
                    r[(size_t)y * w + x] = (unsigned char)nn;
                    g[(size_t)y * w + x] = 0;
                    b[(size_t)y * w + x] = 0;
                }
    }
    if (str) /* Compress.h:294-296 */
        sprintf(str, "Step Size (min, max) : %.3f, %.3f", lvl >= 1 ? min_step_size_2d : min_step_size,
                lvl >= 1 ? max_step_size_2d : max_step_size);
    if (str2) /* Compress.h:299-300 */
        sprintf(str2, "RLE Size (Y, Cb, Cr) : %d, %d, %d", (int)k_y, (int)k_cb, (int)k_cr);
    free(step_size_2d);
    return 0;
}

/* Compress.h:330-391 / Decompress.h:128-188: the three stream lengths (bytes) as bars, in unsigned char arithmetic.
 * The reference writes whatever rows and columns its quotient and remainder by the image HEIGHT give, also outside
 * the picture (undefined there); this restatement skips those positions. */
int khgo_vis_bars(int h, int w, const long long *len, uint8_t *r, uint8_t *g, uint8_t *b)
{
    static const int add[3][3] = {{76, 149, 29}, {-43, -84, 127}, {127, -106, -20}};
    for (size_t i = 0; i < (size_t)h * w; i++) r[i] = g[i] = b[i] = 0;
    for (int c = 0; c < 3; c++) {
        long long c_y = len[c] / h, c_x = len[c] % h;
        for (long long y = 0; y < c_y; y++) {
            long long max_x = (y == c_y - 1) ? c_x : w;
            for (long long x = 0; x < max_x; x++) {
                if (y >= h || x >= w) continue;
                size_t p = (size_t)y * w + (size_t)x;
                r[p] = (unsigned char)(r[p] + add[c][0]);
                g[p] = (unsigned char)(g[p] + add[c][1]);
                b[p] = (unsigned char)(b[p] + add[c][2]);
            }
        }
    }
    return 0;
}

/* DownSample.h:5-37: bilinear thumbnail of one plane, single precision.  The unsigned char conversion of a NaN or an
 * infinity (division by zero on the source's last row / column) is undefined in C; the reference's x86-64 build
 * yields 0 there (cvttss2si -> INT_MIN, low byte), and so does this restatement. */
static unsigned char khgo_f32_to_u8(float v)
{
    if (!(v > -2147483904.0f && v < 2147483648.0f)) return 0;
    return (unsigned char)(unsigned int)(int)v;
}

int khgo_downsample(const uint8_t *src, int srcHeight, int srcWidth, uint8_t *dst, int dstHeight, int dstWidth)
{
    const float scaleX = (float)srcWidth / (float)dstWidth;   /* DownSample.h:29 */
    const float scaleY = (float)srcHeight / (float)dstHeight; /* DownSample.h:30 */
    for (int yy = 0; yy < dstHeight; ++yy)
        for (int xx = 0; xx < dstWidth; ++xx) {
            volatile float x = xx * scaleX, y = yy * scaleY; /* DownSample.h:34 (float arguments) */
            int x1 = (int)floor(x);                          /* DownSample.h:7 */
            int y1 = (int)floor(y);
            int x2 = KHGO_MIN(x1 + 1, srcWidth - 1);
            int y2 = KHGO_MIN(y1 + 1, srcHeight - 1);
            unsigned char Q11 = src[(size_t)y1 * srcWidth + x1], Q21 = src[(size_t)y1 * srcWidth + x2];
            unsigned char Q12 = src[(size_t)y2 * srcWidth + x1], Q22 = src[(size_t)y2 * srcWidth + x2];
            float R1 = ((x2 - x) / (x2 - x1)) * Q11 + ((x - x1) / (x2 - x1)) * Q21; /* DownSample.h:17 */
            float R2 = ((x2 - x) / (x2 - x1)) * Q12 + ((x - x1) / (x2 - x1)) * Q22; /* DownSample.h:18 */
            float P = ((y2 - y) / (y2 - y1)) * R1 + ((y - y1) / (y2 - y1)) * R2;    /* DownSample.h:20 */
            dst[(size_t)yy * dstWidth + xx] = khgo_f32_to_u8(P);
        }
    return 0;
}
