// khg_metrics.h -- SSIM over non-overlapping 8x8 blocks (SSIM.h:5-72) and the squared-error
// sums behind GetPsnr (KhuGleBase.cpp:602-627).
#pragma once
#include "khg_common.h"

struct KhgSsimArgs {
    const uint8_t *a, *b;  // two planes, width x height, row-major
    int width, height;
    double* blocks;        // [(height/8) * (width/8)] per-block SSIM in raster order
};

// Dot product of the four bytes of two words.
#ifdef KHG_SIM
KHG_DEV uint32_t khg_dot4(uint32_t x, uint32_t y)
{
    uint32_t s = 0;
    for (int i = 0; i < 4; ++i) s += ((x >> (8 * i)) & 255u) * ((y >> (8 * i)) & 255u);
    return s;
}
#else
KHG_DEV uint32_t khg_dot4(uint32_t x, uint32_t y) { return __builtin_amdgcn_udot4(x, y, 0u, false); }
#endif

// One thread per block.  All sums are exact in f64 (integers, or multiples of 1/4096 below 2^28),
// so only the final expression of SSIM.h:52-53 rounds, in the reference's order.
KHG_DEV void khg_ssim_body(const KhgSsimArgs& a, uint32_t bid)
{
    const int bw = a.width / 8, bh = a.height / 8;
    KHG_FOR_THREADS(256)
    {
        size_t blk = (size_t)bid * 256 + tid;
        if (blk < (size_t)bw * bh) {
            int by = (int)(blk / bw), bx = (int)(blk - (size_t)by * bw);
            const uint8_t* pa = a.a + (size_t)by * 8 * a.width + bx * 8;
            const uint8_t* pb = a.b + (size_t)by * 8 * a.width + bx * 8;
            // Every f64 value the reference forms up to here is exact: the sums of the 64 values (integers), the means
            // (over 64), the differences from them (multiples of 1/64 below 256), their squares and products
            // (multiples of 1/4096 below 2^16) and the running sums of those (below 2^22).  Exact values do not depend
            // on how they were added up, so they are taken from integer dot products of the rows' bytes:
            //   sum (p - m)^2 = (64 sum p^2 - (sum p)^2) / 64,   sum (p - m)(q - n) = (64 sum p q - sum p sum q) / 64.
            uint32_t sa = 0, sb = 0, saa = 0, sbb = 0, sab = 0;
            KHG_UNROLL
            for (int i = 0; i < 8; ++i) {
                uint32_t x[2], y[2];
                __builtin_memcpy(x, pa + (size_t)i * a.width, 8);
                __builtin_memcpy(y, pb + (size_t)i * a.width, 8);
                KHG_UNROLL
                for (int h = 0; h < 2; ++h) {
                    sa += khg_dot4(x[h], 0x01010101u);
                    sb += khg_dot4(y[h], 0x01010101u);
                    saa += khg_dot4(x[h], x[h]);
                    sbb += khg_dot4(y[h], y[h]);
                    sab += khg_dot4(x[h], y[h]);
                }
            }
            const double s1 = (double)sa, s2 = (double)sb;
            double avg1 = s1 / 64.0, avg2 = s2 / 64.0;  // SSIM.h:14-21
            const double v1 = (double)(64u * saa - sa * sa) / 64.0;  // SSIM.h:23-30
            const double v2 = (double)(64u * sbb - sb * sb) / 64.0;
            const double cv = (double)((long long)(64u * sab) - (long long)(sa * sb)) / 64.0;  // SSIM.h:32-40
            double var1 = v1 / 64.0, var2 = v2 / 64.0, cov = cv / 64.0;
            double L = 255.0;
            double K1 = 0.01, K2 = 0.03;
            double C1 = (K1 * L) * (K1 * L), C2 = (K2 * L) * (K2 * L);
            a.blocks[blk] = (2.0 * avg1 * avg2 + C1) * (2.0 * cov + C2) /
                            ((avg1 * avg1 + avg2 * avg2 + C1) * (var1 + var2 + C2));  // SSIM.h:52-53
        }
    }
}

// ---- Mean of the per-block values (SSIM.h:66-71).  The reference adds them one after another in raster order, and
// a double sum depends on its order.  Adding one value at a time (one lane, one dependent addition per value) costs
// 8.5 ns per value; instead the additions are REPRODUCED, rounding for rounding, a window of 4096 values at a time:
//   While the running sum s stays in one binade, s = +-S * u with u its unit in the last place and S an integer in
//   [2^52, 2^53], adding x gives +-(S + t) * u exactly, t = +-x / u (a power-of-two scaling, exact), and the rounded
//   result is S + rint(t) unless t lies exactly half way between two integers -- then it is whichever of S + floor(t)
//   and S + floor(t) + 1 is even -- provided S + t is still inside [2^52, 2^53).  Integers below 2^53 add exactly
//   in f64 in any order.  So a window whose values do not tie, and whose sum of |rint(t)| + 2 cannot carry S out of
//   the binade, is one parallel sum of rint(t).  Otherwise the window is scanned: each thread's eight values as a map
//   "parity of S before -> integer added" (a tie rounds by the parity of the sum before it, nothing else of S
//   matters), the maps composed in order, every value replayed from its thread's true S, and the first value the
//   model does not cover (result outside the binade; a huge, infinite or NaN value) located.  Everything before it
//   is committed, and one thread goes on from there with real f64 additions for a short run (64 values; binade
//   changes come in clusters while the sum is small), after which the windows resume in whatever binade the sum is
//   in then.  A sum the model cannot stand for (zero, tiny, near overflow, not finite) is advanced by such runs too.
//   Runs get longer when a window keeps needing them, which bounds the cost on values that defeat the scheme (a sum
//   that keeps returning to zero) at about twice that of adding everything one at a time.
// One workgroup of 512 threads, windows of 4096 values, aligned.  Each thread keeps eight values of the current
// window and of the next two in registers, taken with a stride of 512 (coalesced loads; the parallel sum does not
// care which thread adds what), so no step waits for memory; the scanned window and the runs need the values in
// order and take them from a copy of the window in LDS, made when first needed.  The state (s, pos) is computed by
// every thread alike.
enum { KHG_OSUM_NT = 512, KHG_OSUM_NW = KHG_OSUM_NT / 64, KHG_OSUM_E = 8, KHG_OSUM_W = KHG_OSUM_NT * KHG_OSUM_E };
enum { KHG_OSUM_RUN = 64, KHG_OSUM_RUNS_SHORT = 8 };
#define KHG_OS_MANT 0x000FFFFFFFFFFFFFull
#define KHG_OS_SIGN 0x8000000000000000ull
#define KHG_OS_LO 4503599627370496.0   // 2^52
#define KHG_OS_HI 9007199254740991.0   // 2^53 - 1

KHG_DEV uint64_t khg_f64_bits(double v) { uint64_t b; __builtin_memcpy(&b, &v, 8); return b; }
KHG_DEV double khg_bits_f64(uint64_t b) { double v; __builtin_memcpy(&v, &b, 8); return v; }
#ifdef KHG_SIM
KHG_DEV uint64_t khg_uniform64(uint64_t v) { return v; }
#else
// the same value in every lane, told to the compiler
KHG_DEV uint64_t khg_uniform64(uint64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
#endif
KHG_DEV double khg_rint(double v) { return __builtin_rint(v); }    // to nearest, ties to even (v_rndne_f64)
KHG_DEV double khg_floor(double v) { return __builtin_floor(v); }
KHG_DEV double khg_abs(double v) { return __builtin_fabs(v); }
KHG_DEV double khg_uniform_f64(double v) { return khg_bits_f64(khg_uniform64(khg_f64_bits(v))); }
// 1 for an odd integer (|v| < 2^53), 0 for an even one
KHG_DEV double khg_odd(double v) { return v - 2.0 * khg_floor(v * 0.5); }

// A sum the integer model can stand for: a normal number whose unit in the last place, and its reciprocal, are too.
KHG_DEV bool khg_osum_model(double s)
{
    const uint32_t eb = (uint32_t)(khg_f64_bits(s) >> 52) & 0x7FFu;
    return eb >= 64u && eb <= 2040u;
}

// The integer one value adds to S: rint(t) unless t ties, then by the parity of S before it.  fl = floor(t).
KHG_DEV double khg_osum_step(double t, double par, double* fl)
{
    const double c0 = khg_rint(t);
    *fl = khg_floor(t);
    if (khg_abs(t - c0) == 0.5) return par == 0.0 ? c0 : 2.0 * t - c0;  // c0 is the even neighbour
    return c0;
}

// Where value i of the window lies in the LDS copy: one double of padding after every eight, so that threads reading
// eight consecutive values each meet in a bank two at a time only.
KHG_DEV uint32_t khg_osum_slot(uint32_t i) { return i + (i >> 3); }

struct KhgSsimMeanSmem {
    double s;                 // hand-over of a sum advanced by one thread
    uint32_t bad;             // scanned window: its first value outside the model (offset in the window)
    KhgWgSum2 red[2];
    double data[KHG_OSUM_W + KHG_OSUM_W / 8];  // the current window in order, see khg_osum_slot
    double map[2][2][KHG_OSUM_NT];             // scanned window: [buffer][parity of S before][thread] -> integer added
};
struct KhgSsimMeanArgs {
    const double* blocks;
    size_t nblk;
    double* out;  // [1]
    unsigned long long* stat;  // optional [3]: windows summed in parallel, windows scanned, values added one by one
};

// Thread tid's eight values of the aligned window at wbase: values tid, tid + 512, ... (zeros beyond the end).
KHG_DEV void khg_osum_load(const double* x, size_t n, size_t wbase, int tid, double v[KHG_OSUM_E])
{
    if (wbase + KHG_OSUM_W <= n) {
        const double* w = x + wbase + tid;
        KHG_UNROLL
        for (int e = 0; e < KHG_OSUM_E; ++e) v[e] = w[e * KHG_OSUM_NT];
    } else {
        KHG_UNROLL
        for (int e = 0; e < KHG_OSUM_E; ++e) {
            const size_t i = wbase + (size_t)e * KHG_OSUM_NT + (size_t)tid;
            v[e] = i < n ? x[i] : 0.0;
        }
    }
}

KHG_DEV void khg_ssim_mean_body(const KhgSsimMeanArgs& a, KhgSsimMeanSmem& sh)
{
    const size_t n = a.nblk;
    const double* x = a.blocks;
    double cur[KHG_NSLOT(KHG_OSUM_NT)][KHG_OSUM_E], nx1[KHG_NSLOT(KHG_OSUM_NT)][KHG_OSUM_E],
        nx2[KHG_NSLOT(KHG_OSUM_NT)][KHG_OSUM_E];
    KHG_FOR_THREADS(KHG_OSUM_NT)
    {
        khg_osum_load(x, n, 0, tid, cur[KHG_SLOT]);
        khg_osum_load(x, n, KHG_OSUM_W, tid, nx1[KHG_SLOT]);
        khg_osum_load(x, n, 2 * (size_t)KHG_OSUM_W, tid, nx2[KHG_SLOT]);
    }
    double s = 0.0;
    size_t pos = 0, wbase = 0;
    uint32_t it = 0, runs = 0;  // runs: one-at-a-time runs this window has needed so far
    bool staged = false;        // sh.data holds the current window
    for (;; ++it) {
        if (pos >= n) break;
        while (pos >= wbase + KHG_OSUM_W) {  // the window of pos into cur, one more on its way
            wbase += KHG_OSUM_W;
            runs = 0;
            staged = false;
            KHG_FOR_THREADS(KHG_OSUM_NT)
            {
                KHG_UNROLL
                for (int e = 0; e < KHG_OSUM_E; ++e) {
                    cur[KHG_SLOT][e] = nx1[KHG_SLOT][e];
                    nx1[KHG_SLOT][e] = nx2[KHG_SLOT][e];
                }
                khg_osum_load(x, n, wbase + 2 * (size_t)KHG_OSUM_W, tid, nx2[KHG_SLOT]);
            }
        }
        const size_t wend = n - wbase < (size_t)KHG_OSUM_W ? n : wbase + KHG_OSUM_W;
        bool run = !khg_osum_model(s);
        double before = s;  // the sum a run starts from
        if (!run) {
            const uint64_t sb = khg_f64_bits(s);
            const uint32_t eb = (uint32_t)(sb >> 52) & 0x7FFu;
            const uint64_t sign = sb & KHG_OS_SIGN;
            const double su = khg_bits_f64(((uint64_t)(eb - 52u) << 52) | sign);     // +-u
            const double sinv = khg_bits_f64(((uint64_t)(2098u - eb) << 52) | sign);  // +-1/u
            const double S = s * sinv;                                              // 2^52 .. 2^53 - 1
            // every window: the integers its values add when nothing ties, and how far they can move the sum
            double tsum[KHG_NSLOT(KHG_OSUM_NT)], tmag[KHG_NSLOT(KHG_OSUM_NT)];
            KHG_FOR_THREADS(KHG_OSUM_NT)
            {
                double sum = 0.0, mag = 0.0, dmax = 0.0;
                const uint32_t skip = (uint32_t)(pos - wbase);  // values of the window already added (mostly none)
                if (skip == 0) {
                    KHG_UNROLL
                    for (int e = 0; e < KHG_OSUM_E; ++e) {
                        const double t = cur[KHG_SLOT][e] * sinv, c = khg_rint(t), d = khg_abs(t - c);
                        dmax = d > dmax ? d : dmax;
                        sum += c;
                        mag += khg_abs(c);
                    }
                } else {
                    KHG_UNROLL
                    for (int e = 0; e < KHG_OSUM_E; ++e) {
                        const double xv = (uint32_t)(e * KHG_OSUM_NT + tid) >= skip ? cur[KHG_SLOT][e] : 0.0;
                        const double t = xv * sinv, c = khg_rint(t), d = khg_abs(t - c);
                        dmax = d > dmax ? d : dmax;
                        sum += c;
                        mag += khg_abs(c);
                    }
                }
                // mag only grows, so a total below 2^52 means every partial sum of it, and of sum, was an integer
                // below 2^53 and exact.  A NaN or an infinity among the values makes mag NaN or infinite, and a value
                // beyond the model makes it huge: all fail the test below.
                tsum[KHG_SLOT] = sum;
                tmag[KHG_SLOT] = dmax < 0.5 ? mag : KHG_OS_LO;
            }
            double wsum, wmag;
            KHG_WG_SUM2(tsum, tmag, sh.red[it & 1u], KHG_OSUM_NW, wsum, wmag);
            // S + (any prefix of the integers) + floor(t) of the next value stays within
            // [S - (sum of the negative ones) - 1, S + (sum of the positive ones)]
            const double up = (wmag + wsum) * 0.5, down = (wmag - wsum) * 0.5 + 1.0;
            if (wmag < KHG_OS_LO && S - down >= KHG_OS_LO && S + up <= KHG_OS_HI) {
                if (a.stat) {
                    KHG_FOR_THREADS(KHG_OSUM_NT) { if (tid == 0) a.stat[0] += 1; }
                }
                s = khg_uniform_f64(s + wsum * su);
                pos = wend;
                continue;
            }
            // scanned window.  Each thread takes eight consecutive values from the copy in LDS ...
            if (!staged) {
                KHG_FOR_THREADS(KHG_OSUM_NT)
                {
                    KHG_UNROLL
                    for (int e = 0; e < KHG_OSUM_E; ++e)
                        sh.data[khg_osum_slot((uint32_t)(e * KHG_OSUM_NT + tid))] = cur[KHG_SLOT][e];
                    if (tid == 0) sh.bad = 0xFFFFFFFFu;
                }
                KHG_BARRIER();
                staged = true;
            } else {
                KHG_FOR_THREADS(KHG_OSUM_NT) { if (tid == 0) sh.bad = 0xFFFFFFFFu; }
            }
            double cv[KHG_NSLOT(KHG_OSUM_NT)][KHG_OSUM_E];
            // ... as a map from the parity of S before them to the integer they add
            KHG_FOR_THREADS(KHG_OSUM_NT)
            {
                const size_t base = wbase + (size_t)tid * KHG_OSUM_E;
                double d0 = 0.0, d1 = 0.0, p0 = 0.0, p1 = 1.0;
                KHG_UNROLL
                for (int e = 0; e < KHG_OSUM_E; ++e) {
                    const double xv = sh.data[khg_osum_slot((uint32_t)(tid * KHG_OSUM_E + e))];
                    cv[KHG_SLOT][e] = xv;
                    const double t = (base + e >= pos && base + e < n ? xv : 0.0) * sinv;
                    double fl;
                    const double c0 = khg_osum_step(t, p0, &fl), c1 = khg_osum_step(t, p1, &fl);
                    d0 += c0;
                    d1 += c1;
                    p0 = khg_odd(p0 + khg_odd(c0));
                    p1 = khg_odd(p1 + khg_odd(c1));
                }
                sh.map[0][0][tid] = d0;
                sh.map[0][1][tid] = d1;
            }
            KHG_BARRIER();
            int cb = 0;
            for (int d = 1; d < KHG_OSUM_NT; d <<= 1) {  // inclusive scan under composition: the lower threads' map first
                KHG_FOR_THREADS(KHG_OSUM_NT)
                {
                    double o0 = sh.map[cb][0][tid], o1 = sh.map[cb][1][tid];
                    if (tid >= d) {
                        const double a0 = sh.map[cb][0][tid - d], a1 = sh.map[cb][1][tid - d];
                        const double b0 = o0, b1 = o1;
                        o0 = a0 + (khg_odd(a0) != 0.0 ? b1 : b0);
                        o1 = a1 + (khg_odd(a1) != 0.0 ? b0 : b1);
                    }
                    sh.map[cb ^ 1][0][tid] = o0;
                    sh.map[cb ^ 1][1][tid] = o1;
                }
                KHG_BARRIER();
                cb ^= 1;
            }
            // replay every value from its thread's true S; a thread that meets a value outside the model leaves S
            // before it in the spare buffer.  (Values after one outside the model may have made maps of rubbish, NaN
            // included: those only reach threads behind it, whose claims lose against the earlier one.)
            const int pS = khg_odd(S) != 0.0 ? 1 : 0;
            KHG_FOR_THREADS(KHG_OSUM_NT)
            {
                const size_t base = wbase + (size_t)tid * KHG_OSUM_E;
                double cs = S + (tid ? sh.map[cb][pS][tid - 1] : 0.0);
                for (int e = 0; e < KHG_OSUM_E; ++e) {
                    if (base + e < pos) continue;
                    if (base + e >= n) break;
                    const double t = cv[KHG_SLOT][e] * sinv;
                    double fl;
                    const double c = khg_osum_step(t, khg_odd(cs), &fl);
                    const double v = cs + fl;
                    if (!(v >= KHG_OS_LO && v <= KHG_OS_HI)) {
                        khg_lds_min(&sh.bad, (uint32_t)(tid * KHG_OSUM_E + e));
                        sh.map[cb ^ 1][0][tid] = cs;
                        break;
                    }
                    cs += c;
                }
            }
            KHG_BARRIER();
            const uint32_t bad = sh.bad;
            if (a.stat) {
                KHG_FOR_THREADS(KHG_OSUM_NT) { if (tid == 0) a.stat[1] += 1; }
            }
            if (bad == 0xFFFFFFFFu) {
                s = s + sh.map[cb][pS][KHG_OSUM_NT - 1] * su;
                pos = wend;
            } else {
                before = sh.map[cb ^ 1][0][bad / KHG_OSUM_E] * su;
                pos = wbase + bad;
                run = true;
            }
            s = khg_uniform_f64(s);
            before = khg_uniform_f64(before);
            pos = (size_t)khg_uniform64(pos);
            KHG_BARRIER();  // the buffers are free again
        }
        if (run) {  // real additions, one thread, from `before` at pos
            if (!staged) {
                KHG_FOR_THREADS(KHG_OSUM_NT)
                {
                    KHG_UNROLL
                    for (int e = 0; e < KHG_OSUM_E; ++e)
                        sh.data[khg_osum_slot((uint32_t)(e * KHG_OSUM_NT + tid))] = cur[KHG_SLOT][e];
                }
                KHG_BARRIER();
                staged = true;
            }
            const size_t len = (size_t)KHG_OSUM_RUN << (runs < KHG_OSUM_RUNS_SHORT ? 0u : runs - KHG_OSUM_RUNS_SHORT + 1u);
            const size_t end = wend - pos < len ? wend : pos + len;
            ++runs;
            KHG_FOR_THREADS(KHG_OSUM_NT)
            {
                if (tid == 0) {
                    double t = before;
                    uint32_t i = (uint32_t)(pos - wbase);
                    const uint32_t iend = (uint32_t)(end - wbase);
                    for (; i + 8 <= iend; i += 8) {
                        double v[8];
                        KHG_UNROLL
                        for (int e = 0; e < 8; ++e) v[e] = sh.data[khg_osum_slot(i + e)];
                        KHG_UNROLL
                        for (int e = 0; e < 8; ++e) t += v[e];
                    }
                    for (; i < iend; ++i) t += sh.data[khg_osum_slot(i)];
                    if (a.stat) a.stat[2] += end - pos;
                    sh.s = t;
                }
            }
            KHG_BARRIER();
            s = khg_uniform_f64(sh.s);
            pos = end;
            KHG_BARRIER();
        }
    }
    KHG_FOR_THREADS(KHG_OSUM_NT) { if (tid == 0) a.out[0] = s / (double)a.nblk; }
}

struct KhgSseArgs {
    const uint8_t* in[3];
    const uint8_t* out[3];
    size_t n;                  // pixels per plane
    unsigned long long* sse;   // [3], zeroed by the caller
};

struct KhgSseSmem {
    unsigned long long part[3][256];
};

// Integer sums of (I-O)^2 per channel: exactly what GetMse accumulates in a double (every
// partial sum is an integer below 2^53).
// Sum of (x - y)^2 over the four bytes of two words: sum x^2 + sum y^2 - 2 sum x y, three byte dot products.
KHG_DEV uint32_t khg_sqdiff4(uint32_t x, uint32_t y) { return khg_dot4(x, x) + khg_dot4(y, y) - 2u * khg_dot4(x, y); }

KHG_DEV void khg_sse_body(const KhgSseArgs& a, uint32_t bid, uint32_t nblk, KhgSseSmem& sh)
{
    // sixteen pixels per thread and step when the planes allow 16-byte loads (device allocations and planes of whole
    // images do), one pixel otherwise
    bool wide = (a.n & 15u) == 0;
    for (int c = 0; c < 3; ++c) wide = wide && (((uintptr_t)a.in[c] | (uintptr_t)a.out[c]) & 15u) == 0;
    KHG_FOR_THREADS(256)
    {
        unsigned long long s[3] = {0, 0, 0};
        if (wide) {
            for (size_t i = (size_t)bid * 256 + tid; i < a.n / 16; i += (size_t)nblk * 256)
                for (int c = 0; c < 3; ++c) {
                    uint32_t x[4], y[4];
                    __builtin_memcpy(x, a.in[c] + 16 * i, 16);
                    __builtin_memcpy(y, a.out[c] + 16 * i, 16);
                    s[c] += (unsigned long long)(khg_sqdiff4(x[0], y[0]) + khg_sqdiff4(x[1], y[1]) + khg_sqdiff4(x[2], y[2]) +
                                                 khg_sqdiff4(x[3], y[3]));
                }
        } else {
            for (size_t i = (size_t)bid * 256 + tid; i < a.n; i += (size_t)nblk * 256)
                for (int c = 0; c < 3; ++c) {
                    int d = (int)a.in[c][i] - (int)a.out[c][i];
                    s[c] += (unsigned long long)(d * d);
                }
        }
        for (int c = 0; c < 3; ++c) sh.part[c][tid] = s[c];
    }
    KHG_BARRIER();
    KHG_FOR_THREADS(256)
    {
        if (tid < 3) {
            unsigned long long t = 0;
            for (int i = 0; i < 256; ++i) t += sh.part[tid][i];
#ifdef KHG_SIM
            a.sse[tid] += t;
#else
            atomicAdd(&a.sse[tid], t);
#endif
        }
    }
}
