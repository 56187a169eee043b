// khg_metrics.h -- SSIM over non-overlapping 8x8 blocks (SSIM.h:5-72) and the squared-error
// sums behind GetPsnr (KhuGleBase.cpp:602-627).
#pragma once
#include "khg_common.h"

struct KhgSsimArgs {
    const uint8_t *a, *b;  // two planes, width x height, row-major
    int width, height;
    double* blocks;        // [(height/8) * (width/8)] per-block SSIM in raster order
};

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
            double s1 = 0.0, s2 = 0.0;
            for (int i = 0; i < 8; ++i)
                for (int j = 0; j < 8; ++j) {
                    s1 += pa[(size_t)i * a.width + j];
                    s2 += pb[(size_t)i * a.width + j];
                }
            double avg1 = s1 / 64.0, avg2 = s2 / 64.0;  // SSIM.h:14-21
            double v1 = 0.0, v2 = 0.0, cv = 0.0;
            for (int i = 0; i < 8; ++i)
                for (int j = 0; j < 8; ++j) {
                    double da = pa[(size_t)i * a.width + j] - avg1, db = pb[(size_t)i * a.width + j] - avg2;
                    v1 += da * da;  // SSIM.h:23-30
                    v2 += db * db;
                    cv += da * db;  // SSIM.h:32-40
                }
            double var1 = v1 / 64.0, var2 = v2 / 64.0, cov = cv / 64.0;
            double L = 255.0;
            double K1 = 0.01, K2 = 0.03;
            double C1 = (K1 * L) * (K1 * L), C2 = (K2 * L) * (K2 * L);
            a.blocks[blk] = (2.0 * avg1 * avg2 + C1) * (2.0 * cov + C2) /
                            ((avg1 * avg1 + avg2 * avg2 + C1) * (var1 + var2 + C2));  // SSIM.h:52-53
        }
    }
}

// Mean of the per-block values (SSIM.h:66-71): the reference adds them one after another in raster order, and a
// double sum depends on its order, so one lane does exactly that; the other threads only bring the values into
// LDS, a coalesced chunk at a time.  One workgroup of 256 threads.
enum { KHG_SSIM_CHUNK = 2048 };
struct KhgSsimMeanSmem {
    double v[KHG_SSIM_CHUNK];
    double sum;
};
struct KhgSsimMeanArgs {
    const double* blocks;
    size_t nblk;
    double* out;  // [1]
};
KHG_DEV void khg_ssim_mean_body(const KhgSsimMeanArgs& a, KhgSsimMeanSmem& sh)
{
    KHG_FOR_THREADS(256) { if (tid == 0) sh.sum = 0.0; }
    for (size_t c0 = 0; c0 < a.nblk; c0 += KHG_SSIM_CHUNK) {
        const int n = (int)(a.nblk - c0 < (size_t)KHG_SSIM_CHUNK ? a.nblk - c0 : (size_t)KHG_SSIM_CHUNK);
        KHG_FOR_THREADS(256)
        {
            for (int i = tid; i < n; i += 256) sh.v[i] = a.blocks[c0 + i];
        }
        KHG_BARRIER();
        KHG_FOR_THREADS(256)
        {
            if (tid == 0) {
                double s = sh.sum;
                int i = 0;
                // the additions form one dependent chain (the reference's order); the next sixteen values are
                // requested from LDS before the current sixteen are added, so the chain never waits for a fetch
                double t[16];
                if (n >= 16) {
                    KHG_UNROLL
                    for (int u = 0; u < 16; ++u) t[u] = sh.v[u];
                    for (; i + 32 <= n; i += 16) {
                        double nx[16];
                        KHG_UNROLL
                        for (int u = 0; u < 16; ++u) nx[u] = sh.v[i + 16 + u];
                        KHG_UNROLL
                        for (int u = 0; u < 16; ++u) s += t[u];
                        KHG_UNROLL
                        for (int u = 0; u < 16; ++u) t[u] = nx[u];
                    }
                    KHG_UNROLL
                    for (int u = 0; u < 16; ++u) s += t[u];
                    i += 16;
                }
                for (; i < n; ++i) s += sh.v[i];
                sh.sum = s;
            }
        }
        KHG_BARRIER();
    }
    KHG_FOR_THREADS(256) { if (tid == 0) a.out[0] = sh.sum / (double)a.nblk; }
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
KHG_DEV void khg_sse_body(const KhgSseArgs& a, uint32_t bid, uint32_t nblk, KhgSseSmem& sh)
{
    KHG_FOR_THREADS(256)
    {
        unsigned long long s[3] = {0, 0, 0};
        for (size_t i = (size_t)bid * 256 + tid; i < a.n; i += (size_t)nblk * 256)
            for (int c = 0; c < 3; ++c) {
                int d = (int)a.in[c][i] - (int)a.out[c][i];
                s[c] += (unsigned long long)(d * d);
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
