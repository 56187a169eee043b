// khg_common.h -- shared definitions for the gfx950 kernels of the KhuGle codec path.
//
// Kernel bodies are written as sequences of barrier-separated phases.  Built by hipcc
// they run one workgroup per body call with KHG_BARRIER() = __syncthreads().  Built
// with -DKHG_SIM by a host compiler (tests only) the same bodies run their threads
// one after another inside each phase, which checks indexing and arithmetic order on
// a machine without a GPU.  The product library is only ever built without KHG_SIM.
#pragma once

#include <stdint.h>
#include <stddef.h>

#ifdef KHG_SIM
#include <math.h>
#include <string.h>
#define KHG_DEV inline
#define KHG_HOST_DEV inline
#define KHG_NSLOT(NT) (NT)
#define KHG_FOR_THREADS(NT) for (int tid = 0; tid < (NT); ++tid)
#define KHG_SLOT tid
#define KHG_BARRIER() ((void)0)
#define KHG_OPAQUE(x) ((void)0)
#define KHG_KEEP_SCALAR(x) ((void)0)
#define KHG_TOUCH(reg, ptr) ((void)(reg), (void)(ptr))
#define KHG_TOUCH_DONE(reg) ((void)(reg))
#define KHG_UNROLL
#define KHG_NOUNROLL
#define KHG_ASSUME_ALIGNED(p, n) (p)  // the simulator's buffers are not aligned like the device workspace
// Workgroup collective (256 threads, like KHG_BARRIER not inside KHG_FOR_THREADS): excl[t] = val[0] + .. + val[t-1],
// total = the sum of all; lds4: four LDS words nobody else touches until the next barrier.
#define KHG_WG_EXCL_SCAN(excl, val, lds4, total)   \
    {                                              \
        uint32_t run_ = 0;                         \
        for (int t_ = 0; t_ < 256; ++t_) {         \
            const uint32_t v_ = (val)[t_];         \
            (excl)[t_] = run_;                     \
            run_ += v_;                            \
        }                                          \
        (total) = run_;                            \
    }
// The same for a workgroup of NW waves: lds: NW words nobody else touches until the next barrier.
#define KHG_WG_EXCL_SCAN_NW(NW, excl, val, lds, total) \
    {                                                  \
        uint32_t run_ = 0;                             \
        for (int t_ = 0; t_ < 64 * (NW); ++t_) {       \
            const uint32_t v_ = (val)[t_];             \
            (excl)[t_] = run_;                         \
            run_ += v_;                                \
        }                                              \
        (void)(lds);                                   \
        (total) = run_;                                \
    }
// N such scans behind one barrier: excl[j][t], val[j][t], lds[j][4], total[j] for j < N.
#define KHG_WG_EXCL_SCAN_N(N, excl, val, lds, total)                                           \
    {                                                                                          \
        for (int j_ = 0; j_ < (N); ++j_) KHG_WG_EXCL_SCAN((excl)[j_], (val)[j_], (lds)[j_], (total)[j_]) \
    }
// ... and for a workgroup of NW waves (lds[j][NW]).
#define KHG_WG_EXCL_SCAN_N_NW(N, NW, excl, val, lds, total)                                    \
    {                                                                                          \
        for (int j_ = 0; j_ < (N); ++j_) KHG_WG_EXCL_SCAN_NW(NW, (excl)[j_], (val)[j_], (lds)[j_], (total)[j_]) \
    }
// Workgroup collective (NW waves, not inside KHG_FOR_THREADS): the sums of two doubles over all threads, to every
// thread, for values whose sum does not depend on the order (integers below 2^53).  lds: a KhgWgSum2 nobody else
// touches until the next barrier but one.
#define KHG_WG_SUM2(av, bv, lds, NW, asum, bsum)       \
    {                                                   \
        double a_ = 0, b_ = 0;                          \
        for (int t_ = 0; t_ < 64 * (NW); ++t_) {        \
            a_ += (av)[t_];                             \
            b_ += (bv)[t_];                             \
        }                                               \
        (void)(lds);                                    \
        (asum) = a_;                                    \
        (bsum) = b_;                                    \
    }
static inline uint32_t khg_atomic_add(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
static inline uint32_t khg_atomic_min(uint32_t* p, uint32_t v) { uint32_t o = *p; if (v < o) *p = v; return o; }
static inline uint32_t khg_atomic_or(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }
static inline int khg_atomic_max_i(int* p, int v) { int o = *p; if (v > o) *p = v; return o; }
static inline void khg_lds_or(uint32_t* p, uint32_t v) { *p |= v; }
static inline void khg_lds_add(uint32_t* p, uint32_t v) { *p += v; }
static inline void khg_lds_min(uint32_t* p, uint32_t v) { if (v < *p) *p = v; }
static inline uint32_t khg_bswap32(uint32_t v) { return __builtin_bswap32(v); }
static inline double khg_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
static inline unsigned long long khg_clock(void) { return 0; }
static inline void khg_atomic_max64(unsigned long long* p, unsigned long long v) { if (v > *p) *p = v; }
static inline uint64_t khg_chain_load(const uint64_t* p) { return *p; }
static inline void khg_chain_store(uint64_t* p, uint64_t v) { *p = v; }
#else
#include <hip/hip_runtime.h>
#define KHG_DEV __device__ __forceinline__
#define KHG_HOST_DEV __host__ __device__ __forceinline__  // also used by the launch planning on the host
#define KHG_NSLOT(NT) 1
#define KHG_FOR_THREADS(NT) for (int tid = (int)threadIdx.x, khg_once_ = 1; khg_once_; khg_once_ = 0)
#define KHG_SLOT 0
#define KHG_BARRIER() __syncthreads()
// Makes a 32-bit value opaque to the optimiser at this point: what is computed from it afterwards cannot be hoisted
// out of the (rarely taken) path it sits on into the common path, where it would only cost registers.
#define KHG_OPAQUE(x) asm volatile("" : "+v"(x))
// Pins a wave-uniform 32-bit value read from the kernel's argument block to a scalar register: the compiler can then
// no longer fetch it from the argument block again wherever it is used (inside a loop that is a load and a wait per
// trip); under register pressure it parks it in a lane of a vector register instead.
#define KHG_KEEP_SCALAR(x) asm volatile("" : "+s"(x))
// Cache warming: a 4-byte load whose result nobody waits for (gfx9 has no prefetch instruction).  `reg` is a 32-bit
// variable that lives from before the first KHG_TOUCH to KHG_TOUCH_DONE, so the register the data eventually lands
// in is never handed to anything else; KHG_TOUCH_DONE drains the memory counter before the variable dies.
#define KHG_TOUCH(reg, ptr) asm volatile("global_load_dword %0, %1, off" : "+v"(reg) : "v"(ptr) : "memory")
#define KHG_TOUCH_DONE(reg) asm volatile("s_waitcnt vmcnt(0)" : "+v"(reg) : : "memory")
#define KHG_UNROLL _Pragma("unroll")
#define KHG_NOUNROLL _Pragma("nounroll")
#define KHG_ASSUME_ALIGNED(p, n) __builtin_assume_aligned(p, n)
// Inclusive prefix sum over the lanes of a wave.
// v + (v of the lanes the DPP control names); lanes without a source (shifted in from outside a row, or excluded by
// the row mask) add zero.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t khg_dpp_add_u32(uint32_t v)
{
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
// Six DPP additions, no LDS traffic: shifts by 1, 2, 4, 8 inside the rows of 16 lanes, then each row's total into
// the rows above it.
__device__ __forceinline__ uint32_t khg_wave_incl_scan(uint32_t v)
{
    v = khg_dpp_add_u32<0x111, 0xf>(v);  // row_shr:1
    v = khg_dpp_add_u32<0x112, 0xf>(v);  // row_shr:2
    v = khg_dpp_add_u32<0x114, 0xf>(v);  // row_shr:4
    v = khg_dpp_add_u32<0x118, 0xf>(v);  // row_shr:8
    v = khg_dpp_add_u32<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = khg_dpp_add_u32<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}
// Workgroup collective (256 threads, like KHG_BARRIER not inside KHG_FOR_THREADS): excl[0] = the sum of val of all
// lower threads, total = the sum of all; lds4: four LDS words nobody else touches until the next barrier.
#define KHG_WG_EXCL_SCAN(excl, val, lds4, total)                      \
    {                                                                 \
        const uint32_t inc_ = khg_wave_incl_scan((val)[0]);           \
        const uint32_t w_ = threadIdx.x >> 6;                         \
        if ((threadIdx.x & 63u) == 63u) (lds4)[w_] = inc_;            \
        __syncthreads();                                              \
        uint32_t base_ = 0, tot_ = 0;                                 \
        _Pragma("unroll") for (uint32_t i_ = 0; i_ < 4; ++i_)        \
        {                                                             \
            const uint32_t s_ = (lds4)[i_];                           \
            base_ += i_ < w_ ? s_ : 0u;                               \
            tot_ += s_;                                               \
        }                                                             \
        (excl)[0] = base_ + inc_ - (val)[0];                          \
        (total) = tot_;                                               \
    }
// The same for a workgroup of NW waves: lds: NW words nobody else touches until the next barrier.
#define KHG_WG_EXCL_SCAN_NW(NW, excl, val, lds, total)                \
    {                                                                 \
        const uint32_t inc_ = khg_wave_incl_scan((val)[0]);           \
        const uint32_t w_ = threadIdx.x >> 6;                         \
        if ((threadIdx.x & 63u) == 63u) (lds)[w_] = inc_;             \
        __syncthreads();                                              \
        uint32_t base_ = 0, tot_ = 0;                                 \
        _Pragma("unroll") for (uint32_t i_ = 0; i_ < (NW); ++i_)     \
        {                                                             \
            const uint32_t s_ = (lds)[i_];                            \
            base_ += i_ < w_ ? s_ : 0u;                               \
            tot_ += s_;                                               \
        }                                                             \
        (excl)[0] = base_ + inc_ - (val)[0];                          \
        (total) = tot_;                                               \
    }
// N such scans behind one barrier, for a workgroup of NW waves: excl[j][0], val[j][0], lds[j][NW], total[j] for j < N.
#define KHG_WG_EXCL_SCAN_N_NW(N, NW, excl, val, lds, total)                        \
    {                                                                              \
        uint32_t inc_[N];                                                          \
        const uint32_t w_ = threadIdx.x >> 6;                                      \
        _Pragma("unroll") for (int j_ = 0; j_ < (N); ++j_)                        \
        {                                                                          \
            inc_[j_] = khg_wave_incl_scan((val)[j_][0]);                           \
            if ((threadIdx.x & 63u) == 63u) (lds)[j_][w_] = inc_[j_];              \
        }                                                                          \
        __syncthreads();                                                           \
        _Pragma("unroll") for (int j_ = 0; j_ < (N); ++j_)                        \
        {                                                                          \
            uint32_t base_ = 0, tot_ = 0;                                          \
            _Pragma("unroll") for (uint32_t i_ = 0; i_ < (NW); ++i_)              \
            {                                                                      \
                const uint32_t s_ = (lds)[j_][i_];                                 \
                base_ += i_ < w_ ? s_ : 0u;                                        \
                tot_ += s_;                                                        \
            }                                                                      \
            (excl)[j_][0] = base_ + inc_[j_] - (val)[j_][0];                       \
            (total)[j_] = tot_;                                                    \
        }                                                                          \
    }
#define KHG_WG_EXCL_SCAN_N(N, excl, val, lds, total) KHG_WG_EXCL_SCAN_N_NW(N, 4, excl, val, lds, total)
// v + (v of the lanes the DPP control names); lanes the row mask excludes add zero.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double khg_dpp_add(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(hi, lo);
}
// The sum of v over the wave, valid in lane 63 (the order of the additions is the hardware's butterfly: for values
// whose sum is exact in any order).
__device__ __forceinline__ double khg_wave_sum_to_last(double v)
{
    v = khg_dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v = khg_dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v = khg_dpp_add<0x141, 0xf>(v);  // row_half_mirror
    v = khg_dpp_add<0x140, 0xf>(v);  // row_mirror: every lane holds its row's sum
    v = khg_dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = khg_dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}
// Workgroup collective (NW waves, like KHG_BARRIER not inside KHG_FOR_THREADS): the sums of two doubles over all
// threads, to every thread, for values whose sum does not depend on the order (integers below 2^53).  lds: a
// KhgWgSum2 nobody else touches until the next barrier but one.
#define KHG_WG_SUM2(av, bv, lds, NW, asum, bsum)                    \
    {                                                                \
        const double a_ = khg_wave_sum_to_last((av)[0]);             \
        const double b_ = khg_wave_sum_to_last((bv)[0]);             \
        if ((threadIdx.x & 63u) == 63u) {                            \
            (lds).a[threadIdx.x >> 6] = a_;                          \
            (lds).b[threadIdx.x >> 6] = b_;                          \
        }                                                            \
        __syncthreads();                                             \
        double as_ = 0, bs_ = 0;                                     \
        _Pragma("unroll") for (int w_ = 0; w_ < (NW); ++w_)         \
        {                                                            \
            as_ += (lds).a[w_];                                      \
            bs_ += (lds).b[w_];                                      \
        }                                                            \
        (asum) = as_;                                                \
        (bsum) = bs_;                                                \
    }
__device__ __forceinline__ uint32_t khg_atomic_add(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
__device__ __forceinline__ uint32_t khg_atomic_min(uint32_t* p, uint32_t v) { return atomicMin(p, v); }
__device__ __forceinline__ uint32_t khg_atomic_or(uint32_t* p, uint32_t v) { return atomicOr(p, v); }
__device__ __forceinline__ int khg_atomic_max_i(int* p, int v) { return atomicMax(p, v); }
__device__ __forceinline__ void khg_lds_or(uint32_t* p, uint32_t v) { atomicOr(p, v); }  // result unused: ds_or_b32, no wait
// Workgroup-scope forms for LDS words: a different scope than the device-wide atomics above, so the compiler cannot
// merge an LDS update and a global one into a single flat atomic behind a pointer select.
__device__ __forceinline__ void khg_lds_add(uint32_t* p, uint32_t v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void khg_lds_min(uint32_t* p, uint32_t v) { __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t khg_bswap32(uint32_t v) { return __builtin_bswap32(v); }
__device__ __forceinline__ double khg_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }  // explicit: not subject to -ffp-contract
__device__ __forceinline__ unsigned long long khg_clock(void) { return (unsigned long long)wall_clock64(); }  // constant 100 MHz, chip-wide
__device__ __forceinline__ void khg_atomic_max64(unsigned long long* p, unsigned long long v) { atomicMax(p, v); }
// State words that workgroups of one launch publish to each other (each word is self-contained: no other memory is
// ordered by them).
__device__ __forceinline__ uint64_t khg_chain_load(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void khg_chain_store(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif

struct KhgWgSum2 {
    double a[16], b[16];
};

// ---------------------------------------------------------------- constants

enum {
    KHG_NBINS = 16384,      // dense histogram over symbols [-8192, 8191]
    KHG_SYM_BIAS = 8192,
    KHG_CHUNK = 2048,       // symbols per bit-packing chunk (256 threads x 8)
    KHG_MAX_CODE_BITS = 56  // code + length share one 64-bit table word
};

// Device-side result block of one encode or decode.  A kernel that detects a problem
// stores its (negative) khg_status code in `status`; the first code written stays.
struct KhgResult {
    int status;            // 0 or a negative khg_status
    int pad_;
    uint64_t total_len;    // .comp bytes
    int32_t bits[3];       // padded bit counts
    int32_t k[3];          // table sizes
    int32_t pads[3];
    int32_t ok_to_pack;    // 1 when offsets are valid and the output fits
    // measurement aid (khg_profile_luma_forward): device clock (100 MHz) at which the luma level-1 forward
    // launch's first workgroup started, stored inverted so that zero-initialised memory and a max reduce give
    // the minimum, and at which its last workgroup finished
    unsigned long long prof_not_start, prof_end;
};

KHG_DEV void khg_set_status(int* status, int code)
{
    if (*status == 0) *status = code;  // racing writers all carry an error code: any of them will do
}

// x / c, correctly rounded, for a constant c whose correctly rounded reciprocal rc = RN(1/c) is known
// (Markstein: q = RN(x*rc) is within an ulp, the residual x - q*c is exact in one fused multiply-add, and
// one correction step lands on RN(x/c)).  Holds away from overflow and underflow, where every sample of
// this codec lives; the sign of a zero quotient may differ, which no later step can observe.
// Three instructions instead of the ~35 of an IEEE f64 division.
KHG_DEV double khg_div_const(double x, double c, double rc)
{
    double q = x * rc;
    double r = khg_fma(-q, c, x);
    return khg_fma(r, rc, q);
}

// Workgroups are dealt out to the chip's 8 XCDs (each with its own L2) round robin by their id.  This maps a
// launch's workgroup id to a work index such that runs of 8 consecutive pieces of work go to one XCD: neighbouring
// pieces -- which share cache lines at their edges and read each other's halo -- then meet in one L2 instead of
// two (a line written in part from two XCDs goes to memory as two masked writes).  It transposes every aligned
// block of 64 ids as an 8 x 8 matrix and leaves a last partial block alone, so the order in which kinds of work
// reach the CUs stays what the launch plan laid out.  A bijection on [0, n) for any n; where the hardware deals
// differently the result is as correct, only slower.
KHG_DEV int khg_xcd_remap(int wg, int n)
{
    if ((wg | 63) >= n) return wg;
    return (wg & ~63) | ((wg & 7) << 3) | ((wg >> 3) & 7);
}

// Rows [*r0, *r1) of segment `seg` of `nseg` over the even row range [begin, end): the row pairs are dealt out evenly,
// segments differ by at most one pair.
KHG_DEV void khg_segment_rows(int begin, int end, int nseg, int seg, int* r0, int* r1)
{
    const long long hp = (end - begin) / 2;
    *r0 = begin + 2 * (int)(hp * seg / nseg);
    *r1 = begin + 2 * (int)(hp * (seg + 1) / nseg);
}

// Whole-sample symmetric extension of index i onto [0, n), any i (n even, >= 2).
// The reference's boundary lines x[n-1] += 2*a*x[n-2] and x[0] += 2*a*x[1] (DWT.h:18,26,34,42)
// equal the interior formula on this extension bit for bit: 2*a*x == a*(x + x) exactly.
KHG_DEV int khg_reflect(int i, int n)
{
    if ((unsigned)i < (unsigned)n) return i;
    int p = 2 * (n - 1);
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - i;
}

// Per-plane-type quantiser constants (Compress.h:172-209 / 223-226), computed on the host.
struct KhgQuant {
    double step[7];
    double a[7];
    double b[7];
};

// Compress.h:215-218: q = (int)round(coef / (step * (lvl >= 1 ? a*(x+y)+b : 1)))
KHG_HOST_DEV double khg_divisor(const KhgQuant& q, int band, int x, int y, int lvl)
{
    double div = q.step[band];
    if (lvl >= 1) {
        double mul = q.a[band] * (double)(x + y) + q.b[band];
        div = q.step[band] * mul;
    }
    return div;
}
