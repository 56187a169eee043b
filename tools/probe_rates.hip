// Issue cost of the VALU instructions the codec's f64 kernels are made of (developer probe, gfx950).
// Each wave runs a long loop of 8 independent chains of ONE instruction; with W waves per SIMD the SIMD's issue
// slot is saturated, so (elapsed core cycles * SIMDs in use) / (wave-instructions) is the cycles one wave64
// instruction holds the pipe.  Build: hipcc --offload-arch=gfx950 -O2 -o probe_rates tools/probe_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(double* out, int iters, uint32_t seed, unsigned long long* clk)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double d[8];
    uint32_t u[8];
    float f[8];
    for (int i = 0; i < 8; ++i) {
        d[i] = 1.0 + 0.001 * (threadIdx.x + i) + seed;
        u[i] = threadIdx.x * 2654435761u + i + seed;
        f[i] = 1.0f + i + seed;
    }
    const double c = 1.0000001;
    for (int it = 0; it < iters; ++it) {
#define ONE(i)                                                                                          \
    if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c));                            \
    if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c));                            \
    if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(c));                        \
    if (OP == 3) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(u[i]));                         \
    if (OP == 4) asm volatile("v_trunc_f64 %0, %0" : "+v"(d[i]));                                       \
    if (OP == 5) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[i]) : "v"(u[i] & 1));                   \
    if (OP == 6) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[i]) : "v"(d[i]));                         \
    if (OP == 7) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f[i]) : "v"(u[i]));                      \
    if (OP == 8) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));                         \
    if (OP == 9) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(u[i]));                                   \
    if (OP == 10) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(d[i]), "v"(c) : "vcc");              \
    if (OP == 11) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(d[i]));                                 \
    if (OP == 12) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(d[i]) : "v"(c));                   \
    if (OP == 13) asm volatile("v_mov_b64 %0, %1" : "=v"(d[i]) : "v"(c));                               \
    if (OP == 14) asm volatile("v_add_u32 %0, %0, %0" : "+v"(u[i]));                                    \
    if (OP == 15) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(u[i]));                        \
    if (OP == 16) asm volatile("v_rndne_f64 %0, %0" : "+v"(d[i]));                                      \
    if (OP == 17) asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(u[i]));                                 \
    if (OP == 18) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(c));                        \
    if (OP == 19) asm volatile("v_add_f64 %0, -|%0|, 0.5" : "+v"(d[i]));                                \
    if (OP == 20) asm volatile("v_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i])); \
    if (OP == 21) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));          \
    if (OP == 22) asm volatile("v_max3_u32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));         \
    if (OP == 23) asm volatile("v_alignbyte_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));    \
    if (OP == 24) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 7]) : "vcc"); \
    if (OP == 25) asm volatile("v_mad_i32_i24 %0, %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));      \
    if (OP == 26) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f[i]) : "v"(u[i]));                     \
    if (OP == 27) asm volatile("v_max_f64 %0, |%0|, |%1|" : "+v"(d[i]) : "v"(c));                       \
    if (OP == 28) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        REP8(ONE)
        REP8(ONE)
#undef ONE
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += d[i] + u[i] + f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    // shader cycles and 100 MHz ticks this workgroup lived: their ratio is the clock the chip held under this load
    if (threadIdx.x == 0 && clk) {
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

static const char* names[] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_cvt_f64_u32", "v_trunc_f64", "v_ldexp_f64",
                              "v_cvt_i32_f64", "v_cvt_f32_ubyte1", "v_cvt_f64_f32", "v_bfe_u32", "v_cmp_gt_f64",
                              "v_lshlrev_b64", "v_lshl_add_u64", "v_mov_b64", "v_add_u32", "v_cvt_f64_i32",
                              "v_rndne_f64", "v_mul_lo_u32", "v_pk_add_f32", "v_add_f64 mods", "v_mov_b32_dpp",
                              "v_or3_b32", "v_max3_u32", "v_alignbyte_b32", "v_cndmask_b32", "v_mad_i32_i24",
                              "v_cvt_f32_ubyte1 b", "v_max_f64 abs", "v_lshl_or_b32"};

template <int OP>
static void run(double* out, int blocks, double ghz)
{
    const int iters = 4096;
    static unsigned long long* clk = nullptr;
    if (!clk) hipMalloc(&clk, sizeof(unsigned long long) * 2 * blocks);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    rate_kernel<OP><<<blocks, 256>>>(out, 64, 1, nullptr);
    for (int w = 0; w < 40; ++w) rate_kernel<OP><<<blocks, 256>>>(out, iters, 1, nullptr);  // let the clock settle under this load
    hipDeviceSynchronize();
    hipEventRecord(a);
    rate_kernel<OP><<<blocks, 256>>>(out, iters, 1, clk);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    // waves per SIMD = blocks * 4 waves / (256 CUs * 4 SIMDs); every SIMD issues waves_per_simd * iters * 16 instr
    double waves_per_simd = blocks * 4.0 / 1024.0;
    double instr = waves_per_simd * iters * 16.0;
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    std::vector<double> g;
    for (int i = 0; i < blocks; ++i) g.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);  // GHz in this workgroup
    std::sort(g.begin(), g.end());
    const double held = g[g.size() / 2];
    printf("%-20s %8.3f ms  %5.2f ns per wave64 instruction and SIMD = %5.2f cycles at the %.2f GHz held in the kernel "
           "(nominal %.2f GHz, %g waves/SIMD)\n", names[OP], ms, ms * 1e6 / instr, ms * 1e-3 * held * 1e9 / instr, held, ghz,
           waves_per_simd);
    hipEventDestroy(a);
    hipEventDestroy(b);
}

int main()
{
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    double ghz = clk_khz / 1e6;
    const int blocks = 256 * 4;  // 4 workgroups per CU = 4 waves per SIMD
    double* out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    run<0>(out, blocks, ghz);
    run<1>(out, blocks, ghz);
    run<2>(out, blocks, ghz);
    run<3>(out, blocks, ghz);
    run<4>(out, blocks, ghz);
    run<5>(out, blocks, ghz);
    run<6>(out, blocks, ghz);
    run<7>(out, blocks, ghz);
    run<8>(out, blocks, ghz);
    run<9>(out, blocks, ghz);
    run<10>(out, blocks, ghz);
    run<11>(out, blocks, ghz);
    run<12>(out, blocks, ghz);
    run<13>(out, blocks, ghz);
    run<14>(out, blocks, ghz);
    run<15>(out, blocks, ghz);
    run<16>(out, blocks, ghz);
    run<17>(out, blocks, ghz);
    run<18>(out, blocks, ghz);
    run<19>(out, blocks, ghz);
    run<20>(out, blocks, ghz);
    run<21>(out, blocks, ghz);
    run<22>(out, blocks, ghz);
    run<23>(out, blocks, ghz);
    run<24>(out, blocks, ghz);
    run<25>(out, blocks, ghz);
    run<26>(out, blocks, ghz);
    run<27>(out, blocks, ghz);
    run<28>(out, blocks, ghz);
    hipFree(out);
    return 0;
}
