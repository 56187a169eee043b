// How the SIMD issues the instruction mix of the codec's f64 kernels (developer probe, gfx950): core cycles
// (s_memtime) per instruction for independent / dependent f64 chains, 32-bit ops, and f64 interleaved with scalar
// ALU work and branches, at 1, 2 and 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/probe_issue tools/probe_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP4(X) X X X X
#define REP16(X) REP4(REP4(X))

template <int OP>
__global__ __launch_bounds__(256) void issue_kernel(double* out, long long* cyc, int iters, uint32_t seed)
{
    double d0 = 1.0 + seed, d1 = 2.0 + seed, d2 = 3.0 + seed, d3 = 4.0 + seed;
    uint32_t u0 = threadIdx.x + seed, u1 = seed * 3, u2 = seed * 5, u3 = seed * 7;
    uint32_t s0 = seed, s1 = seed + 1;
    const double c = 1.0000001;
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) {  // 4 independent f64 chains
            REP16(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(c));)
        } else if (OP == 1) {  // one dependent f64 chain
            REP16(asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1"
                               : "+v"(d0) : "v"(c));)
        } else if (OP == 2) {  // 4 independent 32-bit chains
            REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                               : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(seed));)
        } else if (OP == 3) {  // f64 with one scalar ALU op after each
            REP16(asm volatile("v_add_f64 %0, %0, %4\n s_mov_b32 s20, s21\n v_add_f64 %1, %1, %4\n s_mov_b32 s22, s23\n"
                               "v_add_f64 %2, %2, %4\n s_mov_b32 s21, s20\n v_add_f64 %3, %3, %4\n s_mov_b32 s23, s22"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(c) : "s20", "s21", "s22", "s23");)
        } else if (OP == 4) {  // f64 with an untaken branch after each pair
            REP16(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n s_cbranch_execz 1f\n1:\n"
                               "v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n s_cbranch_execz 2f\n2:"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(c));)
        } else if (OP == 5) {  // f64 mul/add dependent pairs like a lifting step: x + a*(l + r)
            REP16(asm volatile("v_add_f64 %0, %1, %2\n v_mul_f64 %0, %0, %4\n v_add_f64 %0, %3, %0\n"
                               "v_add_f64 %1, %0, %2\n v_mul_f64 %1, %1, %4\n v_add_f64 %1, %3, %1"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(c));)
        } else if (OP == 6) {  // f64 followed by v_cmp + s_and_saveexec + s_or exec (predicated region skeleton)
            REP16(asm volatile("v_add_f64 %0, %0, %4\n v_cmp_gt_f64 vcc, %0, %4\n s_and_saveexec_b64 s[20:21], vcc\n"
                               "v_add_f64 %1, %1, %4\n s_or_b64 exec, exec, s[20:21]\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(c) : "vcc", "scc", "s20", "s21");)
        } else if (OP == 7) {  // 32-bit and f64 alternating
            REP16(asm volatile("v_add_f64 %0, %0, %5\n v_add_u32 %4, %4, %6\n v_add_f64 %1, %1, %5\n v_add_u32 %4, %4, %6\n"
                               "v_add_f64 %2, %2, %5\n v_add_u32 %4, %4, %6\n v_add_f64 %3, %3, %5\n v_add_u32 %4, %4, %6"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(u0) : "v"(c), "v"(seed));)
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = d0 + d1 + d2 + d3 + u0 + u1 + u2 + u3 + s0 + s1;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static const char* names[] = {"f64 add x4 independent", "f64 add dependent chain", "u32 add x4 independent",
                              "f64 add + s_add each", "f64 add x2 + untaken branch", "f64 lifting-like dependent",
                              "f64 + cmp/saveexec/restore", "f64 + u32 alternating"};
static const int per_iter[] = {64, 64, 64, 128, 96, 96, 112, 128};
static const int valu_per_iter[] = {64, 64, 64, 64, 64, 96, 80, 128};

template <int OP>
static void run(double* out, long long* cyc, int wg_per_cu)
{
    const int iters = 2000;
    const int blocks = 256 * wg_per_cu;
    issue_kernel<OP><<<blocks, 256>>>(out, cyc, 10, 1);
    hipDeviceSynchronize();
    issue_kernel<OP><<<blocks, 256>>>(out, cyc, iters, 1);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost);
    double sum = 0;
    for (long long v : h) sum += (double)v;
    double per_wave = sum / h.size();
    printf("%-30s %d waves/SIMD: %7.2f cycles per instruction per wave, SIMD: %5.2f cycles per instruction, %5.2f per VALU\n",
           names[OP], wg_per_cu, per_wave / ((double)iters * per_iter[OP]),
           per_wave / ((double)iters * per_iter[OP] * wg_per_cu), per_wave / ((double)iters * valu_per_iter[OP] * wg_per_cu));
    fflush(stdout);
}

int main()
{
    double* out;
    long long* cyc;
    hipMalloc(&out, sizeof(double) * 1024 * 256 * 2);
    hipMalloc(&cyc, sizeof(long long) * 1024 * 4 * 2);
    for (int w : {1, 2, 4}) {
        run<0>(out, cyc, w);
        run<1>(out, cyc, w);
        run<2>(out, cyc, w);
        run<3>(out, cyc, w);
        run<4>(out, cyc, w);
        run<5>(out, cyc, w);
        run<6>(out, cyc, w);
        run<7>(out, cyc, w);
    }
    return 0;
}
