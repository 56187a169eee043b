#include <hip/hip_runtime.h>
__device__ __forceinline__ double right(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);  // wave_shl:1 : lane l takes lane l+1
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double left(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);  // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__global__ void k(const double* in, double* outl, double* outr) {
    int t = threadIdx.x;
    double v = in[t];
    outl[t] = left(v);
    outr[t] = right(v);
}
int main() {
    double h[128], l[128], r[128], *d, *dl, *dr;
    for (int i = 0; i < 128; ++i) h[i] = i + 0.5;
    hipMalloc(&d, sizeof h); hipMalloc(&dl, sizeof h); hipMalloc(&dr, sizeof h);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 128>>>(d, dl, dr);
    hipMemcpy(l, dl, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(r, dr, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 128; ++i) {
        double wl = (i % 64) ? h[i - 1] : h[i], wr = (i % 64) < 63 ? h[i + 1] : h[i];
        if (l[i] != wl || r[i] != wr) { if (bad < 10) printf("lane %d: left %g (want %g) right %g (want %g)\n", i, l[i], wl, r[i], wr); ++bad; }
    }
    printf("bad %d\n", bad);
    return bad != 0;
}
