// Developer microbenchmark (round 5): what a wave64 VALU instruction COSTS a SIMD of gfx950, by instruction class -- the weights of
// bench.py's `roofline.issue` (FP64 instructions "counted at their real cost").
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// Every wave runs ITERS x 16 INDEPENDENT instructions of one class (16 accumulators: no dependent-issue stall); the grid puts
// W = 1, 2, 4, 8 waves on every SIMD of the chip.  Reported: ns per wave-instruction per SIMD (time x SIMDs / wave-instructions) and
// the same in cycles of the 2.4 GHz peak clock (the clock under load is lower: the ns figure is the one bench.py uses).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#define REP16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

template <int KIND>
__global__ __launch_bounds__(256) void rate(double* out, int iters, double seed)
{
    double a[16];
    float f[16];
    unsigned u[16];
    const double b = seed + 1e-9 * threadIdx.x, c = 1.0 - 1e-12;
    const float fb = (float)b, fc = 0.999999f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { a[k] = b + k; f[k] = fb + k; u[k] = threadIdx.x + k; }
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
#define OP(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c), "v"(b));
            REP16(OP)
#undef OP
        } else if (KIND == 1) {
#define OP(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            REP16(OP)
#undef OP
        } else if (KIND == 2) {
#define OP(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
            REP16(OP)
#undef OP
        } else if (KIND == 3) {
#define OP(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(fc), "v"(fb));
            REP16(OP)
#undef OP
        } else if (KIND == 4) {
#define OP(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
            REP16(OP)
#undef OP
        } else if (KIND == 5) {
#define OP(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) & 15]) : );
            REP16(OP)
#undef OP
        } else if (KIND == 6) {
#define OP(k) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");
            REP16(OP)
#undef OP
        } else if (KIND == 7) {
#define OP(k) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            REP16(OP)
#undef OP
        } else if (KIND == 8) {
#define OP(k) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[k]));
            REP16(OP)
#undef OP
        } else if (KIND == 9) {
#define OP(k) asm volatile("v_mov_b32 %0, %1" : "=v"(u[k]) : "v"(u[(k + 1) & 15]));
            REP16(OP)
#undef OP
        } else if (KIND == 10) {
#define OP(k) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[k]) : "v"(f[k]));
            REP16(OP)
#undef OP
        }
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += a[k] + (double)f[k] + (double)u[k];
    if (s == 12345.678) out[0] = s;
}

template <int KIND>
void bench(const char* name, int cus, double* d_out)
{
    const int iters = 4096;
    printf("%-16s", name);
    for (int w : {1, 2, 4, 8}) {
        const int grid = cus * w;                 // workgroups of 4 waves: one per SIMD, w workgroups per CU => w waves per SIMD
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        rate<KIND><<<grid, 256>>>(d_out, 64, 1.0);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            rate<KIND><<<grid, 256>>>(d_out, iters, 1.0);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double wave_insts = (double)grid * 4.0 * iters * 16.0;
        const double ns = best * 1e6 * (cus * 4.0) / wave_insts;
        printf("  W=%d: %.3f ns (%.2f cyc @2.4GHz)", w, ns, ns * 2.4);
    }
    printf("\n");
}

int main()
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz; ns per wave64 instruction per SIMD, W waves per SIMD\n", p.name, cus, p.clockRate);
    double* d_out;
    CK(hipMalloc(&d_out, 64));
    bench<0>("v_fma_f64", cus, d_out);
    bench<1>("v_add_f64", cus, d_out);
    bench<2>("v_mul_f64", cus, d_out);
    bench<7>("v_max_f64", cus, d_out);
    bench<6>("v_cmp_lt_f64", cus, d_out);
    bench<8>("v_rcp_f64", cus, d_out);
    bench<10>("v_cvt_f64_f32", cus, d_out);
    bench<3>("v_fma_f32", cus, d_out);
    bench<4>("v_add_u32", cus, d_out);
    bench<5>("v_cndmask_b32", cus, d_out);
    bench<9>("v_mov_b32", cus, d_out);
    return 0;
}
