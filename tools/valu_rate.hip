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
    unsigned u[16], w[16];
    unsigned long long m[2] = {0x5555555555555555ull, ~0ull};            // lane masks in SGPR pairs for the VOP3 selects
    const double b = seed + 1e-9 * threadIdx.x, c = 1.0 - 1e-12;
    const float fb = (float)b, fc = 0.999999f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { a[k] = b + k; f[k] = fb + k; u[k] = threadIdx.x + k; w[k] = threadIdx.x * 3 + k; }
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
        } else if (KIND == 11) {     // what the compiler makes of (X * ct + Y) * ct + Z: a 32 x 32 -> 64-bit multiply-add
#define OP(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[k]) : "v"(u[k]), "v"(u[(k + 1) & 15]) : "vcc");
            REP16(OP)
#undef OP
        } else if (KIND == 12) {
#define OP(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
            REP16(OP)
#undef OP
        } else if (KIND == 13) {     // the 24-bit multiply-add (operands below 2^24: voxel coordinates are below 2^9)
#define OP(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) & 15]), "v"(u[(k + 2) & 15]));
            REP16(OP)
#undef OP
        } else if (KIND == 16) {     // the select as the compiler emits it in divergent loops: the mask in an SGPR pair
#define OP(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) & 15]), "s"(m[0]));
            REP16(OP)
#undef OP
        } else if (KIND == 17) {     // the select as a bit-field insert on a 0 / -1 mask held in a VGPR
#define OP(k) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(u[k]) : "v"(u[15]), "v"(u[(k + 1) & 15]));
            REP16(OP)
#undef OP
        } else if (KIND == 18) {     // v_cndmask on vcc with sources that no neighbour writes
#define OP(k) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[k]) : "v"(f[k]), "v"(f[(k + 1) & 15]) : );
            REP16(OP)
#undef OP
        } else if (KIND == 19) {     // ... vcc written by a VALU compare first (as in real code), 16 selects behind it
            asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(u[0]), "v"(u[1]) : "vcc");
#define OP(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) & 15]) : );
            REP16(OP)
#undef OP
        } else if (KIND == 20) {     // the VOP3 encoding with vcc as its mask operand
#define OP(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) & 15]) : );
            REP16(OP)
#undef OP
        } else if (KIND == 21) {     // one compare per select (the walk step's pattern: cmp, select, cmp, select)
#define OP(k) asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) & 15]), "v"(u[(k + 2) & 15]) : "vcc");
            REP16(OP)
#undef OP
        } else if (KIND == 22) {
#define OP(k) asm volatile("v_cmp_lt_u32 %3, %1, %2\n\tv_cndmask_b32_e64 %0, %0, %1, %3" : "+v"(u[k]) : "v"(u[(k + 1) & 15]), "v"(u[(k + 2) & 15]), "s"(m[1]) : );
            REP16(OP)
#undef OP
        } else if (KIND == 23) {     // e32 selects on vcc with one independent VALU instruction between them (2 instructions per OP)
#define OP(k) asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n\tv_add_u32 %1, %1, %2" : "+v"(u[k]), "+v"(w[k]) : "v"(u[(k + 1) & 15]) : );
            REP16(OP)
#undef OP
        } else if (KIND == 24) {     // a 64-bit select as the compiler writes it: two e32 selects back to back, then two adds (4 instructions per OP)
#define OP(k) asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc\n\tv_add_u32 %0, %0, %2\n\tv_add_u32 %1, %1, %2" : "+v"(u[k]), "+v"(w[k]) : "v"(f[(k + 1) & 15]) : );
            REP16(OP)
#undef OP
        } else if (KIND == 25) {     // the same with the VOP3 encoding
#define OP(k) asm volatile("v_cndmask_b32_e64 %0, %0, %2, vcc\n\tv_cndmask_b32_e64 %1, %1, %2, vcc\n\tv_add_u32 %0, %0, %2\n\tv_add_u32 %1, %1, %2" : "+v"(u[k]), "+v"(w[k]) : "v"(f[(k + 1) & 15]) : );
            REP16(OP)
#undef OP
        }
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += a[k] + (double)f[k] + (double)u[k] + (double)w[k];
    if (s == 12345.678 || m[0] + m[1] == 7ull) out[0] = s;
}

template <int KIND>
void bench(const char* name, int cus, double* d_out)
{
    const int iters = 4096;
    printf("%-22s", name);
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
    bench<16>("v_cndmask e64 sgpr", cus, d_out);
    bench<17>("v_bfi_b32", cus, d_out);
    bench<18>("v_cndmask indep", cus, d_out);
    bench<19>("cmp; 16 cndmask vcc", cus, d_out);
    bench<20>("v_cndmask e64 vcc", cus, d_out);
    bench<21>("cmp+cndmask vcc (x2)", cus, d_out);
    bench<22>("cmp+cndmask sgpr(x2)", cus, d_out);
    bench<23>("cndmask e32; add (x2)", cus, d_out);
    bench<24>("2 cndm e32; 2 add(x4)", cus, d_out);
    bench<25>("2 cndm e64; 2 add(x4)", cus, d_out);
    bench<11>("v_mad_u64_u32", cus, d_out);
    bench<12>("v_mul_lo_u32", cus, d_out);
    bench<13>("v_mad_u32_u24", cus, d_out);
    return 0;
}
