// valu_rate.hip — issue rate of plain VALU instructions on gfx950: wave-instructions per clock per SIMD.
// One workgroup of 256 threads per (CU, wave slot): grid = 256 CUs x WAVES_PER_SIMD; every wave runs N iterations of 16 independent
// instructions of one kind (inline asm, so nothing is folded).  Prints clocks per wave-instruction per SIMD at the measured time,
// assuming 2.4 GHz (the ratio between kinds is what matters).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int KIND> __global__ __launch_bounds__(256) void k(float *out, int n, float b, float c) {
    float a[16]; int ia[16];
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; ia[i] = (int)threadIdx.x + i; }
    int ib = (int)(b * 3.f) + 1;
    for (int it = 0; it < n; ++it) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ib));
#define CVT(i) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(ia[i]) : "v"(a[i]));
#define FLOOR(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
#define MINU(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ib));
#define BFE(i) asm volatile("v_bfe_u32 %0, %0, %1, 5" : "+v"(ia[i]) : "v"(ib));
#define CMPSEL(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double *)&a[(i) & 14]) : "v"(*(double *)&a[14]), "v"(*(double *)&a[12]));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(*(double *)&a[(i) & 14]) : "v"(*(double *)&a[14]));
#define LSHL_ADD(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(ia[i]) : "v"(ib));
#define AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(ia[i]) : "v"(ib));
#define MAD_I(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(ia[i]) : "v"(ib));
#define MINF(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define SUBF(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define FMAC(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define CNDM(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
#define CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
#define XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(ia[i]) : "v"(ib));
#define LSHL(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(ia[i]));
#define MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(*(double *)&a[(i) & 14]) : "v"(*(double *)&a[14]));
#define ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(*(double *)&a[(i) & 14]) : "v"(*(double *)&a[14]));
#define CVT64(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(*(double *)&a[(i) & 14]) : "v"(b));
#define MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define SUBU(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ib));
#define CVTF(i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(ia[i]));
#define READL(i) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a[i]) : "s20");
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ib));
        if (KIND == 0) { REP16(FMA) } else if (KIND == 1) { REP16(MUL) } else if (KIND == 2) { REP16(ADDU) } else if (KIND == 3) { REP16(CVT) }
        else if (KIND == 4) { REP16(FLOOR) } else if (KIND == 5) { REP16(MINU) } else if (KIND == 6) { REP16(BFE) } else if (KIND == 7) { REP16(CMPSEL) }
        else if (KIND == 8) { REP16(PKFMA) } else if (KIND == 9) { REP16(RCP) } else if (KIND == 10) { REP16(SQRT) } else if (KIND == 11) { REP16(FMA64) }
        else if (KIND == 16) { REP16(MINF) } else if (KIND == 17) { REP16(MAX3) } else if (KIND == 18) { REP16(SUBF) } else if (KIND == 19) { REP16(FMAC) }
        else if (KIND == 20) { REP16(CNDM) } else if (KIND == 21) { REP16(CMP) } else if (KIND == 22) { REP16(MOV) } else if (KIND == 23) { REP16(XOR) }
        else if (KIND == 24) { REP16(LSHL) } else if (KIND == 25) { REP16(MUL64) } else if (KIND == 26) { REP16(ADD64) } else if (KIND == 27) { REP16(CVT64) }
        else if (KIND == 28) { REP16(MED3) } else if (KIND == 29) { REP16(SUBU) } else if (KIND == 30) { REP16(CVTF) } else if (KIND == 31) { REP16(READL) }
        else if (KIND == 12) { REP16(LSHL_ADD) } else if (KIND == 13) { REP16(AND) } else if (KIND == 14) { REP16(MAD_I) } else if (KIND == 15) { REP16(MULLO) }
    }
    float s = 0.f; for (int i = 0; i < 16; ++i) s += a[i] + (float)ia[i];
    if (s == 12345.678f) out[0] = s;
}
template <int KIND> void run(const char *name, int wps, int per_iter) {
    float *d; (void)hipMalloc(&d, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int n = 5000, grid = 256 * wps;   // wps = 8: every SIMD holds 8 waves at once; 32: four rounds of that
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d, 100, 1.0001f, 0.5f);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d, n, 1.0001f, 0.5f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: wps waves x n iterations x per_iter instructions
    double clk = ms * 1e-3 * 2.4e9, inst = (double)wps * n * per_iter;
    printf("%-16s waves/SIMD %d: %8.3f ms  %.2f clocks per wave-instruction per SIMD (at 2.4 GHz)\n", name, wps, ms, clk / inst);
    (void)hipFree(d);
}
int main() {
    for (int wps : {32}) {
        run<0>("v_fma_f32", wps, 16); run<1>("v_mul_f32", wps, 16); run<2>("v_add_u32", wps, 16); run<3>("v_cvt_i32_f32", wps, 16);
        run<4>("v_floor_f32", wps, 16); run<5>("v_min_u32", wps, 16); run<6>("v_bfe_u32", wps, 16); run<7>("v_cmp+v_cndmask", wps, 32);
        run<8>("v_pk_fma_f32", wps, 16); run<9>("v_rcp_f32", wps, 16); run<10>("v_sqrt_f32", wps, 16); run<11>("v_fma_f64", wps, 16);
        run<12>("v_lshl_add_u32", wps, 16); run<13>("v_and_b32", wps, 16); run<14>("v_mad_u32_u24", wps, 16); run<15>("v_mul_lo_u32", wps, 16);
        run<16>("v_min_f32", wps, 16); run<17>("v_max3_f32", wps, 16); run<18>("v_sub_f32", wps, 16); run<19>("v_fmac_f32", wps, 16);
        run<20>("v_cndmask_b32", wps, 16); run<21>("v_cmp_lt_f32", wps, 16); run<22>("v_mov_b32", wps, 16); run<23>("v_xor_b32", wps, 16);
        run<24>("v_lshlrev_b32", wps, 16); run<25>("v_mul_f64", wps, 16); run<26>("v_add_f64", wps, 16); run<27>("v_cvt_f64_f32", wps, 16);
        run<28>("v_med3_f32", wps, 16); run<29>("v_sub_u32", wps, 16); run<30>("v_cvt_f32_u32", wps, 16); run<31>("v_readlane_b32", wps, 16);
    }
    return 0;
}
