// valu_rates.hip -- per-instruction issue cost on gfx950 for the ops the pair kernel
// uses.  Each kernel runs ITER x UNROLL independent-chain instances of one op at full
// occupancy; output = ns per wave-instruction per SIMD and cycles at the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 4096;

template <int OP, int LANES = 64>
__global__ __launch_bounds__(256) void k(float *out, float seed, unsigned long long *clk)
{
    if ((int)(threadIdx.x & 63) >= LANES) return;   // the rest of the wave runs with a partial EXEC mask
    float a[8];
    double d[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x * 1e-3f + i; d[i] = a[i]; }
    const float c1 = seed * 0.999f, c2 = seed * 1e-3f;
    float b[8];
    for (int i = 0; i < 8; i++) b[i] = a[i] * 0.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = __builtin_fmaf(a[i], c1, c2);
            if (OP == 1) a[i] = a[i] * c1;
            if (OP == 2) a[i] = a[i] + c2;
            if (OP == 3) a[i] = __builtin_amdgcn_rsqf(a[i]);
            if (OP == 4) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if (OP == 5) d[i] = d[i] + (double)c2;
            if (OP == 6) { d[i] = (double)a[i]; asm volatile("" : "+v"(d[i])); a[i] = a[i] + c2; }       // cvt_f64_f32 (+1 add)
            if (OP == 7) { a[i] = (float)d[i]; asm volatile("" : "+v"(a[i])); d[i] = d[i] + (double)c2; } // cvt_f32_f64 (+1 add_f64)
            if (OP == 8) a[i] = fminf(a[i], fminf(c1, a[(i + 1) & 7]));
            if (OP == 9) a[i] = __builtin_sqrtf(a[i]);   // v_sqrt_f32 + fixups (correctly rounded)
            if (OP == 10) d[i] = __builtin_fma(d[i], (double)c1, (double)c2);
        }
        // round 4: does a transcendental overlap with plain VALU work?  8 v_rsq + 32 v_fma per iteration:
        // 13 interleaved in one wave (1 : 4), 14 grouped in one wave, 15 the waves of even workgroups only v_rsq (x 8),
        // those of odd workgroups only v_fma (x 32) -- a SIMD holds both kinds
        if (OP == 13 || OP == 14 || OP == 15) {
            const bool do_rsq = OP != 15 || (blockIdx.x & 1) == 0, do_fma = OP != 15 || (blockIdx.x & 1) == 1;
            if (OP == 13) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
#pragma unroll
                    for (int r = 0; r < 4; r++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(b[(4 * i + r) & 7]) : "v"(c1), "v"(c2));
                }
            } else {
                if (do_rsq) {
#pragma unroll
                    for (int i = 0; i < 8; i++) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
                }
                if (do_fma) {
#pragma unroll
                    for (int r = 0; r < 32; r++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(b[r & 7]) : "v"(c1), "v"(c2));
                }
            }
        }
        // round 5: what a DPP-modified addition costs (the other lane of a pair hands its term over inside the add)
        if (OP == 16) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
        }
        if (OP == 17) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
        }
        if (OP == 18) {   // the same dependent chain without DPP
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
        }
        if (OP == 19) {   // one accumulator, eight DPP additions in a row (the ordered sum's shape)
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf" : "+v"(a[0]) : "v"(b[i]));
        }
        if (OP == 20) {   // ... and without DPP
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[0]) : "v"(b[i]));
        }
        if (OP == 21) {   // v_mov_b32_dpp + plain add
#pragma unroll
            for (int i = 0; i < 8; i++) { float t; asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(b[i])); asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(t)); }
        }
        if (OP == 11) {   // packed fp32 fma: 4 x v_pk_fma_f32 on 8 floats
            typedef float v2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                v2 x = {a[i], a[i + 1]}, y = {c1, c1}, z = {c2, c2};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(x) : "v"(x), "v"(y), "v"(z));
                a[i] = x.x; a[i + 1] = x.y;
            }
        }
        if (OP == 12) {   // packed fp32 mul
            typedef float v2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                v2 x = {a[i], a[i + 1]}, y = {c1, c1};
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));
                a[i] = x.x; a[i + 1] = x.y;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; double sd = 0;
    for (int i = 0; i < 8; i++) s += b[i];
    for (int i = 0; i < 8; i++) { s += a[i]; sd += d[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)sd;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int OP, int LANES = 64>
int run(const char *name, int per_iter, float *out, unsigned long long *clk, int waves_per_simd = 8)
{
    const int blocks = 256 * waves_per_simd, threads = 256;   // one 256-thread workgroup = one wave on each SIMD of a CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<OP, LANES><<<blocks, threads>>>(out, 1.0001f, clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<OP, LANES><<<blocks, threads>>>(out, 1.0001f, clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c = 0;
    CHECK(hipMemcpy(&c, clk, sizeof c, hipMemcpyDeviceToHost));
    const double waves = (double)blocks * threads / 64, simds = 256.0 * 4;
    const double insts_per_simd = waves * ITER * per_iter / simds;
    const double ns_per_inst = ms * 1e6 / insts_per_simd;
    // s_memtime ticks at 100 MHz on gfx9: clock = cycles... report ns only plus the wave's own view
    printf("%-28s %8.3f ms  %7.3f ns/wave-inst/SIMD  (%.2f cyc @2.4GHz, %.2f @2.1GHz) memtime=%llu\n", name, ms, ns_per_inst,
           ns_per_inst * 2.4, ns_per_inst * 2.1, c);
    return 0;
}

int main()
{
    float *out; unsigned long long *clk;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
    CHECK(hipMalloc(&clk, 8));
    run<0>("v_fma_f32", 8, out, clk);
    run<1>("v_mul_f32", 8, out, clk);
    run<2>("v_add_f32", 8, out, clk);
    run<3>("v_rsq_f32", 8, out, clk);
    run<4>("v_rcp_f32", 8, out, clk);
    run<5>("v_add_f64", 8, out, clk);
    run<6>("v_cvt_f64_f32 + v_add_f32", 16, out, clk);
    run<7>("v_cvt_f32_f64 + v_add_f64", 16, out, clk);
    run<8>("v_min3_f32", 8, out, clk);
    run<9>("sqrtf correctly rounded", 8, out, clk);
    run<10>("v_fma_f64", 8, out, clk);
    run<11>("v_pk_fma_f32 (2 fma)", 4, out, clk);
    run<12>("v_pk_mul_f32 (2 mul)", 4, out, clk);
    run<0, 32>("v_fma_f32, 32 lanes active", 8, out, clk);
    run<0, 16>("v_fma_f32, 16 lanes active", 8, out, clk);
    run<11, 32>("v_pk_fma_f32, 32 lanes active", 4, out, clk);
    run<3, 32>("v_rsq_f32, 32 lanes active", 8, out, clk);
    for (int w = 1; w <= 4; w *= 2) {
        char name[96];
        snprintf(name, sizeof name, "v_add_f32_dpp quad_perm, %d w/SIMD", w); run<16>(name, 8, out, clk, w);
        snprintf(name, sizeof name, "v_add_f32_dpp row_shr:1, %d w/SIMD", w); run<17>(name, 8, out, clk, w);
        snprintf(name, sizeof name, "v_add_f32 (same shape), %d w/SIMD", w); run<18>(name, 8, out, clk, w);
        snprintf(name, sizeof name, "one chain of 8 v_add_f32_dpp, %d w/SIMD", w); run<19>(name, 8, out, clk, w);
        snprintf(name, sizeof name, "one chain of 8 v_add_f32, %d w/SIMD", w); run<20>(name, 8, out, clk, w);
        snprintf(name, sizeof name, "v_mov_b32_dpp + v_add_f32, %d w/SIMD", w); run<21>(name, 16, out, clk, w);
    }
    // how fast can ONE wave issue (8 independent chains each)?  waves per SIMD = 1, 2, 3, 4
    for (int w = 1; w <= 4; w++) {
        char name[64];
        snprintf(name, sizeof name, "v_fma_f32, %d wave(s)/SIMD", w); run<0>(name, 8, out, clk, w);
        snprintf(name, sizeof name, "v_pk_fma_f32, %d wave(s)/SIMD", w); run<11>(name, 4, out, clk, w);
        snprintf(name, sizeof name, "v_rsq_f32, %d wave(s)/SIMD", w); run<3>(name, 8, out, clk, w);
    }
    // round 4: transcendental beside plain VALU work (per iteration 8 v_rsq + 32 v_fma = 40 instructions;
    // no overlap: 8 x 4 + 32 = 64 issue units, full overlap: 32)
    for (int w = 1; w <= 8; w *= 2) {
        char name[64];
        snprintf(name, sizeof name, "rsq+4fma interleaved, %d w/SIMD", w); run<13>(name, 40, out, clk, w);
        snprintf(name, sizeof name, "8rsq then 32fma, %d w/SIMD", w); run<14>(name, 40, out, clk, w);
        if (w >= 2) { snprintf(name, sizeof name, "rsq waves | fma waves, %d w/SIMD", w); run<15>(name, 20, out, clk, w); }
    }
    return 0;
}
