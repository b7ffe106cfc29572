// How fast are 16-byte LDS reads when the 64 lanes of a wave read 1, 2 or 4 distinct
// addresses (lane groups of 64 / 32 / 16), with the groups' addresses in the same banks
// or skewed by 16 bytes?   hipcc --offload-arch=gfx950 -O3 lds_groups.hip -o lds_groups
#include <hip/hip_runtime.h>
#include <cstdio>

template <int GSHIFT, int STRIDE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[4][4 * 264];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = lane; i < 4 * 264; i += 64) lds[wave][i] = (float)i;
    __syncthreads();
    const float *base = lds[wave] + (lane >> GSHIFT) * STRIDE;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 64; j += 8) {
            const float4 a = *reinterpret_cast<const float4 *>(base + j);
            const float4 b = *reinterpret_cast<const float4 *>(base + j + 4);
            const float4 c = *reinterpret_cast<const float4 *>(base + 64 + j);
            const float4 d = *reinterpret_cast<const float4 *>(base + 64 + j + 4);
            const float4 e = *reinterpret_cast<const float4 *>(base + 128 + j);
            const float4 f = *reinterpret_cast<const float4 *>(base + 128 + j + 4);
            const float4 g = *reinterpret_cast<const float4 *>(base + 192 + j);
            const float4 h = *reinterpret_cast<const float4 *>(base + 192 + j + 4);
            acc.x += a.x + b.y + c.z + d.w; acc.y += e.x + f.y + g.z + h.w;
            acc.z += a.w + c.x + e.y + g.x; acc.w += b.x + d.y + f.z + h.y;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int GSHIFT, int STRIDE>
static void run(const char *name, float *out)
{
    const int blocks = 256 * 8, iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<GSHIFT, STRIDE><<<blocks, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<GSHIFT, STRIDE><<<blocks, 256>>>(out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double reads = (double)blocks * 4 * iters * 64;     // wave-level b128 reads
    printf("%-34s %.3f ms  %.2f ns per wave-read per CU-resident wave set\n", name, ms, ms * 1e6 / (reads / (256.0 * 4)));
}

int main()
{
    float *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    run<6, 0>("1 address (broadcast)", out);
    run<5, 256>("2 groups, same banks", out);
    run<5, 260>("2 groups, skewed 16 B", out);
    run<4, 256>("4 groups, same banks", out);
    run<4, 260>("4 groups, skewed 16 B", out);
    run<4, 264>("4 groups, skewed 32 B", out);
    return 0;
}
