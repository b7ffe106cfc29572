// Exhaustive search: which cheap formulas reproduce RN(1 / RN(sqrt a)) for every float a in
// [2^-62, 2^62]?   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
//   -fhip-fp32-correctly-rounded-divide-sqrt invsqrt_search.hip -o invsqrt_search
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ float sqrt_rn(float a, float &r_out)
{
    const float r = __builtin_amdgcn_rsqf(a);
    const float g = a * r, h = 0.5f * r;
    r_out = r;
    return __builtin_fmaf(__builtin_fmaf(-g, g, a), h, g);
}
__device__ __forceinline__ float newton(float s, float y) { return __builtin_fmaf(__builtin_fmaf(-s, y, 1.0f), y, y); }

#define NV 8
__device__ __forceinline__ void variants(float a, float *v)
{
    float r;
    const float s = sqrt_rn(a, r);
    v[0] = newton(s, __builtin_amdgcn_rcpf(s));              // in use: v_rcp + 1 step
    v[1] = newton(s, r);                                     // rsq start, 1 step
    v[2] = newton(s, newton(s, r));                          // rsq start, 2 steps
    {   // refine r with the sqrt residual first (r1 = r + r*(d*h*r)), then 1 step
        const float g = a * r, h = 0.5f * r, d = __builtin_fmaf(-g, g, a);
        const float r1 = __builtin_fmaf(d * h, r * r, r);
        v[3] = newton(s, r1);
        v[4] = r1;                                           // RN(1/sqrt(a))-ish, no step at all
    }
    {   // residual against the exact product s*y in two pieces
        const float y = newton(s, r);
        const float e = __builtin_fmaf(-s, y, 1.0f);
        v[5] = __builtin_fmaf(e, y, y);
    }
    {   // v_rcp of the ESTIMATE g = a*r (issued before the sqrt correction: shorter chain), 1 step
        const float g = a * r;
        v[6] = newton(s, __builtin_amdgcn_rcpf(g));
    }
    v[7] = newton(s, newton(s, __builtin_amdgcn_rcpf(a * r)));
}

__global__ void k(uint32_t lo, uint32_t hi, unsigned long long *out)
{
    const uint64_t span = (uint64_t)hi - lo + 1;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad[NV] = {};
    for (; i < span; i += stride) {
        const float a = __uint_as_float(lo + (uint32_t)i);
        const float ref = 1.0f / sqrtf(a);
        float v[NV];
        variants(a, v);
        for (int j = 0; j < NV; j++) if (__float_as_uint(v[j]) != __float_as_uint(ref)) bad[j]++;
    }
    for (int j = 0; j < NV; j++) if (bad[j]) atomicAdd(&out[j], bad[j]);
}

int main()
{
    unsigned long long *out, h[NV];
    hipMalloc(&out, sizeof(h)); hipMemset(out, 0, sizeof(h));
    float lo = 0x1p-62f, hi = 0x1p62f; uint32_t lb, hb; memcpy(&lb, &lo, 4); memcpy(&hb, &hi, 4);
    k<<<4096, 256>>>(lb, hb, out);
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const char *name[NV] = {"v_rcp(s) + 1 step (in use)", "rsq + 1 step", "rsq + 2 steps", "refined rsq + 1 step",
                            "refined rsq, no step", "rsq + 2 steps (alt)", "v_rcp(a*r) + 1 step", "v_rcp(a*r) + 2 steps"};
    for (int j = 0; j < NV; j++) printf("%-32s mismatches %llu\n", name[j], h[j]);
    return 0;
}
