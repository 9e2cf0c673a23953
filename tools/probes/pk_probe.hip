// Diagnostic: do the packed fp32 instructions of gfx950 read BOTH halves of a 64-bit SGPR pair operand?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_probe tools/probes/pk_probe.hip && /tmp/pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const f2 *box, const f2 *sw, const f2 *so, float *out)
{
    const int l = threadIdx.x;
    f2 b = box[0];                                 // uniform address -> an SGPR pair
    unsigned long long bs = __builtin_amdgcn_readfirstlane(((const unsigned *)box)[0]) | ((unsigned long long)__builtin_amdgcn_readfirstlane(((const unsigned *)box)[1]) << 32);
    f2 r, r2;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(sw[l]), "s"(bs), "v"(so[l]));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r2) : "v"(sw[l]), "s"(bs), "v"(so[l]));
    float m;
    asm volatile("v_max3_f32 %0, %1, %2, 0" : "=v"(m) : "v"(r.x), "v"(r2.y));
    out[l * 8 + 0] = r.x; out[l * 8 + 1] = r.y; out[l * 8 + 2] = r2.x; out[l * 8 + 3] = r2.y; out[l * 8 + 4] = m;
    const f2 c = __builtin_elementwise_fma(sw[l], b, -so[l]);
    out[l * 8 + 5] = c.x; out[l * 8 + 6] = c.y;
}
int main()
{
    f2 *box, *sw, *so; float *out;
    hipMallocManaged(&box, 64); hipMallocManaged(&sw, 64 * 8); hipMallocManaged(&so, 64 * 8); hipMallocManaged(&out, 64 * 8 * 4);
    box[0] = f2{3.0f, 5.0f};
    for (int l = 0; l < 64; ++l) { sw[l] = f2{2.0f + l, 10.0f}; so[l] = f2{1.0f, 7.0f}; }
    k<<<1, 64>>>(box, sw, so, out);
    hipDeviceSynchronize();
    for (int l = 0; l < 2; ++l)
        std::printf("lane %d: sw.lo*3-1 = %g (want %g)  sw.hi*5-7 = %g (want 43)  1-sw.lo*3 = %g  7-sw.hi*5 = %g (want -43)  max3 = %g  compiler: %g %g\n", l, out[l*8], (2.0f + l) * 3 - 1, out[l*8+1], out[l*8+2], out[l*8+3], out[l*8+4], out[l*8+5], out[l*8+6]);
    return 0;
}
