// Diagnostic: what makes the first reads after a kernel boundary slow?  Back-to-back launches of 744 workgroups; each
// reads R lines (128 B) that the PREVIOUS launch wrote and rewrites them for the next.  Varied: who wrote the line a
// workgroup reads (a workgroup of the same XCD or of another), whole-line or one-word writes, how many workgroups read
// the same line, and the store / load cache policy.
// Measured on MI355X (round 3): a chain of dependent loads costs ~75-125 ns per load on lines nobody rewrote (they stay in
// the L2 across launches), ~170 ns on lines the same XCD rewrote in the previous launch and ~260-430 ns on lines another
// XCD rewrote; independent loads: 64 warm lines per workgroup cost nothing over 1, 64 rewritten lines +8 us (own lines) /
// +5 us (the same 64 lines for all workgroups of an XCD) -- about 1.3 ns per cold REQUEST and XCD, whether or not the lines
// are distinct.  (Whether the 12 buffers a workgroup reads are separate allocations or slices of one arena made no
// difference: 5.1 against 5.4 us -- not a TLB effect.)
//   hipcc --offload-arch=gfx950 -O3 -o build/cold_probe tools/probes/cold_probe.hip && ./build/cold_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Cfg { int R, share, writer_shift, full_line, store_policy, load_policy, rewrite; };

// line a workgroup owns for slot k: share = 1: its own; share = 93: one per XCD (all workgroups of an XCD read the same
// lines); share = 744: the same lines for everybody
__device__ __forceinline__ int line_of(int wg, int k, int R, int share)
{
    const int owner = share == 1 ? wg : (share == 93 ? (wg & 7) : 0);
    return owner * R + k;
}

__global__ __launch_bounds__(256) void k_cold(double *buf, double *sink, Cfg c, int step)
{
    const int tid = threadIdx.x, wg = blockIdx.x;
    double acc = 0.0;
    if (tid < 64) {
        if (c.load_policy == 3) {
            // independent loads: all requested before any is waited for (R <= 64)
            double v[64];
#pragma unroll
            for (int k = 0; k < 64; ++k) v[k] = k < c.R ? buf[(size_t)line_of(wg, k, c.R, c.share) * 16 + (tid & 15)] : 0.0;
#pragma unroll
            for (int k = 0; k < 64; ++k) acc += v[k];
        } else
        for (int k = 0; k < c.R; ++k) {
            const double *p = buf + (size_t)line_of(wg, k, c.R, c.share) * 16 + (tid & 15);
            double v;
            if (c.load_policy == 1) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
            else if (c.load_policy == 2) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
            else asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (a DEPENDENT chain of R loads: time / R = one load's latency)
            acc += v;
        }
    }
    // the writer of line (wg, k) for the NEXT launch: workgroup wg - writer_shift (1: another XCD, 8: the same XCD)
    if (c.rewrite && tid < 16 && (c.share == 1 || (c.share == 93 ? wg < 8 + c.writer_shift && wg >= c.writer_shift : wg == c.writer_shift))) {
        const int target = (wg - c.writer_shift + 744) % 744;
        for (int k = 0; k < c.R; ++k) {
            double *p = buf + (size_t)line_of(target, k, c.R, c.share) * 16 + tid;
            const double v = acc * 1e-30 + step;
            if (c.full_line || tid == 5) {
                if (c.store_policy == 1) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
                else if (c.store_policy == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
                else *p = v;
            }
        }
    }
    if (acc == 12345.678) sink[0] = acc;
}

int main()
{
    const size_t lines = 744 * 64;
    double *buf, *sink;
    CK(hipMalloc(&buf, lines * 128)); CK(hipMemset(buf, 0, lines * 128)); CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 1500;
    struct { const char *name; Cfg c; } cases[] = {
        {"chain R=16 own lines, warm", {16, 1, 1, 1, 0, 0, 0}},
        {"chain R=16 own lines, written by the same XCD", {16, 1, 8, 1, 0, 0, 1}},
        {"chain R=16 own lines, written by another XCD", {16, 1, 1, 1, 0, 0, 1}},
        {"independent R=1  own line, other XCD wrote", {1, 1, 1, 1, 0, 3, 1}},
        {"independent R=4  own lines, other XCD wrote", {4, 1, 1, 1, 0, 3, 1}},
        {"independent R=16 own lines, other XCD wrote", {16, 1, 1, 1, 0, 3, 1}},
        {"independent R=64 own lines, other XCD wrote", {64, 1, 1, 1, 0, 3, 1}},
        {"independent R=64 own lines, same XCD wrote", {64, 1, 8, 1, 0, 3, 1}},
        {"independent R=64 own lines, warm", {64, 1, 1, 1, 0, 3, 0}},
        {"independent R=16 lines shared by an XCD's workgroups, other XCD wrote", {16, 93, 1, 1, 0, 3, 1}},
        {"independent R=64 lines shared by an XCD's workgroups, other XCD wrote", {64, 93, 1, 1, 0, 3, 1}},
        {"independent R=64 lines shared by an XCD's workgroups, warm", {64, 93, 1, 1, 0, 3, 0}},
        {"independent R=64 lines shared by ALL workgroups, rewritten", {64, 744, 1, 1, 0, 3, 1}},
        {"independent R=64 lines shared by ALL workgroups, warm", {64, 744, 1, 1, 0, 3, 0}},
    };
    for (auto &cs : cases) {
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_cold, dim3(744), dim3(256), 0, 0, buf, sink, cs.c, i);
        CK(hipEventRecord(e0));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_cold, dim3(744), dim3(256), 0, 0, buf, sink, cs.c, i);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("%-72s %.2f us per launch\n", cs.name, 1e3 * ms / N);
    }
    return 0;
}
