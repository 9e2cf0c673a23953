// Diagnostic: cost of a per-step rendezvous among the workgroups of ONE XCD (L2-local atomics and loads) against a
// device-wide one -- the question a persistent sampler kernel (walker <-> XCD) would turn on.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/sync_probe tools/probes/sync_probe.hip && /tmp/sync_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ double load_l2(const double *p)
{
    double v;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned load_l2(const unsigned *p)
{
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// mode 0: rendezvous per XCD (workgroup-scope atomic = executed in the XCD's L2), mode 1: device-wide (agent scope)
__global__ __launch_bounds__(256) void k_sync(unsigned *counters, double *partials, double *out, int n_iter, int per_group, int mode, unsigned *xcc_seen, int *err)
{
    __shared__ double s_tot;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = mode == 0 ? (blockIdx.x & 7) : 0, idx = mode == 0 ? (blockIdx.x >> 3) : blockIdx.x;
    if (tid == 0) { unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc_seen[blockIdx.x] = xcc & 15u; }
    unsigned *ctr = counters + grp * 64;                     // one counter per group, 256 B apart
    double *part = partials + (size_t)grp * 2 * per_group * 4;
    double acc = 0.0;
    for (int it = 0; it < n_iter; ++it) {
        double *pp = part + (size_t)(it & 1) * per_group * 4;
        if (lane == 0) pp[idx * 4 + wave] = (double)(it + idx + wave);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (mode == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else { __threadfence(); __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
        if (wave == 0) {
            const unsigned want = (unsigned)per_group * (unsigned)(it + 1);
            int spins = 0;
            for (;;) {
                const unsigned c = mode == 0 ? load_l2(ctr) : __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (c >= want) break;
                if (++spins > 2000000) { if (lane == 0) *err = 1; break; }         // (an exit every wave reaches)
                __builtin_amdgcn_s_sleep(1);
            }
            if (mode == 1) __threadfence();
            double t = 0.0;
            for (int j = lane; j < per_group * 4; j += 64) t += mode == 0 ? load_l2(pp + j) : __hip_atomic_load(pp + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
            if (lane == 0) s_tot = t;
        }
        __syncthreads();
        acc += s_tot;
    }
    if (tid == 0) out[blockIdx.x] = acc;
}

int main()
{
    const int n_wg = 744, n_iter = 200;
    unsigned *ctr, *xcc; double *part, *out; int *err;
    CK(hipMalloc(&ctr, 8 * 64 * sizeof(unsigned)));
    CK(hipMalloc(&xcc, n_wg * sizeof(unsigned)));
    CK(hipMalloc(&part, sizeof(double) * 8 * 2 * n_wg * 4));
    CK(hipMalloc(&out, sizeof(double) * n_wg));
    CK(hipMalloc(&err, sizeof(int)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        const int per_group = mode == 0 ? n_wg / 8 : n_wg;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(ctr, 0, 8 * 64 * sizeof(unsigned)));
            CK(hipMemset(err, 0, sizeof(int)));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_sync, dim3(n_wg), dim3(256), 0, 0, ctr, part, out, n_iter, per_group, mode, xcc, err);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            int h_err = 0; CK(hipMemcpy(&h_err, err, sizeof(int), hipMemcpyDeviceToHost));
            std::vector<double> h(n_wg); CK(hipMemcpy(h.data(), out, sizeof(double) * n_wg, hipMemcpyDeviceToHost));
            std::vector<unsigned> hx(n_wg); CK(hipMemcpy(hx.data(), xcc, sizeof(unsigned) * n_wg, hipMemcpyDeviceToHost));
            int bad_xcc = 0; for (int b = 0; b < n_wg; ++b) bad_xcc += (int)hx[b] != (b & 7);
            // expected total of one iteration's partials of a group: sum over idx, wave of (it + idx + wave)
            double want = 0.0;
            for (int it = 0; it < n_iter; ++it) for (int i = 0; i < per_group; ++i) for (int wv = 0; wv < 4; ++wv) want += it + i + wv;
            int wrong = 0; for (int b = 0; b < n_wg; ++b) wrong += h[b] != want;
            std::printf("mode %d (%s, %d workgroups per rendezvous): %.2f us per iteration; err %d, wrong sums %d, workgroups off their XCD %d\n",
                        mode, mode == 0 ? "per XCD, L2-local" : "device-wide, agent scope", per_group, 1e3 * ms / n_iter, h_err, wrong, bad_xcc);
        }
    }
    return 0;
}
