// Diagnostic: what a "last workgroup to arrive finishes the job" costs against a second launch.
//   mode 0  work launch + a separate merge launch (what a split k_star_marg + k_marg_merge pair does)
//   mode 1  the work launch's workgroups publish their shares (agent-scope release), count themselves in on a per-chunk
//           counter, and the last of a chunk's n_split workgroups merges the chunk (agent-scope acquire)
//   mode 2  one counter for the whole grid: the last workgroup sums every partial in a fixed order (a one-launch b9_logpost)
//   mode 3  mode 2's work + a separate one-workgroup finalize launch (what b9_logpost does now)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/last_probe tools/probes/last_probe.hip && /tmp/last_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ double fake_work(const double *table, int chunk, int rounds)
{
    const int tid = threadIdx.x;
    double acc = 0.0;
    for (int r = 0; r < rounds; ++r) {
        const double v = table[((size_t)(chunk * 37 + r) * 256 + tid) & ((1u << 18) - 1)];
        double x = v;
#pragma unroll
        for (int k = 0; k < 24; ++k) x = fma(x, 0.999, 0.001);
        acc += x;
    }
    return acc;
}

// mode 4: the same without any fence -- shares written through (sc1 stores), a wait for their completion, the counter, and
// sc1 loads in the last arriver; every share is checked against its expected value (stale reads show in *err)
__global__ __launch_bounds__(256) void k_work_wt(const double *table, double *shares, unsigned *counters, double *out, int n_split, int rounds, int iter, unsigned *err)
{
    const int xcd = blockIdx.x & 7, i_x = blockIdx.x >> 3;
    const int i_s = i_x / n_split, split = i_x - i_s * n_split;
    const int chunk = i_s * 8 + xcd, tid = threadIdx.x;
    const double acc = fake_work(table, chunk, rounds);
    double *sh = shares + ((size_t)chunk * n_split + split) * 128;
    if (tid < 128) __hip_atomic_store(sh + tid, acc + tid + iter + split, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __shared__ int s_last;
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(counters + chunk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == (unsigned)n_split - 1u;
    }
    __syncthreads();
    if (!s_last) return;
    if (tid < 64) {
        double t = 0.0;
        unsigned bad = 0;
        for (int k = 0; k < n_split; ++k) {
            const double *p = shares + ((size_t)chunk * n_split + k) * 128;
            const double a = __hip_atomic_load(p + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(p + 64 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bad += (a != acc + tid + iter + k) + (b != acc + 64 + tid + iter + k);
            t += a + b;
        }
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (bad) atomicAdd(err, bad);
        if (tid == 0) { out[chunk] = t; __hip_atomic_store(counters + chunk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
}

__global__ __launch_bounds__(256) void k_work(const double *table, double *shares, unsigned *counters, double *out, int n_split, int rounds, int mode)
{
    const int xcd = blockIdx.x & 7, i_x = blockIdx.x >> 3;
    const int i_s = i_x / n_split, split = i_x - i_s * n_split;
    const int chunk = i_s * 8 + xcd, tid = threadIdx.x;
    const double acc = fake_work(table, chunk, rounds);
    double *sh = shares + ((size_t)chunk * n_split + split) * 128;
    if (tid < 128) sh[tid] = acc + tid;
    if (mode == 0) return;
    __shared__ int s_last;
    __threadfence();                                         // release: this workgroup's shares
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(counters + chunk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == (unsigned)n_split - 1u;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();                                         // acquire: the other workgroups' shares
    if (tid < 64) {
        double t = 0.0;
        for (int k = 0; k < n_split; ++k) {
            const double *p = shares + ((size_t)chunk * n_split + k) * 128;
            t += __hip_atomic_load(p + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + __hip_atomic_load(p + 64 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (tid == 0) { out[chunk] = t; __hip_atomic_store(counters + chunk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
}

__global__ __launch_bounds__(64) void k_merge(const double *shares, double *out, int n_split)
{
    const int chunk = blockIdx.x, tid = threadIdx.x;
    double t = 0.0;
    for (int k = 0; k < n_split; ++k) {
        const double *p = shares + ((size_t)chunk * n_split + k) * 128;
        t += p[tid] + p[64 + tid];
    }
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (tid == 0) out[chunk] = t;
}

// whole-grid counter: every workgroup leaves 4 partials; the last one sums all of them (fixed order) into host-mapped memory
__global__ __launch_bounds__(256) void k_work_all(const double *table, double *partial, unsigned *counter, double *out, int rounds, int mode)
{
    const int tid = threadIdx.x, n_wg = gridDim.x;
    const double acc = fake_work(table, blockIdx.x, rounds);
    double t = acc;
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if ((tid & 63) == 0) partial[blockIdx.x * 4 + (tid >> 6)] = t;
    if (mode == 3) return;
    __shared__ int s_last;
    __shared__ double s_red[4];
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == (unsigned)n_wg - 1u;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    double a = 0.0;
    for (int j = tid; j < n_wg * 4; j += 256) a += __hip_atomic_load(partial + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((tid & 63) == 0) s_red[tid >> 6] = a;
    __syncthreads();
    if (tid == 0) { *out = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]); __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}

__global__ __launch_bounds__(256) void k_final(const double *partial, double *out, int n)
{
    __shared__ double s_red[4];
    const int tid = threadIdx.x;
    double a = 0.0;
    for (int j = tid; j < n; j += 256) a += partial[j];
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((tid & 63) == 0) s_red[tid >> 6] = a;
    __syncthreads();
    if (tid == 0) *out = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

int main()
{
    const int n_chunks = 160, n_launch = 300;
    double *table, *shares, *out, *partial, *h_out, *h_out_dev; unsigned *counters;
    CK(hipMalloc(&table, sizeof(double) << 18));
    CK(hipMemset(table, 0, sizeof(double) << 18));
    CK(hipMalloc(&shares, sizeof(double) * n_chunks * 16 * 128));
    CK(hipMalloc(&out, sizeof(double) * n_chunks));
    CK(hipMalloc(&partial, sizeof(double) * 8192 * 4));
    CK(hipMalloc(&counters, sizeof(unsigned) * (n_chunks + 1)));
    CK(hipMemset(counters, 0, sizeof(unsigned) * (n_chunks + 1)));
    CK(hipHostMalloc((void **)&h_out, 64, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **)&h_out_dev, h_out, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int rounds : {8, 40}) for (int n_split : {8}) for (int mode = 0; mode < 2; ++mode) for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < n_launch; ++i) {
            hipLaunchKernelGGL(k_work, dim3(n_chunks * n_split), dim3(256), 0, s, table, shares, counters, out, n_split, rounds, mode);
            if (mode == 0) hipLaunchKernelGGL(k_merge, dim3(n_chunks), dim3(64), 0, s, shares, out, n_split);
        }
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<double> h(n_chunks); CK(hipMemcpy(h.data(), out, sizeof(double) * n_chunks, hipMemcpyDeviceToHost));
        double chk = 0; for (double v : h) chk += v;
        std::printf("split launch, %d workgroups, rounds %d, mode %d (%s): %.2f us per step   (checksum %.6g)\n", n_chunks * n_split, rounds, mode,
                    mode == 0 ? "work + merge launch" : "last arriver merges", 1e3 * ms / n_launch, chk);
    }
    unsigned *err; CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    for (int rounds : {8, 40}) for (int n_split : {8, 16}) for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < n_launch; ++i)
            hipLaunchKernelGGL(k_work_wt, dim3(n_chunks * n_split), dim3(256), 0, s, table, shares, counters, out, n_split, rounds, i, err);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h_err = 0; CK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
        std::vector<double> h(n_chunks); CK(hipMemcpy(h.data(), out, sizeof(double) * n_chunks, hipMemcpyDeviceToHost));
        double chk = 0; for (double v : h) chk += v;
        std::printf("split launch, %d workgroups, rounds %d, mode 4 (write-through shares, no fence): %.2f us per step   (checksum %.6g, stale or wrong shares %u)\n",
                    n_chunks * n_split, rounds, 1e3 * ms / n_launch, chk, h_err);
    }
    for (int n_wg : {160, 800}) for (int rounds : {4, 16}) for (int mode = 2; mode < 4; ++mode) for (int rep = 0; rep < 2; ++rep) {
        // host-visible completion per step, as b9_logpost: launch, (finalize), synchronize
        double tot = 0.0;
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < n_launch; ++i) {
            hipLaunchKernelGGL(k_work_all, dim3(n_wg), dim3(256), 0, s, table, partial, counters + n_chunks, h_out_dev, rounds, mode);
            if (mode == 3) hipLaunchKernelGGL(k_final, dim3(1), dim3(256), 0, s, partial, h_out_dev, n_wg * 4);
        }
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        tot = *h_out;
        std::printf("one-counter launch, %d workgroups, rounds %d, mode %d (%s): %.2f us per step   (value %.6g)\n", n_wg, rounds, mode,
                    mode == 2 ? "last arriver sums" : "work + finalize launch", 1e3 * ms / n_launch, tot);
    }
    return 0;
}
