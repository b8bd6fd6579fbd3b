// random64.hip — what does MI355X deliver for random 64-byte record reads?  (The BVH walk's access pattern.)
//   indep   : every lane reads records at hashed indices, no dependency between reads
//   chain   : the next index depends on the loaded data (one outstanding read per lane, like a BVH walk)
//   coop    : four lanes read one record together (16 lines per wave instruction instead of 64), independent
// usage: random64 [records] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ __launch_bounds__(256) void k_indep(const float4 *__restrict__ rec, uint32_t nrec, int iters, float *out) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    uint32_t s = mix(tid * 2654435761u + 1u);
    for (int it = 0; it < iters; ++it) {
        s = mix(s + 0x9e3779b9u);
        const float4 *p = rec + 4 * (size_t)(s % nrec);
        float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 123.456f) out[tid] = acc;
}

__global__ __launch_bounds__(256) void k_chain(const float4 *__restrict__ rec, uint32_t nrec, int iters, float *out) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    uint32_t s = mix(tid * 2654435761u + 1u);
    for (int it = 0; it < iters; ++it) {
        const float4 *p = rec + 4 * (size_t)(s % nrec);
        float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z;
        s = mix(s + __float_as_uint(d.w) + 0x9e3779b9u);
    }
    if (acc == 123.456f) out[tid] = acc;
}

// four lanes per record: lane l reads float4 (l & 3) of the record chosen by its group of four
__global__ __launch_bounds__(256) void k_coop(const float4 *__restrict__ rec, uint32_t nrec, int iters, float *out) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    uint32_t s = mix((tid >> 2) * 2654435761u + 1u);
    for (int it = 0; it < iters; ++it) {
        s = mix(s + 0x9e3779b9u);
        const float4 *p = rec + 4 * (size_t)(s % nrec) + (tid & 3);
        float4 a = p[0];
        acc += a.x + a.w;
    }
    if (acc == 123.456f) out[tid] = acc;
}

// chain with one 16-byte read per step (is the cost per line or per instruction?)
__global__ __launch_bounds__(256) void k_chain16(const float4 *__restrict__ rec, uint32_t nrec, int iters, float *out) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    uint32_t s = mix(tid * 2654435761u + 1u);
    for (int it = 0; it < iters; ++it) {
        const float4 *p = rec + 4 * (size_t)(s % nrec);
        float4 d = p[3];
        acc += d.x;
        s = mix(s + __float_as_uint(d.w) + 0x9e3779b9u);
    }
    if (acc == 123.456f) out[tid] = acc;
}

// chain over records of V float4 (64 B * V / 4): does a bigger record cost more per random access?
template <int V>
__global__ __launch_bounds__(256) void k_chainV(const float4 *__restrict__ rec, uint32_t nrec, int iters, float *out) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    uint32_t s = mix(tid * 2654435761u + 1u);
    uint32_t nr = (uint32_t)((size_t)nrec * 4 / V);
    for (int it = 0; it < iters; ++it) {
        const float4 *p = rec + V * (size_t)(s % nr);
        float4 v[V];
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = p[j];
        uint32_t h = 0;
#pragma unroll
        for (int j = 0; j < V; ++j) { acc += v[j].x; h += __float_as_uint(v[j].w); }
        s = mix(s + h + 0x9e3779b9u);
    }
    if (acc == 123.456f) out[tid] = acc;
}

int main(int argc, char **argv) {
    uint32_t nrec = argc > 1 ? (uint32_t)atoll(argv[1]) : 10000000u;
    int iters = argc > 2 ? atoi(argv[2]) : 256;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int ncu = prop.multiProcessorCount;
    float4 *rec; float *out;
    CK(hipMalloc(&rec, (size_t)nrec * 64));
    CK(hipMalloc(&out, (size_t)ncu * 8 * 256 * 4 * 4));
    {   // fill with small integers so the chain's data-dependent index is well mixed
        std::vector<uint32_t> h((size_t)nrec * 16);
        uint32_t x = 12345u;
        for (size_t i = 0; i < h.size(); ++i) { x = x * 1664525u + 1013904223u; h[i] = x >> 9; }
        CK(hipMemcpy(rec, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("records %u (%.0f MB), iters %d, CUs %d\n", nrec, nrec * 64.0 / 1e6, iters, ncu);
    const char *names[6] = {"indep", "chain", "coop", "chain16", "chain128", "chain256"};
    for (int k = 0; k < 6; ++k)
        for (int bpc = 2; bpc <= 8; bpc *= 4) {
            int grid = ncu * bpc;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, 0));
                if (k == 0) hipLaunchKernelGGL(k_indep, dim3(grid), dim3(256), 0, 0, rec, nrec, iters, out);
                if (k == 1) hipLaunchKernelGGL(k_chain, dim3(grid), dim3(256), 0, 0, rec, nrec, iters, out);
                if (k == 2) hipLaunchKernelGGL(k_coop, dim3(grid), dim3(256), 0, 0, rec, nrec, iters, out);
                if (k == 3) hipLaunchKernelGGL(k_chain16, dim3(grid), dim3(256), 0, 0, rec, nrec, iters, out);
                if (k == 4) hipLaunchKernelGGL(k_chainV<8>, dim3(grid), dim3(256), 0, 0, rec, nrec, iters, out);
                if (k == 5) hipLaunchKernelGGL(k_chainV<16>, dim3(grid), dim3(256), 0, 0, rec, nrec, iters, out);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            double recs = (double)grid * 256 * iters / (k == 2 ? 4 : 1);
            double bytes = recs * (k == 3 ? 16 : k == 4 ? 128 : k == 5 ? 256 : 64);
            printf("%-8s blocks/CU %d: %8.3f ms  %7.2f Grec/s  %6.2f TB/s\n", names[k], bpc, best, recs / best / 1e6, bytes / best / 1e9);
            fflush(stdout);
        }
    CK(hipFree(rec)); CK(hipFree(out));
    return 0;
}
