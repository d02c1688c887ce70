// calib_records.hip -- calibration of the memory path for the traversal kernels' access shape (tools, not product).
//
// k_trace reads scattered 64-B records (BVH nodes) and 48-B ones (triangles) from tables far larger than the caches.
// This program reads N random 64-B records of a 1 GiB table in four ways and times them:
//   lane4   : every lane fetches its own record with four 16-B loads (round 1's k_trace shape)
//   coopdma : four adjacent lanes fetch the four quarters of one record, straight into LDS (global_load_lds_dwordx4),
//             and every lane then reads its own record back from LDS with four ds_read_b128
//   coopreg : the same cooperative loads into registers (no transpose; the load path alone)
//   stream  : 16 B per lane, consecutive (the guide's calibrated shape)
// Run under `rocprofv3 --pmc FETCH_SIZE` it also calibrates FETCH_SIZE for the scattered shapes against the known byte
// counts (MI355X_MICROARCH.md says only wide coalesced streaming reads are calibrated: x2).
//   hipcc --offload-arch=gfx950 -O3 -o calib_records tools/calib_records.hip && ./calib_records
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }

constexpr int kIters = 64;           // records per lane

__global__ void __launch_bounds__(64) k_lane4(const float4* __restrict__ tab, uint32_t mask, float* out) {
    const uint32_t gid = blockIdx.x * 64u + threadIdx.x;
    float acc = 0.0f;
    for (int it = 0; it < kIters; it++) {
        const uint32_t idx = hash32(gid * kIters + it) & mask;
        const float4* p = tab + 4 * (size_t)idx;
        const float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
    }
    out[gid] = acc;
}

__global__ void __launch_bounds__(64) k_coopreg(const float4* __restrict__ tab, uint32_t mask, float* out) {
    const uint32_t gid = blockIdx.x * 64u + threadIdx.x;
    const uint32_t lane = threadIdx.x;
    float acc = 0.0f;
    for (int it = 0; it < kIters; it++) {
        const uint32_t idx = hash32(gid * kIters + it) & mask;       // this lane's record
        float4 v[4];
        for (int k = 0; k < 4; k++) {
            const uint32_t src = (uint32_t)__shfl((int)idx, 16 * k + (int)(lane >> 2));
            v[k] = tab[4 * (size_t)src + (lane & 3u)];
        }
        acc += v[0].x + v[1].y + v[2].z + v[3].w;
    }
    out[gid] = acc;
}

__global__ void __launch_bounds__(64) k_coopdma(const float4* __restrict__ tab, uint32_t mask, float* out) {
    __shared__ float4 img[256];                                        // 64 records x 64 B
    const uint32_t gid = blockIdx.x * 64u + threadIdx.x;
    const uint32_t lane = threadIdx.x;
    float acc = 0.0f;
    for (int it = 0; it < kIters; it++) {
        const uint32_t idx = hash32(gid * kIters + it) & mask;
        for (int k = 0; k < 4; k++) {
            const uint32_t src = (uint32_t)__shfl((int)idx, 16 * k + (int)(lane >> 2));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tab + 4 * (size_t)src + (lane & 3u)),
                                             (__attribute__((address_space(3))) void*)(img + 64 * k), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);                            // vmcnt(0)
        __syncthreads();
        const float4 a = img[4 * lane], b = img[4 * lane + 1], c = img[4 * lane + 2], d = img[4 * lane + 3];
        acc += a.x + b.y + c.z + d.w;
        __syncthreads();
    }
    out[gid] = acc;
}

__global__ void __launch_bounds__(64) k_stream(const float4* __restrict__ tab, uint32_t nvec, float* out) {
    const uint32_t gid = blockIdx.x * 64u + threadIdx.x;
    float acc = 0.0f;
    const uint32_t total = gridDim.x * 64u;
    for (int it = 0; it < 4 * kIters; it++) {
        const uint32_t i = (gid + (uint32_t)it * total) % nvec;
        const float4 a = tab[i];
        acc += a.x + a.w;
    }
    out[gid] = acc;
}

int main(int argc, char** argv) {
    const size_t bytes = 1ull << 30;
    const uint32_t nrec = (uint32_t)(bytes / 64); uint32_t mask = nrec - 1;
    float4* tab; float* out;
    const int waves = 256 * 16 * 8;                                    // 8 waves' worth of work per resident wave slot
    CK(hipMalloc((void**)&tab, bytes)); CK(hipMalloc((void**)&out, (size_t)waves * 64 * 4));
    CK(hipMemset(tab, 0x3c, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double nlane = (double)waves * 64 * kIters;
    struct Run { const char* name; int kind; double bytes; } runs[] = {
        {"lane4   (4 x 16 B per lane, one record per lane)", 0, nlane * 64},
        {"coopreg (4 lanes per record, registers)", 1, nlane * 64},
        {"coopdma (4 lanes per record, LDS-DMA + ds_read)", 2, nlane * 64},
        {"stream  (16 B per lane, consecutive)", 3, nlane * 64},
    };
    // the same shapes on a table that stays in the caches (argv[1] = log2 of the table bytes, default 30): what the load path
    // of a CU sustains when memory is not the limit
    if (argc > 1) { const int lg = atoi(argv[1]); mask = (uint32_t)((1ull << lg) / 64) - 1u; printf("table %llu KiB\n", (1ull << lg) >> 10); }
    for (int rep = 0; rep < 2; rep++)
        for (auto& r : runs) {
            CK(hipEventRecord(e0, 0));
            if (r.kind == 0) hipLaunchKernelGGL(k_lane4, dim3(waves), dim3(64), 0, 0, tab, mask, out);
            if (r.kind == 1) hipLaunchKernelGGL(k_coopreg, dim3(waves), dim3(64), 0, 0, tab, mask, out);
            if (r.kind == 2) hipLaunchKernelGGL(k_coopdma, dim3(waves), dim3(64), 0, 0, tab, mask, out);
            if (r.kind == 3) hipLaunchKernelGGL(k_stream, dim3(waves), dim3(64), 0, 0, tab, (mask + 1u) * 4u, out);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 1) printf("%-55s %8.3f ms  %7.1f GB/s  %7.1f M records/s  (%.3f GB requested)\n", r.name, ms, r.bytes / ms / 1e6, nlane / ms / 1e3, r.bytes / 1e9);
        }
    return 0;
}
