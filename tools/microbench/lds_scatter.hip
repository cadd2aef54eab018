// LDS scatter-add rates on gfx950, chip-wide, timed with hipEvents (Gadd/s over all 256 CUs): what bounds the
// accumulate step of the blocked-postings walk (csrc/bp_scan.h).  Ops: ds_add_{f64,u64,f32,u32}, ds_add_rtn_u32,
// plain read-modify-write (racy, for reference), with random slot addresses drawn from a table (like document ids).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/lds_scatter.hip -o /tmp/lds_scatter && /tmp/lds_scatter
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int kIters = 1024;
constexpr int kU = 8;

// OP: 0 f64, 1 u64, 2 f32, 3 u32, 4 u32 returning, 5 racy u32 rmw, 6 pk 2 x u32 via one u64 add
template <int OP, int THREADS>
__global__ __launch_bounds__(THREADS) void scatter(const uint32_t* idx, int slots, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* d = reinterpret_cast<double*>(smem);
    float* f = reinterpret_cast<float*>(smem);
    unsigned long long* u = reinterpret_cast<unsigned long long*>(smem);
    uint32_t* w = reinterpret_cast<uint32_t*>(smem);
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384; i += THREADS) u[i] = 0;
    __syncthreads();
    uint32_t a[kU];
    // slots is a power of two
    for (int j = 0; j < kU; ++j) a[j] = (idx[(blockIdx.x * 7919 + tid * kU + j) & 65535] & (uint32_t)(slots - 1));
    uint32_t got = 0;
    for (int it = 0; it < kIters; ++it) {
        const uint32_t itm = ((uint32_t)it * 977u) & (uint32_t)(slots - 1);
#pragma unroll
        for (int j = 0; j < kU; ++j) {
            const uint32_t x = a[j] ^ itm;                      // stays inside the slot range: one VALU per address
            if (OP == 0) atomicAdd(&d[x], 1.0);
            if (OP == 1) atomicAdd(&u[x], 1ull);
            if (OP == 2) atomicAdd(&f[x], 1.0f);
            if (OP == 3) atomicAdd(&w[x], 1u);
            if (OP == 4) got += atomicAdd(&w[x], 1u);
            if (OP == 5) w[x] = w[x] + 1u;
            if (OP == 6) atomicAdd(&u[x], 0x100000001ull);
        }
    }
    __syncthreads();
    if (w[tid] == 0x12345u || got == 0x7777u) sink[0] = got;
}

template <int OP, int THREADS>
void run(const char* name, const uint32_t* didx, uint32_t* sink, int slots, int wg_per_cu) {
    hipFuncSetAttribute((const void*)scatter<OP, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / wg_per_cu);
    const size_t lds = 128 * 1024 / wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu * 4;
    hipLaunchKernelGGL((scatter<OP, THREADS>), dim3(grid), dim3(THREADS), lds, 0, didx, slots, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((scatter<OP, THREADS>), dim3(grid), dim3(THREADS), lds, 0, didx, slots, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double adds = (double)grid * THREADS * kIters * kU;
    printf("%-10s threads %4d wg/cu %d slots %6d : %8.1f Gadd/s  (%.2f lanes/clk/CU at 2.4 GHz)\n", name, THREADS, wg_per_cu, slots,
           adds / ms * 1e-6, adds / (ms * 1e-3) / 256.0 / 2.4e9);
}

int main() {
    std::vector<uint32_t> h(65536);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 8; }
    uint32_t *didx, *sink;
    hipMalloc(&didx, 65536 * 4); hipMalloc(&sink, 64);
    hipMemcpy(didx, h.data(), 65536 * 4, hipMemcpyHostToDevice);
    for (int slots : {8192, 1024}) {
        run<0, 1024>("f64", didx, sink, slots, 1);
        run<1, 1024>("u64", didx, sink, slots, 1);
        run<6, 1024>("u64x2", didx, sink, slots, 1);
        run<2, 1024>("f32", didx, sink, slots, 1);
        run<3, 1024>("u32", didx, sink, slots, 1);
        run<4, 1024>("u32rtn", didx, sink, slots, 1);
        run<5, 1024>("rmw32", didx, sink, slots, 1);
    }
    run<3, 1024>("u32", didx, sink, 32768, 1);
    run<2, 1024>("f32", didx, sink, 32768, 1);
    run<3, 512>("u32", didx, sink, 8192, 1);
    run<3, 512>("u32", didx, sink, 8192, 2);
    run<3, 256>("u32", didx, sink, 8192, 4);
    run<0, 512>("f64", didx, sink, 8192, 2);
    run<0, 256>("f64", didx, sink, 8192, 4);
    return 0;
}
