// LDS atomic throughput on gfx950: cycles per wave-instruction for ds_add_{f64,u64,f32,u32} and plain stores, with
// conflict-free, random and same-address patterns, 16 waves per CU (one 1024-thread workgroup per CU).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/lds_atomics.hip -o tools/microbench/lds_atomics && ./lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int kIters = 2048;

template <int OP>   // 0 f64 add, 1 u64 add, 2 f32 add, 3 u32 add, 4 store b64, 5 store b32
__global__ __launch_bounds__(1024) void bench(const uint32_t* idx, long long* cycles, int stride_mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* d = reinterpret_cast<double*>(smem);
    float* f = reinterpret_cast<float*>(smem);
    unsigned long long* u = reinterpret_cast<unsigned long long*>(smem);
    uint32_t* w = reinterpret_cast<uint32_t*>(smem);
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 1024) d[i] = 0.0;
    __syncthreads();
    uint32_t a[8];
    for (int j = 0; j < 8; ++j) a[j] = stride_mode == 0 ? (uint32_t)((tid & 63) + 64 * j)            // conflict-free
                                     : stride_mode == 1 ? idx[(tid * 8 + j) & 65535] & 8191u        // random over 8192 slots
                                                        : (uint32_t)j;                               // all lanes same address
    const long long t0 = clock64();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t x = (a[j] + it) & 8191u;
            if (OP == 0) atomicAdd(&d[x], 1.0);
            if (OP == 1) atomicAdd(&u[x], 1ull);
            if (OP == 2) atomicAdd(&f[x], 1.0f);
            if (OP == 3) atomicAdd(&w[x], 1u);
            if (OP == 4) d[x] = (double)it;
            if (OP == 5) f[x] = (float)it;
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if (d[tid] == 12345.0) cycles[0] = 0;
}

template <int OP>
void run(const char* name, const uint32_t* didx, long long* dc) {
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(bench<OP>, dim3(256), dim3(1024), 65536, 0, didx, dc, mode);
        hipDeviceSynchronize();
        std::vector<long long> h(256);
        hipMemcpy(h.data(), dc, 256 * 8, hipMemcpyDeviceToHost);
        double avg = 0;
        for (auto v : h) avg += (double)v;
        avg /= 256;
        // 16 waves x kIters x 8 wave-instructions per CU
        const double per_inst = avg / (16.0 * kIters * 8.0);
        printf("%-12s %-14s %8.2f clk64-ticks per wave-instruction per CU  (%.2f lanes/tick)\n", name,
               mode == 0 ? "conflict-free" : mode == 1 ? "random" : "same-address", per_inst, 64.0 / per_inst);
    }
}

int main() {
    std::vector<uint32_t> h(65536);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 8; }
    uint32_t* didx; long long* dc;
    hipMalloc(&didx, 65536 * 4); hipMalloc(&dc, 256 * 8);
    hipMemcpy(didx, h.data(), 65536 * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)bench<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    run<0>("ds_add_f64", didx, dc); run<1>("ds_add_u64", didx, dc); run<2>("ds_add_f32", didx, dc); run<3>("ds_add_u32", didx, dc);
    run<4>("ds_write_b64", didx, dc); run<5>("ds_write_b32", didx, dc);
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    printf("wall clock rate (clock64 ticks) = %d kHz\n", rate);
    return 0;
}
