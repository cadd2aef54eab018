// A value loaded from memory and held in a VGPR pair across a long inline-asm statement: does it survive when several processes share the
// GPU?  (docs/EXPERIMENTS.md round 5: the bag-of-token walk's next block base, loaded ahead of the walk statement, "came back wrong" in
// 10 - 25 % of the searches with four processes on the GPU, never with one or two.)  This is that shape on its own:
//   persistent 1024-thread workgroups (one per CU: 100 KB of LDS), per iteration
//     next = tab[it + 1]                       a wave-uniform 8-byte global load issued by the compiler's code
//     bq_walk_asm(...)                         THE library's generated walk statement (vsearch_amd/csrc/bp_bq_asm.h: named VGPRs v64 .. v92,
//                                              counted s_waitcnt vmcnt / lgkmcnt, s_and_saveexec inside) over a table of real descriptors
//     check next == f(it + 1)                  mismatches counted per cause
// MODE 0: as above -- the load is IN FLIGHT when the statement starts (the compiler waits for it where the value is first used, behind the
//         statement: check with -S);  1: s_waitcnt vmcnt(0) before the statement (the value sits in its VGPR pair across it);
//         2: waited for and moved to SGPRs (readfirstlane) before the statement;  3: as 0 with the statement's trips = 0 (no loop: the
//         value only crosses the prologue / epilogue of the statement).
// Run as N concurrent processes: tools/microbench/run_vgpr_across_asm.sh.
//   hipcc -O3 --offload-arch=gfx950 -I vsearch_amd/csrc tools/microbench/vgpr_across_asm.hip -o tools/microbench/bin/vgpr_across_asm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include "bp_bq_asm.h"

using namespace vs;

constexpr int kCols = 4096, kPlane = (8192 + 256) * 4, kTrips = 48;
__host__ __device__ inline unsigned long long expect(unsigned long long i) { return i * 0x9E3779B97F4A7C15ull + 0x1234567ull; }

template <int MODE>
__global__ __launch_bounds__(1024) void hold(const char* chunks, const unsigned long long* tab, int iters, unsigned long long* bad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* acc = reinterpret_cast<int32_t*>(smem);                        // two planes from LDS address 0
    uint2* desc = reinterpret_cast<uint2*>(smem + 2 * kPlane);             // 16 waves x 4 lists x (kTrips + over-read) steps
    const uint32_t desc_lds = 2 * kPlane;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n_desc = 64 * (kTrips + kBqOverRead);
    for (int i = tid; i < 2 * kPlane / 4; i += 1024) acc[i] = 0;
    for (int i = tid; i < n_desc; i += 1024) desc[i] = make_uint2((uint32_t)((i * 37 + blockIdx.x * 11) % kCols) | ((uint32_t)((i & 1) * kPlane) << 16), 1u);
    __syncthreads();
    const uint32_t list_lds = desc_lds + (uint32_t)n_desc * 8u + (uint32_t)wv * 1024u;
    const uint32_t g8 = (uint32_t)(lane / kBqGroupLanes) * 8u, l4 = (uint32_t)(lane % kBqGroupLanes) * (uint32_t)(kBqLaneDwords * 4);
    unsigned long long cur = tab[0];
    unsigned long long n_bad = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned long long next = tab[it + 1];                              // the load under test
        if (MODE == 1 || MODE == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(next) :: "memory");
        if (MODE == 2) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)next), hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(next >> 32));
            next = ((unsigned long long)hi << 32) | lo;
        }
        const char* base = chunks + (size_t)((cur >> 40) & 7u) * (size_t)kCols * 64;
        (void)bq_walk_asm(desc_lds + (uint32_t)wv * 32u + g8, MODE == 3 ? 1u : (uint32_t)kTrips, base, l4, (uint32_t)kCols, list_lds, 16u);
        if (next != expect((unsigned long long)it + 1)) ++n_bad;
        cur = next;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (n_bad) atomicAdd(bad, n_bad);
    if (tid == 0 && acc[1] == 0x7FFFFFFF) bad[1] = 1;                       // (keeps the sums alive)
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int iters = argc > 2 ? atoi(argv[2]) : 20000;
    const int launches = argc > 3 ? atoi(argv[3]) : 10;
    const int n_streams = argc > 4 ? atoi(argv[4]) : 1;       // > 1: the launches go round the streams (more queues than the hardware has slots for -> the scheduler time-slices them)
    std::vector<unsigned long long> tab((size_t)iters + 2);
    for (size_t i = 0; i < tab.size(); ++i) tab[i] = expect(i);
    std::vector<uint32_t> chunks((size_t)8 * kCols * 16);
    for (size_t i = 0; i < chunks.size(); ++i) { const uint32_t d0 = (uint32_t)((i * 2) % 8192), d1 = (uint32_t)((i * 2 + 1) % 8192); chunks[i] = d0 | (d1 << 16); }
    char* d_chunks; unsigned long long* d_tab; unsigned long long* d_bad;
    (void)hipMalloc(&d_chunks, chunks.size() * 4); (void)hipMalloc(&d_tab, tab.size() * 8); (void)hipMalloc(&d_bad, 16);
    (void)hipMemcpy(d_chunks, chunks.data(), chunks.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemset(d_bad, 0, 16);
    const size_t lds = 2 * (size_t)kPlane + (size_t)64 * (kTrips + kBqOverRead) * 8 + 16 * 1024;
    void (*k)(const char*, const unsigned long long*, int, unsigned long long*) = mode == 0 ? hold<0> : mode == 1 ? hold<1> : mode == 2 ? hold<2> : hold<3>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    std::vector<hipStream_t> streams((size_t)n_streams);
    for (auto& st : streams) (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(k, dim3(prop.multiProcessorCount), dim3(1024), lds, n_streams > 1 ? streams[(size_t)l % streams.size()] : 0, d_chunks, d_tab, iters, d_bad);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long bad[2] = {0, 0};
    (void)hipMemcpy(bad, d_bad, 16, hipMemcpyDeviceToHost);
    printf("mode %d, %d streams: %d launches x %d iterations x %d workgroups x 16 waves in %.0f ms: %llu wrong values (thread-iterations)\n", mode, n_streams, launches, iters, prop.multiProcessorCount, ms, bad[0]);
    return bad[0] ? 3 : 0;
}
