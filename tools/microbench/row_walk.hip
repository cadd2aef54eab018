// The inner loop of the row walk (tools/microbench/bp_row_asm.h) on its own: every CU walks `n_desc` posting lists (~50 postings
// each, sorted offsets with gaps -- a tile's 21 % of the columns) of NBLK blocks of 5.9 MB and scatter-adds them into LDS.
// Prints cycles per list and CU, checks workgroup 0's sums against the host.  Mode 1: the workgroup first COPIES its descriptor
// table with vector stores (alternating between two sources), then reads the copy with scalar loads -- the coherence the real
// kernel's plan phase relies on (stores -> fence -> barrier -> s_dcache_inv -> s_load).
//   hipcc -O3 --offload-arch=gfx950 -I vsearch_amd/csrc tools/microbench/row_walk.hip -o tools/microbench/bin/row_walk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "bp_row_asm.h"

using namespace vs;

constexpr int kDocs = 1920, kCols = 29523, kAccDw = (kDocs / 16) * 144;
constexpr size_t kRegion = (size_t)kCols * 52 * 4;       // bytes of one block's postings: a 52-posting stride per column

__global__ __launch_bounds__(1024) void walk(const uint32_t* post, const uint2* tabs, uint2* scratch, int n_chunks, int nblk, int mode, long long* cycles,
                                             int* acc_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* acc = reinterpret_cast<int*>(smem);
    const int tid = threadIdx.x, w = tid >> 6;
    for (int i = tid; i < kAccDw; i += 1024) acc[i] = 0;
    __syncthreads();
    const size_t tab_dw = (size_t)(n_chunks + 16 * (kRowOverRead + 1)) * 8;            // descriptors per table, null chunks included
    const uint2* src0 = tabs + (size_t)(2 * blockIdx.x) * tab_dw;
    const uint2* src1 = tabs + (size_t)(2 * blockIdx.x + 1) * tab_dw;
    uint2* mine = scratch + (size_t)blockIdx.x * tab_dw;
    const uint32_t n_mine = (uint32_t)((n_chunks - w + 15) / 16);
    const long long t0 = clock64();
    for (int b = 0; b < nblk; ++b) {
        const uint2* tab = (b & 1) ? src1 : src0;
        if (mode == 1) {
            for (size_t i = tid; i < tab_dw; i += 1024) mine[i] = tab[i];
            __threadfence();
            __syncthreads();
            asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
            tab = mine;
        }
        const char* base = reinterpret_cast<const char*>(post) + (size_t)b * kRegion;
        row_rsrc_t rs;
        rs.x = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)base);
        rs.y = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)base >> 32)) & 0xFFFFu;
        rs.z = (uint32_t)kRegion;
        rs.w = 0x00020000u;
        row_walk_asm(tab + (size_t)w * 8, n_mine, rs, (uint32_t)(tid & 63) * 4u);
        __syncthreads();
    }
    const long long t1 = clock64();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if (blockIdx.x == 0) for (int i = tid; i < kAccDw; i += 1024) acc_out[i] = acc[i];
}

static uint32_t rs_ = 99991u;
static uint32_t rnd() { rs_ = rs_ * 1664525u + 1013904223u; return rs_ >> 8; }
static uint16_t f2h(float f) {          // fp32 -> fp16 bits, round to nearest even, normal range only
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t e = ((x >> 23) & 0xFF) - 127 + 15, m = x & 0x7FFFFF;
    uint32_t h = (e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1FFF;
    if (rem > 0x1000 || (rem == 0x1000 && (h & 1))) ++h;
    return (uint16_t)h;
}
static float h2f(uint16_t h) { uint32_t x = ((uint32_t)((h >> 10) & 31) - 15 + 127) << 23 | (uint32_t)(h & 1023) << 13; float f; memcpy(&f, &x, 4); return f; }

int main(int argc, char** argv) {
    const int nblk = argc > 1 ? atoi(argv[1]) : 24;
    const int n_list = argc > 2 ? atoi(argv[2]) : 6208;
    const int nwg = 256;
    const int al8 = argc > 4 ? atoi(argv[4]) : 1;          // lists start on multiples of al8 x 8 bytes
    const bool l1 = argc > 3 && !strcmp(argv[3], "l1");      // every list inside the first 13 KB of the block: no L2 traffic
    // postings: one dword each
    std::vector<uint32_t> post((size_t)nblk * kRegion / 4);
    for (auto& p : post) {
        const uint32_t doc = rnd() % kDocs;
        const float v = 0.01f + 3.0f * (float)(rnd() & 0xFFFF) / 65536.f;
        p = ((doc >> 4) * 144 + (doc & 15)) | ((uint32_t)f2h(v) << 16);
    }
    const int n_chunks = (n_list + 7) / 8;
    const size_t tab_dw = (size_t)(n_chunks + 16 * (kRowOverRead + 1)) * 8;
    std::vector<uint2> tabs((size_t)2 * nwg * tab_dw);
    for (int t = 0; t < 2 * nwg; ++t) {
        uint2* T = tabs.data() + (size_t)t * tab_dw;
        // a sorted random subset of the columns
        std::vector<int> cols(kCols);
        for (int i = 0; i < kCols; ++i) cols[i] = i;
        for (int i = 0; i < n_list; ++i) std::swap(cols[i], cols[i + rnd() % (kCols - i)]);
        std::sort(cols.begin(), cols.begin() + n_list);
        for (size_t i = 0; i < tab_dw; ++i) {
            uint32_t off8 = 0, cnt = 1, slot = 0; float wq = 0.f;
            if ((int)i < n_list) {
                off8 = (uint32_t)(l1 ? (int)(i % 64) : cols[i]) * 26u / al8 * al8;                          // 52 postings = 208 bytes = 26 units per column
                cnt = 43 + rnd() % 10;                                   // 43 .. 52
                slot = rnd() & 7;
                wq = (0.01f + 3.0f * (float)(rnd() & 0xFFFF) / 65536.f) * 64.f;
            }
            uint32_t wb; memcpy(&wb, &wq, 4);
            T[i] = make_uint2((off8 << 12) | (slot << 6) | (64 - cnt), wb);
        }
    }
    uint32_t* d_post; uint2 *d_tabs, *d_scr; long long* d_cyc; int* d_acc;
    hipMalloc(&d_post, post.size() * 4); hipMalloc(&d_tabs, tabs.size() * 8); hipMalloc(&d_scr, (size_t)nwg * tab_dw * 8);
    hipMalloc(&d_cyc, nwg * 8); hipMalloc(&d_acc, kAccDw * 4);
    hipMemcpy(d_post, post.data(), post.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_tabs, tabs.data(), tabs.size() * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)walk, hipFuncAttributeMaxDynamicSharedMemorySize, kAccDw * 4);
    // expected sums of workgroup 0
    std::vector<int> want(kAccDw, 0);
    for (int b = 0; b < nblk; ++b) {
        const uint2* T = tabs.data() + (size_t)(b & 1) * tab_dw;
        for (int i = 0; i < n_list; ++i) {
            const uint32_t d0 = T[i].x; float wq; memcpy(&wq, &T[i].y, 4);
            const uint32_t cnt = 64 - (d0 & 63), slot = (d0 >> 6) & 7;
            const uint32_t* p = post.data() + (size_t)b * kRegion / 4 + (size_t)(d0 >> 12) * 2;
            for (uint32_t l = 0; l < cnt; ++l) want[(p[l] & 0xFFFF) + slot * 16] += (int)(wq * h2f((uint16_t)(p[l] >> 16)));
        }
    }
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(walk, dim3(nwg), dim3(1024), kAccDw * 4, 0, d_post, d_tabs, d_scr, n_chunks, nblk, mode, d_cyc, d_acc);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> c(nwg); std::vector<int> got(kAccDw);
            hipMemcpy(c.data(), d_cyc, nwg * 8, hipMemcpyDeviceToHost);
            hipMemcpy(got.data(), d_acc, kAccDw * 4, hipMemcpyDeviceToHost);
            double avg = 0; for (auto x : c) avg += (double)x; avg /= nwg;
            size_t bad = 0; for (int i = 0; i < kAccDw; ++i) bad += got[i] != want[i];
            printf("mode %d: %.3f ms, %.0f cycles per block and CU, %.2f cycles per list and CU (%d lists, %d blocks); sums of workgroup 0: %zu of %d differ\n", mode, ms,
                   avg / nblk, avg / nblk / n_list, n_list, nblk, bad, kAccDw);
        }
    }
    return 0;
}
