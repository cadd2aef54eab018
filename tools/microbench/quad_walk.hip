// The inner loop of the quad walk (vsearch_amd/csrc/bp_quad_loop.h) on its own: every CU walks n_list posting lists (one 256-byte
// chunk each, sorted offsets with gaps -- a tile's 21 % of the columns) of NBLK blocks and scatter-adds them into LDS; the
// descriptor table is copied into LDS per block (the real kernel's plan phase writes it there).  Prints cycles per list and CU and
// checks workgroup 0's sums against the host.
//   hipcc -O3 --offload-arch=gfx950 -I vsearch_amd/csrc tools/microbench/quad_walk.hip -o tools/microbench/bin/quad_walk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "bp_quad_loop.h"
#include "bp_quad_asm.h"

using namespace vs;

constexpr int kDocs = 2048, kCols = 29523, kAccDw = (kDocs / 16) * kQuadGroupDw;
constexpr size_t kRegion = (size_t)kCols * kQuadChunkBytes;
#ifndef QD
#define QD 4
#endif

constexpr int kListCap = 48;                      // descriptors of a wave's link list (+ 4 * kQuadOverRead null descriptors behind it)
constexpr int kListBytes = (kListCap + 4 * kQuadOverRead) * 8;
__global__ __launch_bounds__(1024) void walk(const char* post, const uint2* tabs, int n_desc, int nblk, int mode, long long* cycles, int* acc_out, int* dropped) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* acc = reinterpret_cast<int*>(smem);
    uint2* desc = reinterpret_cast<uint2*>(smem + (size_t)kAccDw * 4);
    const uint32_t desc_lds = (uint32_t)((size_t)kAccDw * 4), lists_lds = desc_lds + (uint32_t)(n_desc + 64 * kQuadOverRead) * 8u;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    uint2* listA = reinterpret_cast<uint2*>(smem + lists_lds + (size_t)w * 2 * kListBytes);
    const uint32_t la = lists_lds + (uint32_t)w * 2u * kListBytes, lb = la + kListBytes;
    for (int i = tid; i < kAccDw; i += 1024) acc[i] = 0;
    __syncthreads();
    const long long t0 = clock64();
    for (int b = 0; b < nblk; ++b) {
        const uint2* tab = tabs + (size_t)(2 * blockIdx.x + (b & 1)) * n_desc;
        for (int i = tid; i < n_desc; i += 1024) desc[i] = tab[i];
        for (int i = tid; i < 64 * kQuadOverRead; i += 1024) desc[n_desc + i] = make_uint2(0u, 0u);      // null steps behind the table
        __syncthreads();
        const char* base = post + (size_t)b * kRegion;
        const unsigned long long pb = (unsigned long long)base;
        const char* ub = (const char*)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pb >> 32)) << 32) |
                                       (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pb));
        const uint32_t g8 = (uint32_t)(lane >> 4) * 8u, s16 = (uint32_t)(lane & 15) * 16u;
#ifdef QUAD_CXX
        if (mode == 0) quad_walk<QD, 16>(desc, n_desc / 4, w, lane, ub);
#else
        if (mode == 0) {
            const uint32_t n_link = quad_walk_asm(desc_lds + (uint32_t)w * 32u + g8, (uint32_t)(n_desc / 64), ub, s16, la, (uint32_t)kListCap);
            // the overflow chunks the wave's chunks linked to (the links of THOSE chunks are dropped here: capacity 0)
            const int n_ovf = (int)min(n_link, (uint32_t)kListCap);
            if (n_link > (uint32_t)kListCap && lane == 0) atomicAdd(dropped, (int)n_link - kListCap);
            const int n_pad = (n_ovf + 3) & ~3;
            if (lane < n_pad - n_ovf + 4 * kQuadOverRead) listA[n_ovf + lane] = make_uint2(0u, 0u);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (n_pad > 0) (void)quad_list_asm(la + g8, (uint32_t)(n_pad / 4), ub, s16, lb, 0u);
        }
#endif
        __syncthreads();
    }
    const long long t1 = clock64();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if (blockIdx.x == 0) for (int i = tid; i < kAccDw; i += 1024) acc_out[i] = acc[i];
}

static uint32_t rs_ = 99991u;
static uint32_t rnd() { rs_ = rs_ * 1664525u + 1013904223u; return rs_ >> 8; }
static uint16_t f2h(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t e = ((x >> 23) & 0xFF) - 127 + 15, m = x & 0x7FFFFF;
    uint32_t h = (e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1FFF;
    if (rem > 0x1000 || (rem == 0x1000 && (h & 1))) ++h;
    return (uint16_t)h;
}
static float h2f(uint16_t h) { if (!(h & 0x7FFF)) return 0.f; uint32_t x = ((uint32_t)((h >> 10) & 31) - 15 + 127) << 23 | (uint32_t)(h & 1023) << 13; float f; memcpy(&f, &x, 4); return f; }

int main(int argc, char** argv) {
    const int nblk = argc > 1 ? atoi(argv[1]) : 24;
    const int n_list = argc > 2 ? atoi(argv[2]) : 6630;
    const bool l1 = argc > 3 && !strcmp(argv[3], "l1");
    const bool rndbank = argc > 4 && !strcmp(argv[4], "rnd");      // documents at random: what an un-arranged list looks like
    const int nwg = 256;
    const int link_pct = argc > 5 ? atoi(argv[5]) : 7;       // chunks that continue in an overflow chunk
    std::vector<uint32_t> post((size_t)nblk * kRegion / 4);
    for (size_t c = 0; c < post.size() / 64; ++c) {
        uint32_t* P = post.data() + c * 64;
        for (int j = 0; j < 4; ++j) {
            int banks[32];
            for (int i = 0; i < 32; ++i) banks[i] = i;
            for (int i = 0; i < 16; ++i) std::swap(banks[i], banks[i + rnd() % (32 - i)]);
            for (int l = 0; l < 16; ++l) {
                const uint32_t doc = rndbank ? rnd() % kDocs : (uint32_t)banks[l] + 32u * (rnd() % (kDocs / 32));
                const bool pad = (rnd() % 64) < 14;
                const float v = 0.01f + 3.0f * (float)(rnd() & 0xFFFF) / 65536.f;
                P[l * 4 + j] = quad_acc_index(doc) | ((pad ? 0u : (uint32_t)f2h(v)) << 16);
            }
        }
        if ((int)(rnd() % 100) < link_pct) {
            const uint32_t link = (uint32_t)((c % kCols + 7919) % kCols);        // (a chunk of the same block)
            P[62] = (link >> 14) & 0x3FFFu;
            P[63] = 0x80000000u | (link & 0x3FFFu);
        }
    }
    const int n_desc = (n_list + 63) / 64 * 64;
    std::vector<uint2> tabs((size_t)2 * nwg * n_desc);
    for (int t = 0; t < 2 * nwg; ++t) {
        uint2* T = tabs.data() + (size_t)t * n_desc;
        std::vector<int> cols(kCols);
        for (int i = 0; i < kCols; ++i) cols[i] = i;
        for (int i = 0; i < n_list; ++i) std::swap(cols[i], cols[i + rnd() % (kCols - i)]);
        std::sort(cols.begin(), cols.begin() + n_list);
        for (int i = 0; i < n_desc; ++i) {
            uint32_t chunk = 0, slot = 0; float wq = 0.f;
            if (i < n_list) {
                chunk = (uint32_t)(l1 ? i % 48 : cols[i]);
                slot = rnd() & 7;
                wq = (0.01f + 3.0f * (float)(rnd() & 0xFFFF) / 65536.f) * 64.f;
            }
            uint32_t wb; memcpy(&wb, &wq, 4);
            T[i] = make_uint2(chunk * 256u | slot * 16u, wb);
        }
    }
    char* d_post; uint2* d_tabs; long long* d_cyc; int* d_acc; int* d_drop;
    (void)hipMalloc(&d_drop, 4); (void)hipMemset(d_drop, 0, 4);
    (void)hipMalloc(&d_post, post.size() * 4); (void)hipMalloc(&d_tabs, tabs.size() * 8); (void)hipMalloc(&d_cyc, nwg * 8); (void)hipMalloc(&d_acc, kAccDw * 4);
    (void)hipMemcpy(d_post, post.data(), post.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_tabs, tabs.data(), tabs.size() * 8, hipMemcpyHostToDevice);
    const size_t lds = (size_t)kAccDw * 4 + (size_t)(n_desc + 64 * kQuadOverRead) * 8 + (size_t)16 * 2 * kListBytes;
    (void)hipFuncSetAttribute((const void*)walk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    std::vector<int> want(kAccDw, 0);
    for (int b = 0; b < nblk; ++b) {
        const uint2* T = tabs.data() + (size_t)(b & 1) * n_desc;
        for (int i = 0; i < n_desc; ++i) {
            float wq; memcpy(&wq, &T[i].y, 4);
            const uint32_t slot = (T[i].x & 0xFF) / 16;
            const uint32_t* p = post.data() + (size_t)b * kRegion / 4 + (size_t)(T[i].x >> 8) * 64;
            for (int l = 0; l < 64; ++l) want[(p[l] & 0xFFFF) + slot * 16] += (int)(wq * h2f((uint16_t)(p[l] >> 16)));
            if (p[63] >> 31) {                                              // a link: the overflow chunk's cells count too
                const uint32_t link = (p[63] & 0x3FFFu) | ((p[62] & 0x3FFFu) << 14);
                const uint32_t* o = post.data() + (size_t)b * kRegion / 4 + (size_t)link * 64;
                for (int l = 0; l < 64; ++l) want[(o[l] & 0xFFFF) + slot * 16] += (int)(wq * h2f((uint16_t)(o[l] >> 16)));
            }
        }
    }
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(walk, dim3(nwg), dim3(1024), lds, 0, d_post, d_tabs, n_desc, nblk, mode, d_cyc, d_acc, d_drop);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> c(nwg); std::vector<int> got(kAccDw);
            (void)hipMemcpy(c.data(), d_cyc, nwg * 8, hipMemcpyDeviceToHost);
            (void)hipMemcpy(got.data(), d_acc, kAccDw * 4, hipMemcpyDeviceToHost);
            double avg = 0; for (auto x : c) avg += (double)x; avg /= nwg;
            size_t bad = 0; for (int i = 0; i < kAccDw; ++i) bad += got[i] != want[i];
            int drop = 0; (void)hipMemcpy(&drop, d_drop, 4, hipMemcpyDeviceToHost); (void)hipMemset(d_drop, 0, 4);
            if (drop) printf("(%d links beyond a wave's list capacity dropped)\n", drop);
            printf("%s: %.3f ms, %.0f cycles per block and CU, %.2f cycles per list and CU (%d lists, %d blocks, D = %d); sums of workgroup 0: %zu of %d differ\n",
                   mode == 0 ? "walk" : "table copy only", ms, avg / nblk, avg / nblk / n_list, n_list, nblk, QD, bad, kAccDw);
        }
    return 0;
}
