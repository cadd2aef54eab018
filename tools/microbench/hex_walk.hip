// The inner loop of the 16-slot walk (vsearch_amd/csrc/bp_hex_loop.h, tools/gen_hex_asm.py) on its own, next to tools/microbench/quad_walk.hip:
// every CU walks n_list posting lists (one 128-byte chunk each; 16 queries' worth, slot by slot, columns sorted inside a slot) of NBLK
// blocks and scatter-adds them into LDS.  Prints cycles per block and CU and per 256 cells (= one quad-walk step of 4 chunks), and checks
// workgroup 0's sums against the host.
//   hipcc -O3 --offload-arch=gfx950 -I vsearch_amd/csrc tools/microbench/hex_walk.hip -o tools/microbench/bin/hex_walk
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "bp_hex_loop.h"
#include "bp_hex_asm.h"

using namespace vs;

constexpr int kDocs = 1024, kCols = 29523, kAccDw = (kDocs / 16) * kHexGroupDw;
constexpr size_t kRegion = (size_t)kCols * kHexChunkBytes;
constexpr int kListCap = 96;
constexpr int kListBytes = (kListCap + 4 * kHexOverRead) * 8;

__global__ __launch_bounds__(1024) void walk(const char* post, const uint32_t* tabs, const uint32_t* bnds, int n_desc, int nblk, int mode, float hmul, long long* cycles, int* acc_out, int* dropped) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* acc = reinterpret_cast<int*>(smem);
    uint32_t* desc = reinterpret_cast<uint32_t*>(smem + (size_t)kAccDw * 4);
    const uint32_t desc_lds = (uint32_t)((size_t)kAccDw * 4), lists_lds = desc_lds + (uint32_t)(n_desc + 64 * kHexOverRead) * 4u;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    uint2* listA = reinterpret_cast<uint2*>(smem + lists_lds + (size_t)w * 2 * kListBytes);
    const uint32_t la = lists_lds + (uint32_t)w * 2u * kListBytes, lb = la + kListBytes;
    for (int i = tid; i < kAccDw; i += 1024) acc[i] = 0;
    __syncthreads();
    const long long t0 = clock64();
    for (int b = 0; b < nblk; ++b) {
        const uint32_t* tab = tabs + (size_t)(2 * blockIdx.x + (b & 1)) * n_desc;
        const uint32_t vbnd = lane < 15 ? bnds[(size_t)(2 * blockIdx.x + (b & 1)) * 16 + lane] : 0xFFFFFFFFu;
        for (int i = tid; i < n_desc; i += 1024) desc[i] = tab[i];
        for (int i = tid; i < 64 * kHexOverRead; i += 1024) desc[n_desc + i] = 0u;      // null steps behind the table
        __syncthreads();
        const char* base = post + (size_t)b * kRegion;
        const uint32_t g4 = (uint32_t)(lane >> 4) * 4u, s8 = (uint32_t)(lane & 15) * 8u;
        if (mode == 0) {
            const uint32_t n_link = hex_walk_asm(desc_lds + (uint32_t)w * 16u + g4, (uint32_t)(n_desc / 64), (uint32_t)w, vbnd, hmul, base, s8, la, (uint32_t)kListCap);
            const int n_ovf = (int)min(n_link, (uint32_t)kListCap);
            if (n_link > (uint32_t)kListCap && lane == 0) atomicAdd(dropped, (int)n_link - kListCap);
            const int n_pad = (n_ovf + 3) & ~3;
            if (lane < n_pad - n_ovf + 4 * kHexOverRead) listA[n_ovf + lane] = make_uint2(0u, 0u);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (n_pad > 0) (void)hex_list_asm(la + g4 * 2u, (uint32_t)(n_pad / 4), base, s8, lb, 0u);
        }
        __syncthreads();
    }
    const long long t1 = clock64();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if (blockIdx.x == 0) for (int i = tid; i < kAccDw; i += 1024) acc_out[i] = acc[i];
}

static uint32_t rs_ = 99991u;
static uint32_t rnd() { rs_ = rs_ * 1664525u + 1013904223u; return rs_ >> 8; }
static uint16_t f2h(float f) { return __half_as_ushort(__float2half_rn(f)); }
static float h2f(uint16_t h) { return __half2float(__ushort_as_half(h)); }

int main(int argc, char** argv) {
    const int nblk = argc > 1 ? atoi(argv[1]) : 24;
    const int n_per = argc > 2 ? atoi(argv[2]) : 776;           // entries per query slot
    const bool l1 = argc > 3 && !strcmp(argv[3], "l1");
    const bool rndbank = argc > 4 && !strcmp(argv[4], "rnd");
    const int link_pct = argc > 5 ? atoi(argv[5]) : 7;
    const int nwg = 256;
    const float hmul = 4.f;                                      // 2^(16 - ve) with ve = 14
    std::vector<uint32_t> post((size_t)nblk * kRegion / 4);
    for (size_t c = 0; c < post.size() / 32; ++c) {
        uint32_t* P = post.data() + c * 32;
        for (int j = 0; j < 2; ++j) {
            int banks[32];
            for (int i = 0; i < 32; ++i) banks[i] = i;
            for (int i = 0; i < 16; ++i) std::swap(banks[i], banks[i + rnd() % (32 - i)]);
            for (int l = 0; l < 16; ++l) {
                const uint32_t doc = rndbank ? rnd() % kDocs : (uint32_t)banks[l] + 32u * (rnd() % (kDocs / 32));
                const bool pad = (rnd() % 64) < 12;
                const float v = 0.01f + 3.0f * (float)(rnd() & 0xFFFF) / 65536.f;
                P[l * 2 + j] = hex_acc_index(doc) | ((pad ? 0u : (uint32_t)f2h(v)) << 16);
            }
        }
        if ((int)(rnd() % 100) < link_pct) {
            const uint32_t link = (uint32_t)((c % kCols + 7919) % kCols);        // (a chunk of the same block)
            P[30] = (link >> 14) & 0x3FFFu;
            P[31] = 0x80000000u | (link & 0x3FFFu);
        }
    }
    const int per_pad = (n_per + 3) & ~3;
    const int n_list = 16 * per_pad;
    const int n_desc = (n_list + 63) / 64 * 64;
    std::vector<uint32_t> tabs((size_t)2 * nwg * n_desc, 0u), bnds((size_t)2 * nwg * 16, 0xFFFFFFFFu);
    for (int t = 0; t < 2 * nwg; ++t) {
        uint32_t* T = tabs.data() + (size_t)t * n_desc;
        for (int s = 0; s < 16; ++s) {
            std::vector<int> cols(kCols);
            for (int i = 0; i < kCols; ++i) cols[i] = i;
            for (int i = 0; i < n_per; ++i) std::swap(cols[i], cols[i + rnd() % (kCols - i)]);
            std::sort(cols.begin(), cols.begin() + n_per);
            for (int i = 0; i < per_pad; ++i) {
                uint32_t dsc = 0u;
                if (i < n_per) {
                    const float wq = (0.01f + 3.0f * (float)(rnd() & 0xFFFF) / 65536.f) * 16.f;
                    dsc = (uint32_t)(l1 ? (s * per_pad + i) % 48 : cols[i]) | ((uint32_t)f2h(wq) << 16);
                }
                T[s * per_pad + i] = dsc;
            }
            if (s < 15) bnds[(size_t)t * 16 + s] = (uint32_t)((s + 1) * per_pad / 4);
        }
    }
    char* d_post; uint32_t* d_tabs; uint32_t* d_bnds; long long* d_cyc; int* d_acc; int* d_drop;
    (void)hipMalloc(&d_drop, 4); (void)hipMemset(d_drop, 0, 4);
    (void)hipMalloc(&d_post, post.size() * 4); (void)hipMalloc(&d_tabs, tabs.size() * 4); (void)hipMalloc(&d_bnds, bnds.size() * 4); (void)hipMalloc(&d_cyc, nwg * 8); (void)hipMalloc(&d_acc, kAccDw * 4);
    (void)hipMemcpy(d_post, post.data(), post.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_tabs, tabs.data(), tabs.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_bnds, bnds.data(), bnds.size() * 4, hipMemcpyHostToDevice);
    const size_t lds = (size_t)kAccDw * 4 + (size_t)(n_desc + 64 * kHexOverRead) * 4 + (size_t)16 * 2 * kListBytes;
    (void)hipFuncSetAttribute((const void*)walk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    std::vector<int> want(kAccDw, 0);
    for (int b = 0; b < nblk; ++b) {
        const uint32_t* T = tabs.data() + (size_t)(b & 1) * n_desc;
        for (int i = 0; i < n_list; ++i) {
            const float wq = h2f((uint16_t)(T[i] >> 16)) * hmul;
            const uint32_t slot = (uint32_t)(i / per_pad);
            const uint32_t* p = post.data() + (size_t)b * kRegion / 4 + (size_t)(T[i] & 0xFFFFu) * 32;
            for (int l = 0; l < 32; ++l) want[(p[l] & 0xFFFF) + slot * 16] += (int)(wq * h2f((uint16_t)(p[l] >> 16)));
            if (p[31] >> 31) {
                const uint32_t link = (p[31] & 0x3FFFu) | ((p[30] & 0x3FFFu) << 14);
                const uint32_t* o = post.data() + (size_t)b * kRegion / 4 + (size_t)link * 32;
                for (int l = 0; l < 32; ++l) want[(o[l] & 0xFFFF) + slot * 16] += (int)(wq * h2f((uint16_t)(o[l] >> 16)));
            }
        }
    }
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(walk, dim3(nwg), dim3(1024), lds, 0, d_post, d_tabs, d_bnds, n_desc, nblk, mode, hmul, d_cyc, d_acc, d_drop);
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
            printf("%s: %.3f ms, %.0f cycles per block and CU, %.2f cycles per 256 cells and CU (%d lists, %d blocks, S = %d); sums of workgroup 0: %zu of %d differ\n",
                   mode == 0 ? "walk" : "table copy only", ms, avg / nblk, avg / nblk / n_list * 8.0, n_list, nblk, kHexSets, bad, kAccDw);
        }
    return 0;
}
