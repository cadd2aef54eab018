// What an LDS bank conflict costs a ds_add_u32 on gfx950: cycles per wave-instruction as a function of the WORST bank load inside a
// 32-lane half (1 = conflict-free ... 8), of the number of active lanes, and for the address patterns the postings walks produce:
//   "walk"      4 records x 8 postings per half, documents uniformly random (the list walk of bp_walk.h),
//   "arranged"  the same with the 8 postings of a record in 8 different banks (bp_arrange_kernel),
//   "rowlist"   one list per wave instruction, its postings dealt so that no bank takes more than 2 lanes of a half (bp_row.h).
// 16 waves per CU (one 1024-thread workgroup per CU), all CUs.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/lds_conflicts.hip -o tools/microbench/bin/lds_conflicts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>

constexpr int kIters = 1024;
constexpr int kSets = 8;      // address sets per lane, cycled

__global__ __launch_bounds__(1024) void bench(const uint32_t* addr, long long* cycles, int active) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* w = reinterpret_cast<uint32_t*>(smem);
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384; i += 1024) w[i] = 0;
    __syncthreads();
    uint32_t a[kSets];
    for (int j = 0; j < kSets; ++j) a[j] = addr[(size_t)j * 1024 + tid];
    const bool on = (tid & 63) < active;
    const long long t0 = clock64();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int j = 0; j < kSets; ++j) {
            // (+ 32 * it keeps the bank, moves the row)
            const uint32_t x = (a[j] + 32u * (uint32_t)it) & 16383u;
            if (on && a[j] != 0xFFFFFFFFu) atomicAdd(&w[x], 1u);
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if (w[tid] == 0x12345u) cycles[0] = 0;
}

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

// a 32-lane half whose worst bank takes exactly `load` lanes: lanes are dealt to banks round-robin over 32 / load banks... (distinct rows)
static void fill_load(std::vector<uint32_t>& v, int load) {
    for (int j = 0; j < kSets; ++j)
        for (int t = 0; t < 1024; ++t) {
            const int l = t & 31;
            const int nb = 32 / load;                     // banks in use
            const int bank = l % nb;                      // each bank in use takes `load` lanes
            const int row = (l / nb) + load * (int)(rnd() % 32);                 // different rows: different addresses
            v[(size_t)j * 1024 + t] = (uint32_t)(row * 32 + bank);
        }
}
static void fill_walk(std::vector<uint32_t>& v, bool arranged, int pitch /*dwords per document*/) {
    for (int j = 0; j < kSets; ++j)
        for (int t0 = 0; t0 < 1024; t0 += 8) {
            uint32_t used = 0;
            const uint32_t slot = rnd() % 8;
            for (int i = 0; i < 8; ++i) {
                uint32_t doc;
                do { doc = rnd() % 1920; } while (arranged && (used & (1u << (doc & 31))));
                used |= 1u << (doc & 31);
                v[(size_t)j * 1024 + t0 + i] = pitch == 1 ? slot * 2048 + doc : doc * pitch + slot;
            }
        }
}
// one list of n postings per wave instruction, dealt greedily: a posting goes to the half whose bank count is lower (ties: fewer lanes)
static double fill_rowlist(std::vector<uint32_t>& v, int n_mean, int* worst_hist) {
    double lanes = 0;
    for (int j = 0; j < kSets; ++j)
        for (int t0 = 0; t0 < 1024; t0 += 64) {
            int n = n_mean + (int)(rnd() % 15) - 7;
            if (n > 64) n = 64;
            int cnt[2][32] = {{0}}, fill[2] = {0, 0};
            uint32_t out[2][32];
            const uint32_t slot = rnd() % 8;
            for (int p = 0; p < n; ++p) {
                const uint32_t doc = rnd() % 1920, bk = doc & 31;
                int h = cnt[0][bk] < cnt[1][bk] ? 0 : cnt[1][bk] < cnt[0][bk] ? 1 : (fill[0] <= fill[1] ? 0 : 1);
                if (fill[h] >= 32) h ^= 1;
                out[h][fill[h]++] = slot * 2048 + doc;
                ++cnt[h][bk];
            }
            for (int h = 0; h < 2; ++h) {
                int wmax = 0;
                for (int b = 0; b < 32; ++b) wmax = std::max(wmax, cnt[h][b]);
                ++worst_hist[std::min(wmax, 7)];
                for (int i = 0; i < 32; ++i) v[(size_t)j * 1024 + t0 + h * 32 + i] = i < fill[h] ? out[h][i] : 0xFFFFFFFFu;
            }
            lanes += n;
        }
    return lanes / (kSets * 16.0);
}

static double run(const char* name, const std::vector<uint32_t>& h, uint32_t* d, long long* dc, int active, double lanes = 64.0) {
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(bench, dim3(256), dim3(1024), 65536, 0, d, dc, active);
    hipDeviceSynchronize();
    std::vector<long long> c(256);
    hipMemcpy(c.data(), dc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto x : c) avg += (double)x;
    avg /= 256;
    const double per = avg / (16.0 * kIters * kSets);
    printf("%-44s %7.2f cycles per wave-instruction per CU   %6.2f adds per cycle and CU\n", name, per, std::min(lanes, (double)active) / per);
    return per;
}

int main() {
    std::vector<uint32_t> h((size_t)kSets * 1024);
    uint32_t* d; long long* dc;
    hipMalloc(&d, h.size() * 4); hipMalloc(&dc, 256 * 8);
    hipFuncSetAttribute((const void*)bench, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    char name[128];
    for (int load : {1, 2, 4, 8}) {
        fill_load(h, load);
        snprintf(name, sizeof name, "worst bank of a 32-lane half = %d", load);
        run(name, h, d, dc, 64);
    }
    // 3-way: banks 0..9 take 3 lanes, banks 10, 11 take 1
    for (int j = 0; j < kSets; ++j) for (int t = 0; t < 1024; ++t) { const int l = t & 31; h[(size_t)j * 1024 + t] = (uint32_t)((l / 11 + 3 * (rnd() % 32)) * 32 + l % 11); }
    run("worst bank of a 32-lane half = 3", h, d, dc, 64);
    fill_load(h, 1);
    for (int act : {56, 48, 32, 16}) { snprintf(name, sizeof name, "conflict-free, %d of 64 lanes active", act); run(name, h, d, dc, act); }
    fill_walk(h, false, 9);  run("walk: 4 x 8 random documents per half, pitch 9", h, d, dc, 64);
    fill_walk(h, true, 9);   run("arranged: records bank-distinct, pitch 9", h, d, dc, 64);
    fill_walk(h, false, 1);  run("walk, slot-major accumulators", h, d, dc, 64);
    fill_walk(h, true, 1);   run("arranged, slot-major accumulators", h, d, dc, 64);
    for (int n : {50, 40, 57}) {
        int hist[8] = {0};
        const double lanes = fill_rowlist(h, n, hist);
        snprintf(name, sizeof name, "rowlist: one ~%d-posting list per instruction", n);
        run(name, h, d, dc, 64, lanes);
        printf("    halves by worst bank: 1:%d 2:%d 3:%d 4:%d 5+:%d   (%.1f postings per instruction)\n", hist[1], hist[2], hist[3], hist[4], hist[5] + hist[6] + hist[7], lanes);
    }
    return 0;
}
