// How many cache-line requests can a CU's L1 keep going to L2?  The yardstick of the walks: every walk of this library ends
// at ~ 0.19 - 0.20 L1 -> L2 requests per clock and CU whatever its instruction mix -- is that the hardware's rate for scattered lines?
//
// One workgroup of W waves per CU (persistent, 256 workgroups).  A wave keeps K loads in flight: every load instruction touches random
// lines of a buffer of `footprint` bytes (one resident in every XCD's 4 MB L2 / in the 256 MB Infinity Cache / in HBM only):
//   mode 0: 64 lanes x dword, every lane its own line                      (64 lines an instruction)
//   mode 1: 16-lane groups x dwordx4 = 4 runs of 256 bytes                  (8 lines an instruction: the quad walk's load)
//   mode 2: 16-lane groups x dword   = 4 runs of 64 bytes                   (4 lines an instruction: the bag-of-token walk's load)
// The loop is inline asm with K register sets and a counted s_waitcnt vmcnt(K-1), as the walks' generated loops are.
// Output: lines per clock and CU, bytes per clock and CU, and -- by Little's law with the measured round trip of a dependent chain --
// the lines in flight per CU.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/l2_requests.hip -o /tmp/l2_requests && /tmp/l2_requests
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// K loads in flight per wave; MODE as above; `iters` trips of K loads each
template <int K, int MODE>
__global__ __launch_bounds__(1024) void requests(const char* buf, uint32_t line_mask, int iters, uint32_t* sink) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    // a lane's line sequence: an LCG per group (mode 1, 2: the 16 lanes of a group share the line run) or per lane (mode 0)
    const uint32_t grp = MODE == 0 ? tid : (tid >> 4);
    uint32_t x = (blockIdx.x * 1024u + grp) * 2654435761u + 12345u;
    const uint32_t sub = MODE == 0 ? 0u : MODE == 1 ? (lane & 15u) * 16u : (lane & 15u) * 4u;
    uint32_t acc = 0;
    uint32_t off[K];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        x = x * 1664525u + 1013904223u;
        off[k] = (((x >> 8) & line_mask) << (MODE == 1 ? 8 : MODE == 2 ? 6 : 7)) + sub;
    }
    // prologue: K loads
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v[k]) : "v"(off[k]), "s"(buf) : "memory");
        else asm volatile("global_load_dword %0, %1, %2" : "=v"(v[k].x) : "v"(off[k]), "s"(buf) : "memory");
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            x = x * 1664525u + 1013904223u;
            const uint32_t o = (((x >> 8) & line_mask) << (MODE == 1 ? 8 : MODE == 2 ? 6 : 7)) + sub;
            // the oldest load has landed when at most K - 1 are outstanding
            if (MODE == 1) {
                asm volatile("s_waitcnt vmcnt(%2)\n\tglobal_load_dwordx4 %0, %1, %3"          // (the landed set is simply loaded again)
                             : "+v"(v[k]) : "v"(o), "n"(K - 1), "s"(buf) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(%3)\n\tv_xor_b32 %0, %0, %1\n\tglobal_load_dword %1, %2, %4"
                             : "+v"(acc), "+v"(v[k].x) : "v"(o), "n"(K - 1), "s"(buf) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < K; ++k) acc ^= v[k].x;
    if (acc == 0x12345678u) sink[0] = acc;
}

// the SCALAR path (scalar cache -> L2): a wave issues K s_load_dwordx16 (64 bytes each) of random 64-byte lines and waits for all of them
// (scalar loads return out of order: lgkmcnt(0)); W waves per CU.  Is it a second road into L2 beside the L1's?
template <int K>
__global__ __launch_bounds__(1024) void scalar_requests(const char* buf, uint32_t line_mask, int iters, uint32_t* sink) {
    const uint32_t wave = (blockIdx.x * 1024u + threadIdx.x) >> 6;
    uint32_t x = (uint32_t)__builtin_amdgcn_readfirstlane(wave * 2654435761u + 999u);
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t o[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { x = x * 1664525u + 1013904223u; o[k] = ((x >> 8) & line_mask) << 6; }
        typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
        u32x16 r[K];
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(r[k]) : "s"(buf), "s"(o[k]) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < K; ++k) acc ^= r[k].x;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// round trip of ONE dependent load chain per wave (latency under no load, and -- launched beside nothing else -- the base line)
__global__ void chase(const uint32_t* buf, int iters, uint32_t* sink, unsigned long long* clocks) {
    uint32_t p = threadIdx.x * 32u;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) p = buf[p];
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (p == 0xFFFFFFFFu) sink[0] = p;
    if (threadIdx.x == 0) clocks[0] = t1 - t0;
}

template <int K, int MODE>
static int run(const char* d_buf, size_t footprint, int waves, int cus, double clk_hz, uint32_t* d_sink, const char* what) {
    const int line_shift = MODE == 1 ? 8 : MODE == 2 ? 6 : 7;
    const uint32_t mask = (uint32_t)((footprint >> line_shift) - 1);
    const int iters = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((requests<K, MODE>), dim3(cus), dim3(waves * 64), 0, 0, d_buf, mask, 50, d_sink);      // warm-up (and fills the caches)
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((requests<K, MODE>), dim3(cus), dim3(waves * 64), 0, 0, d_buf, mask, iters, d_sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = (double)cus * waves * (double)(iters + 1) * K;
    const double lines_per = MODE == 0 ? 64.0 : MODE == 1 ? 8.0 : 4.0;            // 128-byte lines an instruction touches
    const double bytes_per = MODE == 0 ? 64.0 * 4 : MODE == 1 ? 1024.0 : 256.0;   // bytes it delivers
    const double clocks = ms * 1e-3 * clk_hz;
    printf("%-22s K %2d waves %2d: %8.3f ms  %6.3f instr/clk/CU  %6.3f lines/clk/CU  %6.1f useful B/clk/CU  (%.1f TB/s of lines chip-wide)\n", what, K, waves, ms,
           instr / cus / clocks, instr * lines_per / cus / clocks, instr * bytes_per / cus / clocks, instr * lines_per * 128.0 / (ms * 1e-3) / 1e12);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clk_hz = (double)prop.clockRate * 1e3;
    printf("%s: %d CUs, shader clock %.0f MHz\n", prop.gcnArchName, cus, clk_hz / 1e6);
    uint32_t* d_sink;
    CK(hipMalloc(&d_sink, 64));
    for (size_t footprint : {(size_t)2 << 20, (size_t)128 << 20, (size_t)2 << 30}) {
        char* d_buf;
        CK(hipMalloc(&d_buf, footprint + 4096));
        CK(hipMemset(d_buf, 1, footprint + 4096));
        printf("== footprint %zu MB (%s)\n", footprint >> 20, footprint <= ((size_t)4 << 20) ? "in every XCD's L2" : footprint <= ((size_t)256 << 20) ? "Infinity Cache" : "HBM");
        for (int waves : {4, 16}) {
            if (run<1, 1>(d_buf, footprint, waves, cus, clk_hz, d_sink, "16 lanes x dwordx4")) return 1;
            if (run<2, 1>(d_buf, footprint, waves, cus, clk_hz, d_sink, "16 lanes x dwordx4")) return 1;
            if (run<4, 1>(d_buf, footprint, waves, cus, clk_hz, d_sink, "16 lanes x dwordx4")) return 1;
            if (run<8, 1>(d_buf, footprint, waves, cus, clk_hz, d_sink, "16 lanes x dwordx4")) return 1;
        }
        for (int waves : {4, 16}) {
            if (run<2, 2>(d_buf, footprint, waves, cus, clk_hz, d_sink, "16 lanes x dword")) return 1;
            if (run<8, 2>(d_buf, footprint, waves, cus, clk_hz, d_sink, "16 lanes x dword")) return 1;
        }
        for (int waves : {1, 4}) {
            const uint32_t mask = (uint32_t)((footprint >> 6) - 1) & 0xFFFFFFu;
            const int iters = 2000;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL((scalar_requests<4>), dim3(cus), dim3(waves * 64), 0, 0, d_buf, mask, 20, d_sink);
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((scalar_requests<4>), dim3(cus), dim3(waves * 64), 0, 0, d_buf, mask, iters, d_sink);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double n = (double)cus * waves * iters * 4.0, clocks = ms * 1e-3 * clk_hz;
            printf("scalar s_load_dwordx16  K  4 waves %2d: %8.3f ms  %6.3f 64-byte lines/clk/CU  (%.2f TB/s chip-wide)\n", waves, ms, n / cus / clocks, n * 64.0 / (ms * 1e-3) / 1e12);
        }
        if (run<1, 0>(d_buf, footprint, 16, cus, clk_hz, d_sink, "64 lanes x dword")) return 1;
        if (run<4, 0>(d_buf, footprint, 16, cus, clk_hz, d_sink, "64 lanes x dword")) return 1;
        // dependent chain: the round trip
        if (footprint <= ((size_t)128 << 20)) {
            const size_t n = footprint / 4;
            std::vector<uint32_t> h(n, 0);
            // a stride-32-dword (128 B) cycle over the footprint's lines in a scrambled order
            const size_t lines = n / 32;
            std::vector<uint32_t> perm(lines);
            for (size_t i = 0; i < lines; ++i) perm[i] = (uint32_t)i;
            uint64_t s = 88172645463325252ull;
            for (size_t i = lines - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; const size_t j = s % (i + 1); std::swap(perm[i], perm[j]); }
            {
                for (size_t i = 0; i < lines; ++i) h[(size_t)perm[i] * 32] = perm[(i + 1) % lines] * 32;
                CK(hipMemcpy(d_buf, h.data(), n * 4, hipMemcpyHostToDevice));
                unsigned long long* d_clk;
                CK(hipMalloc(&d_clk, 8));
                hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, 0, (const uint32_t*)d_buf, 2000, d_sink, d_clk);
                hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, 0, (const uint32_t*)d_buf, 20000, d_sink, d_clk);
                unsigned long long c = 0;
                CK(hipMemcpy(&c, d_clk, 8, hipMemcpyDeviceToHost));
                printf("dependent chain, one lane: %.0f ns a load = %.0f shader clocks (s_memrealtime at 100 MHz)\n", (double)c * 10.0 / 20000.0, (double)c * 10e-9 / 20000.0 * clk_hz);
                CK(hipFree(d_clk));
            }
        }
        CK(hipFree(d_buf));
    }
    return 0;
}
