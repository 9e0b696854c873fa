// zy_litmus.hip -- message-passing litmus for the hand-offs the library ships by default.
//
// The one-launch YZ stage (csrc/dfft_zy.hip) hands a unit's results from one workgroup to another INSIDE a launch with
//     producer : 16-byte `sc1` buffer stores -> s_waitcnt vmcnt(0) -> workgroup barrier -> relaxed agent-scope increment of a counter
//     consumer : relaxed agent-scope poll of the counter -> workgroup barrier -> 16-byte `sc1` buffer loads
// and no fences -- the pattern MI355X_MICROARCH.md gives for inter-workgroup hand-offs, not LLVM's agent-scope release / acquire
// (buffer_wbl2 / buffer_inv sc1).  969+ parity tests say it works; this program is what fails FIRST if a ROCm update changes what a
// completed sc1 store means.  Producer and consumer of a pair sit on DIFFERENT XCDs (different L2s; checked with HW_REG_XCC_ID), every
// hand-off is followed by one in the opposite direction (ping-pong, so no slot is overwritten before it has been read), delays are
// randomised, and every 16-byte element read is compared with the sequence number it must carry.
//
//   mode 0  the shipped pattern                                                    -> expected: 0 stale reads
//   mode 1  negative control: the consumer reads with PLAIN loads                  -> expected: stale reads (its L2 still holds the line)
//   mode 2  negative control: the producer writes with PLAIN stores                -> expected: stale reads (the data is still in its L2)
//   mode 3  the part counter of the overlapped pipeline (round 6, SIG launches): a persistent producer writes with `sc1 nt` stores and
//           counts; on ANOTHER stream a one-wave kernel waits for the count and a kernel launched behind it reads with PLAIN loads,
//           as an exchange would (kernel-boundary acquire)                          -> expected: 0 stale reads
//
// usage: zy_litmus <iterations> <mode> [payload KiB per hand-off, default 16]      exit 0 = the expectation held
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/zy_litmus.hip -o tools/bin/zy_litmus
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(stmt)                                                                                 \
    do {                                                                                            \
        hipError_t e_ = (stmt);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            fprintf(stderr, "[%s:%d] %s: %s\n", __FILE__, __LINE__, #stmt, hipGetErrorString(e_)); \
            return 2;                                                                               \
        }                                                                                           \
    } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define AGENT __HIP_MEMORY_SCOPE_AGENT
constexpr int THREADS = 512;

struct Stats {
    unsigned long long handoffs, stale, same_xcd_pairs, timeouts;
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
template <int AUX> __device__ __forceinline__ void store16(void* base, unsigned bytes, unsigned off, u32x4 v, bool plain) {
    if (plain) {
        *reinterpret_cast<u32x4*>((char*)base + off) = v;
    } else {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, AUX);
    }
}
__device__ __forceinline__ u32x4 load16(void* base, unsigned bytes, unsigned off, bool plain) {
    if (plain) return *reinterpret_cast<const u32x4*>((const char*)base + off);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16 /* sc1 */));
}

// One hand-off, producer side: payload of `elems` 16-byte elements carrying (seq, pair, index), then the publish sequence of dfft_zy.hip.
__device__ void produce(void* buf, unsigned elems, unsigned seq, unsigned pair, unsigned* flag, bool plain_store) {
    for (unsigned i = threadIdx.x; i < elems; i += THREADS) store16<16>(buf, elems * 16u, i * 16u, u32x4{seq, pair, i, ~seq}, plain_store);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: the results have left this CU   (dfft_zy.hip: publish)
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, AGENT);
}
// ... consumer side: poll as dfft_zy.hip's ready() does (thread 0, relaxed agent loads, s_sleep), broadcast through LDS, then the loads.
__device__ unsigned consume(void* buf, unsigned elems, unsigned seq, unsigned pair, unsigned* flag, unsigned want, bool plain_load, unsigned* shw,
                            unsigned* timeouts) {
    if (threadIdx.x == 0) {
        unsigned ok = 0;
        for (unsigned polls = 0; polls < (64u << 20); ++polls) {
            if (__hip_atomic_load(flag, __ATOMIC_RELAXED, AGENT) >= want) {
                ok = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) atomicAdd(timeouts, 1u);
        shw[0] = ok;
    }
    __syncthreads();
    const unsigned ok = shw[0];
    __syncthreads();
    if (!ok) return 0;
    unsigned bad = 0;
    for (unsigned i = threadIdx.x; i < elems; i += THREADS) {
        const u32x4 v = load16(buf, elems * 16u, i * 16u, plain_load);
        if (v.x != seq || v.y != pair || v.z != i || v.w != ~seq) ++bad;
    }
    return bad;
}

// modes 0-2: workgroups 2p (A) and 2p + 1 (B) form pair p; iteration t: A -> B with sequence 2t + 1, then B -> A with 2t + 2.
__global__ void __launch_bounds__(THREADS) pingpong_kernel(char* bufs, unsigned* flags, unsigned* xcds, Stats* st, unsigned iters, unsigned elems, int mode) {
    __shared__ unsigned shw[2];
    const unsigned pair = blockIdx.x >> 1, side = blockIdx.x & 1;
    char*          ab = bufs + (size_t)pair * 2 * elems * 16, *ba = ab + (size_t)elems * 16;
    unsigned *     fab = flags + pair * 64, *fba = fab + 32;  // (two cache lines per pair)
    if (threadIdx.x == 0) xcds[blockIdx.x] = xcc_id();
    const bool plain_load = mode == 1, plain_store = mode == 2;
    unsigned   bad = 0, rnd = 12345u + 977u * blockIdx.x;
    unsigned   touts = 0;
    for (unsigned t = 0; t < iters; ++t) {
        rnd = rnd * 1664525u + 1013904223u;
        if ((rnd >> 28) == 0) __builtin_amdgcn_s_sleep(64);  // now and then one side is late
        else if ((rnd >> 27) == 1) __builtin_amdgcn_s_sleep(8);
        if (side == 0) {
            produce(ab, elems, 2 * t + 1, pair, fab, plain_store);
            bad += consume(ba, elems, 2 * t + 2, pair, fba, t + 1, plain_load, shw, &touts);
        } else {
            bad += consume(ab, elems, 2 * t + 1, pair, fab, t + 1, plain_load, shw, &touts);
            produce(ba, elems, 2 * t + 2, pair, fba, plain_store);
        }
    }
    // reduce the stale-read count of the workgroup
    __shared__ unsigned total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    atomicAdd(&total, bad);
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&st->stale, (unsigned long long)total);
        atomicAdd(&st->handoffs, (unsigned long long)iters);
        atomicAdd(&st->timeouts, (unsigned long long)touts);
    }
}

// mode 3: persistent producer (one workgroup per pair), `sc1 nt` stores like store_cols of a SIG launch, counts, waits for the reader's ack
__global__ void __launch_bounds__(THREADS) part_producer_kernel(char* bufs, unsigned* counters, unsigned* acks, unsigned iters, unsigned elems) {
    __shared__ unsigned shw[1];
    char*               buf = bufs + (size_t)blockIdx.x * elems * 16;
    for (unsigned t = 0; t < iters; ++t) {
        for (unsigned i = threadIdx.x; i < elems; i += THREADS) store16<18>(buf, elems * 16u, i * 16u, u32x4{t + 1, blockIdx.x, i, ~(t + 1)}, false);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&counters[blockIdx.x * 32], 1u, __ATOMIC_RELAXED, AGENT);
            unsigned ok = 0;
            for (unsigned polls = 0; polls < (64u << 20); ++polls) {  // the payload is overwritten only after it has been read
                if (__hip_atomic_load(&acks[blockIdx.x * 32], __ATOMIC_RELAXED, AGENT) >= t + 1) {
                    ok = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            shw[0] = ok;
        }
        __syncthreads();
        if (!shw[0]) return;
        __syncthreads();
    }
}
__global__ void part_wait_kernel(const unsigned* counters, unsigned npairs, unsigned target) {  // (zy_part_wait_kernel)
    if (threadIdx.x >= npairs) return;
    for (unsigned polls = 0; polls < (64u << 20); ++polls) {
        if ((int)(__hip_atomic_load(&counters[threadIdx.x * 32], __ATOMIC_RELAXED, AGENT) - target) >= 0) return;
        __builtin_amdgcn_s_sleep(8);
    }
}
// launched BEHIND the wait kernel: plain loads, every workgroup reads every pair's payload (so every XCD's L2 sees every line, and a line
// left over from the previous iteration would be found stale), then the last workgroup to finish acknowledges
__global__ void __launch_bounds__(THREADS) part_reader_kernel(const char* bufs, unsigned* acks, unsigned* done, Stats* st, unsigned npairs, unsigned elems, unsigned seq) {
    unsigned bad = 0;
    for (unsigned p = 0; p < npairs; ++p) {
        const u32x4* b = reinterpret_cast<const u32x4*>(bufs + (size_t)p * elems * 16);
        for (unsigned i = threadIdx.x; i < elems; i += THREADS) {
            const u32x4 v = b[i];
            if (v.x != seq || v.y != p || v.z != i || v.w != ~seq) ++bad;
        }
    }
    __shared__ unsigned total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    atomicAdd(&total, bad);
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&st->stale, (unsigned long long)total);
        atomicAdd(&st->handoffs, (unsigned long long)npairs);
        if (__hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, AGENT) == gridDim.x * seq - 1)  // last workgroup of this launch
            for (unsigned p = 0; p < npairs; ++p) __hip_atomic_store(&acks[p * 32], seq, __ATOMIC_RELAXED, AGENT);
    }
}

int main(int argc, char** argv) {
    const unsigned iters = argc > 1 ? (unsigned)atoll(argv[1]) : 4000u;
    const int      mode = argc > 2 ? atoi(argv[2]) : 0;
    const unsigned kib = argc > 3 ? (unsigned)atoi(argv[3]) : 16u, elems = kib * 64u;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const unsigned cus = (unsigned)prop.multiProcessorCount, npairs = cus / 2;
    char*          bufs = nullptr;
    unsigned *     flags = nullptr, *xcds = nullptr;
    Stats*         st = nullptr;
    CHECK(hipMalloc((void**)&bufs, (size_t)npairs * 2 * elems * 16));
    CHECK(hipMalloc((void**)&flags, (size_t)npairs * 64 * sizeof(unsigned) + 1024));
    CHECK(hipMalloc((void**)&xcds, cus * sizeof(unsigned)));
    CHECK(hipMalloc((void**)&st, sizeof(Stats)));
    CHECK(hipMemset(bufs, 0, (size_t)npairs * 2 * elems * 16));
    CHECK(hipMemset(flags, 0, (size_t)npairs * 64 * sizeof(unsigned) + 1024));
    CHECK(hipMemset(st, 0, sizeof(Stats)));
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    Stats                 h{};
    std::vector<unsigned> hx(cus, 0);
    unsigned long long    cross = 0;
    if (mode <= 2) {
        // one workgroup per CU: neighbours 2p, 2p + 1 are dealt to different XCDs (workgroup b runs on XCD b mod 8)
        CHECK(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(pingpong_kernel, dim3(2 * npairs), dim3(THREADS), 0, nullptr, bufs, flags, xcds, st, iters, elems, mode);
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(e1, nullptr));
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(hx.data(), xcds, cus * sizeof(unsigned), hipMemcpyDeviceToHost));
        for (unsigned p = 0; p < npairs; ++p) cross += hx[2 * p] != hx[2 * p + 1];
    } else {
        hipStream_t sa, sb;
        CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
        CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
        const unsigned np = 32;  // producers (persistent), readers: 16 workgroups per launch
        unsigned *     counters = flags, *acks = flags + np * 32, *done = flags + 2 * np * 32;
        CHECK(hipEventRecord(e0, sb));
        hipLaunchKernelGGL(part_producer_kernel, dim3(np), dim3(THREADS), 0, sa, bufs, counters, acks, iters, elems);
        CHECK(hipGetLastError());
        for (unsigned t = 0; t < iters; ++t) {
            hipLaunchKernelGGL(part_wait_kernel, dim3(1), dim3(64), 0, sb, counters, np, t + 1);
            hipLaunchKernelGGL(part_reader_kernel, dim3(16), dim3(THREADS), 0, sb, bufs, acks, done, st, np, elems, t + 1);
        }
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(e1, sb));
        CHECK(hipDeviceSynchronize());
        cross = np;
    }
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(&h, st, sizeof(Stats), hipMemcpyDeviceToHost));
    const bool expect_clean = mode == 0 || mode == 3;
    const bool held = h.timeouts == 0 && (expect_clean ? h.stale == 0 : h.stale > 0) && h.handoffs > 0;
    printf("zy_litmus mode %d (%s): %llu hand-offs of %u KiB, %llu stale 16-byte reads, %llu time-outs, %llu of %u pairs across XCDs, %.1f ms -- %s\n", mode,
           mode == 0 ? "shipped pattern: sc1 stores, vmcnt(0), relaxed agent increment | relaxed poll, sc1 loads"
           : mode == 1 ? "negative control: plain loads on the consumer side"
           : mode == 2 ? "negative control: plain stores on the producer side"
                       : "part counter: sc1 nt stores + count | wait kernel, reader kernel with plain loads on another stream",
           h.handoffs, kib, h.stale, h.timeouts, cross, mode <= 2 ? npairs : 32u, ms,
           held ? (expect_clean ? "PASS (no stale read)" : "PASS (the control fails as it must)") : "FAIL");
    return held ? 0 : 1;
}
