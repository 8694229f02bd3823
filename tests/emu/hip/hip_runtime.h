// tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny host-side lock-step emulation of the HIP device constructs that opensot_amd/csrc/osot_*.h use,
// so the kernel bodies can be executed (and debugged) on a machine without a GPU.  Each lane of a
// 64-lane workgroup is a ucontext fiber on ONE OS thread; every cross-lane collective (__shfl*) and
// every team_sync() is a rendezvous of the participating lane group, so a missing synchronisation shows
// up as a wrong result here even where the real hardware's lock-step execution would hide it.
// It is NOT a compatibility layer of the product: nothing under opensot_amd/ includes it.
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <functional>
#include <cstdint>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

namespace emu {
struct Dim3 { unsigned x, y, z; };
struct Lane {
    ucontext_t ctx;
    std::vector<char> stack;
    Dim3 tid;
    bool done = false;
    unsigned long long seq = 0;        // number of collectives entered
    unsigned char slot[2][16];
};
struct State {
    ucontext_t sched;
    std::vector<Lane> lanes;
    int cur = 0;
    Dim3 bid{0, 0, 0}, bdim{64, 1, 1}, gdim{1, 1, 1};
    std::vector<char> dyn_smem;
    int team_width = 64;
    std::function<void()> body;
};
inline State& S() { static State s; return s; }
inline void yield() { State& s = S(); swapcontext(&s.lanes[s.cur].ctx, &s.sched); }

// rendezvous of the `width`-lane group of the calling lane, exchanging 8..16 bytes
inline void exchange(const void* in, void* out, size_t bytes, int src_abs_lane, int width) {
    State& s = S();
    const int me = s.cur;
    Lane& L = s.lanes[me];
    const unsigned long long q = ++L.seq;
    memcpy(L.slot[q & 1], in, bytes);
    const int g0 = (me / width) * width;
    for (;;) {
        bool all = true;
        for (int i = g0; i < g0 + width; ++i)
            if (s.lanes[i].seq < q) {
                if (s.lanes[i].done) { fprintf(stderr, "emu: lane %d exited before a collective of its group\n", i); abort(); }
                all = false; break;
            }
        if (all) break;
        yield();
    }
    if (out) memcpy(out, s.lanes[src_abs_lane].slot[q & 1], bytes);
}
// rendezvous of the whole 64-lane wave; every lane receives all 64 deposited values
inline void allgather(const void* in, void* out, size_t bytes) {
    State& s = S();
    const int me = s.cur;
    Lane& L = s.lanes[me];
    const unsigned long long q = ++L.seq;
    memcpy(L.slot[q & 1], in, bytes);
    for (;;) {
        bool all = true;
        for (int i = 0; i < 64; ++i)
            if (s.lanes[i].seq < q) {
                if (s.lanes[i].done) { fprintf(stderr, "emu: lane %d exited before a wave collective\n", i); abort(); }
                all = false; break;
            }
        if (all) break;
        yield();
    }
    for (int i = 0; i < 64; ++i) memcpy((char*)out + i * bytes, s.lanes[i].slot[q & 1], bytes);
}
inline void trampoline() {
    State& s = S();
    s.body();
    s.lanes[s.cur].done = true;
    swapcontext(&s.lanes[s.cur].ctx, &s.sched);
}
// run one workgroup of 64 lanes
inline void run_block(unsigned block, unsigned grid, size_t dyn_bytes, int team_width, std::function<void()> body) {
    State& s = S();
    s.lanes.clear();
    s.lanes.resize(64);
    s.bid = {block, 0, 0};
    s.gdim = {grid, 1, 1};
    s.dyn_smem.assign(dyn_bytes + 64, 0);
    s.team_width = team_width;
    s.body = body;
    for (int i = 0; i < 64; ++i) {
        Lane& L = s.lanes[i];
        L.stack.resize(512 * 1024);
        L.tid = {(unsigned)i, 0, 0};
        getcontext(&L.ctx);
        L.ctx.uc_stack.ss_sp = L.stack.data();
        L.ctx.uc_stack.ss_size = L.stack.size();
        L.ctx.uc_link = &s.sched;
        makecontext(&L.ctx, (void (*)())trampoline, 0);
    }
    for (;;) {
        bool any = false;
        for (int i = 0; i < 64; ++i) {
            if (s.lanes[i].done) continue;
            any = true;
            s.cur = i;
            swapcontext(&s.sched, &s.lanes[i].ctx);
        }
        if (!any) break;
    }
}
template <class K, class... Args>
inline void launch(K kernel, unsigned grid, size_t dyn_bytes, int team_width, Args... args) {
    for (unsigned b = 0; b < grid; ++b) run_block(b, grid, dyn_bytes, team_width, [&]() { kernel(args...); });
}
inline char* dyn_smem_ptr() {
    char* p = S().dyn_smem.data();
    return (char*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
}
}  // namespace emu

#define threadIdx (emu::S().lanes[emu::S().cur].tid)
#define blockIdx (emu::S().bid)
#define blockDim (emu::S().bdim)
#define gridDim (emu::S().gdim)

template <class V>
inline V __shfl(V v, int src, int width) {
    const int me = emu::S().cur;
    V out;
    emu::exchange(&v, &out, sizeof(V), (me / width) * width + (src & (width - 1)), width);
    return out;
}
template <class V>
inline V __shfl_xor(V v, int mask, int width) {
    const int me = emu::S().cur;
    V out;
    emu::exchange(&v, &out, sizeof(V), (me / width) * width + ((me ^ mask) & (width - 1)), width);
    return out;
}
template <class V>
inline V __shfl_down(V v, int delta, int width) {
    const int me = emu::S().cur;
    const int l = me & (width - 1);
    const int src = (l + delta < width) ? l + delta : l;
    V out;
    emu::exchange(&v, &out, sizeof(V), (me / width) * width + src, width);
    return out;
}
inline long long clock64() { return 0; }
inline long long wall_clock64() { return 0; }
inline void __syncthreads() { int z = 0; emu::exchange(&z, nullptr, sizeof(int), 0, 64); }
