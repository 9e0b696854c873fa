// dfft_trace.cpp -- a small always-on event ring for the control plane (rendezvous, registrations, exchange rounds, RCCL calls,
// executes), so that a rank that stalls or fails can say WHERE: the last events of the process, each with its wall-clock offset.
//
// Why it exists: the multi-process paths (reference analogue: the MPI / hipMemcpyPeer exchange and the rendezvous around it,
// fft_mpi_3d_api.cpp:610-672, fftSpeed3d_c2c.cpp:18-26) failed twice in round 4 by stalling for minutes with nothing on stderr.
//   * every DFFT_ECOMM / DFFT_ERCCL failure prints the ring to stderr (DFFT_TRACE_ON_ERROR=0 switches that off);
//   * DFFT_TRACE_SIGNAL=1 installs a SIGUSR2 handler that prints the ring and a native backtrace of the interrupted thread
//     (tools/stall_hunt.py sends it to every rank of a launch that exceeds its time limit);
//   * dfft_trace_dump() prints it on demand through the C-ABI.
// Cost: one relaxed fetch_add and five stores per event; events are control-plane operations (microseconds apart at most).
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "dfft_internal.h"

namespace {
struct Event {
    double      t;     // seconds since the first event of the process
    const char* what;  // string literal
    long long   a, b;
};
constexpr unsigned RING = 256;
Event                  g_ring[RING];
std::atomic<unsigned>  g_next{0};
const auto             g_t0 = std::chrono::steady_clock::now();
std::atomic<bool>      g_signal_checked{false};

int trace_rank() {
    for (const char* n : {"DFFT_RANK", "RANK", "PMI_RANK", "OMPI_COMM_WORLD_RANK"}) {
        const char* v = getenv(n);
        if (v && *v) return atoi(v);
    }
    return 0;
}

void dump(const char* why, bool with_backtrace) {
    const unsigned n = g_next.load(std::memory_order_acquire);
    const unsigned first = n > RING ? n - RING : 0;
    const double   now = std::chrono::duration<double>(std::chrono::steady_clock::now() - g_t0).count();
    fprintf(stderr, "[dfft trace] rank %d pid %d: %s -- last %u of %u control-plane events (now = %.3f s)\n", trace_rank(), (int)getpid(), why, n - first,
            n, now);
    for (unsigned i = first; i < n; ++i) {
        const Event& e = g_ring[i % RING];
        fprintf(stderr, "[dfft trace]   %10.4f s  %-34s %lld %lld\n", e.t, e.what ? e.what : "?", e.a, e.b);
    }
    if (with_backtrace) {
        void*     frames[48];
        const int k = backtrace(frames, 48);
        fprintf(stderr, "[dfft trace] native backtrace of the interrupted thread (%d frames):\n", k);
        fflush(stderr);
        backtrace_symbols_fd(frames, k, 2);
    }
    fflush(stderr);
}

void on_sigusr2(int) { dump("SIGUSR2", true); }
}  // namespace

namespace dfft {

void trace(const char* what, long long a, long long b) {
    if (!g_signal_checked.exchange(true)) {
        const char* e = getenv("DFFT_TRACE_SIGNAL");
        if (e && *e && *e != '0') {
            struct sigaction sa;
            std::memset(&sa, 0, sizeof(sa));
            sa.sa_handler = on_sigusr2;
            sigaction(SIGUSR2, &sa, nullptr);
        }
    }
    const unsigned i = g_next.fetch_add(1, std::memory_order_acq_rel);
    Event&         e = g_ring[i % RING];
    e.t = std::chrono::duration<double>(std::chrono::steady_clock::now() - g_t0).count();
    e.what = what;
    e.a = a;
    e.b = b;
}

void trace_on_error(int code, const std::string& msg) {
    if (code != DFFT_ECOMM && code != DFFT_ERCCL) return;
    static const bool on = [] {
        const char* e = getenv("DFFT_TRACE_ON_ERROR");
        return !(e && *e == '0');
    }();
    if (!on) return;
    trace("error", code, 0);
    dump(msg.c_str(), false);
}

}  // namespace dfft

extern "C" int dfft_trace_dump(void) {
    dump("dfft_trace_dump", false);
    return DFFT_OK;
}
