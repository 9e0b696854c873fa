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
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "dfft_internal.h"

namespace {
struct Event {
    std::atomic<unsigned> seq{0};  // index + 1 of the event this slot holds, published last (release); 0 while it is being rewritten
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

// The dump is written with write(2) from a stack buffer, formatted by hand: it also runs inside the SIGUSR2 handler, where stdio,
// malloc and iostreams are off limits (a signal that arrives inside malloc or fprintf would dead-lock the rank it was sent to
// diagnose).  clock_gettime, getpid, write and -- once libgcc is loaded, see install_handler -- backtrace / backtrace_symbols_fd are
// async-signal-safe.
struct Line {
    char buf[320];
    int  n = 0;
    void str(const char* p) {
        for (; p && *p && n < (int)sizeof(buf) - 1; ++p) buf[n++] = *p;
    }
    void num(long long v) {
        char tmp[24];
        int  k = 0;
        unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
        do tmp[k++] = (char)('0' + u % 10); while ((u /= 10) != 0);
        if (v < 0) tmp[k++] = '-';
        while (k > 0 && n < (int)sizeof(buf) - 1) buf[n++] = tmp[--k];
    }
    void secs(double t) {  // seconds with four decimals
        if (t < 0) t = 0;
        const long long whole = (long long)t;
        long long       frac = (long long)((t - (double)whole) * 10000.0);
        num(whole);
        str(".");
        for (long long d = 1000; d >= 1; d /= 10) {
            if (n < (int)sizeof(buf) - 1) buf[n++] = (char)('0' + frac / d);
            frac %= d;
        }
    }
    void flush() {
        if (n < (int)sizeof(buf)) buf[n++] = '\n';
        ssize_t r = ::write(2, buf, (size_t)n);
        (void)r;
        n = 0;
    }
};
int g_rank = 0;  // read from the environment once, outside any handler (trace())

void dump(const char* why, bool with_backtrace) {
    const unsigned n = g_next.load(std::memory_order_acquire);
    const unsigned first = n > RING ? n - RING : 0;
    const double   now = std::chrono::duration<double>(std::chrono::steady_clock::now() - g_t0).count();
    Line l;
    l.str("[dfft trace] rank "); l.num(g_rank); l.str(" pid "); l.num((long long)getpid()); l.str(": "); l.str(why);
    l.str(" -- last "); l.num(n - first); l.str(" of "); l.num(n); l.str(" control-plane events (now = "); l.secs(now); l.str(" s)");
    l.flush();
    for (unsigned i = first; i < n; ++i) {
        const Event& e = g_ring[i % RING];
        // several threads write the ring: a slot is shown only if it holds event i completely (its sequence word is published last)
        if (e.seq.load(std::memory_order_acquire) != i + 1) {
            l.str("[dfft trace]   (event "); l.num(i); l.str(" is being overwritten)");
            l.flush();
            continue;
        }
        const double t = e.t;
        const char*  w = e.what;
        const long long a = e.a, b = e.b;
        if (e.seq.load(std::memory_order_acquire) != i + 1) continue;  // overwritten while it was read
        l.str("[dfft trace]   "); l.secs(t); l.str(" s  "); l.str(w ? w : "?"); l.str("  "); l.num(a); l.str(" "); l.num(b);
        l.flush();
    }
    if (with_backtrace) {
        void*     frames[48];
        const int k = backtrace(frames, 48);
        l.str("[dfft trace] native backtrace of the interrupted thread ("); l.num(k); l.str(" frames):");
        l.flush();
        backtrace_symbols_fd(frames, k, 2);
    }
}

void on_sigusr2(int) {
    const int saved = errno;
    dump("SIGUSR2", true);
    errno = saved;
}
}  // namespace

namespace dfft {

void trace(const char* what, long long a, long long b) {
    if (!g_signal_checked.exchange(true)) {
        g_rank = trace_rank();
        const char* e = getenv("DFFT_TRACE_SIGNAL");
        if (e && *e && *e != '0') {
            void* warm[4];
            (void)backtrace(warm, 4);  // the first call loads libgcc_s (dlopen, malloc): do it here, not inside the handler
            struct sigaction sa;
            std::memset(&sa, 0, sizeof(sa));
            sa.sa_handler = on_sigusr2;
            sigaction(SIGUSR2, &sa, nullptr);
        }
    }
    const unsigned i = g_next.fetch_add(1, std::memory_order_acq_rel);
    Event&         e = g_ring[i % RING];
    e.seq.store(0, std::memory_order_release);
    e.t = std::chrono::duration<double>(std::chrono::steady_clock::now() - g_t0).count();
    e.what = what;
    e.a = a;
    e.b = b;
    e.seq.store(i + 1, std::memory_order_release);
}

void trace_on_error(int code, const std::string& msg) {
    if (code != DFFT_ECOMM && code != DFFT_ERCCL) return;
    static const bool on = [] {
        const char* e = getenv("DFFT_TRACE_ON_ERROR");
        return !(e && *e == '0');
    }();
    if (!on) return;
    trace("error", code, 0);
    // one ring per burst of failures: cascaded errors (every plan of a communicator whose peer died) and tests that provoke
    // DFFT_ECOMM on purpose would otherwise print up to 256 lines each.  Later failures within 5 s get one line.
    static std::atomic<long long> last_ms{-1000000};
    const long long now_ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - g_t0).count();
    const long long prev = last_ms.load();
    if (now_ms - prev >= 5000) last_ms.store(now_ms);
    if (now_ms - prev < 5000) {
        Line l;
        l.str("[dfft trace] rank "); l.num(g_rank); l.str(": "); l.str(msg.c_str()); l.str(" (event ring printed "); l.num(now_ms - prev); l.str(" ms ago, not repeated)");
        l.flush();
        return;
    }
    dump(msg.c_str(), false);
}

}  // namespace dfft

extern "C" int dfft_trace_dump(void) {
    dump("dfft_trace_dump", false);
    return DFFT_OK;
}
