// dfft_bootstrap.cpp -- minimal TCP rendezvous (star over rank 0) for multi-process launches without MPI.
//
// The reference driver uses MPI for four control-plane things only (fftSpeed3d_c2c.cpp:18-26, 120-124, and the dead
// ncclUniqueId broadcast in fft_mpi_3d_api.cpp:29-37): rank/size, a broadcast, a barrier and a MAX reduction.  The data
// plane (t2) is RCCL here, so a ~200-line socket layer replaces the MPI dependency.  Not a general MPI: blocking, root 0
// hub, intended for <= a few dozen ranks on one node.
#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "dfft_internal.h"

namespace {

struct Boot {
    bool             inited = false;
    int              rank = 0, size = 1;
    int              hub = -1;        // non-root: socket to rank 0
    std::vector<int> peers;           // root: socket per rank (index = rank; [0] unused)
    bool             broken = false;  // a collective failed (time-out / dead peer): the byte streams may be mid-message, nothing more is read from them
    uint32_t         magic = 0;       // hello word of THIS job (job_magic)
} g;

const char* env_first(std::initializer_list<const char*> names) {
    for (const char* n : names) {
        const char* v = getenv(n);
        if (v && *v) return v;
    }
    return nullptr;
}

bool send_all(int fd, const void* buf, size_t n) {
    const char* p = (const char*)buf;
    while (n) {
        ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL);
        if (k < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += k;
        n -= (size_t)k;
    }
    return true;
}
// Every wait of the rendezvous is bounded (DFFT_BOOT_TIMEOUT_S, default 180 s; 0 = wait for ever): a peer that died, or that
// left the collective call sequence, turns into DFFT_ECOMM with the process's last control-plane events on stderr
// (dfft_trace.cpp) instead of a rank that hangs in recv() -- the MPI calls this layer replaces (fftSpeed3d_c2c.cpp:18-26,
// 120-124) would hang the same way, which is how round 4's stalled multi-process cases left no trace of their cause.
int boot_timeout_ms() {
    static const int ms = [] {
        const char* e = getenv("DFFT_BOOT_TIMEOUT_S");
        const double s = e && *e ? atof(e) : 180.0;
        return s <= 0 ? -1 : (int)(s * 1000.0);
    }();
    return ms;
}
int  g_recv_state = 0;  // why the last recv_all failed: 1 = time-out, 2 = peer closed the connection, 3 = socket error
bool recv_all(int fd, void* buf, size_t n, int timeout_ms = -2) {
    char* p = (char*)buf;
    if (timeout_ms == -2) timeout_ms = boot_timeout_ms();
    g_recv_state = 0;
    while (n) {
        pollfd pf{fd, POLLIN, 0};
        const int pr = ::poll(&pf, 1, timeout_ms);
        if (pr < 0) {
            if (errno == EINTR) continue;  // (a signal, e.g. the trace dump's SIGUSR2: keep waiting)
            g_recv_state = 3;
            return false;
        }
        if (pr == 0) {
            g_recv_state = 1;
            return false;
        }
        ssize_t k = ::recv(fd, p, n, 0);
        if (k < 0) {
            if (errno == EINTR) continue;
            g_recv_state = 3;
            return false;
        }
        if (k == 0) {
            g_recv_state = 2;
            return false;
        }
        p += k;
        n -= (size_t)k;
    }
    return true;
}
const char* recv_why() {
    return g_recv_state == 1   ? "no answer within DFFT_BOOT_TIMEOUT_S (default 180 s): the peer is stuck, or is not in the same collective call"
           : g_recv_state == 2 ? "the peer closed the connection (it exited, or failed and left the rendezvous)"
                               : "socket error";
}
// Hello word: "DFFT" mixed with a hash of what every rank of ONE job agrees on (rendezvous address, port, world size; DFFT_JOB_ID if the
// launcher sets one), so that a rank of another dfft job aimed at the same port is turned away like any other stranger.
uint32_t job_magic(const char* addr, int port, int size) {
    uint32_t h = 2166136261u;  // FNV-1a
    auto mix = [&](const char* p) {
        for (; p && *p; ++p) h = (h ^ (unsigned char)*p) * 16777619u;
    };
    mix(addr);
    mix(":");
    mix(std::to_string(port).c_str());
    mix("/");
    mix(std::to_string(size).c_str());
    mix(getenv("DFFT_JOB_ID"));
    return 0x44464654u ^ h;
}
unsigned long long g_ops = 0;                  // collective operations entered by this process (all ranks count alike)

}  // namespace

using namespace dfft;

extern "C" {

int dfft_boot_init(void) {
    if (g.inited) return DFFT_OK;
    const char* r = env_first({"DFFT_RANK", "RANK", "PMI_RANK", "OMPI_COMM_WORLD_RANK"});
    const char* s = env_first({"DFFT_WORLD_SIZE", "WORLD_SIZE", "PMI_SIZE", "OMPI_COMM_WORLD_SIZE"});
    g.rank = r ? atoi(r) : 0;
    g.size = s ? atoi(s) : 1;
    if (g.size < 1 || g.rank < 0 || g.rank >= g.size) return fail(DFFT_ECOMM, "dfft_boot_init: inconsistent rank/size");
    if (g.size == 1) {
        g.inited = true;
        return DFFT_OK;
    }
    const char* addr = env_first({"DFFT_MASTER_ADDR", "MASTER_ADDR"});
    const char* port = env_first({"DFFT_MASTER_PORT", "MASTER_PORT"});
    if (!addr) addr = "127.0.0.1";
    int portno = port ? atoi(port) : 29533;
    if (!getenv("DFFT_MASTER_PORT") && getenv("MASTER_PORT")) portno += 1;  // stay off torchrun's own store port

    g.magic = job_magic(addr, portno, g.size);
    g.broken = false;

    if (g.rank == 0) {
        int ls = ::socket(AF_INET, SOCK_STREAM, 0);
        if (ls < 0) return fail(DFFT_ECOMM, "dfft_boot_init: socket()");
        int one = 1;
        setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        sockaddr_in sa;
        std::memset(&sa, 0, sizeof(sa));
        sa.sin_family = AF_INET;
        // listen on the rendezvous address only (loopback unless DFFT_MASTER_ADDR / MASTER_ADDR names a local interface):
        // the hello is unauthenticated and the channel carries the RCCL id and hipIpc handles, so it must not be reachable
        // from interfaces the job does not use.  An address that is not local falls back to loopback.
        if (inet_pton(AF_INET, addr, &sa.sin_addr) != 1) {
            sa.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
            addrinfo hints, *res = nullptr;  // a host name: listen where the peers will resolve it to
            std::memset(&hints, 0, sizeof(hints));
            hints.ai_family = AF_INET;
            hints.ai_socktype = SOCK_STREAM;
            if (getaddrinfo(addr, nullptr, &hints, &res) == 0 && res) {
                sa.sin_addr = ((sockaddr_in*)res->ai_addr)->sin_addr;
                freeaddrinfo(res);
            }
        }
        sa.sin_port = htons((uint16_t)portno);
        bool bound = ::bind(ls, (sockaddr*)&sa, sizeof(sa)) == 0;
        int  bind_errno = bound ? 0 : errno;
        // Loopback is a substitute only when the named address does not exist on this host (EADDRNOTAVAIL: a container
        // whose hostname resolves to an address of another namespace) -- any other failure (port in use, permission) is
        // reported as it is: listening on loopback while the peers connect to the real address would hang the job.
        if (!bound && bind_errno == EADDRNOTAVAIL && sa.sin_addr.s_addr != htonl(INADDR_LOOPBACK)) {
            sa.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
            bound = ::bind(ls, (sockaddr*)&sa, sizeof(sa)) == 0;
            if (bound && getenv("DFFT_DEBUG"))
                fprintf(stderr, "[dfft] rendezvous: %s is not a local address, listening on 127.0.0.1:%d (single-node jobs only)\n", addr, portno);
        }
        if (!bound || ::listen(ls, g.size) != 0) {
            const int err = bound ? errno : bind_errno;
            ::close(ls);
            return fail(DFFT_ECOMM, std::string("dfft_boot_init: bind/listen on ") + addr + ":" + std::to_string(portno) + ": " +
                                        strerror(err));
        }
        if (getenv("DFFT_DEBUG")) {
            char txt[INET_ADDRSTRLEN] = "?";
            inet_ntop(AF_INET, &sa.sin_addr, txt, sizeof(txt));
            fprintf(stderr, "[dfft] rendezvous: rank 0 listening on %s:%d for %d ranks\n", txt, portno, g.size);
        }
        g.peers.assign(g.size, -1);
        trace("boot: rank 0 listening", portno, g.size);
        for (int joined = 1; joined < g.size;) {
            pollfd pf{ls, POLLIN, 0};
            const int pr_ = ::poll(&pf, 1, boot_timeout_ms());
            if (pr_ < 0 && errno == EINTR) continue;
            if (pr_ <= 0) {
                ::close(ls);
                return fail(DFFT_ECOMM, "dfft_boot_init: rank 0 saw only " + std::to_string(joined) + " of " + std::to_string(g.size) +
                                            " ranks connect within DFFT_BOOT_TIMEOUT_S on port " + std::to_string(portno));
            }
            int fd = ::accept(ls, nullptr, nullptr);
            if (fd < 0) {
                if (errno == EINTR || errno == ECONNABORTED) continue;
                ::close(ls);
                return fail(DFFT_ECOMM, std::string("dfft_boot_init: accept(): ") + strerror(errno));
            }
            setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
            // hello = {magic, rank}; anything else (a port scanner, another job's client that guessed our port) is dropped
            // without failing the rendezvous
            uint32_t hello[2] = {0, 0};
            if (!recv_all(fd, hello, sizeof(hello), 5000) || hello[0] != g.magic || hello[1] < 1 || (int)hello[1] >= g.size) {
                trace("boot: dropped a connection that is not a rank of this job", hello[0], hello[1]);
                ::close(fd);
                continue;
            }
            const uint32_t ack[2] = {g.magic, (uint32_t)g.size};
            if (!send_all(fd, ack, sizeof(ack))) {
                ::close(fd);
                continue;
            }
            // a second hello of a rank that is registered already: the client gave up on its first connection (its wait for the ack
            // timed out while this loop was busy) and came back -- the NEW socket is the live one, the old one is closed on its side
            const bool again = g.peers[hello[1]] != -1;
            if (again) ::close(g.peers[hello[1]]);
            g.peers[hello[1]] = fd;
            trace(again ? "boot: rank re-joined on a new connection" : "boot: rank joined", hello[1], joined);
            if (!again) ++joined;
        }
        ::close(ls);
    } else {
        addrinfo hints, *res = nullptr;
        std::memset(&hints, 0, sizeof(hints));
        hints.ai_family = AF_INET;
        hints.ai_socktype = SOCK_STREAM;
        const std::string ps = std::to_string(portno);
        if (getaddrinfo(addr, ps.c_str(), &hints, &res) != 0 || !res)
            return fail(DFFT_ECOMM, std::string("dfft_boot_init: cannot resolve ") + addr);
        int fd = -1;
        int one = 1;
        trace("boot: connecting to rank 0", portno, g.rank);
        int strangers = 0;
        // rank 0 may start later: retry until DFFT_BOOT_TIMEOUT_S has elapsed in all (at least 60 s), whatever an attempt costs
        const auto t_start = std::chrono::steady_clock::now();
        const long long budget_ms = boot_timeout_ms() < 0 ? -1 : (boot_timeout_ms() < 60000 ? 60000 : boot_timeout_ms());
        for (int attempt = 0;; ++attempt) {
            if (attempt > 0 && budget_ms >= 0 &&
                std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_start).count() > budget_ms)
                break;
            fd = ::socket(AF_INET, SOCK_STREAM, 0);
            if (fd >= 0 && ::connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
                // the listener must answer the hello with this job's magic and world size: a port that some OTHER listener
                // happens to own (torchrun's store is one port below, RCCL's bootstrap sockets take ephemeral ports) accepts
                // the connection too, and a rank that took it for rank 0 would wait in its first collective for ever
                setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
                const uint32_t hello[2] = {g.magic, (uint32_t)g.rank};
                uint32_t       ack[2] = {0, 0};
                if (send_all(fd, hello, sizeof(hello)) && recv_all(fd, ack, sizeof(ack), 10000) && ack[0] == g.magic &&
                    (int)ack[1] == g.size)
                    break;
                ++strangers;
                trace("boot: the listener on the rendezvous port is not rank 0 of this job", ack[0], ack[1]);
            }
            if (fd >= 0) ::close(fd);
            fd = -1;
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
        freeaddrinfo(res);
        if (fd < 0)
            return fail(DFFT_ECOMM, std::string("dfft_boot_init: cannot reach rank 0 at ") + addr + ":" + ps +
                                        (strangers ? " (a listener that is not this job's rank 0 answered there)" : ""));
        g.hub = fd;
        trace("boot: joined", g.rank, g.size);
    }
    g.inited = true;
    return DFFT_OK;
}

int dfft_boot_rank(void) { return g.rank; }
int dfft_boot_size(void) { return g.size; }

static int boot_fail(const char* op, int peer) {
    g.broken = true;  // the stream to that peer may have stopped mid-message: no later collective may read from it
    return fail(DFFT_ECOMM, std::string(op) + " (collective #" + std::to_string(g_ops) + " of rank " + std::to_string(g.rank) + "): waiting for rank " +
                                std::to_string(peer) + ": " + recv_why());
}

int dfft_boot_bcast(void* buf, size_t bytes, int root) {
    if (!g.inited) return fail(DFFT_ECOMM, "dfft_boot: not initialised");
    if (g.broken) return fail(DFFT_ECOMM, "dfft_boot_bcast: an earlier collective of this rendezvous failed (time-out or dead peer); it cannot be used again");
    if (g.size == 1 || bytes == 0) return DFFT_OK;
    ++g_ops;
    trace("boot: bcast enter", (long long)g_ops, root);
    if (root != 0) {  // route through the hub
        if (g.rank == root) {
            if (!send_all(g.hub, buf, bytes)) return fail(DFFT_ECOMM, "dfft_boot_bcast: send to rank 0 failed (it has gone)");
        } else if (g.rank == 0) {
            if (!recv_all(g.peers[root], buf, bytes)) return boot_fail("dfft_boot_bcast", root);
        }
    }
    if (g.rank == 0) {
        for (int i = 1; i < g.size; ++i)
            if (i != root || root == 0)
                if (!send_all(g.peers[i], buf, bytes)) return fail(DFFT_ECOMM, "dfft_boot_bcast: send to rank " + std::to_string(i) + " failed (it has gone)");
    } else if (g.rank != root) {
        if (!recv_all(g.hub, buf, bytes)) return boot_fail("dfft_boot_bcast", 0);
    }
    trace("boot: bcast done", (long long)g_ops, root);
    return DFFT_OK;
}

int dfft_boot_allreduce_max(double* v, int n) {
    if (!g.inited) return fail(DFFT_ECOMM, "dfft_boot: not initialised");
    if (g.broken) return fail(DFFT_ECOMM, "dfft_boot_allreduce_max: an earlier collective of this rendezvous failed (time-out or dead peer); it cannot be used again");
    if (g.size == 1 || n <= 0) return DFFT_OK;
    const size_t bytes = sizeof(double) * (size_t)n;
    ++g_ops;
    trace("boot: allreduce/barrier enter", (long long)g_ops, n);
    if (g.rank == 0) {
        std::vector<double> tmp(n);
        for (int i = 1; i < g.size; ++i) {
            if (!recv_all(g.peers[i], tmp.data(), bytes)) return boot_fail("dfft_boot_allreduce_max / barrier", i);
            for (int k = 0; k < n; ++k)
                if (tmp[k] > v[k]) v[k] = tmp[k];
        }
        for (int i = 1; i < g.size; ++i)
            if (!send_all(g.peers[i], v, bytes)) return fail(DFFT_ECOMM, "dfft_boot_allreduce_max: send to rank " + std::to_string(i) + " failed (it has gone)");
    } else {
        if (!send_all(g.hub, v, bytes)) return fail(DFFT_ECOMM, "dfft_boot_allreduce_max: send to rank 0 failed (it has gone)");
        if (!recv_all(g.hub, v, bytes)) return boot_fail("dfft_boot_allreduce_max / barrier", 0);
    }
    trace("boot: allreduce/barrier done", (long long)g_ops, n);
    return DFFT_OK;
}

int dfft_boot_barrier(void) {
    double z = 0;
    return dfft_boot_allreduce_max(&z, 1);
}

int dfft_boot_finalize(void) {
    if (!g.inited) return DFFT_OK;
    if (g.size > 1) dfft_boot_barrier();
    if (g.hub >= 0) ::close(g.hub);
    for (int fd : g.peers)
        if (fd >= 0) ::close(fd);
    g.peers.clear();
    g.hub = -1;
    g.inited = false;
    return DFFT_OK;
}

}  // extern "C"
