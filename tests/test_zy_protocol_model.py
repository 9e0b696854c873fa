"""CPU model of the one-launch YZ stage's work distribution (distributedfft_amd/csrc/dfft_zy.hip): tickets, per-plane "published"
counters, prefetch-only-what-is-ready, and the lazy-publish rule "never wait for another workgroup while holding an unpublished
producer unit".  The kernel's two unit loops (eager and lazy) are restated statement by statement as Python generators that yield
wherever another workgroup's action could interleave; a random scheduler (including workgroups that start late, as a grid larger
than the resident set does) drives them.  Checked under many interleavings:
  * the launch terminates (no deadlock), every item is processed exactly once;
  * a consumer unit is loaded only after all producer units of its plane have published;
  * every producer has published when the launch ends (the next execute's done_base arithmetic relies on it);
  * the ticket counter ends at exactly items + 2 * workgroups -- zy_tickets(), what the host adds to its running ticket base.
Negative controls: the lazy loop WITHOUT the flush in front of its blocking wait deadlocks; a loop that prefetches without the
dependency poll loads a consumer unit too early.  (Test infrastructure: a model of the protocol, not the kernel's arithmetic --
that is what the -m gpu parity tests are for.)"""
import random

import pytest

NONE, PROD, CONS = 0, 1, 2


class Launch:
    def __init__(self, nplanes, chunk, ua, ub, groups):
        self.nplanes, self.ch, self.ua, self.ub, self.groups = nplanes, chunk, ua, ub, groups
        self.bb = ua + ub
        self.nchunks = (nplanes + chunk - 1) // chunk
        self.total = self.nchunks * chunk * self.bb  # dfft_zy.hip: `total`
        self.ticket = 0
        self.done = [0] * nplanes
        self.processed = {}
        self.violations = []

    def decode(self, t):  # dfft_zy.hip: decode()
        if t >= self.total:
            return (t, NONE, 0, 0)
        c, r = divmod(t, self.ch * self.bb)
        if r < self.ch * self.ua:
            pl = c * self.ch + r // self.ua
            return (t, PROD, pl, r % self.ua) if pl < self.nplanes else (t, NONE, 0, 0)
        r2 = r - self.ch * self.ua
        pl = c * self.ch + r2 // self.ub
        return (t, CONS, pl, r2 % self.ub) if pl < self.nplanes else (t, NONE, 0, 0)

    def is_ready(self, it):
        return it[1] != CONS or self.done[it[2]] >= self.ua

    def load(self, it):
        if it[1] == CONS and self.done[it[2]] < self.ua:
            self.violations.append(("consumer loaded before its plane was published", it))

    def process(self, it):
        key = (it[1], it[2], it[3])
        self.processed[key] = self.processed.get(key, 0) + 1


def workgroup(L, lazy, flush_before_wait=True, poll_before_prefetch=True):
    """One workgroup's life, as a generator; yields ('run',) between steps and ('wait', item) while it spins on a dependency."""
    def take():
        t = L.ticket
        L.ticket += 1
        return t

    def wait_ready(it):
        while not L.is_ready(it):
            yield ("wait", it)

    cur = L.decode(take())
    yield ("run",)
    nxt = L.decode(take())
    yield ("run",)
    if cur[1] != NONE:
        yield from wait_ready(cur)
        L.load(cur)
    pend = None
    nxt_ok = False
    while cur[0] < L.total:
        t2 = take()
        yield ("run",)
        loaded = False
        if nxt[1] != NONE and ((lazy and nxt_ok) or not poll_before_prefetch or L.is_ready(nxt)):
            L.load(nxt)
            loaded = True
        yield ("run",)
        nn_ok = False
        if cur[1] != NONE:
            L.process(cur)  # compute_unit
            yield ("run",)
            if lazy:
                if pend is not None:  # flush(): publish the previous producer unit at the quiet point
                    L.done[pend] += 1
                    pend = None
                nn = L.decode(t2)
                nn_ok = L.is_ready(nn)
                yield ("run",)
                # store_unit
                if cur[1] == PROD:
                    pend = cur[2]
            else:
                # store_unit, then publish at once
                if cur[1] == PROD:
                    L.done[cur[2]] += 1
                nn = L.decode(t2)
        else:
            nn = L.decode(t2)
        yield ("run",)
        if nxt[1] != NONE and not loaded:
            if lazy and flush_before_wait and pend is not None:
                L.done[pend] += 1
                pend = None
            yield from wait_ready(nxt)
            L.load(nxt)
        cur, nxt, nxt_ok = nxt, nn, nn_ok
        yield ("run",)
    if lazy and pend is not None:
        L.done[pend] += 1


def run(L, lazy, seed, late=0, **kw):
    """Drive L.groups workgroups with a random scheduler; `late` of them only start once an earlier one has finished.
    Returns 'ok' or 'deadlock'."""
    rng = random.Random(seed)
    gens = [workgroup(L, lazy, **kw) for _ in range(L.groups)]
    state = ["new"] * L.groups  # new / run / wait / done
    held_back = set(range(L.groups - late, L.groups))
    steps = 0
    while any(s != "done" for s in state):
        steps += 1
        assert steps < 2_000_000, "model did not terminate"
        if any(s == "done" for s in state):
            held_back = set()  # a resident slot became free
        cand = [g for g in range(L.groups) if state[g] != "done" and g not in held_back]
        if not cand:
            held_back = set()
            continue
        # a deadlock: every startable workgroup spins on a dependency nobody left can satisfy
        runnable = [g for g in cand if state[g] != "wait"]
        if not runnable:
            progressed = False
            for g in cand:
                ev = next(gens[g], None)
                if ev is None:
                    state[g], progressed = "done", True
                elif ev[0] != "wait":
                    state[g], progressed = "run", True
            if not progressed and not held_back:
                return "deadlock"
            if not progressed:
                held_back = set()
            continue
        g = rng.choice(cand)
        ev = next(gens[g], None)
        state[g] = "done" if ev is None else ("wait" if ev[0] == "wait" else "run")
    return "ok"


GEOMS = [  # planes, planes per phase, producer units per plane, consumer units per plane, workgroups
    (5, 2, 3, 2, 1), (5, 2, 3, 2, 2), (7, 3, 4, 4, 3), (9, 4, 2, 5, 4), (6, 6, 3, 3, 5), (4, 1, 2, 2, 3), (10, 3, 1, 1, 6),
]


@pytest.mark.parametrize("lazy", [False, True], ids=["eager", "lazy"])
@pytest.mark.parametrize("geom", GEOMS)
def test_every_interleaving_tried_terminates_and_respects_the_dependencies(geom, lazy):
    nplanes, ch, ua, ub, groups = geom
    for seed in range(60):
        for late in sorted({0, groups // 2, groups - 1}):
            L = Launch(nplanes, ch, ua, ub, groups)
            assert run(L, lazy, seed, late) == "ok", (geom, seed, late)
            assert not L.violations, L.violations[:3]
            want = {(PROD, p, u): 1 for p in range(nplanes) for u in range(ua)}
            want.update({(CONS, p, u): 1 for p in range(nplanes) for u in range(ub)})
            assert L.processed == want
            assert L.done == [ua] * nplanes                 # every producer published by the end of the launch
            assert L.ticket == L.total + 2 * groups         # zy_tickets(): the host's running ticket base


def test_lazy_loop_without_the_flush_before_a_blocking_wait_deadlocks():
    """One workgroup, one plane per phase: after its last producer unit the workgroup's next item is the consumer of that very
    plane -- holding the unit unpublished while waiting for it is a deadlock, and the flush in front of the wait removes it."""
    assert run(Launch(4, 1, 2, 2, 1), True, 0, flush_before_wait=False) == "deadlock"
    assert run(Launch(4, 1, 2, 2, 1), True, 0) == "ok"
    hits = sum(run(Launch(6, 2, 2, 3, 2), True, s, flush_before_wait=False) == "deadlock" for s in range(40))
    assert hits > 0


def test_prefetch_without_the_dependency_poll_is_caught():
    bad = 0
    for seed in range(40):
        L = Launch(7, 3, 4, 4, 3)
        run(L, False, seed, poll_before_prefetch=False)
        bad += bool(L.violations)
    assert bad > 0
