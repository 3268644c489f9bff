"""Wave-level model of the ping-pong conv kernels' schedules (conv_igemm_bf16_pp.hip, conv3x3_dwr_bf16.hip): CPU-only, no GPU code.

What a kernel run cannot show reliably -- a read of an LDS region that races with the LDS-DMA that (re)fills it -- this model
shows deterministically.  It executes the eight waves' phase programs (the same sequence of fragment reads, LDS-DMA issues, counted
s_waitcnt vmcnt(N), lgkmcnt(0) and s_barrier as the kernels, wave group 1 one barrier behind group 0, the re-alignment around the
epilogue) under random interleavings of the waves between barriers and with ADVERSARIAL landing times of every piece:
  early  a piece lands the moment it is issued          -> catches write-after-read (a region re-filled while still being read)
  late   a piece lands only when its wave's vmcnt forces it -> catches read-after-write (a region read before its pieces landed)
  random anything in between.
Every fragment read is checked twice (at issue and at the lgkmcnt(0) that retires it): all eight waves' pieces of the region must
hold the chunk / step the reader expects.  Barrier counts must agree at the end (no deadlock).  tests/test_pp_schedule_cpu.py runs it
and checks that deliberately broken schedules ARE detected.
"""
import random


def simulate(make_prog, mode, seed=0):
    progs = [list(make_prog(w)) for w in range(8)]
    rnd = random.Random(seed)
    pc = [0] * 8
    landed = {}
    queue = [[] for _ in range(8)]
    pending = [[] for _ in range(8)]
    nbar = [0] * 8
    at_bar = [False] * 8
    done = [False] * 8

    def land(w, force_until=None):
        q = queue[w]
        if force_until is not None:
            while len(q) > force_until:
                r, b, c = q.pop(0)
                landed[(r, b, w)] = c
        if mode == "early":
            while q:
                r, b, c = q.pop(0)
                landed[(r, b, w)] = c
        elif mode == "random":
            while q and rnd.random() < 0.3:
                r, b, c = q.pop(0)
                landed[(r, b, w)] = c

    def check(w, region, buf, ident):
        for ww in range(8):
            got = landed.get((region, buf, ww))
            if got != ident:
                raise AssertionError("wave %d reads %s of buffer %s for %s but wave %d's pieces hold %s" % (w, region, buf, ident, ww, got))

    while not all(done):
        progressed = False
        order = list(range(8))
        rnd.shuffle(order)
        for w in order:
            if done[w] or at_bar[w]:
                continue
            n = rnd.randint(1, 6)
            while n > 0 and not at_bar[w] and not done[w]:
                n -= 1
                if pc[w] >= len(progs[w]):
                    done[w] = True
                    break
                ins = progs[w][pc[w]]
                pc[w] += 1
                progressed = True
                if ins[0] == "issue":                     # (issue, region, buffer, identity, pieces)
                    for _ in range(ins[4]):
                        queue[w].append((ins[1], ins[2], ins[3]))
                    land(w)
                elif ins[0] == "wait":
                    land(w, force_until=ins[1])
                elif ins[0] == "read":
                    check(w, ins[1], ins[2], ins[3])
                    pending[w].append(ins)
                elif ins[0] == "lgk":
                    for r in pending[w]:
                        check(w, r[1], r[2], r[3])
                    pending[w] = []
                elif ins[0] == "slabw":                   # epilogue slabs inside an activation region: (slabw, region, buffer, identity)
                    for ww in range(8):
                        landed[(ins[1], ins[2], ww)] = ins[3]
                elif ins[0] == "slabr":                   # ... still intact when the epilogue ends (no piece landed on them)
                    check(w, ins[1], ins[2], ins[3])
                elif ins[0] == "bar":
                    at_bar[w] = True
                    nbar[w] += 1
                for ww in range(8):
                    land(ww)
        if all(at_bar[w] or done[w] for w in range(8)):
            if any(done) and any(at_bar):
                raise AssertionError("deadlock: some waves finished while others wait at a barrier %s %s" % (done, nbar))
            if all(at_bar):
                assert len(set(nbar)) == 1, nbar
                at_bar = [False] * 8
                progressed = True
        if not progressed and not all(done):
            raise AssertionError("stuck")
    assert len(set(nbar)) == 1, nbar
    return nbar[0]


def _tile_end(prog, g, last):
    if g == 0:
        prog.append(("bar",))
    prog.append(("epi",))
    if not last and g == 1:
        prog.append(("bar",))


def pp_program(w, nk, ntiles, bug=None):
    """conv_igemm_bf16_pp.hip: phases (j0,h0) (j1,h0) (j0,h1) (j1,h1); B0 / B1 of chunk G+1 in phases 1 / 2, A0 + A1 of chunk G+2 in
    phase 4; vmcnt(6) in phases 1 and 4; lgkmcnt(0) in front of the barrier in phase 3 only."""
    g = w >> 2
    total = nk * ntiles
    prog = [("issue", r, 0, 0, 2) for r in ("A0", "A1", "B0", "B1")] + [("issue", "A0", 1, 1, 2), ("issue", "A1", 1, 1, 2)]
    prog += [("wait", 4), ("bar",)]
    if g == 1:
        prog.append(("bar",))
    buf = 0
    for G in range(total):
        prog += [("read", "A0", buf, G), ("read", "A1", buf, G), ("read", "B0", buf, G), ("issue", "B0", buf ^ 1, G + 1, 2),
                 ("wait", 8 if bug == "wait1" else 6), ("bar",), ("lgk",), ("mfma",), ("bar",)]
        prog += [("read", "B1", buf, G), ("issue", "B1", buf ^ 1, G + 1, 2), ("bar",), ("lgk",), ("mfma",), ("bar",)]
        prog += [("read", "A0", buf, G), ("read", "A1", buf, G), ("read", "B0", buf, G)]
        prog += [("bar",), ("lgk",), ("mfma",), ("bar",)] if bug == "lgk3" else [("lgk",), ("bar",), ("mfma",), ("bar",)]
        prog += [("read", "B1", buf, G), ("issue", "A0", buf, G + 2, 2), ("issue", "A1", buf, G + 2, 2), ("wait", 8 if bug == "wait4" else 6),
                 ("bar",), ("lgk",), ("mfma",), ("bar",)]
        buf ^= 1
        if (G + 1) % nk == 0:
            if bug == "norealign":
                prog.append(("epi",))
            else:
                _tile_end(prog, g, G + 1 == total)
            if G + 1 == total:
                break
    return prog


def dwr_program(w, nsteps, ntiles, APW=2, BPW=2, bug=None):
    """conv3x3_dwr_bf16.hip: a step = chunks dw 0, 1, 2 on ONE activation load; A1 of step t+1 in phase 4 of (t, dw 0), A0 of step t+2
    in phase 4 of (t, dw 2); weights per chunk; counted waits BPW + (APW | 0), see DwrGeom::wait_p1 / wait_p4."""
    g = w >> 2
    total = nsteps * ntiles
    prog = [("issue", "A0", 0, ("s", 0), APW), ("issue", "A1", 0, ("s", 0), APW), ("issue", "B0", 2, ("c", 0), BPW), ("issue", "B1", 2, ("c", 0), BPW),
            ("issue", "A0", 1, ("s", 1), APW), ("wait", APW), ("bar",)]
    if g == 1:
        prog.append(("bar",))
    G = 0
    for t in range(total):
        ab = t & 1
        for dw in range(3):
            bb = 2 + (G & 1)
            n1 = BPW + (0 if dw == 2 else APW) + (1 if bug == "wait1" else 0)
            n4 = BPW + (0 if dw == 1 else APW) + (1 if bug == "wait4" else 0)
            prog += [("read", "A0", ab, ("s", t)), ("read", "A1", ab, ("s", t)), ("read", "B0", bb, ("c", G)), ("issue", "B0", bb ^ 1, ("c", G + 1), BPW),
                     ("wait", n1), ("bar",), ("lgk",), ("mfma",), ("bar",)]
            prog += [("read", "B1", bb, ("c", G)), ("issue", "B1", bb ^ 1, ("c", G + 1), BPW), ("bar",), ("lgk",), ("mfma",), ("bar",)]
            prog += [("read", "A0", ab, ("s", t)), ("read", "A1", ab, ("s", t)), ("read", "B0", bb, ("c", G))]
            prog += [("lgk",), ("bar",), ("mfma",), ("bar",)] if dw == 2 and bug != "lgk3" else [("bar",), ("lgk",), ("mfma",), ("bar",)]
            prog.append(("read", "B1", bb, ("c", G)))
            if dw == 0:
                prog.append(("issue", "A1", ab ^ 1, ("s", t + 1), APW))
            if dw == 2:
                prog.append(("issue", "A0", ab, ("s", t + 2), APW))
            prog += [("wait", n4), ("bar",), ("lgk",), ("mfma",), ("bar",)]
            G += 1
        if (t + 1) % nsteps == 0:
            _tile_end(prog, g, t + 1 == total)
            if t + 1 == total:
                break
    return prog


def dwr64_program(w, nsteps, ntiles, bug=None):
    """conv3x3_dwr64_bf16_kernel (512 x 64 tiles, wave tile 128 x 32): TWO phases per chunk (K halves), ONE weight region per chunk in a
    ring of four (B of chunk G+3 goes out in phase X of chunk G), the activations of step t+1 in phases X / Y of (t, dw 0); counted
    waits in every Y phase: 10, 10, 2; every Y phase waits for its fragment reads in front of its barrier."""
    g = w >> 2
    total = nsteps * ntiles
    prog = [("issue", "A0", 0, ("s", 0), 4), ("issue", "A1", 0, ("s", 0), 4)] + [("issue", "B", 10 + c, ("c", c), 1) for c in range(3)]
    prog += [("wait", 2), ("bar",)]
    if g == 1:
        prog.append(("bar",))
    G = 0
    for t in range(total):
        ab = t & 1
        for dw in range(3):
            bb = 10 + (G & 3)
            prog += [("read", "A0", ab, ("s", t)), ("read", "A1", ab, ("s", t)), ("read", "B", bb, ("c", G)), ("issue", "B", 10 + ((G + 3) & 3), ("c", G + 3), 1)]
            if dw == 0:
                prog.append(("issue", "A0", ab ^ 1, ("s", t + 1), 4))
            prog += [("bar",), ("lgk",), ("mfma",), ("bar",)]
            prog += [("read", "A0", ab, ("s", t)), ("read", "A1", ab, ("s", t)), ("read", "B", bb, ("c", G))]
            if dw == 0:
                prog.append(("issue", "A1", ab ^ 1, ("s", t + 1), 4))
            n = (2 if dw == 2 else 10) + (1 if bug == "wait" else 0)
            prog.append(("wait", n))
            prog += [("bar",), ("lgk",), ("mfma",), ("bar",)] if bug == "lgk" else [("lgk",), ("bar",), ("mfma",), ("bar",)]
            G += 1
        if (t + 1) % nsteps == 0:                       # the slabs live in region A1 of the buffer this tile has just finished with
            if g == 0 and bug != "norealign":
                prog.append(("bar",))
            prog += [("slabw", "A0" if bug == "slab" else "A1", ab, ("slab", t)), ("slabr", "A0" if bug == "slab" else "A1", ab, ("slab", t))]
            if t + 1 == total:
                break
            if g == 1 and bug != "norealign":
                prog.append(("bar",))
    return prog


def sweep(make, seeds=10):
    for mode in ("early", "late", "random"):
        for seed in range(seeds):
            simulate(make, mode, seed)


if __name__ == "__main__":
    for nk in (2, 3, 4, 9):
        for nt in (1, 2, 3):
            sweep(lambda w: pp_program(w, nk, nt))
    for apw, bpw in ((2, 2), (4, 1)):
        for ns in (1, 2, 3, 6):
            for nt in (1, 2, 3):
                sweep(lambda w: dwr_program(w, ns, nt, apw, bpw))
    for ns in (1, 2, 3, 6):
        for nt in (1, 2, 3):
            sweep(lambda w: dwr64_program(w, ns, nt))
    print("schedule model: ok")
