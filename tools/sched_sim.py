#!/usr/bin/env python3
"""Offline lane-planning experiments on a dumped dependency graph (tools/critical_path.py --json): the cost model of
acrmi_program.hip build_schedule (measured on MI355X: a dependent kernel on the SAME stream starts when its producer ends - the
launch gap is inside the profiled op time - through an event on ANOTHER stream ~16 us later; each other lane waited for costs
the consumer's queue ~2 us) applied to different planners:

  greedy     the round-5 planner: ops in program order, each to the lane where it can START first
  rank       the same rule with ops taken in order of their upward rank (longest remaining path): the critical chain claims
             its lane first, side chains fill the others
  refine     rank + local search: single-op and chain moves between lanes, accepted when the simulated makespan drops

    python tools/sched_sim.py gpurun_out/r06_dag_b1.json [--lanes 4] [--wait-us 16]
"""
import argparse
import json


def load(path):
    d = json.load(open(path))
    ops = d['ops']
    idx = {o['idx']: k for k, o in enumerate(ops)}
    ms = [o['ms'] for o in ops]
    deps = [[idx[x] for x in o['deps'] if x in idx] for o in ops]
    return ops, ms, deps


def simulate(order, lane, ms, deps, L, wait, sync=0.002, event=0.002):
    """makespan of a lane assignment; every lane runs its ops in `order`"""
    fin = [0.0] * len(ms)
    free = [0.0] * L
    for j in order:
        l = lane[j]
        start = free[l]
        others = set()
        for d in deps[j]:
            if lane[d] != l:
                others.add(lane[d])
                start = max(start, fin[d] + wait)
        start += sync * len(others)
        fin[j] = start + max(ms[j] - event, 0.002)
        free[l] = fin[j] + 0.5 * sync
    return max(fin), fin


def plan(order, ms, deps, L, wait, sync=0.002, event=0.002):
    n = len(ms)
    lane = [0] * n
    fin = [0.0] * n
    free = [0.0] * L
    for j in order:
        best = None
        for l in range(L):
            start = free[l]
            others = set()
            prod = False
            for d in deps[j]:
                if lane[d] == l:
                    prod = True
                    continue
                others.add(lane[d])
                start = max(start, fin[d] + wait)
            start += sync * len(others)
            key = (round(start, 6), 0 if prod else 1, l)
            if best is None or key < best[0]:
                best = (key, l, start)
        lane[j] = best[1]
        fin[j] = best[2] + max(ms[j] - event, 0.002)
        free[best[1]] = fin[j] + 0.5 * sync
    return lane


def upward_rank(ms, deps, wait_share=0.0):
    n = len(ms)
    succ = [[] for _ in range(n)]
    for j in range(n):
        for d in deps[j]:
            succ[d].append(j)
    ru = [0.0] * n
    for j in reversed(range(n)):
        ru[j] = ms[j] + max([ru[s] + wait_share for s in succ[j]] or [0.0])
    return ru, succ


def refine(order, lane, ms, deps, L, wait, iters=4000, seed=0):
    import random
    rs = random.Random(seed)
    n = len(ms)
    succ = [[] for _ in range(n)]
    for j in range(n):
        for d in deps[j]:
            succ[d].append(j)
    best, _ = simulate(order, lane, ms, deps, L, wait)
    lane = list(lane)
    for it in range(iters):
        j = rs.randrange(n)
        # a chain move: j and its followers on the same lane while they are single-successor links
        chain = [j]
        if rs.random() < 0.5:
            k = j
            while True:
                nxt = [s for s in succ[k] if lane[s] == lane[j]]
                if len(nxt) != 1:
                    break
                k = nxt[0]
                chain.append(k)
                if len(chain) >= rs.choice((2, 4, 8, 16)):
                    break
        to = rs.randrange(L)
        if to == lane[j]:
            continue
        old = [lane[k] for k in chain]
        for k in chain:
            lane[k] = to
        m, _ = simulate(order, lane, ms, deps, L, wait)
        if m < best - 1e-9:
            best = m
        else:
            for k, o in zip(chain, old):
                lane[k] = o
    return lane, best


def cross_edges(lane, deps):
    return sum(1 for j in range(len(lane)) for d in deps[j] if lane[d] != lane[j])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dag')
    ap.add_argument('--lanes', type=int, default=4)
    ap.add_argument('--wait-us', type=float, default=16.0)
    ap.add_argument('--iters', type=int, default=4000)
    a = ap.parse_args()
    ops, ms, deps = load(a.dag)
    n = len(ms)
    W = a.wait_us * 1e-3
    prog = list(range(n))
    print('%d ops, single stream %.3f ms' % (n, sum(ms)))
    ru, succ = upward_rank(ms, deps)
    print('longest chain %.3f ms' % max(ru))
    for L in sorted({1, 2, a.lanes, 8}):
        g = plan(prog, ms, deps, L, W)
        mg, _ = simulate(prog, g, ms, deps, L, W)
        order_r = sorted(prog, key=lambda j: (-ru[j], j))
        r = plan(order_r, ms, deps, L, W)
        mr, _ = simulate(order_r, r, ms, deps, L, W)
        ru2, _ = upward_rank(ms, deps, W * 0.5)
        order_r2 = sorted(prog, key=lambda j: (-ru2[j], j))
        r2 = plan(order_r2, ms, deps, L, W)
        mr2, _ = simulate(order_r2, r2, ms, deps, L, W)
        line = '%d lanes: greedy %.3f ms (%d cross edges) | rank %.3f (%d) | rank+comm %.3f (%d)' % (
            L, mg, cross_edges(g, deps), mr, cross_edges(r, deps), mr2, cross_edges(r2, deps))
        if L > 1 and a.iters:
            f, mf = refine(prog, g, ms, deps, L, W, a.iters)
            f2, mf2 = refine(order_r, r, ms, deps, L, W, a.iters)
            line += ' | greedy+refine %.3f (%d) | rank+refine %.3f (%d)' % (mf, cross_edges(f, deps), mf2, cross_edges(f2, deps))
        print(line)
        if L > 1:
            free, _ = simulate(prog, g, ms, deps, L, 0.0, sync=0.0)
            print('         (greedy plan with free edges: %.3f ms)' % free)


if __name__ == '__main__':
    main()
