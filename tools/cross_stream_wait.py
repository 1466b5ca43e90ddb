#!/usr/bin/env python
"""What a cross-stream dependency costs on this box (GPU only): N tiny kernels in one stream vs the same N kernels
alternating between two streams with an event record + wait between every pair."""
import os, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')   # this probe's own setting (two streams only); the package leaves the variable alone since round 6
import torch

x = torch.zeros(1024, device='cuda')
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 200


def serial():
    with torch.cuda.stream(s1):
        for _ in range(N):
            x.add_(1.0)


def pingpong():
    evs = [torch.cuda.Event() for _ in range(N)]
    for i in range(N):
        s = s1 if i % 2 == 0 else s2
        if i:
            s.wait_event(evs[i - 1])
        with torch.cuda.stream(s):
            x.add_(1.0)
        evs[i].record(s)


def fan(k):
    """one producer kernel, k consumers on k streams wait for it, then the producer stream waits for all k (fork / join)"""
    ss = [torch.cuda.Stream() for _ in range(k)]
    def run():
        for _ in range(N // 4):
            e = torch.cuda.Event()
            with torch.cuda.stream(s1):
                x.add_(1.0)
            e.record(s1)
            for s in ss:
                s.wait_event(e)
                with torch.cuda.stream(s):
                    x.add_(1.0)
                j = torch.cuda.Event()
                j.record(s)
                s1.wait_event(j)
    return run


for name, fn, n in (('serial, one stream', serial, N), ('ping-pong, two streams', pingpong, N), ('fork/join x3', fan(3), N // 4)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        torch.cuda._sleep(20_000_000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s1)
        fn()
        s1.wait_stream(s2)
        e1.record(s1)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print('%-28s %.1f us per step (%d steps)' % (name, 1e3 * best / n, n))
