#!/usr/bin/env python3
"""SIMT cost of the control phases of the go() kernel under different slot-queue keys, from the pc trace of the host
instantiation (tools/memprof/run.py writes /tmp/pctrace.bin).  A wave pops <= 64 slots from the longest queue, runs the
primitive, then every lane walks its pcs until the next primitive request: in lockstep, step k costs the number of DISTINCT pcs
the lanes are at.  usage: simt.py [trace] [slots]"""
import sys
from collections import defaultdict, deque

import numpy as np

path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/pctrace.bin"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
t = np.fromfile(path, dtype=np.uint16)
reads, cur, phase = [], [], []
for v in t:
    v = int(v)
    if v & 0x8000:
        cur.append((tuple(phase), v & 0x7fff)); phase = []
        if (v & 0x7fff) == 0:
            reads.append(cur); cur = []
    else:
        phase.append(v)
print("reads %d, trips/read %.2f, pcs/read %.1f" % (len(reads), sum(len(r) for r in reads) / len(reads), sum(len(p[0]) for r in reads for p in r) / len(reads)))


def cost(paths):
    c, k = 0, 0
    while True:
        s = {p[k] for p in paths if len(p) > k}
        if not s:
            return c
        c += len(s); k += 1


def simulate(keyfn, name):
    queues = defaultdict(deque)
    nxt, free = 0, S
    tot_cost = tot_trips = tot_lanes = lane_steps = 0
    while True:
        best = max(queues.items(), key=lambda kv: len(kv[1]), default=(None, ()))
        bestc = len(best[1])
        more = nxt < len(reads)
        if more and free > 0 and (bestc < 64 or free >= S // 4):
            n = min(64, free, len(reads) - nxt)
            batch = [(i, 0) for i in range(nxt, nxt + n)]
            nxt += n; free -= n
        elif bestc == 0:
            if not more:
                break
            continue
        else:
            q = best[1]
            batch = [q.popleft() for _ in range(min(64, len(q)))]
        paths = [reads[i][ph][0] for i, ph in batch]
        tot_cost += cost(paths); tot_trips += 1; tot_lanes += len(batch); lane_steps += sum(len(p) for p in paths)
        for i, ph in batch:
            op = reads[i][ph][1]
            if op == 0:
                free += 1
            else:
                queues[keyfn(i, ph, op)].append((i, ph + 1))
    print("%-34s wave-trips %7d  lanes/trip %5.1f  control cost (distinct pc bodies) %8d = %6.2f per trip, %5.2f per read; ideal (1 lane alone) %5.2f per read" % (
        name, tot_trips, tot_lanes / tot_trips, tot_cost, tot_cost / tot_trips, tot_cost / len(reads), lane_steps / len(reads)))


simulate(lambda i, ph, op: op, "queue per primitive")
simulate(lambda i, ph, op: (op, reads[i][ph + 1][0][0]), "queue per (primitive, resume pc)")
simulate(lambda i, ph, op: (op, reads[i][ph + 1][0]), "queue per whole pc path (bound)")
