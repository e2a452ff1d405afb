#!/usr/bin/env python3
"""Host model of a two-level TRSV schedule on the block DAG of the shell-like L factors (no GPU).
chunk = CB consecutive blocks (natural order) owned by one workgroup of W wavefronts; inside a chunk the blocks are levelled on
their in-chunk dependencies only, a local level = one or more steps of <= 64 blocks, step s goes to wavefront s % W.
Costs (us) from the traces in profiles/r5/trsv_experiments.txt: work per step, LDS hand-off inside the workgroup, hand-off
through HBM between workgroups, latency of the values of a step (issued when the wavefront finished its previous step)."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import standins

def dag(variant, width=600, n=1508065):
    nb = n // 5
    bi = np.arange(nb, dtype=np.int64)
    j = bi % width
    offs = np.array([-width - 1, -width, -1, 0, 1, width, width + 1], dtype=np.int64)
    dj = np.array([-1, 0, -1, 0, 1, 0, 1], dtype=np.int64)
    bc = bi[:, None] + offs[None, :]
    ok = (bc >= 0) & (bc < nb) & ((j[:, None] + dj[None, :]) >= 0) & ((j[:, None] + dj[None, :]) < width)
    if variant == "unstructured":
        bc, ok = standins._unstructure(bc, ok, 313)
    low = ok & (bc < bi[:, None])
    return nb, bc, low

def simulate(nb, bc, low, CB, W, work, local, remote, vals):
    chunk = np.arange(nb) // CB
    K = bc.shape[1]
    ll = np.zeros(nb, dtype=np.int64)
    for b in range(nb):  # natural order: dependencies first
        m = 0
        for k in range(K):
            if low[b, k]:
                d = bc[b, k]
                if chunk[d] == chunk[b] and ll[d] + 1 > m:
                    m = ll[d] + 1
        ll[b] = m
    fin = np.zeros(nb)
    nsteps = 0
    total = 0.0
    for c in range((nb + CB - 1) // CB):
        b0, b1 = c * CB, min(nb, (c + 1) * CB)
        order = b0 + np.argsort(ll[b0:b1], kind="stable")
        lv = ll[order]
        # steps: runs of equal local level, cut at 64
        steps = []
        s0 = 0
        while s0 < len(order):
            s1 = s0
            while s1 < len(order) and lv[s1] == lv[s0] and s1 - s0 < 64:
                s1 += 1
            steps.append(order[s0:s1]); s0 = s1
        wfree = [0.0] * W
        for s, blks in enumerate(steps):
            w = s % W
            t_issue = wfree[w]
            t = t_issue + vals
            for b in blks:
                for k in range(K):
                    if low[b, k]:
                        d = bc[b, k]
                        r = fin[d] + (local if chunk[d] == c else remote)
                        if r > t:
                            t = r
            f = t + work
            fin[blks] = f
            wfree[w] = f
            if f > total:
                total = f
        nsteps += len(steps)
    return total, nsteps, int(ll.max()) + 1

if __name__ == "__main__":
    variant = sys.argv[1] if len(sys.argv) > 1 else "structured"
    nb, bc, low = dag(variant)
    work = 0.40 if variant == "structured" else 0.76
    for CB in (512, 1024, 2048, 4096):
        for W in (4, 8, 16):
            t, ns, ml = simulate(nb, bc, low, CB, W, work, 0.06, float(os.environ.get("REMOTE", "1.0")), 2.0)
            print(json.dumps({"variant": variant, "chunk_blocks": CB, "waves": W, "ms": round(t / 1e3, 3), "steps": ns,
                              "max_local_levels": ml}), flush=True)
