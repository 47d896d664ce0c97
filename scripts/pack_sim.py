#!/usr/bin/env python3
"""Offline: the device's forest packing (pair_kernels.hip, packing_role) restated in numpy over measured subtree shapes
(profiles/r05/shapes_<name>.npz, dumped by scripts/shape_dump.py on the GPU box), to try packing rules before they are built.  CPU only."""
import sys
import numpy as np

NCAP, ACAP, SLOTS = 432, 64, 1280


def items_of(nodes, atoms, split_nodes, max_parts, fit=True, share=0.9):
    l2 = np.maximum(atoms - 1, 0)
    deep = np.maximum(nodes - 1 - l2, 0)
    p = np.minimum(max_parts, 1 + (nodes >= split_nodes) + (nodes >= 2 * split_nodes) + (nodes >= 3 * split_nodes))
    if fit:
        fit_nodes = 0.85 * (share / 0.9) * NCAP
        need, room = 1.35 * deep, fit_nodes - (1 + l2)
        pf = np.where(need <= room, 1, np.where(need <= 2 * room, 2, np.where(need <= 3 * room, 3, 4)))
        p = np.maximum(p, pf)
    inv_tn, inv_ta = 1024 / (share * NCAP), 1024 / (share * ACAP)
    item_nodes = (1 + l2) + (deep + p - 1) / p
    w = np.minimum(np.maximum(np.maximum(item_nodes * inv_tn, atoms * inv_ta), 128), 2047).astype(int)
    return np.repeat(w, p), np.repeat(item_nodes, p), np.repeat(atoms, p)


def current_rule(w):
    w = np.sort(w)[::-1]
    n = len(w)
    na, nab = int((w > 768).sum()), int((w > 512).sum())
    nb = nab - na
    light = int((w <= 256).sum())
    npair = min(nb, light)
    nc = n - nab - npair
    wc = w[nab:].sum()
    fmin = max(-(-wc // 870), -(-nc // 8))
    rounds = -(-(nab + fmin) // SLOTS)
    fs = min(nc, max(fmin, rounds * SLOTS - nab))
    return dict(items=n, A=na, B=nb, pairs=npair, C_items=nc, fmin=int(fmin), forests_min=int(nab + fmin), forests=int(nab + fs), rounds=int(rounds), total_weight=float(w.sum() / 1024))


def first_fit_decreasing(w, wn, wa, cap_w=1024, max_roots=8):
    """reference point: what a real bin packing reaches (sequential FFD on the weight)"""
    order = np.argsort(-w)
    bins, roots = [], []
    for i in order:
        for b in range(len(bins)):
            if bins[b] + w[i] <= cap_w and roots[b] < max_roots:
                bins[b] += w[i]; roots[b] += 1
                break
        else:
            bins.append(w[i]); roots.append(1)
    return len(bins)


for name in sys.argv[1:]:
    d = np.load(f"profiles/r05/shapes_{name}.npz")
    nodes, atoms = d["nodes"], d["atoms"]
    nh = len(nodes)
    roomy = 2 * nh <= SLOTS
    split_nodes = 48 if roomy else int(0.55 * 0.9 * NCAP)
    max_parts = 4 if roomy else 3
    w, wn, wa = items_of(nodes, atoms, split_nodes, max_parts)
    print(name, "subtrees", nh, "engine forests", int(d["forests"]), current_rule(w))
    print("   weight histogram (1/8 units):", np.histogram(w, bins=[0, 129, 256, 384, 512, 640, 768, 896, 1024, 2048])[0].tolist())
    print("   node-bound vs atom-bound items:", int((wn / (0.9 * NCAP) >= wa / (0.9 * ACAP)).sum()), int((wn / (0.9 * NCAP) < wa / (0.9 * ACAP)).sum()))
    print("   FFD bins at fill target 0.9:", first_fit_decreasing(w, wn, wa))


def rounds_rule(w, F, max_roots=8, cap=1024, quantum=4):
    """Candidate rule: the F heaviest items lead a forest each; then rounds: every forest whose sum still takes the HEAVIEST
    item left (a bound that decouples the forests' decisions) is open, the open forests take the next items in
    serpentine order (heaviest open forest <- lightest item of the round).  Weights rounded UP to the bins' quantum as the
    device knows them.  Returns (items placed, sums)."""
    w = np.sort(w)[::-1]
    wq = ((w + quantum - 1) // quantum) * quantum
    n = len(w)
    sums = wq[:F].astype(np.int64).copy()
    roots = np.ones(F, dtype=int)
    base = F
    while base < n:
        open_ = (sums + wq[base] <= cap) & (roots < max_roots)
        k = int(open_.sum())
        if k == 0:
            return base, sums
        take = min(k, n - base)
        idx = np.flatnonzero(open_)  # forests in descending leader weight
        # heaviest open forest gets the LIGHTEST item of this round's batch
        batch = np.arange(base, base + take)
        fl = idx[:take] if take == k else idx[:take]
        sums[fl] += wq[batch[::-1]]
        roots[fl] += 1
        base += take
    return base, sums


if __name__ == "__main__":
    for name in sys.argv[1:]:
        d = np.load(f"profiles/r05/shapes_{name}.npz")
        nodes, atoms = d["nodes"], d["atoms"]
        nh = len(nodes)
        roomy = 2 * nh <= SLOTS
        w, wn, wa = items_of(nodes, atoms, 48 if roomy else int(0.55 * 0.9 * NCAP), 4 if roomy else 3)
        for F in (1100, 1150, 1200, 1250, 1280, 1400, 2560):
            placed, sums = rounds_rule(w, F)
            print(f"   rounds rule F={F}: placed {placed}/{len(w)}  fill mean {sums.mean() / 1024:.3f} max {sums.max() / 1024:.3f} min {sums.min() / 1024:.3f}")
