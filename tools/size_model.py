"""Per-rank device memory of a factorization, per flow, WITHOUT building the problem (host only, no GPU, seconds even for 256^3).

    python tools/size_model.py [--grid 256] [--kind helmholtz|poisson] [--ranks 8] [--nmax 4096] [--swlevel 6] [--alpha 9.8] [--leaf 512]

The (ni, nb) of every front of the geometric nested dissection of an N^3 grid follow from the boxes alone (the rule of
hierarchicalsolvers.jl_amd/problems.py:grid_nested_dissection -- longest axis, first on ties, split in the middle; bnd(B) = cells of B with a
neighbour outside B; int(parent) = (bnd(l) | bnd(r)) - bnd(parent)); tests/test_host.py checks this restatement against the real tree.
What is EXACT: the dense factor bytes sum((ldl*ni + ni*nb) + inverse blocks) * sizeof(T) per rank and flow (the library's own layout,
hs_api.hip analyze_impl).  What is MODELLED: the bytes of the compressed fronts, from a rank model r(s) = min(s/2, alpha*sqrt(s)) for a cluster
of s DOFs of a 2-D separator (alpha is calibrated on one-GPU runs: Poisson 128^3 at 1e-4: top-level rank 886 for clusters of 8,192 -> 9.8;
Helmholtz 112^3 at 1e-4: see BASELINE.md).  Flows:
  a      exact, a front above the rank cut on the first rank of its group (dist_top = 0)
  b      exact, fronts above the cut REPLICATED on their group (dist_top = 1, hs_dist.h)
  b1d    exact, fronts above the cut stored once, 1-D block-cyclic over the group (what hs_dist.h would need for 256^3: not built)
  mfK    hs_options.mf = K over ranks (the joins ship HSS generators): levels <= swlevel compressed
"""
import argparse
import math


def tree_sizes(N, nmax, dims=3):
    """[(level, ni, nb, box)] in post-order plus child links, from the boxes alone."""
    shape = (N,) * dims if isinstance(N, int) else tuple(N)
    d = len(shape)
    nodes = []

    def nbnd(lo, hi):
        vol, inner = 1, 1
        for a in range(d):
            e = hi[a] - lo[a]
            vol *= e
            inner *= max(e - (1 if lo[a] > 0 else 0) - (1 if hi[a] < shape[a] else 0), 0)
        return vol - inner, vol

    def build(lo, hi, level):
        ext = [hi[a] - lo[a] for a in range(d)]
        nb, vol = nbnd(lo, hi)
        if vol <= nmax or max(ext) < 2:
            nodes.append(dict(level=level, ni=vol - nb, nb=nb, left=-1, right=-1))
            return len(nodes) - 1
        ax = ext.index(max(ext))
        mid = lo[ax] + ext[ax] // 2
        hl, lr = list(hi), list(lo)
        hl[ax] = mid
        lr[ax] = mid
        li = build(lo, tuple(hl), level + 1)
        ri = build(tuple(lr), hi, level + 1)
        ni = nodes[li]["nb"] + nodes[ri]["nb"] - nb
        nodes.append(dict(level=level, ni=ni, nb=nb, left=li, right=ri))
        return len(nodes) - 1

    build(tuple([0] * d), shape, 1)
    return nodes


def owners(nodes, nranks):
    """rank range [lo, lo+cnt) below each node (hs_api.hip build_plan)."""
    n = len(nodes)
    lo, cnt = [0] * n, [nranks] * n
    for i in range(n - 1, -1, -1):
        x = nodes[i]
        if x["left"] >= 0:
            if cnt[i] > 1:
                lo[x["left"]], cnt[x["left"]] = lo[i], cnt[i] // 2
                lo[x["right"]], cnt[x["right"]] = lo[i] + cnt[i] // 2, cnt[i] // 2
            else:
                lo[x["left"]] = lo[x["right"]] = lo[i]
                cnt[x["left"]] = cnt[x["right"]] = 1
    return lo, cnt


def dense_front_elems(ni, nb):
    m = ni + nb
    return m * ni + ni * nb + 2 * ((ni + 31) // 32) * 32 * 32 + 2 * ((ni + 255) // 256) * 256 * 256


def rank_of(s, alpha):
    return int(min(s / 2.0, alpha * math.sqrt(max(s, 1))))


def hss_elems(n, leaf, alpha, factored=False):
    """Generators of an HSS matrix of order n over a bisection tree with leaves <= leaf: leaf blocks, interpolation matrices, couplings."""
    if n <= 0:
        return 0

    def rec(s):
        if s <= leaf:
            r = rank_of(s, alpha)
            return s * s + (s - r) * r, r
        a, ra = rec((s + 1) // 2)
        b, rb = rec(s // 2)
        r = rank_of(s, alpha)
        m = ra + rb
        return a + b + 2 * ra * rb + max(m - r, 0) * min(r, m), min(r, m)

    e, _ = rec(n)
    return int(e * (2.0 if factored else 1.0))  # the ULV-type elimination keeps a front per node: about as much again


def model(N, kind, nranks, nmax, swlevel, alpha, leaf, verbose=False):
    sz = 16 if kind == "helmholtz" else 8
    nodes = tree_sizes(N, nmax)
    lo, cnt = owners(nodes, nranks)
    cut = int(math.log2(nranks)) + 1
    flows = ["a", "b", "b1d", "mf1", "mf2", "mf3"]
    per = {f: [0.0] * nranks for f in flows}      # persistent bytes per rank
    trans = {f: [0.0] * nranks for f in flows}    # largest transient (Schur scratch / front being compressed) per rank
    scratch = {f: {} for f in flows}              # (rank, level) -> bytes of the scratch fronts of the compressed fronts of that level (hs_api.hip NodeH::cfront)
    for i, x in enumerate(nodes):
        ni, nb, lv = x["ni"], x["nb"], x["level"]
        dense = dense_front_elems(ni, nb) * sz
        sb = nb * nb * sz
        group = range(lo[i], lo[i] + cnt[i])
        first = lo[i]
        # exact flows
        per["a"][first] += dense
        trans["a"][first] = max(trans["a"][first], 2 * sb)  # the level's S and the children's (ping-pong scratch)
        for r in group:
            per["b"][r] += dense
            trans["b"][r] = max(trans["b"][r], 2 * sb)
            per["b1d"][r] += dense / cnt[i]
            trans["b1d"][r] = max(trans["b1d"][r], 2 * sb / cnt[i] + 3 * (ni + nb) * 512 * sz)
        # matrix-free flows: levels <= swlevel with a boundary are compressed; the root (nb = 0) takes its children's HSS blocks
        flagged = lv <= swlevel and nb > leaf and lv >= 2
        kids_flagged = x["left"] >= 0 and all(nodes[c]["level"] <= swlevel and nodes[c]["nb"] > leaf for c in (x["left"], x["right"]))
        rL = rR = 0
        if nb > 0:
            rL = rR = min(rank_of(nb // 2, alpha) * 2 + (nb // 2 if False else 0), min(ni, nb))
        for K in (1, 2, 3):
            f = "mf%d" % K
            if not (flagged or kids_flagged):
                per[f][first] += dense
                trans[f][first] = max(trans[f][first], 2 * sb)
                continue
            if not kids_flagged:  # transition front: eliminated on a SCRATCH front, keeps the compact LU of Aii + low-rank L, R; S compressed to HSS afterwards
                per[f][first] += (dense_front_elems(ni, 0) + (ni + nb) * (rL + rR)) * sz
                key = (first, lv)
                scratch[f][key] = scratch[f].get(key, 0.0) + ((ni + nb) * ni + ni * nb) * sz
                trans[f][first] = max(trans[f][first], 2 * sb + hss_elems(nb, leaf, alpha) * sz)
                continue
            lr = (ni + nb) * (rL + rR) * sz  # C_L, Z_L, W = D^-1 C_R, Z_R
            if K == 1:
                d = dense_front_elems(ni, max(rL, rR)) * sz
            elif K == 2:
                d = hss_elems(ni, leaf, alpha * 1.5, factored=True) * sz  # D is compressed at tol * 1e-2: higher ranks
            else:
                n1 = ni // 2
                d = (hss_elems(n1, leaf, alpha, factored=True) + hss_elems(ni - n1, leaf, alpha * 1.5, factored=True)) * sz + n1 * (ni - n1) * sz  # + A11^-1 A12 dense
            s_h = hss_elems(nb, leaf, alpha) * sz
            per[f][first] += d + lr
            kids = sum(hss_elems(nodes[c]["nb"], leaf, alpha) for c in (x["left"], x["right"])) * sz
            trans[f][first] = max(trans[f][first], s_h + kids + 2 * ni * 4096 * sz)
    out = {}
    for f in flows:
        for r in range(nranks):
            trans[f][r] += max([v for (rr, _), v in scratch[f].items() if rr == r], default=0.0)
        tot = [per[f][r] + trans[f][r] for r in range(nranks)]
        out[f] = dict(max_GiB=max(tot) / 2**30, min_GiB=min(tot) / 2**30, persistent_max_GiB=max(per[f]) / 2**30)
    top = sorted(((x["level"], x["ni"], x["nb"]) for x in nodes if x["level"] <= 5), key=lambda t: t[0])
    seen, tops = set(), []
    for lv, ni, nb in top:
        if lv not in seen:
            seen.add(lv)
            tops.append((lv, ni, nb))
    return out, dict(n=N**3, nnodes=len(nodes), depth=max(x["level"] for x in nodes), cut=cut, tops=tops, sizeof=sz)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--kind", default="helmholtz")
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--nmax", type=int, default=4096)
    ap.add_argument("--swlevel", type=int, default=7)
    ap.add_argument("--alpha", type=float, default=9.8)
    ap.add_argument("--leaf", type=int, default=512)
    ap.add_argument("--hbm-gb", type=float, default=288.0)
    ap.add_argument("--scan", action="store_true", help="largest cube (multiples of 16) that fits 0.9 x HBM per flow")
    a = ap.parse_args()
    budget = 0.9 * a.hbm_gb * 1e9 / 2**30
    out, info = model(a.grid, a.kind, a.ranks, a.nmax, a.swlevel, a.alpha, a.leaf)
    print(f"{a.kind} {a.grid}^3: n = {info['n']:,}, {info['nnodes']} fronts, depth {info['depth']}, rank cut at level {info['cut']} ({a.ranks} ranks), sizeof(T) = {info['sizeof']}")
    print("top fronts (level, ni, nb):", info["tops"])
    print(f"budget per rank: 0.9 x {a.hbm_gb:.0f} GB = {budget:.0f} GiB;  rank model r(s) = min(s/2, {a.alpha} sqrt(s)), HSS leaves {a.leaf}, swlevel {a.swlevel}")
    print(f"{'flow':6s} {'busiest rank GiB':>18s} {'(persistent)':>14s} {'idlest rank GiB':>16s}  fits")
    for f, v in out.items():
        print(f"{f:6s} {v['max_GiB']:18.1f} {v['persistent_max_GiB']:14.1f} {v['min_GiB']:16.1f}  {'yes' if v['max_GiB'] <= budget else 'NO'}")
    if a.scan:
        print("largest cube per flow (multiples of 16):")
        for f in out:
            best = None
            for N in range(64, 321, 16):
                o, _ = model(N, a.kind, a.ranks, a.nmax, a.swlevel, a.alpha, a.leaf)
                if o[f]["max_GiB"] <= budget:
                    best = N
            print(f"  {f:6s} {best}^3" if best else f"  {f:6s} none")


if __name__ == "__main__":
    main()
