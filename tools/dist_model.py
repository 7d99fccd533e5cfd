"""Timing MODEL of the multi-rank elimination (no GPU needed; nothing here is measured on more than one GPU).

    python tools/dist_model.py [WORKLOAD] [--bw GB/s] [--nb 1024] [--period 1]

The tree comes from the host layer; the per-kernel parameters are the one-GPU measurements of this round (profiles/r02_*):
  * a rank-local level runs at `rate_low` TFLOP/s on its minimal flop count (levels 5-10 of Poisson 128^3: 44-54),
  * a front eliminated by one rank: max(chain, GEMM) with chain = `chain_us` per 32 columns next to a running GEMM and the GEMM at `rate_top`,
  * a front eliminated by its group (csrc/hs_dist.h): event simulation of the 1-D block-cyclic schedule with look-ahead -- the owner of block j+1
    applies block j to it, factors it (chain), fans it out (L part: (m - c0) * w * 8 bytes at `bw` GB/s to every peer at once), every rank applies
    block j to its own block columns and to its slice of the boundary columns; then the Schur slice and the gather of the slices.
It prints seconds per factorization for N = 1, 2, 4, 8 with the fronts above the cut on the first rank of their group (mode a) and on the whole
group (mode b).  DESIGN.md section 6 quotes its output."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def front_flops(ni, nb):
    return (2.0 / 3.0) * ni**3 + 2.0 * ni * ni * nb + 2.0 * ni * nb * nb


def one_rank_front(ni, nb, P):
    """max(chain, GEMM): the look-ahead hides whichever is shorter, block column by block column."""
    t, NB = 0.0, P["nb_la"]
    m = ni + nb
    for c0 in range(0, ni, NB):
        w = min(NB, ni - c0)
        chain = (w / 32.0) * P["chain_us"] * 1e-6
        gemm = 2.0 * (m - c0 - w) * (ni - c0 - w + nb) * w / (P["rate_top"] * 1e12)
        t += max(chain, gemm)
    return t + 2.0 * nb * nb * ni / (P["rate_schur"] * 1e12)


def group_front(ni, nb, g, P):
    """Event simulation of factor_front_dist (one front, g ranks)."""
    NB, per = P["nb"], P["period"]
    m = ni + nb
    nblk = (ni + NB - 1) // NB
    owner = lambda j: (j // per) % g
    rate = P["rate_k256"] if NB <= 256 else P["rate_top"]
    upd = lambda j, wk: 2.0 * (m - (j + 1) * NB) * wk * min(NB, ni - j * NB) / (rate * 1e12) if (j + 1) * NB < m else 0.0
    chain = lambda j: (min(NB, ni - j * NB) / 32.0) * P["chain_us"] * 1e-6
    msg = lambda j: (m - j * NB) * min(NB, ni - j * NB) * 8.0 / (P["bw"] * 1e9)
    main = [0.0] * g  # when each rank's main stream is free
    have = [[0.0] * g for _ in range(nblk)]
    fact_done = chain(0)
    link_free = 0.0
    for r in range(g):
        have[0][r] = fact_done if r == owner(0) else fact_done + msg(0)
    link_free = fact_done + msg(0)
    bper = (nb + g - 1) // g
    for j in range(nblk):
        for r in range(g):
            t = max(main[r], have[j][r])
            if j + 1 < nblk and owner(j + 1) == r:  # look-ahead: my next block first, factored on the side stream
                t += upd(j, min(NB, ni - (j + 1) * NB))
                fd = t + chain(j + 1)
                start_send = max(fd, link_free)
                link_free = start_send + msg(j + 1)
                for q in range(g):
                    have[j + 1][q] = fd if q == r else link_free
            for k in range(j + 1, nblk):  # my other block columns
                if owner(k) == r and not (k == j + 1):
                    t += upd(j, min(NB, ni - k * NB))
            if nb > 0:
                t += upd(j, bper)
            main[r] = t
    t_end = max(main)
    if nb > 0:
        t_end += 2.0 * nb * bper * ni / (P["rate_schur"] * 1e12)
        t_end += (ni + nb) * bper * 8.0 / (P["bw"] * 1e9)
    return t_end


def group_front_analytic(ni, nb, pr, pc, P):
    """One front on a pr x pc process grid, block-cyclic with blocks of `nb` in both directions (pr = 1: the 1-D column distribution that
    csrc/hs_dist.h implements), block column by block column with one block column of look-ahead: the step costs the longer of
      * the trailing update of one rank:  2 (m_rem / pr) (c_rem / pc) w / rate, and
      * the critical path through the NEXT panel: its own update on its owners, its chain, and the messages that make it usable --
        the L panel along the process row ((m_rem / pr) w 8 bytes to every peer at once) and, in 2-D, the U panel along the process column.
    The chain of a 32-column panel costs `chain_us` on one rank (measured next to a running GEMM).  A COLUMN OF RANKS (pr > 1) adds to every
    32-column panel the pivot reduction over pr ranks (ceil(log2 pr) exchanges of a 32 x 32 candidate block: the tournament of
    kernels_panel.hip across ranks), one exchange of the pivot rows and the broadcast of the factored diagonal block before the rows below
    can be divided: (2 + ceil(log2 pr)) latencies of `lat_us` on the chain.  It removes nothing from it: the eliminations of a panel are serial."""
    import math

    NB = P["nb"]
    m = ni + nb
    rate = (P["rate_k256"] if NB <= 256 else P["rate_top"]) * 1e12
    bw = P["bw"] * 1e9
    extra = (2 + math.ceil(math.log2(pr))) * P["lat_us"] * 1e-6 if pr > 1 else 0.0
    t = 0.0
    for c0 in range(0, ni, NB):
        w = min(NB, ni - c0)
        mrem, crem = m - c0 - w, ni + nb - c0 - w
        chain = (w / 32.0) * (P["chain_us"] * 1e-6 + extra)
        upd = 2.0 * (mrem / pr) * (crem / pc) * w / rate
        nxt = 2.0 * (mrem / pr) * w * w / rate
        msg = (mrem / pr) * w * 8.0 / bw * (1 if pc > 1 else 0) + (w * (crem / pc) * 8.0 / bw if pr > 1 else 0.0)
        t += max(upd, nxt + chain + msg)
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="poisson3d_128")
    ap.add_argument("--bw", type=float, default=45.0, help="point-to-point GB/s per link (what bench.py reports as comm_p2p_GBps)")
    ap.add_argument("--nb", type=int, default=1024)
    ap.add_argument("--period", type=int, default=1)
    ap.add_argument("--chain-us", type=float, default=330.0, help="panel chain per 32 columns of a lone front next to a running GEMM (root of 128^3 at the round-2 head)")
    ap.add_argument("--lat-us", type=float, default=10.0, help="latency of one small inter-GPU exchange on the panel chain of a 2-D layout (pivot reduction, row swap, diagonal block)")
    args = ap.parse_args()
    import hsamd

    hs = hsamd.load()
    A, b, nd = hs.problems.make_problem(args.workload)
    nd, nd_loc = hs.symfact(nd)
    levels = {}

    def walk(x, lv):
        levels.setdefault(lv, []).append((len(x.int), len(x.bnd)))
        if x.left is not None:
            walk(x.left, lv + 1)
            walk(x.right, lv + 1)

    walk(nd, 1)
    P = dict(nb=args.nb, period=args.period, bw=args.bw, chain_us=args.chain_us, nb_la=1024, rate_top=58.0, rate_k256=40.0, rate_schur=64.0, rate_low=48.0, lat_us=args.lat_us)
    print(f"{args.workload}: {len(levels)} levels; model parameters {P}")
    print(f"{'N':>2} {'subtrees':>9} {'top, first rank (a)':>20} {'top, groups (b)':>16} {'total a':>8} {'total b':>8} {'speed-up a / b':>15}")
    base = None
    for N in (1, 2, 4, 8):
        p = N.bit_length() - 1
        cut = p + 1
        # levels >= cut are rank-local: the fronts of a level are dealt over the ranks; big single fronts follow the one-rank model
        t_sub = 0.0
        for lv, fr in levels.items():
            if lv < cut:
                continue
            per_rank = fr[: max(len(fr) // N, 1)]
            if len(per_rank) == 1:
                t_sub += one_rank_front(*per_rank[0], P)
            else:  # a batch: its flops at the measured rate of such levels (57 TF/s for 2-8 fronts, 48 below), never shorter than the longest chain
                rate = 57.0 if len(per_rank) <= 8 else P["rate_low"]
                t_sub += max(sum(front_flops(ni, nb_) for ni, nb_ in per_rank) / (rate * 1e12), max(ni for ni, _ in per_rank) / 32.0 * 150e-6)
        t_a = sum(one_rank_front(*levels[lv][0], P) for lv in range(1, cut))
        t_b = sum(group_front(*levels[lv][0], N >> (lv - 1), P) for lv in range(1, cut))
        ta, tb = t_sub + t_a, t_sub + t_b
        if base is None:
            base = ta
        print(f"{N:>2} {t_sub:9.3f} {t_a:20.3f} {t_b:16.3f} {ta:8.3f} {tb:8.3f} {base / ta:7.2f} / {base / tb:5.2f}")
    # 1-D against 2-D for the fronts above the cut ("justify 1-D vs 2-D with the model including a column-of-ranks panel"): the same
    # analytic step model for both layouts, so that the comparison does not depend on the event simulation above
    print("\nfronts above the cut, analytic step model, seconds (pr x pc process grid; 1 x g = what hs_dist.h implements):")
    for lat in sorted({0.0, args.lat_us, 2 * args.lat_us}):
        P2 = dict(P, lat_us=lat)
        for N in (2, 4, 8):
            p = N.bit_length() - 1
            rows = []
            grids = [(pr, N // pr) for pr in (1, 2, 4, 8) if pr <= N]
            for pr, pc in grids:
                tt = 0.0
                for lv in range(1, p + 1):
                    g = N >> (lv - 1)  # ranks of the group that owns a front of this level
                    gpr = min(pr, g)
                    tt += group_front_analytic(*levels[lv][0], gpr, max(g // gpr, 1), P2)
                rows.append(f"{pr}x{pc}: {tt:.3f}")
            print(f"  lat {lat:4.0f} us, N = {N}:  " + "   ".join(rows))
    ni1, nb1 = levels[1][0]
    steps = (ni1 + P['nb'] - 1) // P['nb']
    print(f"  (the root alone: {steps} block columns x {P['nb'] // 32} panels x {P['chain_us']:.0f} us = {steps * (P['nb'] // 32) * P['chain_us'] * 1e-6:.3f} s of chain on ANY layout; "
          f"its trailing updates at {P['rate_top']:.0f} TF/s on 8 ranks: {front_flops(ni1, nb1) / (P['rate_top'] * 1e12) / 8:.3f} s)")


if __name__ == "__main__":
    main()
