"""Generates visualcloze_amd/csrc/attention64_sched.h: which filler instructions sit in which MFMA gap of the bounded-softmax
tile of attention64.hip (`attn64_kernel<true>`).

The tile is two phases of 32 MFMAs (one wave per SIMD, 32 cycles per MFMA = 7 issue slots of ~4.6 cycles: the MFMA itself and
six others).  Everything else the tile needs is a TOKEN with a measured issue price (kernel header / tools/ubench/a64_gap):

    E0 / E1 (k)   the two v_exp_f32 of probability pair k                          2 each
    A0 / A1 / CV (k)   the two row-sum adds and the bf16 pack of pair k             1 each
    RV / RV2 / RK   one ds_read_b128 (V^T fragment / K fragment)                   3.2
    DMA(i)   one LDS-DMA piece (offset add, M0, global_load_lds)                   ~8

    per tile: 32 pairs x 7 + 32 reads x 3.2 + 8 pieces x 8 = 390 slots for 64 gaps of 6  -> the tile is over-subscribed by a few
    per cent whatever the order; what the order decides is whether some gaps carry 8-13 slots while others carry 3 (round 5's
    stream: 24 phase-A gaps of 7+, eight DMA gaps of 11-13, sixteen gaps of 3.2).

This script deals the tokens out so that every gap carries about the same price, under the data dependences of the tile:

  phase A(t):  S(t+1) = K(t+1) . Q^T (32 MFMAs: 4 chains of 8)  ||  20 pairs of P(t) (8..15, 20..31), the four V^T(t)
               fragments of the first 16-key step, the 4 V^T(t+2) LDS-DMA pieces and one K(t+4) piece
  phase B(t):  O += V^T(t) . P(t)^T, 16-key step s major (MFMA j: s = j >> 3, dt = (j >> 1) & 3, qb = j & 1)  ||  12 pairs
               of P(t+1) written straight into the P registers that the s-major order has retired (pairs 0..3 -> P[0][0] after
               MFMA 6, 16..19 -> P[1][0] after MFMA 7, 4..7 -> P[0][1] after MFMA 14), V^T(t) fragment (dt, s + 1) into the
               register of (dt, s) as it retires (a ring of FOUR fragments: 16 registers), the 16 K(t+2) fragments, 3 K(t+4) pieces
  waits:       lgkmcnt is in order for LDS reads, so each first use of a fragment waits with the COUNT of reads issued after it.

    python tools/gen_a64_sched.py            # rewrites attention64_sched.h (ring of 4 fragment registers), prints the per-gap prices
    python tools/gen_a64_sched.py --ring8    # attention64_sched8.h: ring of 8 (two 16-key steps; -DVC_A64_RING8 builds use it)
"""
import os
import sys

E0, E1, A0, A1, CV, RV, RV2, RK, DMA, WAIT = range(10)
NAMES = ["e", "E", "a", "A", "C", "RV", "RV2_", "RK", "DMA", "WAIT"]
COST = {E0: 2.0, E1: 2.0, A0: 1.0, A1: 1.0, CV: 1.0, RV: 3.2, RV2: 3.2, RK: 3.2, DMA: 8.0, WAIT: 0.2}
EARLY_PAIRS = [0, 1, 2, 3, 16, 17, 18, 19, 4, 5, 6, 7]          # done in phase B of the previous tile
LATE_PAIRS = [k for k in list(range(8, 16)) + list(range(20, 32))]
assert sorted(EARLY_PAIRS + LATE_PAIRS) == list(range(32))


def gap_cycles(price):
    return max(32.0, 4.6 * (1.0 + price))


# instructions a token turns into (what `make audit64` counts between two MFMAs): a DMA piece is offset add (+ clamp for K),
# M0 add, wait state, load
NINSN = {E0: 1, E1: 1, A0: 1, A1: 1, CV: 1, RV: 1, RV2: 1, RK: 1, DMA: 5, WAIT: 1}
MAX_INSN = 6            # per gap, hard (MI355X_MICROARCH.md: <= 5 single-issue fillers hidden per 32x32x16 MFMA, placed)


def deal(tokens, n_gaps=32):
    """tokens: dicts(kind, a, b, earliest, order).  The tokens keep their `order`; the sequence is cut into n_gaps contiguous
    groups (dynamic programme) so that the estimated cycles - sum over gaps of max(32, 4.6 (1 + price)) - are minimal, no
    token lands before its `earliest` gap, and among equal-cost cuts the prices are as even as possible."""
    toks = sorted(tokens, key=lambda t: t["order"])
    n = len(toks)
    pre, cnt = [0.0], [0]
    for t in toks:
        pre.append(pre[-1] + COST[t["kind"]])
        cnt.append(cnt[-1] + NINSN[t["kind"]])
    INF = 1e18
    dp = [[INF] * (n + 1) for _ in range(n_gaps + 1)]
    arg = [[-1] * (n + 1) for _ in range(n_gaps + 1)]
    dp[0][0] = 0.0
    for g in range(n_gaps):
        for i in range(n + 1):
            if dp[g][i] >= INF:
                continue
            j = i
            while True:
                price = pre[j] - pre[i]
                c = dp[g][i] + gap_cycles(price) + 0.02 * price * price
                if cnt[j] - cnt[i] > MAX_INSN and j > i + 1:       # (a single token may exceed the cap by itself)
                    break
                if c < dp[g + 1][j]:
                    dp[g + 1][j], arg[g + 1][j] = c, i
                if j == n or toks[j]["earliest"] > g:
                    break
                j += 1
    assert dp[n_gaps][n] < INF, "no feasible cut"
    gaps, j = [None] * n_gaps, n
    for g in range(n_gaps, 0, -1):
        i = arg[g][j]
        gaps[g - 1] = toks[i:j]
        j = i
    return gaps


def pair_chain(pairs, pos0, pos1):
    """E0 E1 of pair i, then A0 A1 CV of pair i - 1: order keys spread linearly over [pos0, pos1)"""
    seq = []
    for i, k in enumerate(pairs):
        seq += [(E0, k), (E1, k)]
        if i > 0:
            seq += [(A0, pairs[i - 1]), (A1, pairs[i - 1]), (CV, pairs[i - 1])]
    seq += [(A0, pairs[-1]), (A1, pairs[-1]), (CV, pairs[-1])]
    w = [COST[k] for k, _ in seq]
    tot, acc, out = sum(w), 0.0, []
    for (kind, k), c in zip(seq, w):
        out.append((kind, k, pos0 + (acc + c / 2) / tot * (pos1 - pos0)))
        acc += c
    return out


RING = 4            # V^T fragment registers: 4 (one 16-key step; set by main()) or 8 (two)


def vreg(dt, s):
    """register of V^T fragment (dt, s)"""
    return dt if RING == 4 else (s & 1) * 4 + dt


def phase_a():
    toks = [dict(kind=kind, a=k, b=0, earliest=0, order=pos) for kind, k, pos in pair_chain(LATE_PAIRS, 0.0, 32.0)]
    if RING == 8:
        # V^T(t) fragments of the first two 16-key steps, in the order the P.V MFMAs use them; all issued by gap ~27
        for f in range(8):
            toks.append(dict(kind=RV, a=(f & 3) + 4 * (f >> 2), b=vreg(f & 3, f >> 2), earliest=0, order=1.0 + f * 3.6))
        for i in range(4):                                # V^T(t+2) pieces
            toks.append(dict(kind=DMA, a=i, b=0, earliest=0, order=3.0 + i * 8.0))
        return deal(toks)
    # V^T(t) fragments of the first 16-key step (s = 0): register dt; the ring is FOUR fragments deep - fragment (dt, s + 1)
    # is read into register dt as soon as MFMA (s, dt, qb = 1) has issued, eight MFMAs before its first use
    for f in range(4):
        toks.append(dict(kind=RV, a=f, b=vreg(f, 0), earliest=0, order=14.0 + f * 4.0))
    for i in range(5):                                # the 4 V^T(t+2) pieces and the first K(t+4) piece
        toks.append(dict(kind=DMA, a=i, b=0, earliest=0, order=2.0 + i * 6.4))
    return deal(toks)


def pv(j):
    return dict(s=j >> 3, dt=(j >> 1) & 3, qb=j & 1)


def phase_b():
    toks = []

    def p_free_gap(k):      # the SUM of pair k writes P[pq][s]: legal once the last P.V MFMA that reads it has issued
        pq, s = k >> 4, (k & 15) >> 2
        return max(j for j in range(32) if pv(j)["s"] == s and pv(j)["qb"] == pq)
    for kind, k, pos in pair_chain(EARLY_PAIRS, 7.0, 32.0):
        e = p_free_gap(k) + 1 if kind == CV else 0
        toks.append(dict(kind=kind, a=k, b=0, earliest=e, order=max(pos, e + 0.01 * (kind == CV))))
    if RING == 8:
        # fragment (dt, s + 2) into the register of (dt, s) behind MFMA (s, dt, qb = 1) = 8 s + 2 dt + 1
        for s in range(2):
            for dt in range(4):
                j = 8 * s + 2 * dt + 1
                toks.append(dict(kind=RV2, a=dt + 4 * (s + 2), b=vreg(dt, s + 2), earliest=j, order=j + 0.05))
        for i, pos in enumerate([0.0, 2.0, 4.0, 6.0]):
            toks.append(dict(kind=DMA, a=4 + i, b=0, earliest=0, order=pos))
        for ut in range(16):
            toks.append(dict(kind=RK, a=ut, b=0, earliest=0, order=(0.6 + ut * 2.0) if ut < 2 else 7.5 + (ut - 2) * 1.72))
        return deal(toks)
    # V^T(t) fragment (dt, s) for s = 1..3 into register dt behind MFMA (s - 1, dt, qb = 1) = 8 (s - 1) + 2 dt + 1
    for s in range(1, 4):
        for dt in range(4):
            j = 8 * (s - 1) + 2 * dt + 1
            toks.append(dict(kind=RV2, a=dt + 4 * s, b=vreg(dt, s), earliest=j, order=j + 0.05))
    # the other 3 K(t+4) pieces open the phase (its first gaps have no pair work), the 16 K(t+2) fragments are spread over the rest
    for i, pos in enumerate([0.0, 2.5, 5.0]):
        toks.append(dict(kind=DMA, a=5 + i, b=0, earliest=0, order=pos))
    for ut in range(16):
        toks.append(dict(kind=RK, a=ut, b=0, earliest=0, order=(0.6 + ut * 2.0) if ut < 3 else 8.5 + (ut - 3) * 1.78))
    return deal(toks)


def add_waits(ga, gb):
    """lgkmcnt waits: LDS reads return in order, so the first use of a fragment waits until at most N reads issued AFTER it are
    outstanding.  One wait in front of every first use (P.V MFMA 8 s + 2 dt; for MFMA 0 at the end of phase A) that an earlier
    wait does not already cover, besides the lgkmcnt(0) before the barrier."""
    reads = []                                        # tokens in issue order with their (phase, gap)
    for ph, gaps in ((0, ga), (1, gb)):
        for g, toks in enumerate(gaps):
            for t in toks:
                if t["kind"] in (RV, RV2, RK):
                    reads.append((ph, g, t))
    out, done = [], -1                                # reads[0..done] are known complete
    for j in range(0, 32, 2):
        p = pv(j)
        code = p["dt"] + 4 * p["s"]
        tok = next(t for g in ga + gb for t in g if t["kind"] in (RV, RV2) and t["a"] == code)
        where = (0, 31) if j == 0 else (1, j - 1)
        idx = next(i for i, r in enumerate(reads) if r[2] is tok)
        if RING == 8 and p["dt"] == 0:
            # ring of eight: all four fragments of a 16-key step are issued a whole step before its first MFMA - ONE wait per
            # step, for the last of them (free by then), instead of one per fragment
            grp = [next(i for i, r in enumerate(reads) if r[2] is t) for g in ga + gb for t in g
                   if t["kind"] in (RV, RV2) and t["a"] // 4 == p["s"]]
            issued_now = sum(1 for (ph, g, t) in reads if (ph, g) <= where)
            if all(i < issued_now for i in grp):
                idx = max(grp)
        issued = sum(1 for (ph, g, t) in reads if (ph, g) <= where)
        assert idx < issued, (j, idx, issued)
        if idx <= done:
            continue
        n = min(15, issued - 1 - idx)
        done = issued - 1 - n
        out.append((where, n))
    # the counter is 4 bits: never more than 15 reads in flight
    outstanding = 0
    waits = dict(out)
    for ph, gaps in ((0, ga), (1, gb)):
        for g, toks in enumerate(gaps):
            outstanding += sum(1 for t in toks if t["kind"] in (RV, RV2, RK))
            assert outstanding <= 15, ("more than 15 LDS reads in flight at", ph, g)
            if (ph, g) in waits:
                outstanding = min(outstanding, waits[(ph, g)])
    for (ph, g), n in out:
        (ga if ph == 0 else gb)[g].append(dict(kind=WAIT, a=n, b=0))
    return ga, gb


def price(toks):
    return sum(COST[t["kind"]] for t in toks)


def emit(ga, gb, path):
    def arr(name, gaps):
        flat, first = [], [0]
        for g in gaps:
            flat += g
            first.append(len(flat))
        body = ", ".join("{%d, %d, %d}" % (t["kind"], t["a"], t["b"]) for t in flat)
        return (f"constexpr Tok {name}_TOK[{len(flat)}] = {{{body}}};\n"
                f"constexpr int {name}_FIRST[{len(first)}] = {{{', '.join(map(str, first))}}};\n")
    early = ", ".join(map(str, EARLY_PAIRS))
    txt = f"""// GENERATED by tools/gen_a64_sched.py - do not edit: the filler schedule of attn64_kernel<true>'s tile (attention64.hip).
// Token (kind, a, b) executed in the gap BEHIND MFMA g of its phase:  E0 / E1 k (v_exp of the pair's first / second probability) |
// A0 / A1 k (row-sum adds) | CV k (bf16 pack into P) | RV / RV2 (dt + 4 s, register): V^T fragment (dt, s) | RK ut (ds_read_b128) |
// DMA i | WAIT lgkmcnt
#pragma once
namespace a64s {{
enum : int {{ T_E0 = {E0}, T_E1 = {E1}, T_A0 = {A0}, T_A1 = {A1}, T_CV = {CV}, T_RV = {RV}, T_RV2 = {RV2}, T_RK = {RK}, T_DMA = {DMA}, T_WAIT = {WAIT} }};
struct Tok {{ int kind, a, b; }};
constexpr int V_REGS = {RING};                           // V^T fragment registers (u32x4 each)
constexpr int PV_REG[32] = {{{", ".join(str(vreg(pv(j)["dt"], pv(j)["s"])) for j in range(32))}}};      // the one P.V MFMA j reads
constexpr int N_EARLY = {len(EARLY_PAIRS)};
constexpr int EARLY_PAIR[N_EARLY] = {{{early}}};      // pairs of P(t+1) exponentiated in phase B of tile t
constexpr int EARLY_FIRST[2] = {{{next(k for k in EARLY_PAIRS if k < 16)}, {next(k for k in EARLY_PAIRS if k >= 16)}}};      // the first of them per query block
{arr("A", ga)}{arr("B", gb)}}}  // namespace a64s
"""
    open(path, "w").write(txt)


def main():
    global RING
    RING = 8 if "--ring8" in sys.argv else 4
    ga, gb = phase_a(), phase_b()
    ga, gb = add_waits(ga, gb)
    for name, gaps in (("A", ga), ("B", gb)):
        print(f"phase {name}: total {sum(price(g) for g in gaps):.1f} slots, max gap {max(price(g) for g in gaps):.1f}, "
              f"est. cycles {sum(gap_cycles(price(g)) for g in gaps):.0f}")
        for g, toks in enumerate(gaps):
            print(f"  {g:2d} {price(toks):5.1f}  " + " ".join(f"{NAMES[t['kind']]}{t['a']}" + (f">{t['b']}" if t['kind'] in (RV, RV2) else "") for t in toks))
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    emit(ga, gb, os.path.join(here, "visualcloze_amd", "csrc", "attention64_sched.h" if RING == 4 else "attention64_sched8.h"))


if __name__ == "__main__":
    main()
