#!/usr/bin/env python3
"""Offline model of one C4 shard on one MI355X: 1024 SIMDs x 6 waves x 64 lanes, a pixel = an indivisible chain of
closest-hit queries (per-pixel ray counts of a real shard: gpurun_out/shard_waves_c4_s0of8_p0_b0.npz, written by
tools/gpu_shard_waves.py), a global queue in longest-first tile order, and the SIMD's issue arbitration as measured
(DESIGN "What a wave's speed depends on": cycles per query by dispatch round 22/24/29/36/47/60 k with six waves
resident).  Used to price wave-priority policies before spending GPU time on them.
usage: simd_share_sim.py [policy ...]   policies: age | lrpt | lrpt4 | global4"""
import sys
import numpy as np

A = 1.0 / 22.0  # queries per kcycle of a wave that has the SIMD to itself
MEAS = np.array([22., 24., 29., 36., 47., 60.])
CAP = (1.0 / MEAS).sum()
# share of the alone-rate a wave gets as a function of the capacity fraction left by the waves served before it
_left = [1.0]
for r in 1.0 / MEAS:
    _left.append(_left[-1] - r / CAP)
LEFT_PTS = np.array(_left[:6][::-1])           # increasing
RATE_PTS = ((1.0 / MEAS) / A)[::-1]


def rates_for_order(active_sorted_mask):
    """active_sorted_mask [S, 6] bool in SERVICE order -> rates [S, 6] (queries per kcycle) in the same order."""
    S = active_sorted_mask.shape[0]
    left = np.ones(S)
    out = np.zeros((S, 6))
    for i in range(6):
        g = np.interp(left, LEFT_PTS, RATE_PTS, left=0.0)   # fraction of the alone rate
        g = np.where(left < LEFT_PTS[0], left / LEFT_PTS[0] * RATE_PTS[0], g)
        r = np.minimum(g * A, left * CAP) * active_sorted_mask[:, i]
        out[:, i] = r
        left = np.maximum(left - r / CAP, 0.0)
    return out


def simulate(counts, policy, dt_kc=600.0, noise=0.10, seed=1, update_every=1, verbose=False):
    rng = np.random.default_rng(seed)
    n_simd, n_w = 1024, 6
    W = n_simd * n_w
    tiles = counts.reshape(-1, 64).astype(np.float64)
    est = tiles.sum(1) * (1.0 + noise * rng.standard_normal(tiles.shape[0]))
    order = np.argsort(-est, kind="stable")
    # item order: tiles longest first, pixels of a tile in sequence
    items = (order[:, None] * 64 + np.arange(64)[None, :]).reshape(-1)
    flat = counts.astype(np.float64)
    qpos = 0
    # dispatch: wave w of workgroup order; age rank within the SIMD = w // n_simd (round-robin over SIMDs)
    rem = np.zeros((W, 64))
    # initial deal: wave w takes items [64 w, 64 w + 64)
    rem[:] = flat[items[:W * 64]].reshape(W, 64)
    qpos = W * 64
    simd_of = np.arange(W) % n_simd
    age = np.arange(W) // n_simd          # 0 = oldest
    wave_idx = (age[:, None] * 0)  # unused
    byslot = np.arange(W).reshape(n_w, n_simd).T   # [S, 6] wave ids, column = age
    done_wave = np.zeros(W, bool)
    end_t = np.zeros(W)
    wave_q = np.zeros(W)
    t = 0.0
    prio = np.zeros(W, int)
    step = 0
    n_items = items.size
    while not done_wave.all():
        wrem = rem.max(1)                 # the wave's chain: its longest lane (future queue items unknown)
        act = ~done_wave
        if step % update_every == 0:
            if policy == "age":
                key = age.astype(float)
            elif policy == "lrpt":
                key = -wrem + 1e-6 * age
            elif policy == "lrpt4":      # 4 levels by rank within the SIMD, ties by age
                r_s = wrem[byslot] * act[byslot]
                rank = (-r_s).argsort(1).argsort(1)          # 0 = largest remaining
                lvl = np.choose(np.minimum(rank, 5), [3, 2, 1, 1, 0, 0])
                p = np.zeros(W, int)
                p[byslot] = lvl
                key = -p + 1e-3 * age
            elif policy == "global4":    # levels by remaining relative to the frame-wide maximum
                m = wrem[act].max() if act.any() else 1.0
                p = np.digitize(wrem / max(m, 1.0), [0.4, 0.6, 0.8])
                key = -p + 1e-3 * age
            else:
                raise SystemExit("unknown policy " + policy)
        k_s = key[byslot]
        svc = k_s.argsort(1, kind="stable")                  # service order per SIMD
        ids = np.take_along_axis(byslot, svc, 1)
        r = rates_for_order(act[ids])
        rate = np.zeros(W)
        rate[ids] = r
        adv = rate * dt_kc                                   # queries this step
        wave_q += adv
        rem -= adv[:, None]
        t += dt_kc
        # lanes that finished take the next items
        need = (rem <= 0) & act[:, None]
        rem[need] = 0.0
        nn = int(need.sum())
        if nn and qpos < n_items:
            take = min(nn, n_items - qpos)
            wi, li = np.nonzero(need)
            # hand out in wave order (oldest first, like the atomics: whoever asks first)
            rem[wi[:take], li[:take]] = flat[items[qpos:qpos + take]]
            qpos += take
        newly = act & (rem.max(1) <= 0)
        end_t[newly] = t
        done_wave |= newly
        step += 1
    T = t
    return dict(policy=policy, T_Mcycles=T / 1e3, first_end=end_t.min() / 1e3, end_pct=np.percentile(end_t, [10, 50, 90, 99]) / 1e3,
                wave_q_sum=wave_q.sum(), util=flat.sum() / (wave_q.sum() * 64),
                full_rate_T=wave_q.sum() / (n_simd * CAP) / 1e3)


if __name__ == "__main__":
    d = np.load("gpurun_out/shard_waves_c4_s0of8_p0_b0.npz")
    counts = d["counts"].astype(np.int64)
    for pol in (sys.argv[1:] or ["age", "lrpt", "lrpt4", "global4"]):
        res = simulate(counts, pol)
        print({k: (np.round(v, 1).tolist() if isinstance(v, np.ndarray) else (round(v, 3) if isinstance(v, float) else v)) for k, v in res.items()})


def simulate_plan(counts, deal="snake", prio="lrpt", dt_kc=600.0, noise=0.10, seed=1, n_w=6, lanes_pair="asis", levels=None):
    """Static plan: tiles -> W chains by LPT on the (noisy) probe estimate; chains dealt to (SIMD, slot); a wave's
    lanes walk the chain's tiles pixel by pixel (lane l takes pixel l of every tile); no queue.
    prio: 'age' (hardware default) or 'lrpt' (strict, by remaining incl. the estimate of the tiles still to come)."""
    import heapq
    rng = np.random.default_rng(seed)
    n_simd = 1024
    W = n_simd * n_w
    tiles = counts.reshape(-1, 64).astype(np.float64)
    nt = tiles.shape[0]
    est_t = tiles.max(1) * (1.0 + noise * rng.standard_normal(nt))   # wave-level cost of a tile: its longest pixel
    order = np.argsort(-est_t, kind="stable")
    heap = [(0.0, w) for w in range(W)]
    heapq.heapify(heap)
    chains = [[] for _ in range(W)]
    for tix in order:
        tot, w = heapq.heappop(heap)
        chains[w].append(tix)
        heapq.heappush(heap, (tot + est_t[tix], w))
    est_chain = np.array([sum(est_t[c]) for c in chains])
    # deal chains to (simd, slot)
    by_cost = np.argsort(-est_chain, kind="stable")
    slot_of = np.zeros((n_simd, n_w), int)   # chain id at (simd, slot)
    if deal == "snake":
        for k in range(n_w):
            seg = by_cost[k * n_simd:(k + 1) * n_simd]
            slot_of[:, k] = seg if k % 2 == 0 else seg[::-1]
    else:  # dispatch order: wave w = chain by_cost[w], simd = w % n_simd, slot = w // n_simd
        for k in range(n_w):
            slot_of[:, k] = by_cost[k * n_simd:(k + 1) * n_simd]
    maxlen = max(len(c) for c in chains)
    # per wave-slot state
    chain_tiles = -np.ones((W, maxlen), int)
    for s in range(n_simd):
        for k in range(n_w):
            c = chains[slot_of[s, k]]
            chain_tiles[k * n_simd + s, :len(c)] = c
    pos = np.zeros((W, 64), int)                      # which tile of the chain each lane is on
    rem = tiles[chain_tiles[:, 0]].copy()             # [W, 64]
    est_future = np.zeros((W, maxlen + 1))            # estimated cost of tiles from position p on
    for p in range(maxlen - 1, -1, -1):
        valid = chain_tiles[:, p] >= 0
        est_future[:, p] = est_future[:, p + 1] + np.where(valid, est_t[np.maximum(chain_tiles[:, p], 0)], 0.0)
    byslot = np.arange(W).reshape(n_w, n_simd).T
    age = np.arange(W) // n_simd
    done_wave = np.zeros(W, bool)
    end_t = np.zeros(W)
    wave_q = np.zeros(W)
    t = 0.0
    ar = np.arange(W)
    while not done_wave.all():
        act = ~done_wave
        lane_left = rem + est_future[ar[:, None], np.minimum(pos + 1, maxlen)]
        wrem = lane_left.max(1)
        if prio == "age":
            key = age.astype(float)
        elif levels is not None:
            r_s = wrem[byslot] * act[byslot]
            rank = (-r_s).argsort(1).argsort(1)
            p = np.zeros(W, int)
            p[byslot] = np.choose(np.minimum(rank, 5), levels)
            key = -p + 1e-3 * age
        else:
            key = -wrem + 1e-6 * age
        svc = key[byslot].argsort(1, kind="stable")
        ids = np.take_along_axis(byslot, svc, 1)
        r = rates_for_order(act[ids])
        rate = np.zeros(W)
        rate[ids] = r
        adv = rate * dt_kc
        wave_q += adv
        rem -= adv[:, None]
        t += dt_kc
        need = (rem <= 0) & act[:, None]
        if need.any():
            wi, li = np.nonzero(need)
            npos = pos[wi, li] + 1
            ok = npos < maxlen
            nxt = np.where(ok, chain_tiles[wi, np.minimum(npos, maxlen - 1)], -1)
            has = nxt >= 0
            pos[wi, li] = npos
            rem[wi, li] = np.where(has, tiles[np.maximum(nxt, 0), li], 0.0)
            pos[wi[~has], li[~has]] = maxlen  # finished: no future
        fin = (pos >= maxlen) | ((rem <= 0))
        newly = act & (rem.max(1) <= 0) & ((pos >= maxlen) | (chain_tiles[ar[:, None], np.minimum(pos, maxlen - 1)] < 0)).all(1)
        end_t[newly] = t
        done_wave |= newly
    return dict(policy="plan/%s/%s%s" % (deal, prio, levels or ""), T_Mcycles=t / 1e3, first_end=end_t.min() / 1e3,
                end_pct=np.percentile(end_t, [10, 50, 90, 99]) / 1e3, wave_q_sum=wave_q.sum(),
                util=tiles.sum() / (wave_q.sum() * 64), full_rate_T=wave_q.sum() / (n_simd * CAP) / 1e3,
                chain_est=[float(est_chain.min()), float(est_chain.mean()), float(est_chain.max())])
