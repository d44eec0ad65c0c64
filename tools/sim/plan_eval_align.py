#!/usr/bin/env python3
"""Offline: what is left of a planned frame's imbalance, and could the plan remove it?  Same model as plan_eval.py (per-pixel
ray counts of recorded frames, estimates from s1 samples as true x (1 + 1.2 / sqrt(s1) N(0, 1))), plus one more policy:
ALIGN -- when a chain moves on to its next tile, the lanes that are ahead in accumulated (estimated) cost take that tile's
cheapest pixels.  Result (round 4): with perfect knowledge alignment lifts lane utilisation from 0.944 to 0.978 on C2; with
estimates from 16 or 64 samples it does nothing -- the spread of the lanes' totals inside a wave is the Monte-Carlo scatter
of the pixels' own ray counts (1.2 / sqrt(1024) = 3.7 % per pixel, the unluckiest of 64 lanes ends 8 % after the mean),
which no estimate made in advance can know.  That scatter, not the plan, is what bounds C2 and the C4 shards now.
usage: tools/sim/plan_eval_align.py"""
import numpy as np, sys
def deal_eval(counts, s1, align, S=1024, R=6, seed=0, rel=1.2):
    rng=np.random.default_rng(seed)
    px=counts.astype(float)
    est_px = np.maximum(px*(1+rel/np.sqrt(s1)*rng.standard_normal(px.shape)),0) if s1>0 else px
    t_true=px.reshape(-1,64); t_est=est_px.reshape(-1,64)
    cost=t_est.mean(1)
    nt=len(cost); order=np.argsort(-cost,kind='stable')
    ranks=np.arange(nt); k=ranks//S; pos=ranks%S
    s_of=np.where(k%2==1, S-1-pos, pos)
    j=k//R; kr=k%R
    r_of=np.where(j%2==1, R-1-kr, kr)
    chain=(s_of*R+r_of)   # chain id per rank
    W=S*R
    lane_true=np.zeros((W,64)); lane_est=np.zeros((W,64))
    # process ranks in order of chain position j (each chain's tiles in order)
    for jj in range(j.max()+1):
        sel=np.nonzero(j==jj)[0]
        tiles=order[sel]; ch=chain[sel]
        if align and jj>0:
            # lanes sorted by accumulated estimate desc get pixels sorted by estimate asc
            lane_order=np.argsort(-lane_est[ch],axis=1)            # [n,64] lanes heavy->light
            pix_order=np.argsort(t_est[tiles],axis=1)              # pixels light->heavy
            perm=np.empty_like(lane_order)
            np.put_along_axis(perm, lane_order, pix_order, axis=1)  # perm[lane]=pixel
            lane_true[ch]+=np.take_along_axis(t_true[tiles],perm,axis=1)
            lane_est[ch]+=np.take_along_axis(t_est[tiles],perm,axis=1)
        else:
            lane_true[ch]+=t_true[tiles]; lane_est[ch]+=t_est[tiles]
    wave_q=lane_true.max(1).reshape(S,R); simd=wave_q.sum(1)
    util=px.sum()/(wave_q.sum()*64)
    return simd.max()/simd.mean(), util, simd.max()
for f,tag in (('gpurun_out/shard_waves_c2_s0of1_p0_b0_r4f.npz','c2'),('gpurun_out/shard_waves_c4_s1of8_p0_b0_r4f.npz','c4s1')):
    counts=np.load(f)['counts'].astype(np.int64)
    for s1 in (16,64,0):
        for al in (False,True):
            a,u,mx=deal_eval(counts,s1,al)
            print(tag,'s1',s1,'align' if al else 'plain','max/mean %.3f util %.3f  max simd sum %.0f'%(a,u,mx))
