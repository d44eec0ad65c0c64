#!/usr/bin/env python3
"""Offline: how evenly does the planned-chain deal (kernels.hip: chain_link_kernel -- tiles in longest-first order dealt
to the SIMDs in a snake, a SIMD's share to its six waves in a snake again) load the SIMDs, as a function of how good the
cost estimate is?  Input: the per-pixel ray counts of real frames (gpurun_out/shard_waves_*.npz, written by
tools/gpu_shard_waves.py).  A pixel's estimate from s1 probe samples is modelled as true x (1 + 1.2 / sqrt(s1) x N(0, 1));
s1 = 0 is perfect knowledge.  A wave's length is its longest lane's total (lane l renders pixel l of every tile of its
chain), a SIMD's load the sum over its waves.  Prints max / mean and min / mean SIMD load and the heaviest wave's share.
usage: tools/sim/plan_eval.py"""
import numpy as np, sys
def plan_eval(counts, s1, use, S=1024, R=6, seed=0, rel=1.2):
    rng=np.random.default_rng(seed)
    px=counts.astype(float)
    est_px = px*(1+rel/np.sqrt(s1)*rng.standard_normal(px.shape)) if s1>0 else px
    est_px=np.maximum(est_px,0)
    t_true=px.reshape(-1,64); t_est=est_px.reshape(-1,64)
    if use=='mean': cost=t_est.mean(1)
    elif use=='max': cost=t_est.max(1)
    elif use=='p90': cost=np.percentile(t_est,90,axis=1)
    elif use=='blend': cost=0.5*(t_est.mean(1)+t_est.max(1))
    nt=len(cost); order=np.argsort(-cost,kind='stable')
    simd_sum=np.zeros(S); wave_q=np.zeros((S,R))
    # chains
    lane_tot=np.zeros((S,R,64))
    for s in range(S):
        pass
    # vectorised: for each rank -> (s, r)
    ranks=np.arange(nt); k=ranks//S; pos=ranks%S
    s_of=np.where(k%2==1, S-1-pos, pos)
    j=k//R; kr=k%R
    r_of=np.where(j%2==1, R-1-kr, kr)
    np.add.at(lane_tot,(s_of[:,None].repeat(64,1), r_of[:,None].repeat(64,1), np.arange(64)[None,:].repeat(nt,0)), t_true[order])
    wave_q=lane_tot.max(2)
    simd=wave_q.sum(1)
    return simd.mean(), simd.max(), simd.min(), wave_q.max()/simd.mean()
for f,tag in (('gpurun_out/shard_waves_c4_s1of8_p0_b0_simd.npz','c4s1'),('gpurun_out/shard_waves_c2_s0of1_p0_b0_simd.npz','c2')):  # (any records of these two frames do)
    counts=np.load(f)['counts'].astype(np.int64)
    for s1 in (2,8,32,128,0):
        for use in ('mean','max','blend','p90'):
            m,mx,mn,share=plan_eval(counts,s1,use)
            print(tag,'s1',s1,use,'simd sum mean %.0f max/mean %.3f min/mean %.3f  heaviest wave share %.3f'%(m,mx/m,mn/m,share))
