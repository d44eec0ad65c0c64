#!/usr/bin/env python3
"""Digest of gpurun_out/shard_waves_<tag>.npz (tools/gpu_shard_waves.py on a -DRTMI_STATS=9 build): when the waves of a
frame end, per wave and per SIMD (HW_ID / XCC_ID recorded by the kernel)."""
import sys
import numpy as np
for tag in sys.argv[1:]:
    d = np.load('gpurun_out/shard_waves_%s.npz' % tag)
    ws = d['ws']; counts = d['counts'].astype(np.int64)
    q = ws[:, 11].astype(float); t0 = ws[:, 12].astype(float); t1 = ws[:, 13].astype(float); hw = ws[:, 14]
    s = t0.min(); end = (t1 - s) / 100
    print(tag, 'T ms %.1f' % (end.max() / 1e3), 'wave end pct(0,1,10,50,90,99)', np.round(np.percentile(end, [0, 1, 10, 50, 90, 99]) / 1e3, 0),
          'util %.3f' % (counts.sum() / (q.sum() * 64)), 'sumq %.1fM' % (q.sum() / 1e6), 'q pct', np.round(np.percentile(q, [0, 10, 50, 90, 100]) / 1e3, 1))
    key = ((hw >> 32) & 15) << 10 | ((hw >> 8) & 0xff) << 2 | ((hw >> 4) & 3)
    u, cnt = np.unique(key, return_counts=True)
    idx = np.searchsorted(u, key)
    sq = np.bincount(idx, weights=q); emax = np.zeros(len(u)); np.maximum.at(emax, idx, end); emin = np.full(len(u), 1e18); np.minimum.at(emin, idx, end)
    print('   simds', len(u), 'waves/simd', dict(zip(*np.unique(cnt, return_counts=True))), 'per-simd sum q pct(0,10,50,90,100)', np.round(np.percentile(sq, [0, 10, 50, 90, 100]) / 1e3, 0),
          ' simd last-end', np.round(np.percentile(emax, [0, 10, 50, 90, 100]) / 1e3, 0), ' simd first-end', np.round(np.percentile(emin, [0, 10, 50, 90, 100]) / 1e3, 0))
    xcc = (u >> 10) & 15
    print('   per XCC mean simd end:', np.round([emax[xcc == x].mean() / 1e3 for x in np.unique(xcc)], 0), ' mean sum q (k):', np.round([sq[xcc == x].mean() / 1e3 for x in np.unique(xcc)], 0))
