#!/usr/bin/env python3
"""Per-search cost distribution of k_rs_validate (instrumented build, HOPE_RS_TIMING=1) and what queue order would do to its
tail: list-scheduling simulation of the logged search durations on `slots` wave slots in (a) the order the kernel ran them,
(b) longest first.  Usage (GPU box):  HOPE_RS_TIMING=1 python tools/rs_tail.py [--scenes 65536]"""
import argparse
import ctypes as C
import heapq
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('HOPE_RS_TIMING', '1')


def makespan(d, slots):
    h = [0.0] * min(slots, len(d))
    heapq.heapify(h)
    for x in d:
        heapq.heappush(h, heapq.heappop(h) + x)
    return max(h)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scenes', type=int, default=65536)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--slots', type=int, default=3072, help='concurrent waves: 1024 SIMDs x waves per SIMD')
    args = ap.parse_args()
    import torch
    sys.argv = ['bench']
    import bench
    from hope_amd import ParkingBatch, _lib as L
    from hope_amd.scenes import pack_scenes
    rng = np.random.default_rng(42)
    uniq = bench.make_scenes(2048, 'mixed', rng)
    start, dest, bbox, verts, nob, nvert = pack_scenes(uniq, 128)
    N = args.scenes
    reps = (N + len(uniq) - 1) // len(uniq)
    tile = lambda a: np.concatenate([a] * reps, axis=0)[:N]  # noqa: E731
    env = ParkingBatch(N, 128, overlap=False, profile=True)
    for a in range(0, N, 8192):
        sl = slice(a, min(N, a + 8192))
        env.set_scene_arrays(np.arange(sl.start, sl.stop), tile(start)[sl], tile(dest)[sl], tile(bbox)[sl], tile(verts)[sl], tile(nob)[sl])
    g = torch.Generator(device='cuda').manual_seed(1)
    env.reset_obs()
    for _ in range(12):
        env.step(torch.rand((N, 2), device='cuda', generator=g) * 2 - 1, auto_reset=True)
    lib = L.load_library()
    cap = 1 << 21
    buf = np.zeros((cap, 4), np.int32)
    n = C.c_int32(0)
    lib.hope_debug_rs_log(buf.ctypes.data_as(C.c_void_p), cap, C.byref(n), 1)
    env.kernel_ms()
    for _ in range(args.steps):
        env.step(torch.rand((N, 2), device='cuda', generator=g) * 2 - 1, auto_reset=True)
    lib.hope_debug_rs_log(buf.ctypes.data_as(C.c_void_p), cap, C.byref(n), 1)
    km = env.kernel_ms()
    print({k: round(a / max(b, 1), 4) for k, (a, b) in km.items() if b}, 'ms per launch')
    m = min(n.value, cap)
    r = buf[:m]
    cyc = r[:, 0].astype(np.float64)
    words, allowed, big = r[:, 1] & 0xff, (r[:, 1] >> 8) & 0xff, (r[:, 1] >> 16) & 1
    passes, found = r[:, 2], r[:, 3]
    print(f'{m} searches logged over {args.steps} steps ({m / args.steps:.0f} per step); mean {cyc.mean():.0f} cycles, '
          f'words tested {words.mean():.2f} of {allowed.mean():.2f} allowed, passes {passes.mean():.2f}, found {found.mean():.3f}')
    for q in (50, 75, 90, 95, 99, 99.9, 100):
        print(f'  p{q:<5} cycles {np.percentile(cyc, q):10.0f}   passes {np.percentile(passes, q):6.0f}')
    order = np.argsort(-cyc)
    csum = np.cumsum(cyc[order]) / cyc.sum()
    for f in (0.001, 0.01, 0.05, 0.1, 0.25):
        print(f'  the longest {100 * f:.1f} % of the searches hold {100 * csum[int(f * m) - 1]:.1f} % of the cycles')
    for cls in (0, 1):
        s = big == cls
        if s.any():
            print(f'  tile class {cls}: {s.sum() / args.steps:.0f} searches/step, mean {cyc[s].mean():.0f} cycles, found {found[s].mean():.3f}, '
                  f'mean passes found / not found {passes[s & (found == 1)].mean() if (s & (found == 1)).any() else 0:.1f} / {passes[s & (found == 0)].mean():.1f}')
    # queue-order simulation per step and class (the log is in completion order: shuffle = an arbitrary order)
    per = m // args.steps
    for cls in (0, 1):
        d = cyc[big == cls][: (big == cls).sum() // args.steps]
        if len(d) == 0:
            continue
        rs = np.random.default_rng(0)
        arb = np.mean([makespan(rs.permutation(d), args.slots) for _ in range(5)])
        lpt = makespan(np.sort(d)[::-1], args.slots)
        ideal = max(d.sum() / args.slots, d.max())
        # a cost proxy the kernels know before validation: words allowed by the stop rule
        print(f'  class {cls}, {len(d)} searches on {args.slots} wave slots: arbitrary order {arb:.0f} cycles, longest first {lpt:.0f}, '
              f'lower bound {ideal:.0f} (longest search {d.max():.0f})')


if __name__ == '__main__':
    main()
