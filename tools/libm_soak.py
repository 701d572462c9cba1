#!/usr/bin/env python3
"""the independent-math comparison (tests/libm_compare.py: HIP path vs the oracle's glibc-libm build, every step
from the GPU's state) as a SOAK over >= 1e6 scene-steps, the libm oracle under OpenMP; prints every tolerated difference with the
tie it sits on and fails on anything else.
    python tools/libm_soak.py --scenes 8192 --steps 128 > profiles/rNN_libm_soak.txt"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scenes', type=int, default=8192)
    ap.add_argument('--steps', type=int, default=128)
    ap.add_argument('--seed', type=int, default=191)
    args = ap.parse_args()
    from libm_compare import run, check
    r = run(n=args.scenes, steps=args.steps, seed=args.seed, omp=True, restart=True, progress=16)
    print(f'libm soak: {r["scene_steps"]} scene-steps, {r["searches"]} searches; status differences {r["status_bad"]}; '
          f'worst continuous difference {r["worst"]:.3e}; unexplained search differences {len(r["unexplained"])}')
    print(f'tolerated: {len(r["mask_ties"])} mask rows, {r["excused"]} searches ({r["n_twins"]} equal-length twins, {r["n_axis"]} axis-aligned luck)')
    print('mask ties (step, scene, distance of the nearest table entry to its scan value):')
    for m in r['mask_ties']:
        print('  ', m)
    print('search ties (step, scene, GPU word, libm-oracle word, class, relative length gap):')
    for m in r['rs_ties']:
        print('  ', m)
    check(r, min_searches=100000)
    print('OK')


if __name__ == '__main__':
    main()
