O=gpurun_out/final2; rm -rf $O; mkdir -p $O
T="timeout 400"
$T python bench.py 2>/dev/null | tail -1 > $O/r04_bench_default.json
$T python bench.py 2>/dev/null | tail -1 > $O/r04_bench_default_2.json
$T python bench.py --image --no-cpu-baseline 2>/dev/null | tail -1 > $O/r04_bench_image.json
$T python bench.py --rs-join joined --no-cpu-baseline 2>/dev/null | tail -1 > $O/r04_bench_joined.json
$T python bench.py --preroll 0 --no-cpu-baseline --witness 0 2>/dev/null | tail -1 > $O/r04_bench_fresh_episodes.json
