# same-box A/B of memory-side traffic (FETCH_SIZE / WRITE_SIZE passes, tools/pmc_traffic.py) and step time:
# hope_amd/libhope_env.so vs hope_amd/libhope_env_b1.so.  Outputs under gpurun_out/s3/.
R=$PWD; O=$R/gpurun_out/s3; mkdir -p $O
bash tools/exp_ab.sh > $O/ab_time.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for V in A B; do
  LIB=$R/hope_amd/libhope_env.so; [ $V = B ] && LIB=$R/hope_amd/libhope_env_b1.so
  for C in FETCH_SIZE WRITE_SIZE; do
    HOPE_AMD_LIB=$LIB timeout 400 rocprofv3 --pmc $C --output-format csv -d $O/pmc_${V}_$C -- python $R/bench.py --steps 6 --warmup 2 --preroll 60 --no-cpu-baseline --witness 0 --repeat-passes 0 > /dev/null 2>&1
  done
  F=$(find $O/pmc_${V}_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_${V}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  python $R/tools/pmc_traffic.py $F $W $O/traffic_$V.json > /dev/null
  rm -rf $O/pmc_${V}_FETCH_SIZE $O/pmc_${V}_WRITE_SIZE
done
cat $O/ab_time.txt
python - <<PY
import json
for v in 'AB':
    d=json.load(open('$O/traffic_%s.json'%v))['kernels']
    per={'k_env_step':4,'k_kinematics':2,'k_post':2,'k_rs_segs':2,'k_rs_validate':2,'k_rs_words':2,'k_rs_compact':2}
    tot=sum(d[k]['hbm_bytes']*m for k,m in per.items() if k in d)
    print(v, {k:(round(d[k]['fetch_bytes']/1e6,1), round(d[k]['write_bytes']/1e6,1)) for k in d}, 'per step MB', round(tot/1e6,1))
PY
