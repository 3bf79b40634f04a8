# interleaved A/B: tail on / off, 4 rounds
for i in 1 2 3 4; do
  for d in on off $EXTRA; do
    if [ $d = off ]; then NMFAMD_NO_FUSED_TAIL=1 python bench.py --steps 200 --warmup 20 > gpurun_out/ab_${d}_$i.json 2>/dev/null
    elif [ $d = on ]; then python bench.py --steps 200 --warmup 20 > gpurun_out/ab_${d}_$i.json 2>gpurun_out/ab_${d}_$i.err
    else NMFAMD_TAIL_DEBUG=$d python bench.py --steps 200 --warmup 20 > gpurun_out/ab_${d}_$i.json 2>gpurun_out/ab_${d}_$i.err; fi
  done
done
python - <<PY
import json,glob,collections
r=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/ab_*_?.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r[f.split("ab_")[1].rsplit("_",1)[0]].append(round(d["ms_per_step"]*1000,1))
    except Exception as e: r[f.split("ab_")[1].rsplit("_",1)[0]].append("ERR")
for k,v in r.items(): print(k, v)
PY
