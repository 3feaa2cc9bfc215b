mkdir -p gpurun_out/r3op
for v in 1 0 1 0; do
  echo "=== KMC_TILE_LOOP=$v" >> gpurun_out/r3op/ab.txt
  KMC_TILE_LOOP=$v python tools/measure_traj_batch.py 64 1000000 >> gpurun_out/r3op/ab.txt 2>/dev/null
  KMC_TILE_LOOP=$v python tools/measure_traj_batch.py 520 123397 >> gpurun_out/r3op/ab.txt 2>/dev/null
  KMC_TILE_LOOP=$v python tools/measure_small_batches.py >> gpurun_out/r3op/ab.txt 2>/dev/null
  KMC_TILE_LOOP=$v python tools/measure_tiers.py >> gpurun_out/r3op/ab.txt 2>/dev/null
  KMC_TILE_LOOP=$v python bench.py --no-legs --no-cpu-baseline --no-live-traffic --sustained-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('bench', d['value'], d['roofline']['frac'], d['configs3']['value'])" >> gpurun_out/r3op/ab.txt
done
cat gpurun_out/r3op/ab.txt
