#!/bin/bash
# tools/clk_probe.sh -- sample the shader clock while bench.py runs (GPU box)
ls /sys/class/drm/ 2>/dev/null | head; for f in /sys/class/drm/card*/device/pp_dpm_sclk; do echo $f; cat $f; done
python bench.py --steps 400 --warmup 2 --no-cpu-baseline --verify-batch 0 > /tmp/b.json 2>/dev/null &
BP=$!
sleep 12
for i in 1 2 3 4 5; do
  for f in /sys/class/drm/card*/device/pp_dpm_sclk; do grep '\*' $f; done | tr '\n' ' '; echo
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|power\|junction" | tr '\n' ' '; echo
  sleep 0.3
done
wait $BP
tail -c 600 /tmp/b.json
