#!/bin/bash
# Clock and power of the GPU while the planner loop runs (is the fused launch power-bound?): runs a bench configuration in
# the background and samples rocm-smi.  usage: power_probe.sh "<env>" "<bench args>"
ENVV="$1"; ARGS="$2"
( env $ENVV python3 bench.py --steps 6000 --warmup 20 --no-other-configs --no-cpu-baseline --no-parity $ARGS > /tmp/power_probe_bench.json 2>/dev/null ) &
BP=$!
sleep 14
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk|mclk|fclk" | tr -s ' ' | tr '\n' ';'; echo
  sleep 0.15
done
wait $BP
python3 -c "
import json; d=json.load(open('/tmp/power_probe_bench.json')); print('[$ENVV $ARGS]', round(d['value'],1), 'it/s', {k:round(v,5) for k,v in d['kernel_ms_per_step'].items()})"
