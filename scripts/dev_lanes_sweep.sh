#!/bin/bash
# batches in flight / paths per batch against the headline frame
H="python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 4 --warmup 1"
for lanes in 1 2 3 4; do echo -n "lanes $lanes: "; RTGPU_RENDER_LANES=$lanes timeout -k 10 120 $H 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],1), round(j['value']))" || exit 1; done
for bp in 4194304 8388608 33554432; do echo -n "batch $bp: "; RTGPU_BATCH_PATHS=$bp timeout -k 10 120 $H 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],1), round(j['value']))" || exit 1; done
for bp in 8388608; do echo -n "batch $bp lanes 4: "; RTGPU_RENDER_LANES=4 RTGPU_BATCH_PATHS=$bp timeout -k 10 120 $H 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],1), round(j['value']))" || exit 1; done
