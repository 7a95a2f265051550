for l in 2 3 4; do for b in 16777216 33554432; do
echo -n "lanes=$l batch=$b: "; RTGPU_RENDER_LANES=$l RTGPU_BATCH_PATHS=$b timeout -k 10 200 python bench.py --no-extras --no-roofline --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), round(d['value']))" || exit 1
done; done
