fmt() { python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(d.get('config'), 'Mrays/s', d.get('mrays_per_s'), 'kernel_ms', d.get('kernel_ms'), 'lds', d.get('lds_bytes'), 'vgprs', d.get('vgprs'))
"; }
for rep in 1 2 3; do
echo "== shipped (refs b64 for mesh)"; python scripts/bench_configs.py --configs c3,c4 --spp 64 --steps 6 2>&1 | fmt
echo "== variant (full 16-B fetch)"; RTR_HIP_LIBRARY=$PWD/raytracingrust_amd/_lib/librtr_hip_nn.so python scripts/bench_configs.py --configs c3,c4 --spp 64 --steps 6 2>&1 | fmt
done
