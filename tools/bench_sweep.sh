# workload sweep + PCIe-inclusive host-API rate (numbers for DESIGN.md section 8)
run() { python bench.py --cpu-seconds 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['config']['workload'][:2], 'fps', d['value'], 'ms/step', d['ms_per_step'], 'parity', d['parity']['bit_exact_vs_oracle'], 'kp', d['config']['keypoints_per_frame'], 'dom', r['kernel'], r['achieved'], r['stage_ms'])"; }
run --workload C2
run --workload C3
run --workload C4 --batch 128
run --workload C5
run --workload C2 --no-match
python - <<'PY'
import sys, time
sys.path.insert(0, '.')
import numpy as np
from visual_sgraphs_amd import orb, synth
for B in (1, 8, 64):
    ex = orb.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
    imgs = np.stack([synth.sequence_frame(640, 480, 2000, t) for t in range(B)])
    for _ in range(3): ex.extract_batch(imgs)
    n = max(3, 200 // B); t0 = time.perf_counter()
    for _ in range(n): ex.extract_batch(imgs)
    dt = time.perf_counter() - t0
    print(f"host API (pageable numpy in, numpy out, PCIe both ways) batch {B}: {B*n/dt:.0f} frames/s, {dt/n*1e3:.3f} ms per call")
ex = orb.ORBextractor(1000, 1.2, 8, 20, 7)
img = synth.frame(640, 480, 1)
for _ in range(5): ex(img)
t0 = time.perf_counter()
for _ in range(200): ex(img)
print(f"single-frame operator() latency (host image -> host keypoints): {(time.perf_counter()-t0)/200*1e3:.3f} ms")
PY
