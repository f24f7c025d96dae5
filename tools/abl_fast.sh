run() { python bench.py --cpu-seconds 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['parity']['bit_exact_vs_oracle'], d['roofline']['stage_ms'] if d['roofline'] else '')"; }
for v in 0 2; do echo "== VSG_ORIENT_DBG=$v"; VSG_ORIENT_DBG=$v VSG_NO_OVERLAP=1 run --no-match --steps 100 --warmup 10; done
echo "== default"; run --steps 100 --warmup 10
