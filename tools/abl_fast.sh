run() { python bench.py --cpu-seconds 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['parity']['bit_exact_vs_oracle'], d['roofline'])"; }
echo "== default"; run
echo "== serialized"; VSG_NO_OVERLAP=1 run
